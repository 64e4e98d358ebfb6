// kernels_labelling.h -- part of libpiscoseg.so, included by piscoseg.hip (one translation unit).
// The sparse labelling chain on the 1-bit threshold plane: run extraction, look-back scan, LDS union-find per band (k_runs_band), boundary unions, enclosure tree, bbox, area, ranks (device bodies in pseg_core.h).
#pragma once

// ------------------------------------------------------------------------------------------------
// sparse kernels: wrappers around pseg_core.h bodies
// ------------------------------------------------------------------------------------------------
#define GTID ((size_t)blockIdx.x * blockDim.x + threadIdx.x)
// inclusive prefix sum over the wave (DPP); lane 63 holds the total
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) { return wave_sum63(v); }
__device__ __forceinline__ uint32_t lane_shr1(uint32_t v)      // value of lane-1, 0 for lane 0 (wave_shr:1)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
}
// Foreground runs, one wave per group of RUNS_RPW rows (lane = 64-bit word of the current row; rows longer
// than 64 words loop with a carry).  FILL = false: rowcnt[row] = number of runs.  FILL = true: write
// run_x / run_fy from the row's first index; rows without runs are skipped without touching the plane.
#define RUNS_RPW 16
template <bool FILL>
__global__ __launch_bounds__(256) void k_runs(SegState S, int nrows)
{
    const int lane = threadIdx.x & 63;
    // the wave index is uniform: keeping it in a scalar register moves all per-row index arithmetic off the VALU
    const int row0 = (blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * RUNS_RPW;
    if (row0 >= nrows) return;
    unsigned long long todo;
    int startv = 0;                                       // lane i < RUNS_RPW: index of the first run of row row0 + i
    {
        const int row = row0 + lane;
        bool want = lane < RUNS_RPW && row < nrows;
        if (want) {
            const int f = row / S.H;
            if (!S.valid[f]) { want = false; if (!FILL) S.rowcnt[row] = 0; }
            else if (FILL) { want = seg_rowcount(S, f, row - f * S.H) != 0; startv = seg_rowstart(S, f, row - f * S.H); }
        }
        todo = __ballot(want);
    }
    // rows of up to 64 words: the words of all wanted rows are loaded first (one memory latency per wave)
    uint64_t pre[RUNS_RPW];
    const bool small = S.WW <= 64;
    if (small) {
#pragma unroll
        for (int i = 0; i < RUNS_RPW; i++)
            pre[i] = (((todo >> i) & 1ull) && lane < S.WW) ? S.bits[(size_t)(row0 + i) * S.WW + lane] : 0;
    }
#pragma unroll
    for (int i = 0; i < RUNS_RPW; i++) {
        if (!((todo >> i) & 1ull)) continue;
        const int row = row0 + i;
        const int f = row / S.H, y = row - f * S.H;
        const uint64_t *rb = S.bits + (size_t)row * S.WW;
        uint32_t total = 0, carry = 0;
        const int base = FILL ? __builtin_amdgcn_readlane(startv, i) : 0;
        for (int w0 = 0; w0 < S.WW; w0 += 64) {
            const int w = w0 + lane;
            const uint64_t v = small ? pre[i] : (w < S.WW ? rb[w] : 0);
            uint32_t top = (uint32_t)(v >> 63);
            uint32_t prevtop = lane_shr1(top);
            if (lane == 0) prevtop = carry;
            uint64_t st = v & ~((v << 1) | prevtop);
            uint32_t c = (uint32_t)__popcll(st);
            uint32_t incl = wave_scan_incl(c);
            if (FILL) {
                // ones that continue a run from bit 0 of this word on, over the following words of the chunk: suffix
                // doubling over the lanes (bit 31 = "every word so far is full"); only when some word ends set
                uint32_t ext = 0;
                if (__ballot(top != 0)) {
                    ext = (v == ~0ull) ? (0x80000000u | 64u) : (uint32_t)(__ffsll((unsigned long long)~v) - 1);
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        uint32_t o = __shfl_down(ext, d, 64);
                        if (lane + d > 63) o = (w0 + 64 < S.WW) ? 0x80000000u : 0u;   // beyond the chunk: unknown / row end
                        if (ext & 0x80000000u) ext = (ext & 0x7fffffffu) + (o & 0x7fffffffu) + (o & 0x80000000u);
                    }
                }
                uint32_t extn = __shfl_down(ext, 1, 64);           // continuation available in the next word
                if (lane == 63) extn = 0x80000000u;                  // unknown beyond the chunk: read the plane
                int idx = base + (int)total + (int)(incl - c);
                while (st) {
                    int s = __ffsll((unsigned long long)st) - 1;
                    st &= st - 1;
                    uint64_t t = ~(v >> s);                       // zeros mark the run's extent from bit s
                    int ones = t ? __ffsll((unsigned long long)t) - 1 : 64;
                    int xe = 64 * w + s + ones - 1;
                    if (s + ones == 64 && w + 1 < S.WW) {           // run continues into the next words
                        if (!(extn & 0x80000000u)) xe += (int)extn;
                        else {                                      // reaches the end of the chunk (rows wider than 4096 px) or of the row
                            int k = w + 1;
                            if (lane != 63) { xe += (int)(extn & 0x7fffffffu); k = w0 + 64; }
                            for (; k < S.WW; k++) {
                                uint64_t u = rb[k];
                                if (u == ~0ull) { xe += 64; continue; }
                                xe += __ffsll((unsigned long long)~u) - 1;
                                break;
                            }
                        }
                    }
                    S.run_x[idx] = (uint32_t)(64 * w + s) | ((uint32_t)xe << 16);
                    S.run_fy[idx] = (uint32_t)y | ((uint32_t)f << 16);
                    idx++;
                }
            }
            total += __builtin_amdgcn_readlane(incl, 63);
            carry = __builtin_amdgcn_readlane(top, 63);
        }
        if (!FILL && lane == 0) S.rowcnt[row] = total;
    }
}
__global__ void k_init(SegState S, int nfg, int nt) { size_t g = GTID; if (g < (size_t)nt) init_body(S, (int)g, nfg); }
// both union passes in one launch (two independent union-find structures)
__global__ void k_unions(SegState S, int nfg, int nt, int bandh)
{
    size_t g = GTID;
    if (g < (size_t)nfg) fg_union_body(S, (int)g, bandh);
    if (g < (size_t)nt) bg_union_body(S, (int)g, nfg, bandh);
}

#define BAND_H 64
#define BAND_CAP 1024
// The kernels of the labelling chain are queued before the host has seen the run count (no host round trip in the middle
// of a call): they are sized by the capacity of the run tables and do nothing when the runs did not fit (cap) or when a band
// had too many runs for the LDS labelling (bcap); the host then repeats the chain the slow way.  framebase[F] = runs of the
// batch, [F + 1] = largest band, [F + 2] = error flag of the look-back scan.
__device__ __forceinline__ bool chain_ok(const SegState &S, uint32_t cap, uint32_t bcap)
{
    // (the three words are requested together: with && the compiler waits for each before it asks for the next, three scalar-cache
    // round trips at the head of every wave of every kernel of the chain)
    const uint32_t a = S.framebase[S.F], b = S.framebase[S.F + 1], c = S.framebase[S.F + 2];
    return (a <= cap) & (b <= bcap) & (c == 0);             // (c: a scan that timed out left no usable tables)
}
// unions across the band boundaries only: block = the first row of a band (y = BAND_H, 2 BAND_H, ...) of one frame, a
// thread per run (its foreground unions with the row above, and the gap on its left) plus one for the trailing gap.  The
// per-element launch (k_unions over all runs and gaps) has one thread in BAND_H with work.
__global__ __launch_bounds__(64) void k_unions_rows(SegState S, int bandh, uint32_t cap, uint32_t bcap)
{
    // (queued behind k_runs_band before the host has seen the totals: nothing to do when the runs did not fit the tables
    // or a band was too large for the LDS labelling -- the host then takes another path)
    if (cap == 0 || !chain_ok(S, cap, bcap)) return;
    const int f = blockIdx.y, y = ((int)blockIdx.x + 1) * bandh;
    if (y >= S.H) return;
    const int rs = seg_rowstart(S, f, y), n = seg_rowcount(S, f, y);
    for (int j = threadIdx.x; j <= n; j += 64) {
        if (j < n) fg_union_body(S, rs + j);
        bg_union_elem(S, f, y, j);
    }
}

// ---- band-local labelling in LDS ----------------------------------------------------------------------------------
// Block = one band of BAND_H rows of one frame.  The band's runs and gaps are united with a union-find that lives in
// LDS (same rules as fg_union_body / bg_union_elem, local indices), flattened, and written to the global parent arrays
// as global indices; k_unions then only unites across the band boundaries and k_flatten_resolve finishes.  The
// frame-connected background (smallest background index of the frame, seg_b0) is the virtual node -1 here.  Also does
// the work of k_init for the band.  Used when no band of the batch has more than BAND_CAP runs (k_scan_rows reports
// the maximum); otherwise the host takes the global path.
__device__ __forceinline__ int lds_find(int *par, int a)
{
    if (a < 0) return -1;
    int p = par[a];
    while (p != a && p >= 0) {
        int g = par[p];
        if (g != p) par[a] = g;                              // path halving (g may be the virtual root -1)
        a = p; p = g;
    }
    return p < 0 ? -1 : a;
}
__device__ __forceinline__ void lds_union(int *par, int a, int b)
{
    for (int guard = 0; guard < (1 << 24); guard++) {
        a = lds_find(par, a); b = lds_find(par, b);
        if (a == b) return;
        const int hi = a > b ? a : b, lo = a > b ? b : a;    // hi >= 0; lo may be the virtual root
        const int old = atomicMin(&par[hi], lo);
        if (old == hi) return;
        a = old; b = lo;
    }
}
__global__ __launch_bounds__(256) void k_band_label(SegState S)
{
    __shared__ uint32_t lx[BAND_CAP];                     // run_x of the band's runs
    __shared__ uint16_t ly[BAND_CAP];                     // band-local row of every run
    __shared__ int fgp[BAND_CAP], bgp[BAND_CAP + BAND_H];
    __shared__ int lrow[BAND_H + 1];                      // local index of the first run of every row
    const int f = blockIdx.y, y0 = blockIdx.x * BAND_H, H = S.H;
    const int nr = (y0 + BAND_H <= H) ? BAND_H : H - y0;  // rows in this band
    const uint32_t *ro = S.rowoff + (size_t)f * (H + 1) + y0;
    const int R0 = (int)ro[0];                            // global index of the band's first run
    const int nb = (int)(ro[nr] - ro[0]);                 // runs in the band (<= BAND_CAP, checked by the host)
    const int B0g = seg_b0(S, f), Bb = R0 + f * H + y0;   // global background indices: frame-connected / band base
    for (int t = threadIdx.x; t <= nr; t += 256) lrow[t] = (int)(ro[t] - ro[0]);
    __syncthreads();
    // init (k_init's work for this band)
    for (int i = threadIdx.x; i < nb; i += 256) {
        const int r = R0 + i;
        const int y = (int)(S.run_fy[r] & 0xffffu), yl = y - y0;
        lx[i] = S.run_x[r]; ly[i] = (uint16_t)yl; fgp[i] = i;
        const int j = i - lrow[yl];
        bgp[i + yl] = (j == 0 || y == 0 || y == H - 1) ? -1 : i + yl;
        S.bg_left[Bb + i + yl] = j == 0 ? -1 : r - 1;
        S.minx[r] = 0x7fffffff; S.maxx[r] = -1; S.maxy[r] = -1; S.area2[r] = 0;
    }
    for (int yl = threadIdx.x; yl < nr; yl += 256) {      // trailing gap of every row
        const int g = lrow[yl + 1] + yl;
        bgp[g] = -1;
        S.bg_left[Bb + g] = lrow[yl + 1] == lrow[yl] ? -1 : R0 + lrow[yl + 1] - 1;
    }
    __syncthreads();
    // unions with the previous row of the band (rows yl >= 1)
    for (int i = threadIdx.x; i < nb + nr; i += 256) {
        int yl, j, n;
        if (i < nb) { yl = ly[i]; j = i - lrow[yl]; n = lrow[yl + 1] - lrow[yl]; }
        else { yl = i - nb; n = lrow[yl + 1] - lrow[yl]; j = n; }                 // trailing gap of row yl
        if (yl == 0) continue;
        const int ps = lrow[yl - 1], pn = lrow[yl] - ps, rs = lrow[yl];
        if (i < nb && pn > 0) {                              // foreground, 8-connectivity (fg_union_body)
            const int lo = (int)(lx[i] & 0xffffu) - 1, hi = (int)(lx[i] >> 16) + 1;
            int a = ps, b = ps + pn;                         // first q with xe_q >= lo
            while (a < b) { const int m = (a + b) >> 1; if ((int)(lx[m] >> 16) >= lo) b = m; else a = m + 1; }
            for (int q = a; q < ps + pn && (int)(lx[q] & 0xffffu) <= hi; q++) lds_union(fgp, i, q);
        }
        {                                                    // background, 4-connectivity (bg_union_elem)
            const int ga = j == 0 ? -1 : (int)(lx[rs + j - 1] >> 16) + 1;
            const int ge = j == n ? S.W : (int)(lx[rs + j] & 0xffffu) - 1;
            const int g = rs + j + yl, pb = ps + (yl - 1);
            int lo = 0, hi = pn;                             // first gap of the previous row whose right end >= ga
            while (lo < hi) { const int m = (lo + hi) >> 1; if ((int)(lx[ps + m] & 0xffffu) >= ga + 1) hi = m; else lo = m + 1; }
            for (int q = lo; q <= pn; q++) {
                const int qa = q == 0 ? -1 : (int)(lx[ps + q - 1] >> 16) + 1;
                if (qa > ge) break;
                lds_union(bgp, g, pb + q);
            }
        }
    }
    __syncthreads();
    // flatten and publish as global indices
    for (int i = threadIdx.x; i < nb + nr; i += 256) {
        if (i < nb) S.fg_parent[R0 + i] = R0 + lds_find(fgp, i);
        const int rt = lds_find(bgp, i);
        S.bg_parent[Bb + i] = rt < 0 ? B0g : Bb + rt;
    }
}
// ---- run extraction + scan + band labelling in ONE pass over the threshold plane ---------------------------------------
// Block = one band of BAND_H rows of one frame, wave = 16 rows, four lanes per row: a lane requests its quarter of the
// row's 64-bit words at once (one memory latency per wave) and walks them in order, so the carry between words stays in
// the lane; run starts  v & ~(v << 1 | carry)  are rare (one word in fifty).  A DPP scan over the wave (lane order = raster
// order) and the four wave totals give every lane the band-local index of its first run.
// The global index needs the number of runs in all bands before this one (frame-major, band-major = raster order): a
// decoupled look-back scan (Merrill / Garland) over the blocks.  Blocks take a ticket when they start, so every block with
// a smaller ticket is running or done; a block publishes its total (AGGREGATE) as soon as it has counted, wave 0 sums its
// predecessors' totals 64 at a time until it meets one that already knows its INCLUSIVE prefix, and publishes its own.
// The runs are staged in LDS, labelled there (same union-find as k_band_label: 8-connected foreground, 4-connected
// background, the frame-connected background as the virtual node -1) and leave the block once, with global indices, in
// coalesced stores -- the plane is read once, and the run tables are not read back.
// Launched before the host knows the run count, into the buffers as they stand: nothing is written to the run tables
// beyond `cap` entries (the totals still come out right; the host then grows the buffers and launches again).  A band with
// more than BAND_CAP runs skips the LDS labelling and writes its runs directly; bandmax tells the host to label globally.
// scan[0] = ticket counter, scan[2 + t] = state of block t: flag << 62 | value.  All cleared by k_reduce_partials.
// Ticket modes (order): BAND_ORDER_TICKET = one atomic on scan[0] per block, which is what makes the scan safe on any
// dispatcher (a ticket is only ever held by a running block).  8000 blocks asking one word of device memory for a number is
// 45 of the kernel's 158 us, so the default is BAND_ORDER_INDEX: ticket = block index.  That is safe where blocks start in
// index order on every XCD (then the unfinished block with the smallest index is always running: the blocks that hold its
// XCD's slots have smaller indices and wait for nothing unfinished) -- how the hardware dispatches a 1-D grid, but not a
// documented promise, so it is not trusted: the spin bound is short in this mode, and the host answers the timeout flag by
// running the batch again with tickets and keeping them for the rest of the context's life (no wrong result, no hang).
// It also needs this kernel to be the only one of its kind on the device: two index-ordered scans side by side can hold each
// other's slots (A's late blocks fill the XCD that B's earliest unfinished block is waiting to be placed on, and the other way
// round) until the bound ends it.  Inside one process the per-device turn guarantees that; contexts that do not take turns
// (PSEG_EXCLUSIVE=0) start with tickets, and processes that share a GPU find out through the timeout -- seen once in a
// rehearsal with four processes on one GPU (profiles/r02/56), which is why the bound is as short as it is.
// BAND_ORDER_TEST raises the timeout flag without waiting (the host's way back is under test in tests/test_gpu_async.py).
#define SCAN_AGG (1ull << 62)
#define SCAN_INC (2ull << 62)
#define SCAN_SPIN_MAX (1 << 22)
#define SCAN_SPIN_MAX_INDEX (1 << 16)
#define BAND_ORDER_TICKET 0
#define BAND_ORDER_INDEX 1
#define BAND_ORDER_TEST 2
#ifndef LB_K
#define LB_K 1                        // look-back windows (of 256 predecessor blocks) read per round (4: 0.179 ms against 0.157 ms)
#endif
// CAP = runs a band may hold for the LDS labelling: 1024 (15 KB of LDS, 8 blocks per CU) for ordinary frames, BAND_CAP_MID
// (29 KB, 5 blocks per CU) and BAND_CAP_DENSE (57 KB, 2 blocks per CU) for dense ones -- the host moves to the class a batch's
// largest band asks for, and back down after 8 batches that would have fitted the smaller one with a third to spare
#define BAND_CAP_MID 2048
#define BAND_CAP_DENSE 4096
#ifdef BAND_PROF
// -DBAND_PROF: thread 0 of every block adds the time (wall_clock64 ticks of 10 ns) it spent in each phase of k_runs_band;
// pseg_destroy prints the averages.  Not part of the product build.
__device__ unsigned long long g_band_prof[16];
#define BAND_STAMP(k) do { if (threadIdx.x == 0) { const unsigned long long now_ = wall_clock64(); dt_[k] = (uint32_t)(now_ - stamp_); stamp_ = now_; \
                           if ((k) == 5) for (int q_ = 0; q_ < 6; q_++) atomicAdd(&g_band_prof[q_], (unsigned long long)dt_[q_]); } } while (0)
#else
#define BAND_STAMP(k) do { } while (0)
#endif
template <int WPL, int CAP>
__global__ __launch_bounds__(256) void k_runs_band(SegState S, int wpl, uint32_t cap, unsigned long long *__restrict__ scan,
                                                   uint32_t *__restrict__ bandmax, int order)
{
#ifdef BAND_PROF
    unsigned long long stamp_ = wall_clock64();
    uint32_t dt_[6] = {0, 0, 0, 0, 0, 0};
    if (threadIdx.x == 0) atomicAdd(&g_band_prof[15], 1ull);
#endif
    __shared__ uint32_t lx[CAP];                          // run_x of the band's runs
    __shared__ uint16_t ly[CAP];                          // band-local row of every run
    __shared__ int fgp[CAP], bgp[CAP + BAND_H];
    __shared__ int lrow[BAND_H + 1];                      // local index of the first run of every row
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t s_ticket, s_base;
    const int H = S.H, nbpf = (H + BAND_H - 1) / BAND_H;
    if (order == BAND_ORDER_TICKET) {
        if (threadIdx.x == 0) s_ticket = (uint32_t)atomicAdd(&scan[0], 1ull);
        __syncthreads();
    }
    BAND_STAMP(0);                                        // ticket
    const int t = order == BAND_ORDER_TICKET ? (int)s_ticket : (int)blockIdx.x;
    const int spin_max = order == BAND_ORDER_TICKET ? SCAN_SPIN_MAX : SCAN_SPIN_MAX_INDEX;
    const int f = t / nbpf, band = t - f * nbpf, y0 = band * BAND_H;
    const int nr = (y0 + BAND_H <= H) ? BAND_H : H - y0;  // rows in this band
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, part = lane & 3;
    const int yl = wv * 16 + (lane >> 2), y = y0 + yl;
    const bool inband = yl < nr;
    const bool want = inband && S.valid[f];
    const uint64_t *rb = S.bits + ((size_t)f * H + (inband ? y : y0)) * S.WW;
    const int w0 = part * wpl, nw = want ? (w0 + wpl < S.WW ? wpl : S.WW - w0) : 0;     // nw <= wpl <= WPL (may be <= 0)
    uint64_t v[WPL];
#pragma unroll
    for (int k = 0; k < WPL; k++) v[k] = k < nw ? rb[w0 + k] : 0;
    const uint64_t before = (w0 > 0 && nw > 0) ? rb[w0 - 1] : 0;
    uint32_t cnt = 0;
    {
        uint64_t prev = before;
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            cnt += (uint32_t)__popcll(v[k] & ~((v[k] << 1) | (prev >> 63)));
            prev = v[k];
        }
    }
    const uint32_t incl = wave_scan_incl(cnt);            // lane order = (row, quarter) = raster order
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    BAND_STAMP(1);                                        // bit plane loaded, runs counted
    uint32_t wbase = 0;
    for (int w = 0; w < wv; w++) wbase += wtot[w];
    const uint32_t nb = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    if (threadIdx.x == 0) {
        __hip_atomic_store(&scan[2 + t], (t == 0 ? SCAN_INC : SCAN_AGG) | nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) s_base = 0;
        if (nb > *(volatile uint32_t *)bandmax) atomicMax(bandmax, nb);      // (a stale read only costs a redundant atomic)
        if (order == BAND_ORDER_TEST && t == 1) bandmax[1] = 1;
        lrow[BAND_H] = (int)nb;
    }
    const uint32_t li = wbase + incl - cnt;               // band-local index of the lane's first run
    if (part == 0) lrow[yl] = (int)li;
    const bool fits = nb <= CAP;
    // the lane's runs: into LDS when the band fits, else straight to memory once the base is known
    auto emit_runs = [&](auto &&put) {
        uint64_t prev = before;
        uint32_t idx = li;
#pragma unroll
        for (int k = 0; k < WPL; k++) {
            const uint64_t vv = v[k];
            uint64_t st = vv & ~((vv << 1) | (prev >> 63));
            prev = vv;
            const int w = w0 + k;
            while (st) {
                const int s0 = __ffsll((unsigned long long)st) - 1;
                st &= st - 1;
                const uint64_t z = ~(vv >> s0);             // zeros mark the run's extent from bit s0
                const int ones = z ? __ffsll((unsigned long long)z) - 1 : 64;
                int xe = 64 * w + s0 + ones - 1;
                if (s0 + ones == 64)                        // the run goes on into the next words (possibly another lane's)
                    for (int q = w + 1; q < S.WW; q++) {
                        const uint64_t u = rb[q];
                        if (u == ~0ull) { xe += 64; continue; }
                        xe += __ffsll((unsigned long long)~u) - 1;
                        break;
                    }
                put(idx++, (uint32_t)(64 * w + s0) | ((uint32_t)xe << 16));
            }
        }
    };
    if (fits && cnt) emit_runs([&](uint32_t i, uint32_t rx) { lx[i] = rx; ly[i] = (uint16_t)yl; });
    __syncthreads();
    BAND_STAMP(2);                                        // runs in LDS
    // global background index of the band's first gap.  That gap (the leading gap of the band's first row) touches the
    // padding column x = -1, so it stands for the frame-connected background (the virtual node -1) of this band; the
    // representatives of a frame's bands are united by k_unions_rows across the band boundaries, and the smallest one is
    // seg_b0(S, f) -- the frame's first background index, which only the block of band 0 could know here
    if (fits) {
        // init (k_init's work for this band)
        for (int i = threadIdx.x; i < (int)nb; i += 256) {
            const int yy = ly[i], j = i - lrow[yy], ya = y0 + yy;
            fgp[i] = i;
            bgp[i + yy] = (j == 0 || ya == 0 || ya == H - 1) ? -1 : i + yy;
        }
        for (int yy = threadIdx.x; yy < nr; yy += 256) bgp[lrow[yy + 1] + yy] = -1;      // trailing gap of every row
        __syncthreads();
        // unions with the previous row of the band (rows yy >= 1)
        for (int i = threadIdx.x; i < (int)nb + nr; i += 256) {
            int yy, j, n;
            if (i < (int)nb) { yy = ly[i]; j = i - lrow[yy]; n = lrow[yy + 1] - lrow[yy]; }
            else { yy = i - (int)nb; n = lrow[yy + 1] - lrow[yy]; j = n; }                 // trailing gap of row yy
            if (yy == 0) continue;
            const int ps = lrow[yy - 1], pn = lrow[yy] - ps, rs = lrow[yy];
            if (i < (int)nb && pn > 0) {                         // foreground, 8-connectivity (fg_union_body)
                const int lo = (int)(lx[i] & 0xffffu) - 1, hi = (int)(lx[i] >> 16) + 1;
                int a = ps, b = ps + pn;                         // first q with xe_q >= lo
                while (a < b) { const int m = (a + b) >> 1; if ((int)(lx[m] >> 16) >= lo) b = m; else a = m + 1; }
                for (int q = a; q < ps + pn && (int)(lx[q] & 0xffffu) <= hi; q++) lds_union(fgp, i, q);
            }
            {                                                    // background, 4-connectivity (bg_union_elem)
                const int ga = j == 0 ? -1 : (int)(lx[rs + j - 1] >> 16) + 1;
                const int ge = j == n ? S.W : (int)(lx[rs + j] & 0xffffu) - 1;
                const int g = rs + j + yy, pb = ps + (yy - 1);
                int lo = 0, hi = pn;                             // first gap of the previous row whose right end >= ga
                while (lo < hi) { const int m = (lo + hi) >> 1; if ((int)(lx[ps + m] & 0xffffu) >= ga + 1) hi = m; else lo = m + 1; }
                for (int q = lo; q <= pn; q++) {
                    const int qa = q == 0 ? -1 : (int)(lx[ps + q - 1] >> 16) + 1;
                    if (qa > ge) break;
                    lds_union(bgp, g, pb + q);
                }
            }
        }
        __syncthreads();
    }
    BAND_STAMP(3);                                        // labelled
    // decoupled look-back, after the labelling (most predecessors know their inclusive prefix by now): the whole block
    // reads 256 predecessor states per round, every wave sums its 64 up to the nearest inclusive one
    __shared__ uint32_t lb_sum[4 * LB_K], lb_hit[4 * LB_K];
    if (t > 0) {
        uint32_t excl = 0;
        int pos = t - 1;
        bool timeout = false;
        for (;;) {
            // LB_K windows of 256 predecessors per round, all loads in flight together
            unsigned long long sv[LB_K];
#pragma unroll
            for (int k = 0; k < LB_K; k++) {
                const int idx = pos - 256 * k - (int)threadIdx.x;
                sv[k] = idx >= 0 ? __hip_atomic_load(&scan[2 + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : SCAN_INC;   // before block 0: inclusive prefix 0
            }
#pragma unroll
            for (int k = 0; k < LB_K; k++) {
                const int idx = pos - 256 * k - (int)threadIdx.x;
                int spins = 0;
                while ((sv[k] >> 62) == 0 && ++spins < spin_max) {
                    __builtin_amdgcn_s_sleep(2);
                    sv[k] = __hip_atomic_load(&scan[2 + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#ifdef BAND_PROF
                if (spins) { atomicAdd(&g_band_prof[9], (unsigned long long)spins); atomicAdd(&g_band_prof[10], 1ull); }   // polls, threads that polled
#endif
                if ((sv[k] >> 62) == 0) { timeout = true; sv[k] = SCAN_INC; }   // never expected; the host reports it
                const unsigned long long incm = __ballot((sv[k] >> 62) == 2);
                const int L = incm ? (int)__builtin_ctzll(incm) : 64;
                const uint32_t part_sum = wave_sum63(lane <= L ? (uint32_t)sv[k] : 0u);
                if (lane == 63) { lb_sum[4 * k + wv] = part_sum; lb_hit[4 * k + wv] = incm != 0; }
            }
            if (timeout) bandmax[1] = 1;
#ifdef BAND_PROF
            if (threadIdx.x == 0) atomicAdd(&g_band_prof[8], 1ull);                    // look-back rounds
#endif
            __syncthreads();
            bool hit = false;
            for (int w = 0; w < 4 * LB_K && !hit; w++) { excl += lb_sum[w]; hit = lb_hit[w] != 0; }
            __syncthreads();
            if (hit) break;
            pos -= 256 * LB_K;
        }
        if (threadIdx.x == 0) {
            s_base = excl;
            __hip_atomic_store(&scan[2 + t], SCAN_INC | (unsigned long long)(excl + nb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    BAND_STAMP(4);                                        // look-back
    const int R0 = (int)s_base;                           // global index of the band's first run
    // (cap == 0: the tables do not exist yet -- not even the background entries every row owns without any run)
    const bool room = cap > 0 && (unsigned long long)R0 + nb <= (unsigned long long)cap;
    // row table (global indices), frame bases, the grand total
    if (part == 0 && inband) S.rowoff[(size_t)f * (H + 1) + y] = (uint32_t)R0 + li;
    if (threadIdx.x == 0) {
        if (band == nbpf - 1) S.rowoff[(size_t)f * (H + 1) + H] = (uint32_t)R0 + nb;
        if (band == 0) S.framebase[f] = (uint32_t)R0;
        if (t == nbpf * S.F - 1) S.framebase[S.F] = (uint32_t)R0 + nb;
    }
    if (!room) return;
    if (!fits) {                                          // too many runs for the LDS tables: the host labels globally (k_init, k_unions)
        const uint32_t fy = (uint32_t)y | ((uint32_t)f << 16);
        if (cnt) emit_runs([&](uint32_t i, uint32_t rx) { S.run_x[R0 + i] = rx; S.run_fy[R0 + i] = fy; });
        return;
    }
    const int Bb = R0 + f * H + y0;                       // global background index of the band's first gap (see below)
    // flatten and publish: runs, parents (global indices), the per-object accumulators' start values
    for (int i = threadIdx.x; i < (int)nb + nr; i += 256) {
        if (i < (int)nb) {
            const int r = R0 + i, yy = ly[i], j = i - lrow[yy];
            S.run_x[r] = lx[i];
            S.run_fy[r] = (uint32_t)(y0 + yy) | ((uint32_t)f << 16);
            S.fg_parent[r] = R0 + lds_find(fgp, i);
            S.bg_left[Bb + i + yy] = j == 0 ? -1 : r - 1;
            S.minx[r] = 0x7fffffff; S.maxx[r] = -1; S.maxy[r] = -1; S.area2[r] = 0;
        }
        const int rt = lds_find(bgp, i);
        S.bg_parent[Bb + i] = rt < 0 ? Bb : Bb + rt;
    }
    for (int yy = threadIdx.x; yy < nr; yy += 256) {        // trailing gap of every row
        const int g = lrow[yy + 1] + yy;
        S.bg_left[Bb + g] = lrow[yy + 1] == lrow[yy] ? -1 : R0 + lrow[yy + 1] - 1;
    }
    BAND_STAMP(5);                                        // published (stores issued)
}

// flatten + enclosure tree in one launch: every parent is followed to its root on the fly (the parents other threads
// flatten meanwhile are only ever replaced by one of their ancestors, so a chain stays a chain to the same root)
__global__ void k_flatten_resolve(SegState S, uint32_t cap, uint32_t bcap)
{
    if (!chain_ok(S, cap, bcap)) return;
    const int nfg = (int)S.framebase[S.F], nbg = nfg + S.F * S.H;
    size_t g = GTID;
    if (g < (size_t)nbg) S.bg_parent[g] = uf_find(S.bg_parent, (int)g);
    if (g >= (size_t)nfg) return;
    const int r = (int)g;
    const int root = uf_find(S.fg_parent, r);
    int R = root;
    for (int guard = 0; guard < 65536; guard++) {
        const int y = (int)(S.run_fy[R] & 0xffffu), f = (int)(S.run_fy[R] >> 16);
        const int hb = uf_find(S.bg_parent, R + f * S.H + y);   // background around R's raster-first pixel
        if (hb == seg_b0(S, f)) break;                       // frame-connected: R is external
        R = uf_find(S.fg_parent, S.bg_left[hb]);             // component that owns the hole
    }
    S.fg_parent[r] = root;
    S.ext[r] = R;
}
// bounding boxes: the 256 runs of a block are combined per object in an LDS hash table first, so that
// only one set of global atomics is issued per (block, object)
__global__ __launch_bounds__(256) void k_bbox(SegState S, uint32_t cap, uint32_t bcap)
{
    __shared__ int keys[512], vminx[512], vmaxx[512], vmaxy[512];
    if (!chain_ok(S, cap, bcap)) return;
    const int nfg = (int)S.framebase[S.F];
    for (int i = threadIdx.x; i < 512; i += 256) { keys[i] = -1; vminx[i] = 0x7fffffff; vmaxx[i] = -1; vmaxy[i] = -1; }
    __syncthreads();
    size_t g = GTID;
    if (g < (size_t)nfg) {
        int r = (int)g, E = S.ext[r];
        unsigned h = ((unsigned)E * 2654435761u) >> 23;       // 9 bits
        for (;;) {
            int old = atomicCAS(&keys[h], -1, E);
            if (old == -1 || old == E) break;
            h = (h + 1) & 511;
        }
        atomicMin(&vminx[h], run_xs(S, r));
        atomicMax(&vmaxx[h], run_xe(S, r));
        const int y = (int)(S.run_fy[r] & 0xffffu), f = (int)(S.run_fy[r] >> 16);
        atomicMax(&vmaxy[h], y);
        // object that fills the gap on the left of this run (-1: none, or the row's leading gap)
        S.gapl[r] = (r == seg_rowstart(S, f, y)) ? -1 : bg_gid(S, f, r + f * S.H + y);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
        if (keys[i] >= 0) bbox_flush(S, keys[i], vminx[i], vmaxx[i], vmaxy[i]);
}
// contour areas: the contributions of a block's 256 runs are summed per object in an LDS hash table first (a large
// object receives one contribution per row and span: hundreds of atomics on one word otherwise)
__global__ __launch_bounds__(256) void k_area(SegState S, uint32_t cap, uint32_t bcap)
{
    __shared__ int keys[512], vsum[512];
    if (!chain_ok(S, cap, bcap)) return;
    const int nfg = (int)S.framebase[S.F];
    for (int i = threadIdx.x; i < 512; i += 256) { keys[i] = -1; vsum[i] = 0; }
    __syncthreads();
    size_t g = GTID;
    if (g < (size_t)nfg) {
        int E;
        const int add = area_contrib(S, (int)g, &E);
        if (add) {
            unsigned h = ((unsigned)E * 2654435761u) >> 23;   // 9 bits
            for (;;) {
                int old = atomicCAS(&keys[h], -1, E);
                if (old == -1 || old == E) break;
                h = (h + 1) & 511;
            }
            atomicAdd(&vsum[h], add);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
        if (keys[i] >= 0) atomicAdd(&S.area2[keys[i]], vsum[i]);
}

// exclusive scan of rowcnt inside each frame (block per frame); ftot[f] = runs in frame f
__global__ __launch_bounds__(256) void k_scan_rows(SegState S, uint32_t *ftot, uint32_t *bandmax)
{
    __shared__ uint32_t wtot[4];
    __shared__ uint32_t carry;
    const int f = blockIdx.x, H = S.H, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < H; base += 256) {
        const int y = base + threadIdx.x;
        const uint32_t v = y < H ? S.rowcnt[(size_t)f * H + y] : 0;
        uint32_t incl = v;                                   // inclusive scan inside the wave, then across the four waves
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        uint32_t pre = carry;
        for (int w = 0; w < wv; w++) pre += wtot[w];
        if (y < H) S.rowoff[(size_t)f * (H + 1) + y] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { S.rowoff[(size_t)f * (H + 1) + H] = carry; ftot[f] = carry; }
    // largest number of runs in one band of BAND_H rows (decides between k_band_label and the global path)
    __syncthreads();
    uint32_t bm = 0;
    for (int b = threadIdx.x; b * BAND_H < H; b += 256) {
        const int ya = b * BAND_H, yb = ya + BAND_H < H ? ya + BAND_H : H;
        const uint32_t c = S.rowoff[(size_t)f * (H + 1) + yb] - S.rowoff[(size_t)f * (H + 1) + ya];
        bm = c > bm ? c : bm;
    }
    for (int o = 32; o > 0; o >>= 1) { uint32_t t = __shfl_xor(bm, o, 64); bm = t > bm ? t : bm; }
    if (lane == 0) wtot[wv] = bm;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) bm = wtot[w] > bm ? wtot[w] : bm;
        if (bm > *(volatile uint32_t *)bandmax) atomicMax(bandmax, bm);      // (a stale read only costs a redundant atomic)
    }
}

// two-pass path (k_runs): the row table becomes global run indices, like the one k_runs_band writes
__global__ void k_rowoff_global(SegState S)
{
    size_t g = GTID;
    if (g >= (size_t)S.F * (S.H + 1)) return;
    S.rowoff[g] += S.framebase[g / (S.H + 1)];
}

// exclusive scan over the frames by one block: out[0..F] (strided input / output)
__global__ __launch_bounds__(256) void k_scan_small(const uint32_t *in, uint32_t *out, int F, int stride_in, int stride_out)
{
    in += blockIdx.x; out += (size_t)blockIdx.x * (F + 1);  // grid.x > 1: columns in, in+1, .. -> tables out, out + (F+1), ..
    __shared__ uint32_t sh[256];
    __shared__ uint32_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < F; base += 256) {
        int f = base + threadIdx.x;
        uint32_t v = f < F ? in[(size_t)f * stride_in] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int o = 1; o < 256; o <<= 1) {
            uint32_t t = threadIdx.x >= o ? sh[threadIdx.x - o] : 0;
            __syncthreads();
            sh[threadIdx.x] += t;
            __syncthreads();
        }
        uint32_t c0 = carry;
        if (f < F) out[(size_t)f * stride_out] = c0 + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 255) carry = c0 + sh[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)F * stride_out] = carry;
}

#define RANK_IPT 4
// Label-map values before the table exists.  The label of a run is the 1-based OpenCV index of its external contour =
// (external roots of the frame) - (rank of its root among them), and needs neither areas nor boxes: this kernel ranks the
// external roots of every frame (block per frame, as k_rank) and then writes runlabel for the frame's runs, so that the label
// fill can follow the enclosure tree directly and bbox / area / ranks / rows / size scans run BEHIND the fill, beside the next
// batch's background kernel (they were 67 us of the batch's critical path, 0.44 ms on frames of 2000 objects).
// ftot3[f*3] = number of external contours.
__global__ __launch_bounds__(1024) void k_rank_ext(SegState S, uint32_t *ftot3, int32_t *__restrict__ runlabel, uint32_t cap, uint32_t bcap)
{
    __shared__ uint32_t wtot[16];
    __shared__ uint32_t carry;
    if (!chain_ok(S, cap, bcap)) return;
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = (int)S.framebase[f], r1 = (int)S.framebase[f + 1];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = r0; base < r1; base += 1024 * RANK_IPT) {
        const int r = base + RANK_IPT * (int)threadIdx.x;
        const bool full = r + RANK_IPT <= r1;
        int fp[RANK_IPT], ex[RANK_IPT];
        if (full) { __builtin_memcpy(fp, S.fg_parent + r, 4 * RANK_IPT); __builtin_memcpy(ex, S.ext + r, 4 * RANK_IPT); }
        else
#pragma unroll
            for (int j = 0; j < RANK_IPT; j++) { fp[j] = r + j < r1 ? S.fg_parent[r + j] : -1; ex[j] = r + j < r1 ? S.ext[r + j] : -1; }
        uint32_t m0 = 0;                                     // bit j: run r + j is an external root
#pragma unroll
        for (int j = 0; j < RANK_IPT; j++) m0 |= (uint32_t)(fp[j] == r + j && ex[j] == r + j) << j;
        const uint32_t cnt = (uint32_t)__popc(m0);
        const uint32_t incl = wave_scan_incl(cnt);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        uint32_t p0 = carry;
        for (int w = 0; w < wv; w++) p0 += wtot[w];
        const uint32_t excl = p0 + incl - cnt;
#pragma unroll
        for (int j = 0; j < RANK_IPT; j++)
            if ((m0 >> j) & 1u)                             // (only roots are ever looked up; stored past the L1 cache: see below)
                __hip_atomic_store(S.extrank + r + j, (int)(excl + (uint32_t)__popc(m0 & ((1u << j) - 1u))), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (threadIdx.x == 1023) carry = p0 + incl;
        __syncthreads();
    }
    const int next = (int)carry;
    if (threadIdx.x == 0) ftot3[f * 3] = (uint32_t)next;
    // The ranks written above are read back by other threads of this block.  No fence (an agent-scope fence writes the whole
    // L2 of the XCD back: the kernel took 57 us with one): the stores above and the loads below are relaxed agent-scope
    // atomics, which pass the L1 cache both ways, and the barrier waits for the stores' acknowledgements.
    __syncthreads();
    for (int base = r0; base < r1; base += 1024 * RANK_IPT) {
        const int r = base + RANK_IPT * (int)threadIdx.x;
        if (r >= r1) break;
        const bool full = r + RANK_IPT <= r1;
        int fp[RANK_IPT], ex[RANK_IPT], lab[RANK_IPT];
        if (full) { __builtin_memcpy(fp, S.fg_parent + r, 4 * RANK_IPT); __builtin_memcpy(ex, S.ext + r, 4 * RANK_IPT); }
        else
#pragma unroll
            for (int j = 0; j < RANK_IPT; j++) { fp[j] = r + j < r1 ? S.fg_parent[r + j] : -1; ex[j] = r + j < r1 ? S.ext[r + j] : -2; }
        // label of the run's external component, 0 for components nested in holes (their root is not the external one)
#pragma unroll
        for (int j = 0; j < RANK_IPT; j++) lab[j] = (fp[j] == ex[j]) ? next - __hip_atomic_load(S.extrank + ex[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;   // ... and are read from there
        if (full) __builtin_memcpy(runlabel + r, lab, 4 * RANK_IPT);
        else
#pragma unroll
            for (int j = 0; j < RANK_IPT; j++) if (r + j < r1) runlabel[r + j] = lab[j];
    }
}

// ranks of external roots / kept / saved objects inside each frame (block per frame; a thread takes RANK_IPT consecutive
// runs per round, the three counts of a thread travel through the wave scan packed in one register).
// ftot3[f*3+{0,1,2}] = number of external contours, kept, saved.
__global__ __launch_bounds__(1024) void k_rank(SegState S, double min_seg, double min_save, uint32_t *ftot3, uint32_t cap, uint32_t bcap)
{
    __shared__ uint32_t wtot[16];                         // per wave: the three totals, 10 bits each (<= 64 * RANK_IPT)
    __shared__ uint32_t carry[3];
    if (!chain_ok(S, cap, bcap)) return;
    const int f = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int r0 = (int)S.framebase[f], r1 = (int)S.framebase[f + 1];
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int base = r0; base < r1; base += 1024 * RANK_IPT) {
        const int r = base + RANK_IPT * (int)threadIdx.x;
        const bool full = r + RANK_IPT <= r1;
        int fp[RANK_IPT], ex[RANK_IPT];
        if (full) { __builtin_memcpy(fp, S.fg_parent + r, 4 * RANK_IPT); __builtin_memcpy(ex, S.ext + r, 4 * RANK_IPT); }
        else
#pragma unroll
            for (int j = 0; j < RANK_IPT; j++) { fp[j] = r + j < r1 ? S.fg_parent[r + j] : -1; ex[j] = r + j < r1 ? S.ext[r + j] : -1; }
        uint32_t m0 = 0, m1 = 0, m2 = 0;                    // bit j: run r + j is an external root / kept / saved
#pragma unroll
        for (int j = 0; j < RANK_IPT; j++)
            if (fp[j] == r + j && ex[j] == r + j) {         // is_ext_root
                const double area = 0.5 * (double)S.area2[r + j];
                const bool v1 = !(area < min_seg);          // detection.py:223 ("< -> continue")
                m0 |= 1u << j; m1 |= (uint32_t)v1 << j; m2 |= (uint32_t)(v1 && area > min_save) << j;   // detection.py:235
            }
        const uint32_t packed = (uint32_t)__popc(m0) | ((uint32_t)__popc(m1) << 10) | ((uint32_t)__popc(m2) << 20);
        const uint32_t incl = wave_scan_incl(packed);
        if (lane == 63) wtot[wv] = incl;
        __syncthreads();
        uint32_t p0 = carry[0], p1 = carry[1], p2 = carry[2];
        for (int w = 0; w < wv; w++) { const uint32_t t = wtot[w]; p0 += t & 0x3ffu; p1 += (t >> 10) & 0x3ffu; p2 += t >> 20; }
        const uint32_t excl = incl - packed;
        int o0[RANK_IPT], o1[RANK_IPT], o2[RANK_IPT];
#pragma unroll
        for (int j = 0; j < RANK_IPT; j++) {
            const uint32_t below = (1u << j) - 1u;
            o0[j] = (int)(p0 + (excl & 0x3ffu) + (uint32_t)__popc(m0 & below));
            o1[j] = (int)(p1 + ((excl >> 10) & 0x3ffu) + (uint32_t)__popc(m1 & below));
            o2[j] = (int)(p2 + (excl >> 20) + (uint32_t)__popc(m2 & below));
        }
        if (full) {
            __builtin_memcpy(S.extrank + r, o0, 4 * RANK_IPT); __builtin_memcpy(S.keptrank + r, o1, 4 * RANK_IPT);
            __builtin_memcpy(S.savedrank + r, o2, 4 * RANK_IPT);
        } else
#pragma unroll
            for (int j = 0; j < RANK_IPT; j++)
                if (r + j < r1) { S.extrank[r + j] = o0[j]; S.keptrank[r + j] = o1[j]; S.savedrank[r + j] = o2[j]; }
        __syncthreads();
        if (threadIdx.x == 1023) {
            const uint32_t t = wtot[15];
            carry[0] = p0 + (t & 0x3ffu); carry[1] = p1 + ((t >> 10) & 0x3ffu); carry[2] = p2 + (t >> 20);
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) ftot3[f * 3 + threadIdx.x] = carry[threadIdx.x];
}

