// kernels_output.h -- part of libpiscoseg.so, included by piscoseg.hip (one translation unit).
// What leaves the chain: table rows (k_emit), label map (k_label_fill / k_label_fill_rows), size scans, crop bands (k_band_map, k_crops_sums: mask window, three crops, integer sums), convex hulls (k_convex).
#pragma once

#ifndef CROP_BAND
#define CROP_BAND 32768                // (8192 / 16384 / 65536 px: 1.93 / 1.91 / 1.92 ms per step against 1.90 ms)
#endif
#ifndef CROP_THREADS
#define CROP_THREADS 256
#endif
#ifndef CROP_UNROLL
#define CROP_UNROLL 4
#endif
#ifndef CROP_TPR
#define CROP_TPR 4                   // threads that share the runs of one row while the mask is painted (64 rows at a time;
                                     // 16 / 8 / 2 threads per row: 0.384 / 0.351 / 0.338 ms for the crop group against 0.334 ms)
#endif
#ifndef CROP_EXP
#define CROP_EXP 0                   // experiments only (wrong results): 1 no flush of the sums, 2 no mask painting, 4 no pixel loop, 8 no moment / perimeter / Euler loops
#endif
#ifndef CROP_RPT
#define CROP_RPT 4                   // runs of a row a thread requests together while the mask is painted
#endif
#ifndef CROP_LDS_WORDS
#define CROP_LDS_WORDS 2048          // mask words of a band + 2 halo rows on each side (frames up to 8192 px wide: 8 rows of 256 words)
#endif
#define CROP_SMALL_ONE (1ull << 40)   // the band column of the size scan: bands | small windows << 40
#ifndef CROP_SMALL_THREADS
#define CROP_SMALL_THREADS 64
#endif
#ifndef CROP_SMALL_TPR
#define CROP_SMALL_TPR 2
#endif
#define CROP_SMALL_WORDS 256          // LDS words per array of the small class: (124 + 4) rows x 2 words
// a window of one band whose mask rows (+ 4 halo rows) fit CROP_SMALL_WORDS
__host__ __device__ inline bool crop_band_is_small(int w, int h) { return w <= 64 && h <= 124; }
__host__ __device__ inline int crop_rows_per_band(int w)
{
    int r = CROP_BAND / w;
    // (the band-local 32-bit sums of k_crops_sums: sum yl^2 over a full band < w r^3 / 3 < 2^32 for every w with r <= 600,
    // and r = CROP_BAND / w <= 585 where w >= 56)
    if (r > 600) r = 600;
    const int fit = CROP_LDS_WORDS / ((w + 31) >> 5) - 4;  // rows + 4 halo rows of (w + 31) / 32 words fit the LDS arrays
    if (r > fit) r = fit;
    return r < 1 ? 1 : r;
}
// one row per kept object, ordered (frame, c); c counts from the LAST external root (detection.py:222)
__global__ void k_emit(SegState S, uint32_t cap, uint32_t bcap, double min_seg, double min_save, const uint32_t *ftot3,
                       const uint32_t *rowbase, const uint32_t *savedbase, RowsDev R, int32_t *objlabel, int32_t *objflag,
                       int32_t *saved_row, int32_t *geo, unsigned long long *sz, size_t szstride)
{
    if (!chain_ok(S, cap, bcap)) return;
    const int nfg = (int)S.framebase[S.F];
    size_t g = GTID;
    if (g >= (size_t)nfg) return;
    int r = (int)g;
    objflag[r] = 0;
    objlabel[r] = 0;
    int f = (int)(S.run_fy[r] >> 16);
    // (the label-map value of the run, runlabel[r], was written by k_rank_ext ahead of the label fill)
    if (!is_ext_root(S, r)) return;
    int next = (int)ftot3[f * 3], nkept = (int)ftot3[f * 3 + 1], nsaved = (int)ftot3[f * 3 + 2];
    int label = next - S.extrank[r];
    objlabel[r] = label;
    double area = 0.5 * (double)S.area2[r];
    if (area < min_seg) return;
    bool saved = area > min_save;
    int c = nkept - 1 - S.keptrank[r];
    int i = (int)rowbase[f] + c;
    R.frame[i] = f; R.c[i] = c; R.label[i] = label; R.area2[i] = S.area2[r];
    const int y0 = (int)(S.run_fy[r] & 0xffffu);           // the root run lies in the object's first row
    const int x = S.minx[r], w = S.maxx[r] - x + 1, h = S.maxy[r] - y0 + 1;
    R.x[i] = x; R.y[i] = y0;
    R.w[i] = w; R.h[i] = h;
    R.saved[i] = saved; R.root[i] = r;
    objflag[r] = 1 | (saved ? 2 : 0);
    if (saved) {
        // geometry of saved object s: geo[s*8] = f, x, y, w, h, root, words-per-row, row; the three size columns
        // sz[k * szstride + s] = crop bytes | rows (offset of the per-row extremes) | row bands (a band = a run of rows of
        // about CROP_BAND pixels: the unit of work of k_crops_sums), scanned by k_scan_sizes
        const size_t sidx = (size_t)savedbase[f] + (size_t)(nsaved - 1 - S.savedrank[r]);
        saved_row[sidx] = i;
        int32_t *o = geo + 8 * sidx;
        o[0] = f; o[1] = x; o[2] = y0; o[3] = w; o[4] = h; o[5] = r; o[6] = (w + 31) >> 5; o[7] = i;
        const int rpb = crop_rows_per_band(w);
        sz[sidx] = (unsigned long long)w * h;
        sz[szstride + sidx] = (unsigned long long)h;
        // (bit 40 counts the windows of the small class of k_crops_sums, which have one band: the scan of this column then gives
        // every object its bands' place among the bands of its own class -- k_band_map)
        sz[2 * szstride + sidx] = (unsigned long long)((h + rpb - 1) / rpb) | (crop_band_is_small(w, h) ? CROP_SMALL_ONE : 0ull);
    }
}

// label map: one dense pass writes every int32 exactly once.  Wave = one 256-pixel span of a row
// (lane = 4 pixels = one 16-byte store; a block writes 4 consecutive spans, blocks sweep the map in address order --
// the shape that streams at 6.8 TB/s when it only stores).  A span whose threshold bits are all zero is stored at once.
// Otherwise the wave works together: the row's runs and their labels are loaded one per lane, the runs that meet the
// span are broadcast one after the other (typically one or two) and every lane takes what falls on its pixels.
typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));
#define FILL_ROWS_MIN 8
// grid = (ceil(W / 1024), H, frames): row and frame come from the block index (no divisions; at one KiB per wave the
// kernel only streams at full rate while a wave spends a few dozen instructions)
__global__ __launch_bounds__(256) void k_label_fill(SegState S, const int32_t *__restrict__ runlabel, int32_t *__restrict__ labels,
                                                    uint32_t cap, uint32_t bcap)
{
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.z, y = blockIdx.y;
    // (the row's place in the run tables is requested together with the chain's totals, before either is looked at: a wave lives for
    // about a microsecond, most of it dependent loads -- kernel arguments, totals, row offsets, runs -- and the kernel's rate is
    // the number of waves in flight over that time; the offsets lie inside the table whatever the chain did)
    uint32_t t0 = S.framebase[S.F], t1 = S.framebase[S.F + 1], t2 = S.framebase[S.F + 2];
    uint32_t r0 = S.rowoff[f * (S.H + 1) + y], r1 = S.rowoff[f * (S.H + 1) + y + 1];
    asm volatile("" : "+s"(t0), "+s"(t1), "+s"(t2), "+s"(r0), "+s"(r1));      // (all five here: the compiler would sink the offsets below the test)
    if (!((t0 <= cap) & (t1 <= bcap) & (t2 == 0))) return;  // chain_ok (queued with the chain)
    const int rs = (int)r0, n = (int)(r1 - r0);
    const int x0 = blockIdx.x * 1024 + (threadIdx.x >> 6) * 256, x = x0 + 4 * lane;
    if (x0 >= S.W) return;                                   // wave beyond the row end
    const bool in = x < S.W;
    const size_t rowpix = ((size_t)f * S.H + y) * S.W;
    i32x4_t out = {0, 0, 0, 0};
    // the row's run list (L2-resident tables) decides, not the threshold plane: a read stream next to the write stream
    // costs the stores more than the lookups do.  (One bit per row and span left by k_runs_band -- "no foreground here", 2 MB per
    // batch read through the scalar cache, seven spans in ten stored without a look at the run tables -- made the kernel slower:
    // 1.205 against 1.087 ms, four alternating runs on one box, round 3.)
    {
        for (int j0 = 0; j0 < n; j0 += 64) {
            uint32_t rx = 0xffffu; int lab = 0;              // lanes past the row's runs: xs = 65535, xe = 0
            if (j0 + lane < n) { rx = S.run_x[rs + j0 + lane]; lab = runlabel[rs + j0 + lane]; }
            const int xs = (int)(rx & 0xffffu), xe = (int)(rx >> 16);
            unsigned long long hit = __ballot(lab != 0 && xe >= x0 && xs <= x0 + 255);
            while (hit) {
                const int l = __builtin_ctzll(hit);
                hit &= hit - 1;
                const int a = __builtin_amdgcn_readlane(xs, l), b = __builtin_amdgcn_readlane(xe, l), lb = __builtin_amdgcn_readlane(lab, l);
                if (a <= x && x <= b) out.x = lb;
                if (a <= x + 1 && x + 1 <= b) out.y = lb;
                if (a <= x + 2 && x + 2 <= b) out.z = lb;
                if (a <= x + 3 && x + 3 <= b) out.w = lb;
            }
            if (__builtin_amdgcn_readlane(xs, 63) > x0 + 255) break;    // runs are sorted: nothing further meets the span
        }
    }
    if (in) __builtin_nontemporal_store(out, (i32x4_t *)(labels + rowpix + x));
}

// The same map with one wave per ROW (block = 4 consecutive rows): the row's first 64 runs and their labels are loaded
// once and kept in registers while the wave walks the row's 256-pixel spans, so that a wave's two dependent table loads
// are paid once per row (10 KiB at W = 2560) instead of once per KiB.  For rows with many runs (dense frames).
__global__ __launch_bounds__(256) void k_label_fill_rows(SegState S, const int32_t *__restrict__ runlabel, int32_t *__restrict__ labels,
                                                         uint32_t cap, uint32_t bcap)
{
    if (!chain_ok(S, cap, bcap)) return;
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.z, y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= S.H) return;
    const size_t rowpix = ((size_t)f * S.H + y) * S.W;
    const int rs = seg_rowstart(S, f, y), n = seg_rowcount(S, f, y);
    uint32_t rx0 = 0xffffu; int lab0 = 0;                    // lanes past the row's runs: xs = 65535, xe = 0
    if (lane < n) { rx0 = S.run_x[rs + lane]; lab0 = runlabel[rs + lane]; }
    const int xs0 = (int)(rx0 & 0xffffu), xe0 = (int)(rx0 >> 16);
    for (int x0 = 0; x0 < S.W; x0 += 256) {
        const int x = x0 + 4 * lane;
        i32x4_t out = {0, 0, 0, 0};
        unsigned long long hit = __ballot(lab0 != 0 && xe0 >= x0 && xs0 <= x0 + 255);
        while (hit) {
            const int l = __builtin_ctzll(hit);
            hit &= hit - 1;
            const int a = __builtin_amdgcn_readlane(xs0, l), b = __builtin_amdgcn_readlane(xe0, l), lb = __builtin_amdgcn_readlane(lab0, l);
            if (a <= x && x <= b) out.x = lb;
            if (a <= x + 1 && x + 1 <= b) out.y = lb;
            if (a <= x + 2 && x + 2 <= b) out.z = lb;
            if (a <= x + 3 && x + 3 <= b) out.w = lb;
        }
        if (n > 64 && __builtin_amdgcn_readlane(xs0, 63) <= x0 + 255) {       // (runs are sorted) the later batches, from memory
            for (int j0 = 64; j0 < n; j0 += 64) {
                uint32_t rx = 0xffffu; int lab = 0;
                if (j0 + lane < n) { rx = S.run_x[rs + j0 + lane]; lab = runlabel[rs + j0 + lane]; }
                const int xs = (int)(rx & 0xffffu), xe = (int)(rx >> 16);
                hit = __ballot(lab != 0 && xe >= x0 && xs <= x0 + 255);
                while (hit) {
                    const int l = __builtin_ctzll(hit);
                    hit &= hit - 1;
                    const int a = __builtin_amdgcn_readlane(xs, l), b = __builtin_amdgcn_readlane(xe, l), lb = __builtin_amdgcn_readlane(lab, l);
                    if (a <= x && x <= b) out.x = lb;
                    if (a <= x + 1 && x + 1 <= b) out.y = lb;
                    if (a <= x + 2 && x + 2 <= b) out.z = lb;
                    if (a <= x + 3 && x + 3 <= b) out.w = lb;
                }
                if (__builtin_amdgcn_readlane(xs, 63) > x0 + 255) break;
            }
        }
        if (x < S.W) __builtin_nontemporal_store(out, (i32x4_t *)(labels + rowpix + x));
    }
}

// exclusive scans of the three size columns; out[k][0..n].  One block of 1024 threads, or -- where the previous batch saved
// more than SCAN_MULTI_MIN objects (this batch's number is not known on the host when the chain is queued) -- two launches of
// SCAN_BLOCKS blocks: every block takes a contiguous chunk of the objects, the first launch leaves the chunks' column totals
// (and tallest object), the second scans each chunk behind the sum of the chunks before it.  A thread takes 8 consecutive objects, read and written as 16-byte vectors: the columns are stored one after the
// other (stride `szstride`, even, with 8 entries of slack behind each column).  (One block for everything took 0.125 ms for the
// 100 000 saved objects of a dense batch; 0.011 ms for 5 700, as the two launches do.)
#define SCAN_IPT 8
#define SCAN_BLOCKS 128
#define SCAN_MULTI_MIN 16384
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void scan_chunk_of(int n, int &i_begin, int &i_end)
{
    const int nb = (int)gridDim.x;
    const int chunk = (((n + nb - 1) / nb) + SCAN_IPT - 1) & ~(SCAN_IPT - 1);
    i_begin = (int)blockIdx.x * chunk; i_end = i_begin + chunk < n ? i_begin + chunk : n;
}
__device__ __forceinline__ void scan_load8(const unsigned long long *col, int i0, int n, unsigned long long *v)
{
    if (i0 + SCAN_IPT <= n) {
        const u64x2_t *p = (const u64x2_t *)(col + i0);
#pragma unroll
        for (int q = 0; q < SCAN_IPT / 2; q++) { const u64x2_t t = p[q]; v[2 * q] = t.x; v[2 * q + 1] = t.y; }
    } else
#pragma unroll
        for (int j = 0; j < SCAN_IPT; j++) v[j] = i0 + j < n ? col[i0 + j] : 0;
}
// partial[b * 4 + k]: total of column k over chunk b (k < 3), tallest object of the chunk (k = 3)
__global__ __launch_bounds__(1024) void k_scan_sizes_part(SegState S, uint32_t cap, uint32_t bcap, const unsigned long long *sz, size_t szstride,
                                                           const uint32_t *n_p, unsigned long long *partial)
{
    __shared__ unsigned long long wred[16][4];
    if (!chain_ok(S, cap, bcap)) return;
    const int n = (int)*n_p, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int i_begin, i_end;
    scan_chunk_of(n, i_begin, i_end);
    unsigned long long tot[3] = {0, 0, 0}, hmax = 0;
    for (int i0 = i_begin + (int)threadIdx.x * SCAN_IPT; i0 < i_end; i0 += 1024 * SCAN_IPT)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            unsigned long long v[SCAN_IPT];
            scan_load8(sz + k * szstride, i0, i_end, v);
#pragma unroll
            for (int j = 0; j < SCAN_IPT; j++) { tot[k] += v[j]; if (k == 1) hmax = v[j] > hmax ? v[j] : hmax; }   // column 1 = rows of the object
        }
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) tot[k] += __shfl_xor(tot[k], o, 64);
        const unsigned long long t = __shfl_xor(hmax, o, 64);
        hmax = t > hmax ? t : hmax;
    }
    if (lane == 0) { wred[wv][0] = tot[0]; wred[wv][1] = tot[1]; wred[wv][2] = tot[2]; wred[wv][3] = hmax; }
    __syncthreads();
    if (threadIdx.x < 4) {
        unsigned long long r = 0;
        for (int w = 0; w < 16; w++) r = threadIdx.x < 3 ? r + wred[w][threadIdx.x] : (wred[w][3] > r ? wred[w][3] : r);
        partial[blockIdx.x * 4 + threadIdx.x] = r;
    }
}
__global__ __launch_bounds__(1024) void k_scan_sizes(SegState S, uint32_t cap, uint32_t bcap, const unsigned long long *sz, size_t szstride,
                                                      const uint32_t *n_p, long long *crop_off, long long *bits_off,
                                                      long long *band_off, long long *totals, const unsigned long long *partial)
{
    __shared__ unsigned long long wtot[16][3];
    __shared__ unsigned long long carry[3];
    // totals[4..6]: runs of the batch, largest band, scan error flag -- what the host needs to accept or repeat the chain
    if (blockIdx.x == 0 && threadIdx.x < 3) totals[4 + threadIdx.x] = (long long)S.framebase[S.F + threadIdx.x];
    if (!chain_ok(S, cap, bcap)) { if (blockIdx.x == 0 && threadIdx.x < 4) totals[threadIdx.x] = 0; return; }
    const int n = (int)*n_p, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int i_begin, i_end;
    scan_chunk_of(n, i_begin, i_end);
    long long *outp[3] = {crop_off, bits_off, band_off};
    // what lies before this chunk (block 0 also: the grand totals and the tallest object, from all chunks).  A grid of one
    // block (the host's choice for batches with few saved objects: one launch less) finds both on its way.
    const bool alone = gridDim.x == 1;
    unsigned long long hmax1 = 0;
    if (alone) { if (threadIdx.x < 3) carry[threadIdx.x] = 0; }
    else if (wv == 0) {
        const int upto = blockIdx.x == 0 ? (int)gridDim.x : (int)blockIdx.x;
        unsigned long long pre[3] = {0, 0, 0}, hmax = 0;
        for (int b = lane; b < upto; b += 64) {
#pragma unroll
            for (int k = 0; k < 3; k++) pre[k] += partial[b * 4 + k];
            hmax = partial[b * 4 + 3] > hmax ? partial[b * 4 + 3] : hmax;
        }
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int k = 0; k < 3; k++) pre[k] += __shfl_xor(pre[k], o, 64);
            const unsigned long long t = __shfl_xor(hmax, o, 64);
            hmax = t > hmax ? t : hmax;
        }
        if (lane < 3) {
            const unsigned long long mine = lane == 0 ? pre[0] : (lane == 1 ? pre[1] : pre[2]);
            if (blockIdx.x == 0) { outp[lane][n] = (long long)mine; totals[lane] = (long long)mine; carry[lane] = 0; }
            else carry[lane] = mine;
        }
        if (lane == 3 && blockIdx.x == 0) totals[3] = (long long)hmax;       // tallest saved object (LDS size of k_convex)
    }
    __syncthreads();
    for (int base = i_begin; base < i_end; base += 1024 * SCAN_IPT) {
        const int i0 = base + threadIdx.x * SCAN_IPT;
        unsigned long long v[3][SCAN_IPT], tot[3] = {0, 0, 0}, inc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (i0 < i_end) scan_load8(sz + k * szstride, i0, i_end, v[k]);
#pragma unroll
            for (int j = 0; j < SCAN_IPT; j++) { if (i0 + j >= i_end) v[k][j] = 0; tot[k] += v[k][j]; }
        }
#pragma unroll
        for (int j = 0; j < SCAN_IPT; j++) hmax1 = v[1][j] > hmax1 ? v[1][j] : hmax1;   // column 1 = rows of the object
        for (int k = 0; k < 3; k++) {                        // inclusive scan of the thread totals inside the wave
            unsigned long long x = tot[k];
            for (int o = 1; o < 64; o <<= 1) { unsigned long long t = __shfl_up(x, o, 64); if (lane >= o) x += t; }
            inc[k] = x;
        }
        if (lane == 63) for (int k = 0; k < 3; k++) wtot[wv][k] = inc[k];
        __syncthreads();
        unsigned long long pre[3] = {carry[0], carry[1], carry[2]};
        for (int w = 0; w < wv; w++) for (int k = 0; k < 3; k++) pre[k] += wtot[w][k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            unsigned long long run = pre[k] + inc[k] - tot[k];
            if (i0 + SCAN_IPT <= i_end) {
                u64x2_t *o = (u64x2_t *)(outp[k] + i0);
#pragma unroll
                for (int q = 0; q < SCAN_IPT / 2; q++) {
                    u64x2_t t; t.x = run; run += v[k][2 * q]; t.y = run; run += v[k][2 * q + 1];
                    o[q] = t;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SCAN_IPT; j++) { if (i0 + j < i_end) outp[k][i0 + j] = (long long)run; run += v[k][j]; }
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) for (int k = 0; k < 3; k++) carry[k] = pre[k] + inc[k];
        __syncthreads();
    }
    if (alone) {
        if (threadIdx.x < 3) { outp[threadIdx.x][n] = (long long)carry[threadIdx.x]; totals[threadIdx.x] = (long long)carry[threadIdx.x]; }
        for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(hmax1, o, 64); hmax1 = t > hmax1 ? t : hmax1; }
        __syncthreads();
        if (lane == 0) wtot[wv][0] = hmax1;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 16; w++) hmax1 = wtot[w][0] > hmax1 ? wtot[w][0] : hmax1;
            totals[3] = (long long)hmax1;
        }
    }
}

// band -> saved object (one wave per object): every band gets a 64-byte descriptor -- the object's geometry, its index, the
// band's first row, the object's offsets into the crop planes and the row-extremes table -- so that a band's block starts
// with ONE (scalar) load instead of the chain band -> object -> geometry / offsets.
//   desc[0..7] = f, x, y, w, h, root, words per row, object ; [8] = first row ; [10..11] = crop offset ; [12..13] = row offset
#define BAND_DESC_INTS 16
__global__ __launch_bounds__(256) void k_band_map(int n_saved, const long long *__restrict__ band_off, const int32_t *__restrict__ geo,
                                                  const long long *__restrict__ crop_off, const long long *__restrict__ row_off,
                                                  int32_t *__restrict__ band_desc, long long *__restrict__ sums)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= n_saved) return;
    if (lane < PSEG_NSUMS) sums[(size_t)s * PSEG_NSUMS + lane] = lane == 6 ? 255 : 0;   // min intensity starts at 255
    const int4 g0 = *(const int4 *)(geo + 8 * (size_t)s), g1 = *(const int4 *)(geo + 8 * (size_t)s + 4);
    // descriptors of the small class first (one per small window, in object order), then the bands of the other objects:
    // small windows before s = high part of the scanned column, bands before s = low part
    const unsigned long long p0 = (unsigned long long)band_off[s], p1 = (unsigned long long)band_off[s + 1];
    const long long n_small = (long long)((unsigned long long)band_off[n_saved] >> 40);
    const long long sm0 = (long long)(p0 >> 40), nb = (long long)((p1 & (CROP_SMALL_ONE - 1)) - (p0 & (CROP_SMALL_ONE - 1)));
    const bool small = (p1 >> 40) != (p0 >> 40);
    const long long b0 = small ? sm0 : n_small + (long long)(p0 & (CROP_SMALL_ONE - 1)) - sm0, b1 = b0 + nb;
    const long long co = crop_off[s], ro = row_off[s];
    const int rpb = crop_rows_per_band(g0.w);
    for (long long b = b0 + lane; b < b1; b += 64) {
        int4 *d = (int4 *)(band_desc + BAND_DESC_INTS * (size_t)b);
        d[0] = g0;
        d[1] = make_int4(g1.x, g1.y, g1.z, s);
        d[2] = make_int4((int)(b - b0) * rpb, 0, (int)(uint32_t)co, (int)(co >> 32));
        d[3] = make_int4((int)(uint32_t)ro, (int)(ro >> 32), 0, 0);
    }
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Three crops + integer sums of one row band of a saved object (block per band; bands of ~4096 px keep
// every block short, large objects are spread over many CUs).
//  phase 1: the band's cumulative filled mask as bits in LDS, one thread per 32-pixel word: a pixel is set iff
//           it lies in the hole-filled region of a SAVED object iterated at or before this one (root index >=
//           own root) -- detection.py:206,245-246, the frame mask accumulates over the object loop;
//  phase 2: crops (detection.py:243-247) and the sums, combined across bands with integer atomics (exact).
// sums[s*PSEG_NSUMS + k]: 0 N, 1 sum r, 2 sum c, 3 sum r^2, 4 sum c^2, 5 sum r*c, 6 min I, 7 max I, 8 sum I,
//   9 filled area, 10..12 perimeter code counts (weight 1, sqrt2, (1+sqrt2)/2), 13 euler, 14 convex area
// p / d and p % d for small non-negative p (< 2^22) with the reciprocal in single precision and one correction step: the
// integer division of the compiler is some 25 instructions, several of them quarter-rate 32-bit multiplies
__device__ __forceinline__ void divmod_small(int p, int d, float rcp, int &q, int &r)
{
    q = (int)((float)p * rcp);
    r = p - (int)__umul24((uint32_t)q, (uint32_t)d);
    if (r < 0) { q--; r += d; } else if (r >= d) { q++; r -= d; }
}
// Size classes: THREADS / TPR (threads that share a row while painting) / LW (words of each LDS array).  <256, 4, LW>
// serves every band; <64, 2, CROP_SMALL_WORDS> serves the single-band windows of small objects (crop_band_is_small: up to 64
// columns and 124 rows), of which a CU then holds 32 at a time instead of 8 -- on frames of 2000 objects the kernel is a chain of
// latencies per object, and what shortens it is the number of objects in flight.  The fused path launches one grid per class
// over its own stretch of the descriptors (k_band_map puts the small windows first; grids over all bands with the other
// class's blocks returning at once cost 0.15 ms per 100 000 bands).  cls: 0 = every band, 1 = small windows, 2 = the others
// (a guard only: a block whose band is of the other class returns).
template <bool DENSE, int THREADS, int TPR, int LW>
__global__ __launch_bounds__(THREADS) void k_crops_sums(SegState S, int n_saved, const int32_t *__restrict__ geo,
                                                    const long long *__restrict__ crop_off, const long long *__restrict__ band_off,
                                                    const int32_t *__restrict__ objflag, const uint8_t *__restrict__ cleaned,
                                                    const uint8_t *__restrict__ raw, uint8_t *__restrict__ c_raw,
                                                    uint8_t *__restrict__ c_masked, uint8_t *__restrict__ c_mask, long long *sums,
                                                    const long long *__restrict__ row_off, int32_t *__restrict__ rowext,
                                                    const int32_t *__restrict__ band_obj, const uint8_t *__restrict__ dense_mask,
                                                    const int32_t *__restrict__ band_desc, int cls)
{
    __shared__ uint32_t mbuf[LW];             // mask bits of rows r0-2 .. r1+1 (zero outside the window)
    __shared__ uint32_t bd[LW];               // 4-neighbour border bits of rows r0-1 .. r1
    __shared__ uint32_t acc[32];                          // block totals (see the end of the kernel)
    const long long bid = blockIdx.x;
    int s, f, bx, by, w, h, root, wpr, r0;
    long long coff, roff;
    if (band_desc) {                                      // (k_band_map's descriptor: block-uniform, one load)
        const int32_t *d = band_desc + BAND_DESC_INTS * (size_t)bid;
        f = d[0]; bx = d[1]; by = d[2]; w = d[3]; h = d[4]; root = d[5]; wpr = d[6]; s = d[7]; r0 = d[8];
        coff = (long long)(((unsigned long long)(uint32_t)d[11] << 32) | (uint32_t)d[10]);
        roff = (long long)(((unsigned long long)(uint32_t)d[13] << 32) | (uint32_t)d[12]);
    } else {                                              // pseg_regionprops: band -> object -> geometry
        s = band_obj[bid];                                // band_off[s] <= blockIdx.x < band_off[s+1]
        const int32_t *o = geo + 8 * (size_t)s;
        f = o[0]; bx = o[1]; by = o[2]; w = o[3]; h = o[4]; root = o[5]; wpr = o[6];
        r0 = (int)(bid - band_off[s]) * crop_rows_per_band(w);
        coff = crop_off[s]; roff = row_off[s];
    }
    if (cls != 0 && (cls == 1) != crop_band_is_small(w, h)) return;       // (block-uniform)
    const int rpb = crop_rows_per_band(w);
    const int r1 = (r0 + rpb < h) ? r0 + rpb : h;
    const int nrows = r1 - r0;
    const float rcp_wpr = 1.0f / (float)wpr, rcp_wpr1 = 1.0f / (float)(wpr + 1);
    for (int wi = threadIdx.x; wi < (nrows + 4) * wpr; wi += THREADS) mbuf[wi] = 0;
    if (threadIdx.x < 32) acc[threadIdx.x] = threadIdx.x == 15 ? 255u : 0u;    // acc[15] = running minimum
    __syncthreads();
    if (CROP_EXP & 2) {
    } else if (DENSE) {
        // pseg_regionprops: the mask is a dense uint8 image (nonzero = set) of row stride S.W instead of the run tables
        for (int wi = threadIdx.x; wi < (nrows + 4) * wpr; wi += THREADS) {
            const int yl = wi / wpr, wx = wi - yl * wpr, yy = r0 - 2 + yl;
            uint32_t v = 0;
            if (yy >= 0 && yy < h) {
                const uint8_t *mrow8 = dense_mask + (size_t)(by + yy) * S.W + bx + 32 * wx;
                const int nb = w - 32 * wx < 32 ? w - 32 * wx : 32;
                for (int k = 0; k < nb; k++) v |= (uint32_t)(mrow8[k] != 0) << k;
            }
            mbuf[wi] = v;
        }
    } else {
        // TPR threads per row walk the row's runs; a run paints itself and the gap on its left when the object that fills
        // them (ext / gapl) is saved and was iterated at or before this one.  Two dependent global loads per row.
        const int sub = threadIdx.x & (TPR - 1), x_lo = bx, x_hi = bx + w - 1;
        auto paint = [&](uint32_t *mrow, int xa, int xb) {
            int la = (xa > x_lo ? xa : x_lo) - bx, lb = (xb < x_hi ? xb : x_hi) - bx;
            for (int wd = la >> 5; wd <= (lb >> 5); wd++) {
                int ba = la > 32 * wd ? la - 32 * wd : 0, bb = lb < 32 * wd + 31 ? lb - 32 * wd : 31;
                uint32_t m = (bb - ba == 31) ? 0xffffffffu : (((1u << (bb - ba + 1)) - 1u) << ba);
                atomicOr(&mrow[wd], m);
            }
        };
        for (int yl = threadIdx.x / TPR; yl < nrows + 4; yl += THREADS / TPR) {
            const int yy = r0 - 2 + yl;
            if (yy < 0 || yy >= h) continue;
            const int rs = seg_rowstart(S, f, by + yy), n = seg_rowcount(S, f, by + yy);
            uint32_t *mrow = mbuf + yl * wpr;
            // A thread takes CROP_RPT runs of the row per round: their extents (and their left neighbours') are requested
            // together -- one memory latency per round -- and only the runs that meet the window (one or two of a dense
            // frame's twenty per row) go on to the dependent loads: object of the run / of the gap on its left, saved flag.
            // (One run per step with all its loads took 0.39 of the kernel's 0.75 ms on frames of 2000 objects.)
            for (int jb = 0; jb < n; jb += CROP_RPT * TPR) {
                uint32_t rx[CROP_RPT], px[CROP_RPT];
#pragma unroll
                for (int k = 0; k < CROP_RPT; k++) {
                    const int j = jb + k * TPR + sub;
                    rx[k] = j < n ? S.run_x[rs + j] : 0xffffu;                   // past the row's runs: xs = 65535, xe = 0
                    px[k] = (j > 0 && j < n) ? S.run_x[rs + j - 1] : ((uint32_t)x_hi << 16);   // j == 0: the leading gap belongs to no object
                }
                int xs_last = 0;
#pragma unroll
                for (int k = 0; k < CROP_RPT; k++) {
                    const int xs = (int)(rx[k] & 0xffffu), xe = (int)(rx[k] >> 16), pe = (int)(px[k] >> 16);
                    const bool run_in = xe >= x_lo && xs <= x_hi, gap_in = xs - 1 >= x_lo && pe + 1 <= x_hi;
                    xs_last = xs;
                    if (run_in || gap_in) {
                        const int r = rs + jb + k * TPR + sub;
                        const int idr = S.ext[r], idg = S.gapl[r];
                        // (the object itself is saved by definition: the flag is only looked up for other objects inside the window)
                        if (run_in && (idr == root || (idr > root && (objflag[idr] & 2)))) paint(mrow, xs, xe);
                        if (gap_in && (idg == root || (idg > root && (objflag[idg] & 2)))) paint(mrow, pe + 1, xs - 1);
                    }
                }
                if (xs_last > x_hi) break;                   // (runs are sorted) neither the later runs nor the gaps on their left meet the window
            }
        }
    }
    __syncthreads();
    // border = mask minus its erosion by the 4-neighbour cross, zero outside the window
    // (skimage perimeter(image, 4): binary_erosion(..., border_value=0), _regionprops_utils.py:225-231)
    for (int wi = threadIdx.x; wi < (nrows + 2) * wpr; wi += THREADS) {
        int yl, wx;                                        // row r0-1+yl  ->  mbuf row yl+1
        divmod_small(wi, wpr, rcp_wpr, yl, wx);
        const uint32_t *m = mbuf + __umul24((uint32_t)(yl + 1), (uint32_t)wpr);
        uint32_t c = m[wx], up = m[wx - wpr], dn = m[wx + wpr];
        uint32_t lf = (c << 1) | (wx > 0 ? m[wx - 1] >> 31 : 0u);
        uint32_t rt = (c >> 1) | (wx + 1 < wpr ? m[wx + 1] << 31 : 0u);
        bd[wi] = c & ~(up & dn & lf & rt);
    }
    __syncthreads();
    const uint32_t *mb = mbuf + 2 * wpr;                  // row r0 of the band
    const size_t P = (size_t)S.H * S.W;
    const bool same = (raw == cleaned);                   // fused path: the Crops/ plane is cut from the same image
    // block-uniform bases; per-item offsets stay 32-bit
    const uint8_t *sb = cleaned + (size_t)f * P + (size_t)(by + r0) * S.W + bx;
    const uint8_t *rb = raw + (size_t)f * P + (size_t)(by + r0) * S.W + bx;
    const long long off = coff + (long long)r0 * w;
    uint8_t *o_raw = c_raw + off, *o_mask = c_mask + off, *o_masked = c_masked + off;
    uint32_t isum = 0, mn_e = 0x00ff00ffu, mn_o = 0x00ff00ffu, mx_e = 0, mx_o = 0;   // intensities: byte lanes as u16 pairs
    if (CROP_EXP & 4) {
    } else if (w >= 16) {
        // thread = 16 consecutive pixels of a row (the 8-pixel form below with 128-bit accesses)
        const int cpr = (w + 15) >> 4, nitem = nrows * cpr;
        const float rcp = 1.0f / (float)cpr;
#pragma unroll CROP_UNROLL
        for (int p = threadIdx.x; p < nitem; p += THREADS) {
            int yl = (int)((float)p * rcp);
            int c = p - (int)__umul24((uint32_t)yl, (uint32_t)cpr);
            if (c < 0) { yl--; c += cpr; } else if (c >= cpr) { yl++; c -= cpr; }
            int xx = 16 * c;
            const int over = xx - (w - 16);               // > 0 only for the last chunk of a row when w % 16 != 0
            if (over > 0) xx = w - 16;
            const uint32_t *mw = mb + __umul24((uint32_t)yl, (uint32_t)wpr) + (xx >> 5);
            const uint32_t bits16 = __builtin_amdgcn_alignbit(mw[1], mw[0], (uint32_t)(xx & 31)) & 0xffffu;
            const uint32_t so = __umul24((uint32_t)yl, (uint32_t)S.W) + (uint32_t)xx, qo = __umul24((uint32_t)yl, (uint32_t)w) + (uint32_t)xx;
            uint32_t cv[4], rv[4], m[4], cm[4];
            __builtin_memcpy(cv, sb + so, 16);
            if (same) { rv[0] = cv[0]; rv[1] = cv[1]; rv[2] = cv[2]; rv[3] = cv[3]; } else __builtin_memcpy(rv, rb + so, 16);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t m1 = __umul24((bits16 >> (4 * d)) & 0xfu, 0x00204081u) & 0x01010101u;
                m[d] = (m1 << 8) - m1;
                cm[d] = (cv[d] & m[d]) | ~m[d];
            }
            __builtin_memcpy(o_raw + qo, rv, 16);
            __builtin_memcpy(o_mask + qo, m, 16);
            __builtin_memcpy(o_masked + qo, cm, 16);
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const int nb = over - 4 * d;              // bytes of this dword that belong to the neighbour chunk
                const uint32_t keep = nb <= 0 ? 0xffffffffu : (nb >= 4 ? 0u : 0xffffffffu << (8 * nb));
                const uint32_t mk = m[d] & keep, in4 = cv[d] & mk, out4 = cv[d] | ~mk;
                isum = __builtin_amdgcn_sad_u8(in4, 0u, isum);
                mx_e = pkmax(mx_e, __builtin_amdgcn_perm(0, in4, 0x0c020c00)); mx_o = pkmax(mx_o, __builtin_amdgcn_perm(0, in4, 0x0c030c01));
                mn_e = pkmin(mn_e, __builtin_amdgcn_perm(0, out4, 0x0c020c00)); mn_o = pkmin(mn_o, __builtin_amdgcn_perm(0, out4, 0x0c030c01));
            }
        }
    } else if (w >= 8) {
        // thread = 8 consecutive pixels of a row: one (unaligned) 64-bit load / store per plane; the last chunk of a row is
        // shifted back to end at the row end, `keep` drops the bytes it shares with its neighbour from the sums (as below;
        // half the index arithmetic and half the memory instructions per pixel of the 4-pixel form)
        const int cpr = (w + 7) >> 3, nitem = nrows * cpr;
        const float rcp = 1.0f / (float)cpr;
#pragma unroll CROP_UNROLL
        for (int p = threadIdx.x; p < nitem; p += THREADS) {
            int yl = (int)((float)p * rcp);
            int c = p - (int)__umul24((uint32_t)yl, (uint32_t)cpr);
            if (c < 0) { yl--; c += cpr; } else if (c >= cpr) { yl++; c -= cpr; }
            int xx = 8 * c;
            const int over = xx - (w - 8);                // > 0 only for the last chunk of a row when w % 8 != 0
            uint32_t keep_lo = 0xffffffffu, keep_hi = 0xffffffffu;
            if (over > 0) {
                xx = w - 8;
                if (over >= 4) { keep_lo = 0; keep_hi = over > 4 ? 0xffffffffu << (8 * (over - 4)) : 0xffffffffu; }
                else keep_lo <<= 8 * over;
            }
            const uint32_t *mw = mb + __umul24((uint32_t)yl, (uint32_t)wpr) + (xx >> 5);
            const uint32_t bits8 = __builtin_amdgcn_alignbit(mw[1], mw[0], (uint32_t)(xx & 31)) & 0xffu;
            const uint32_t so = __umul24((uint32_t)yl, (uint32_t)S.W) + (uint32_t)xx, qo = __umul24((uint32_t)yl, (uint32_t)w) + (uint32_t)xx;
            const uint32_t m1a = __umul24(bits8 & 0xfu, 0x00204081u) & 0x01010101u, m1b = __umul24(bits8 >> 4, 0x00204081u) & 0x01010101u;
            uint2 m8 = make_uint2((m1a << 8) - m1a, (m1b << 8) - m1b);
            uint2 cv8, rv8;
            __builtin_memcpy(&cv8, sb + so, 8);
            if (same) rv8 = cv8; else __builtin_memcpy(&rv8, rb + so, 8);
            const uint2 cm8 = make_uint2((cv8.x & m8.x) | ~m8.x, (cv8.y & m8.y) | ~m8.y);
            __builtin_memcpy(o_raw + qo, &rv8, 8);
            __builtin_memcpy(o_mask + qo, &m8, 8);
            __builtin_memcpy(o_masked + qo, &cm8, 8);
            const uint32_t mka = m8.x & keep_lo, mkb = m8.y & keep_hi;
            const uint32_t ina = cv8.x & mka, inb = cv8.y & mkb, outa = cv8.x | ~mka, outb = cv8.y | ~mkb;
            isum = __builtin_amdgcn_sad_u8(ina, 0u, __builtin_amdgcn_sad_u8(inb, 0u, isum));
            mx_e = pkmax(mx_e, pkmax(__builtin_amdgcn_perm(0, ina, 0x0c020c00), __builtin_amdgcn_perm(0, inb, 0x0c020c00)));
            mx_o = pkmax(mx_o, pkmax(__builtin_amdgcn_perm(0, ina, 0x0c030c01), __builtin_amdgcn_perm(0, inb, 0x0c030c01)));
            mn_e = pkmin(mn_e, pkmin(__builtin_amdgcn_perm(0, outa, 0x0c020c00), __builtin_amdgcn_perm(0, outb, 0x0c020c00)));
            mn_o = pkmin(mn_o, pkmin(__builtin_amdgcn_perm(0, outa, 0x0c030c01), __builtin_amdgcn_perm(0, outb, 0x0c030c01)));
        }
    } else if (w >= 4) {
        // thread = 4 consecutive pixels of a row: one (unaligned) 32-bit load / store per plane.  The last chunk of a row is
        // shifted back to end at the row end (it rewrites bytes of its neighbour with the same values); `keep` drops the
        // bytes it shares with the neighbour from the sums.
        const int cpr = (w + 3) >> 2, nitem = nrows * cpr;
        const float rcp = 1.0f / (float)cpr;
#pragma unroll CROP_UNROLL
        for (int p = threadIdx.x; p < nitem; p += THREADS) {
            // (products of values below 2^24 go through v_mul_u32_u24: the full 32-bit multiply runs at a quarter of the rate)
            int yl = (int)((float)p * rcp);               // p / cpr, corrected below (p < 2^24)
            int c = p - (int)__umul24((uint32_t)yl, (uint32_t)cpr);
            if (c < 0) { yl--; c += cpr; } else if (c >= cpr) { yl++; c -= cpr; }
            int xx = 4 * c;
            const int over = xx - (w - 4);                // > 0 only for the last chunk of a row when w % 4 != 0
            uint32_t keep = 0xffffffffu;
            if (over > 0) { xx = w - 4; keep <<= 8 * over; }
            const uint32_t *mw = mb + __umul24((uint32_t)yl, (uint32_t)wpr) + (xx >> 5);
            const uint32_t bits4 = __builtin_amdgcn_alignbit(mw[1], mw[0], (uint32_t)(xx & 31)) & 0xfu;
            const uint32_t so = __umul24((uint32_t)yl, (uint32_t)S.W) + (uint32_t)xx, qo = __umul24((uint32_t)yl, (uint32_t)w) + (uint32_t)xx;
            const uint32_t m1 = __umul24(bits4, 0x00204081u) & 0x01010101u;       // bit k -> bit 0 of byte k
            const uint32_t m4 = (m1 << 8) - m1;                                   // ... -> byte k = 255
            uint32_t cv4, rv4;
            __builtin_memcpy(&cv4, sb + so, 4);
            if (same) rv4 = cv4; else __builtin_memcpy(&rv4, rb + so, 4);
            const uint32_t cm4 = (cv4 & m4) | ~m4;        // where(mask, crop, 255)
            __builtin_memcpy(o_raw + qo, &rv4, 4);
            __builtin_memcpy(o_mask + qo, &m4, 4);
            __builtin_memcpy(o_masked + qo, &cm4, 4);
            const uint32_t mk = m4 & keep, in4 = cv4 & mk, out4 = cv4 | ~mk;   // masked-out bytes: 0 for sum / max, 255 for min
            isum = __builtin_amdgcn_sad_u8(in4, 0u, isum);
            mx_e = pkmax(mx_e, __builtin_amdgcn_perm(0, in4, 0x0c020c00)); mx_o = pkmax(mx_o, __builtin_amdgcn_perm(0, in4, 0x0c030c01));
            mn_e = pkmin(mn_e, __builtin_amdgcn_perm(0, out4, 0x0c020c00)); mn_o = pkmin(mn_o, __builtin_amdgcn_perm(0, out4, 0x0c030c01));
        }
    } else {                                              // windows narrower than a dword: thread per pixel
        for (int p = threadIdx.x; p < nrows * w; p += THREADS) {
            const int yl = p / w, xx = p - yl * w;
            const uint32_t bit = (mb[yl * wpr] >> xx) & 1u;
            const uint32_t cv = sb[(size_t)yl * S.W + xx];
            o_raw[p] = same ? (uint8_t)cv : rb[(size_t)yl * S.W + xx];
            o_mask[p] = bit ? 255 : 0;
            o_masked[p] = bit ? (uint8_t)cv : 255;
            if (bit) { isum += cv; mx_e = pkmax(mx_e, cv); mn_e = pkmin(mn_e, cv | 0x00ff0000u); }
        }
    }
    uint32_t imin, imax;
    {
        uint32_t mn = pkmin(mn_e, mn_o), mx = pkmax(mx_e, mx_o);
        imin = (mn & 0xffffu) < (mn >> 16) ? (mn & 0xffffu) : (mn >> 16);
        imax = (mx & 0xffffu) > (mx >> 16) ? (mx & 0xffffu) : (mx >> 16);
    }
    // One pass over the band's words: geometric moments and perimeter codes, both bit-parallel.
    //  moments: for a word with bits at columns 32*wx + k,  sum k = sum_b 2^b popc(v & M_b),
    //           sum k^2 = sum_b 4^b popc(v & M_b) + 2 sum_{b<c} 2^(b+c) popc(v & M_b & M_c); rows are band-local (yl), the
    //           band offset r0 is folded in once per block, so every partial sum of the block fits 32 bits
    //  perimeter codes (skimage perimeter, _regionprops_utils.py:233-249): for a border pixel the 3x3 sum with weights
    //           [[10,2,10],[2,1,2],[10,2,10]] over the border image is 1 + 2*n4 + 10*nd; n4 / nd are counted with bit-sliced adders
    uint32_t mo[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // n, wx n, wx^2 n, sk, wx sk, sk2, yl n, yl^2 n, yl wx n, yl sk
    uint32_t pc1 = 0, pc2 = 0, pc3 = 0;
    for (int wi = threadIdx.x; wi < ((CROP_EXP & 8) ? 0 : nrows * wpr); wi += THREADS) {
        int yl_, wx_;
        divmod_small(wi, wpr, rcp_wpr, yl_, wx_);
        const uint32_t yl = (uint32_t)yl_, wx = (uint32_t)wx_;
        const uint32_t v = mb[wi];
        if (v) {
            const uint32_t M[5] = {0xaaaaaaaau, 0xccccccccu, 0xf0f0f0f0u, 0xff00ff00u, 0xffff0000u};
            uint32_t n1 = (uint32_t)__popc(v), sk = 0, sk2 = 0;
#pragma unroll
            for (int b2 = 0; b2 < 5; b2++) {
                uint32_t pb = (uint32_t)__popc(v & M[b2]);
                sk += pb << b2; sk2 += pb << (2 * b2);
#pragma unroll
                for (int c2 = b2 + 1; c2 < 5; c2++) sk2 += (uint32_t)__popc(v & M[b2] & M[c2]) << (b2 + c2 + 1);
            }
            // (all factors are far below 2^24: 24-bit multiplies)
            const uint32_t wn = __umul24(wx, n1), yn = __umul24(yl, n1);
            mo[0] += n1; mo[1] += wn; mo[2] += __umul24(wx, wn); mo[3] += sk; mo[4] += __umul24(wx, sk); mo[5] += sk2;
            mo[6] += yn; mo[7] += __umul24(yl, yn); mo[8] += __umul24(yl, wn); mo[9] += __umul24(yl, sk);
        }
        const uint32_t *bc = bd + __umul24(yl + 1, (uint32_t)wpr) + wx;     // border row r0+yl
        const uint32_t C = bc[0];
        if (C) {
            const bool hl = wx > 0, hr = wx + 1 < (uint32_t)wpr;
            const uint32_t U = bc[-wpr], D = bc[wpr];
            const uint32_t cl1 = hl ? bc[-1] >> 31 : 0u, cr1 = hr ? bc[1] << 31 : 0u;
            const uint32_t ul1 = hl ? bc[-wpr - 1] >> 31 : 0u, ur1 = hr ? bc[-wpr + 1] << 31 : 0u;
            const uint32_t dl1 = hl ? bc[wpr - 1] >> 31 : 0u, dr1 = hr ? bc[wpr + 1] << 31 : 0u;
            const uint32_t L = (C << 1) | cl1, R = (C >> 1) | cr1;
            const uint32_t UL = (U << 1) | ul1, UR = (U >> 1) | ur1, DL = (D << 1) | dl1, DR = (D >> 1) | dr1;
            // a + b + c + d per bit position -> three bit planes
            auto add4 = [](uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t &s0, uint32_t &s1, uint32_t &s2) {
                uint32_t h1 = a ^ b, c1 = a & b, h2 = c ^ d, c2 = c & d, t = h1 & h2;
                s0 = h1 ^ h2; s1 = c1 ^ c2 ^ t; s2 = c1 & c2;
            };
            uint32_t a0, a1, a2, d0, d1, d2;
            add4(L, R, U, D, a0, a1, a2);
            add4(UL, UR, DL, DR, d0, d1, d2);
            const uint32_t n4_1 = a0 & ~a1 & ~a2, n4_0 = ~(a0 | a1 | a2);
            pc1 += (uint32_t)__popc(C & a1 & ~(d2 | (d1 & d0)));                       // n4 in {2,3}, nd <= 2: codes 5,7,15,17,25,27 -> 1
            pc2 += (uint32_t)__popc(C & ((n4_0 & ~d0 & d1) | (n4_1 & d0 & d1)));       // codes 21,33 -> sqrt(2)
            pc3 += (uint32_t)__popc(C & n4_1 & (d0 ^ d1) & ~d2);                       // codes 13,23 -> (1+sqrt(2))/2
        }
    }
    // Euler number, 8-connectivity (skimage euler_number, _regionprops_utils.py:140-181): 2x2 quads of the
    // zero-padded window; code 8 = only top-left set, 6 = top-right + bottom-left, 14 = all but bottom-right
    int eul = 0;
    {
        const int ylast = (r1 == h) ? nrows + 1 : nrows;   // the last band also owns the quads below the window
        for (int wi = threadIdx.x; wi < ((CROP_EXP & 8) ? 0 : ylast * (wpr + 1)); wi += THREADS) {
            int yl, wx;
            divmod_small(wi, wpr + 1, rcp_wpr1, yl, wx);
            const uint32_t *B = mbuf + __umul24((uint32_t)(yl + 2), (uint32_t)wpr), *T = B - wpr;     // bottom row r0+yl, top row r0+yl-1
            uint32_t tr = wx < wpr ? T[wx] : 0u, br = wx < wpr ? B[wx] : 0u;
            uint32_t tl = (tr << 1) | (wx > 0 ? T[wx - 1] >> 31 : 0u), bl = (br << 1) | (wx > 0 ? B[wx - 1] >> 31 : 0u);
            eul += __popc(tl & ~tr & ~bl & ~br) - __popc(~tl & tr & bl & ~br) - __popc(tl & tr & bl & ~br);
        }
    }
    // first / last set column of every row of the band (input of the convex hull kernel)
    for (int yl = threadIdx.x; yl < nrows; yl += THREADS) {
        const uint32_t *m = mb + yl * wpr;
        int first = 0x7fffffff, last = -1;
        for (int wx = 0; wx < wpr; wx++) if (m[wx]) { first = 32 * wx + __ffs(m[wx]) - 1; break; }
        for (int wx = wpr - 1; wx >= 0; wx--) if (m[wx]) { last = 32 * wx + 31 - __clz(m[wx]); break; }
        rowext[2 * (roff + r0 + yl)] = first;
        rowext[2 * (roff + r0 + yl) + 1] = last;
    }
    // block totals: DPP wave sums (every partial fits 32 bits, see above), lane 63 adds them into LDS, one thread flushes
    // with 64-bit integer atomics (exact, order-independent)
#pragma unroll
    for (int k = 0; k < 10; k++) mo[k] = wave_sum63(mo[k]);
    isum = wave_sum63(isum); pc1 = wave_sum63(pc1); pc2 = wave_sum63(pc2); pc3 = wave_sum63(pc3);
    const uint32_t eulw = wave_sum63((uint32_t)eul);
    for (int o2 = 32; o2 > 0; o2 >>= 1) {
        uint32_t t0 = __shfl_xor(imin, o2, 64), t1 = __shfl_xor(imax, o2, 64);
        imin = t0 < imin ? t0 : imin; imax = t1 > imax ? t1 : imax;
    }
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int k = 0; k < 10; k++) if (mo[k]) atomicAdd(&acc[k], mo[k]);
        if (isum) atomicAdd(&acc[10], isum);
        if (pc1) atomicAdd(&acc[11], pc1);
        if (pc2) atomicAdd(&acc[12], pc2);
        if (pc3) atomicAdd(&acc[13], pc3);
        if (eulw) atomicAdd(&acc[14], eulw);
        atomicMin(&acc[15], imin); atomicMax(&acc[16], imax);
    }
    __syncthreads();
    if (threadIdx.x == 0 && !(CROP_EXP & 1)) {
        unsigned long long *q = (unsigned long long *)(sums + (size_t)s * PSEG_NSUMS);
        const unsigned long long N = acc[0], R0 = (unsigned long long)r0;
        if (r0 == 0 && r1 == h) {
            // the window's only band: plain stores (k_band_map has set the start values; q[9] and q[14] belong to others)
            q[0] = N; q[1] = acc[6]; q[2] = 32ull * acc[1] + acc[3]; q[3] = acc[7];
            q[4] = 1024ull * acc[2] + 64ull * acc[4] + acc[5]; q[5] = 32ull * acc[8] + acc[9];
            q[6] = acc[15]; q[7] = acc[16]; q[8] = acc[10];
            q[10] = acc[11]; q[11] = acc[12]; q[12] = acc[13]; q[13] = (unsigned long long)(long long)(int)acc[14];
        } else if (N) {
            const unsigned long long sx = 32ull * acc[1] + acc[3];                                    // sum c
            const unsigned long long sx2 = 1024ull * acc[2] + 64ull * acc[4] + acc[5];                // sum c^2
            const unsigned long long sy = acc[6], sy2 = acc[7], syx = 32ull * acc[8] + acc[9];        // band-local rows
            atomicAdd(&q[0], N); atomicAdd(&q[1], R0 * N + sy); atomicAdd(&q[2], sx);
            atomicAdd(&q[3], R0 * R0 * N + 2ull * R0 * sy + sy2); atomicAdd(&q[4], sx2); atomicAdd(&q[5], R0 * sx + syx);
        }
        if (!(r0 == 0 && r1 == h)) {
            if (acc[10]) atomicAdd(&q[8], (unsigned long long)acc[10]);
            if (acc[15] < 255u) atomicMin((long long *)&q[6], (long long)acc[15]);
            if (acc[16] > 0u) atomicMax((long long *)&q[7], (long long)acc[16]);
            if (acc[11]) atomicAdd(&q[10], (unsigned long long)acc[11]);
            if (acc[12]) atomicAdd(&q[11], (unsigned long long)acc[12]);
            if (acc[13]) atomicAdd(&q[12], (unsigned long long)acc[13]);
            if (acc[14]) atomicAdd(&q[13], (unsigned long long)(long long)(int)acc[14]);
        }
    }
}

// Convex hull area (skimage convex_hull_image, morphology/convex_hull.py:73-145): hull of the +-0.5 "diamond"
// points of the row-extremal pixels, then every pixel centre inside or on the hull.  In doubled coordinates
// R' = 2*row + 1 (+-1), X = 2*col (+-1) the hull's left chain is the lower convex envelope of the profile L(R') and
// the right chain the upper envelope of U(R'):  odd R' = pixel row r (points (2r, 2c -+ 1)), even R' = between rows
// r-1 and r ((2r-1, 2c) of either row).  Wave 0 builds the left chain, wave 1 the right chain, by divide and conquer:
// every lane takes a contiguous piece of the profile and builds its strict hull with a monotone chain (doubly linked
// list in LDS); six rounds then merge neighbouring hulls pairwise -- one lane per pair walks the two tangent points
// outwards from the junction (cost = vertices removed + 1).  Lane 0 reads the final list into a vertex array, and
// every pixel row finds its hull edge by binary search and gets its column bound with one division.  Exact 32-bit
// cross products: |dX| < 2^15, |dR'| < 2^14 for the heights the LDS admits; collinear points are dropped.
// Dynamic LDS: 56 bytes per row of the tallest object served by the launch (the host launches one size class per
// call so that small objects keep a high occupancy).
#define HULL_LDS_PER_ROW 28
#ifndef HULL_EXP
#define HULL_EXP 0                   // experiments only (wrong results): 1 no chains at all, 2 no merge rounds, 4 no vertex array / row bounds
#endif
typedef short hull_t;                // profile values (|X| <= 2 * 8191 + 1, sentinels +-20000) and list links (< 2 * 2700 + 2) fit 16 bits
template <bool LOWER>
__device__ __forceinline__ bool hull_drop(int ra, int va, int rb, int vb, int rc, int vc)   // b is no strict vertex between a and c
{
    const int d = (rb - ra) * (vc - va) - (vb - va) * (rc - ra);
    return LOWER ? d <= 0 : d >= 0;
}
__device__ __forceinline__ void hull_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <bool LOWER>
__device__ __forceinline__ void hull_chain(const hull_t *pv, int np, volatile hull_t *nxt, volatile hull_t *prv, hull_t *bound, int h, int lane)
{
    const int seg = (np + 63) >> 6;
    {   // strict hull of the lane's piece [s0, s1)
        const int s0 = lane * seg, s1 = s0 + seg < np ? s0 + seg : np;
        if (s0 < np) {
            int top = s0, vtop = pv[s0];
            prv[s0] = -1;
            for (int c = s0 + 1; c < s1; c++) {
                const int vc = pv[c];
                for (;;) {
                    const int p = prv[top];
                    if (p < 0 || !hull_drop<LOWER>(p, pv[p], top, vtop, c, vc)) break;
                    top = p; vtop = pv[p];
                }
                prv[c] = (hull_t)top; nxt[top] = (hull_t)c; top = c; vtop = vc;
            }
            nxt[top] = -1;
        }
    }
    for (int s = 1; s < ((HULL_EXP & 2) ? 0 : 64); s <<= 1) {   // merge the hulls of lanes [l, l+s) and [l+s, l+2s)
        hull_wave_sync();
        const int vstart = (lane + s) * seg;
        if ((lane & (2 * s - 1)) == 0 && vstart < np) {
            int v = vstart, u = vstart - 1, vu = pv[u], vv = pv[v];
            for (;;) {
                bool moved = false;
                int pu = prv[u];
                while (pu >= 0 && hull_drop<LOWER>(pu, pv[pu], u, vu, v, vv)) { u = pu; vu = pv[u]; pu = prv[u]; moved = true; }
                int nv = nxt[v];
                while (nv >= 0 && hull_drop<LOWER>(u, vu, v, vv, nv, pv[nv])) { v = nv; vv = pv[v]; nv = nxt[v]; moved = true; }
                if (!moved) break;
            }
            nxt[u] = (hull_t)v; prv[v] = (hull_t)u;
        }
    }
    hull_wave_sync();
    if (HULL_EXP & 4) return;
    hull_t *vl = (hull_t *)prv;                             // vertex array over the (now dead) back links: vl[k] sits at or before vertex k
    int nv = 0;
    if (lane == 0) for (int i = 0; i >= 0; i = nxt[i]) vl[nv++] = (hull_t)i;
    nv = __builtin_amdgcn_readfirstlane(nv);
    hull_wave_sync();
    for (int r = lane; r < h; r += 64) {                    // pixel row r = profile index 2r+1
        const int i = 2 * r + 1;
        int lo = 0, hi = nv - 1;                            // edge k with vl[k] < i <= vl[k+1]
        while (hi - lo > 1) { const int m = (lo + hi) >> 1; if (vl[m] < i) lo = m; else hi = m; }
        const int ea = vl[lo], eb = vl[hi], xa = pv[ea], xb = pv[eb];
        const int den = eb - ea, twoden = 2 * den;
        const int num = xa * den + (xb - xa) * (i - ea);    // 2 * den * X on the edge; column = X / 2
        const int c = LOWER ? (num >= 0 ? (num + twoden - 1) / twoden : -((-num) / twoden))      // ceil
                            : (num >= 0 ? num / twoden : -((-num + twoden - 1) / twoden));       // floor
        bound[2 * r + (LOWER ? 0 : 1)] = (hull_t)c;
    }
}
__global__ __launch_bounds__(128) void k_convex(int n_saved, const int32_t *__restrict__ geo, const long long *__restrict__ row_off,
                                                const int32_t *__restrict__ rowext, long long *sums, int hlo, int hhi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char hull_smem[];
    const int s = blockIdx.x;
    const int h = geo[8 * (size_t)s + 4];
    if (h <= hlo || h > hhi) return;                        // another size class
    const int np = 2 * h + 1;
    hull_t *fl = (hull_t *)hull_smem;                       // first / last column per row (last = -1: empty row); later the two column bounds
    const int npm = 2 * hhi + 1;
    hull_t *pvl = fl + 2 * hhi, *pvu = pvl + npm, *nxl = pvu + npm, *prl = nxl + npm, *nxu = prl + npm, *pru = nxu + npm;
    long long *q = sums + (size_t)s * PSEG_NSUMS;
    const int32_t *re = rowext + 2 * row_off[s];
    for (int i = threadIdx.x; i < 2 * h; i += 128) { const int v = re[i]; fl[i] = (hull_t)(v > 32767 ? 32767 : v); }   // (first = INT_MAX in empty rows)
    __syncthreads();
    // rows without a set pixel (possible between the components of a dense mask, pseg_regionprops) give no hull points:
    // their profile values lie beyond every real column, small enough for the 32-bit cross products
    const int LINF = 20000, UINF = -20000;
    for (int i = threadIdx.x; i < np; i += 128) {
        const int r = i >> 1;
        int l, u;
        auto lo_of = [&](int row) { return (row >= 0 && row < h && fl[2 * row + 1] >= 0) ? 2 * fl[2 * row] : LINF; };
        auto hi_of = [&](int row) { return (row >= 0 && row < h && fl[2 * row + 1] >= 0) ? 2 * fl[2 * row + 1] : UINF; };
        if (i & 1) {
            const bool has = fl[2 * r + 1] >= 0;
            l = has ? 2 * fl[2 * r] - 1 : LINF; u = has ? 2 * fl[2 * r + 1] + 1 : UINF;
        } else {
            const int la = lo_of(r - 1), lb = lo_of(r), ua = hi_of(r - 1), ub = hi_of(r);
            l = la < lb ? la : lb; u = ua > ub ? ua : ub;
        }
        pvl[i] = (hull_t)l; pvu[i] = (hull_t)u;
    }
    __syncthreads();
    if (!(HULL_EXP & 1)) {
    if (threadIdx.x < 64) hull_chain<true>(pvl, np, nxl, prl, fl, h, threadIdx.x);
    else hull_chain<false>(pvu, np, nxu, pru, fl, h, threadIdx.x - 64);
    }
    __syncthreads();
    unsigned long long cnt = 0;
    for (int r = threadIdx.x; r < h; r += 128)
        if (fl[2 * r + 1] >= fl[2 * r]) cnt += (unsigned long long)(fl[2 * r + 1] - fl[2 * r] + 1);
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd((unsigned long long *)&q[14], cnt);
}
// The same hull for SMALL windows (at most CONVEX_SMALL_H rows), one WAVE per window and no LDS: on frames of 2000 objects a
// batch saves 100 000 windows of 25 - 45 rows, and a 128-thread block with four block barriers, linked lists in LDS and six
// merge rounds per window is all overhead (0.38 ms per batch, the second largest kernel of the tail).  Lane l holds the two
// profile points of row l -- index 2l (between rows l-1 and l) and 2l+1 (on row l) -- in registers.  The strict hull is what
// is left when no point lies on or above the chord of its two surviving neighbours; every round each surviving point looks up
// its neighbours (bit scans over the two ballots of survivors, one ds_bpermute per side for the packed values of another
// lane) and drops out if it does -- dropping a point that lies on or above a chord of two other points of the set is always
// right, so all lanes may decide at once, and when nobody drops the chain is convex.  Typically 5 - 15 rounds of some 40
// instructions.  Same predicate (hull_drop: collinear points dropped) and the same bound arithmetic as k_convex, hence the
// same vertex set and the same count.  The right chain is the left chain of the negated profile (floor x = -ceil -x).
// A lane per window with the stack in LDS was tried first: 0.136 ms for the 5 700 windows of an ordinary batch -- a chain
// of 250 dependent LDS round trips per lane, one wave per SIMD.
#define CONVEX_SMALL_H 63
__device__ __forceinline__ int hull_wave_bound(int v0, int v1, bool has0, bool has1, int lane, int h)   // ceil of the lower hull at 2 lane + 1, /2
{
    bool a0 = has0, a1 = has1;
    const unsigned long long below = (1ull << lane) - 1ull, above = lane == 63 ? 0ull : ~((2ull << lane) - 1ull);
    const int packed = (int)(((uint32_t)v0 & 0xffffu) | ((uint32_t)v1 << 16));
    int ip = 0, vp = 0, in_ = 0, vn = 0;                    // neighbours in other lanes (profile index, value)
    bool hp = false, hn = false;
    for (int guard = 0; guard < 256; guard++) {
        const unsigned long long E = __ballot(a0), O = __ballot(a1), A = E | O;
        const unsigned long long ab = A & below, aa = A & above;
        hp = ab != 0; hn = aa != 0;
        const int pl = hp ? 63 - __builtin_clzll(ab) : 0, nl = hn ? __builtin_ctzll(aa) : 0;
        const int wp = __builtin_amdgcn_ds_bpermute(pl << 2, packed), wn = __builtin_amdgcn_ds_bpermute(nl << 2, packed);
        const bool po = (O >> pl) & 1ull;                   // the last survivor of lane pl: its odd point if that survives
        ip = 2 * pl + (po ? 1 : 0); vp = po ? (wp >> 16) : (int)(short)(wp & 0xffff);
        const bool ne = (E >> nl) & 1ull;                   // the first survivor of lane nl: its even point if that survives
        in_ = 2 * nl + (ne ? 0 : 1); vn = ne ? (int)(short)(wn & 0xffff) : (wn >> 16);
        // point 0 (index 2 lane): between the survivor below and its own odd point, or the survivor above
        const bool d0 = a0 && hp && (a1 || hn) && hull_drop<true>(ip, vp, 2 * lane, v0, a1 ? 2 * lane + 1 : in_, a1 ? v1 : vn);
        // point 1 (index 2 lane + 1): between its own even point, or the survivor below, and the survivor above
        const bool d1 = a1 && (a0 || hp) && hn && hull_drop<true>(a0 ? 2 * lane : ip, a0 ? v0 : vp, 2 * lane + 1, v1, in_, vn);
        if (!__ballot(d0 || d1)) break;
        a0 = a0 && !d0; a1 = a1 && !d1;
    }
    // (the neighbours found in the last round are the final ones: nothing was dropped after they were looked up)
    if (lane >= h) return 0;
    const int i = 2 * lane + 1;
    if (a1) return v1 >= 0 ? (v1 + 1) >> 1 : -((-v1) >> 1);            // a vertex: ceil(v1 / 2)
    const int ea = a0 ? 2 * lane : ip, xa = a0 ? v0 : vp, eb = in_, xb = vn;
    const int den = eb - ea, twoden = 2 * den;
    const int num = xa * den + (xb - xa) * (i - ea);        // 2 * den * X on the edge; column = X / 2
    return num >= 0 ? (num + twoden - 1) / twoden : -((-num) / twoden);
}
__global__ __launch_bounds__(256) void k_convex_small(int n_saved, const int32_t *__restrict__ geo, const long long *__restrict__ row_off,
                                                      const int32_t *__restrict__ rowext, long long *sums)
{
    const int lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (s >= n_saved) return;
    const int h = geo[8 * (size_t)s + 4];
    if (h > CONVEX_SMALL_H) return;                         // k_convex serves the taller ones
    const int32_t *re = rowext + 2 * row_off[s];
    int first = 0, last = -1;                               // first / last column of row `lane` (last = -1: no pixel, no row)
    if (lane < h) { const int2 fl = *(const int2 *)(re + 2 * lane); first = fl.x > 32767 ? 32767 : fl.x; last = fl.y; }
    const int LINF = 20000;
    const int lo = last >= 0 ? 2 * first : LINF, hi = last >= 0 ? -2 * last : LINF;      // doubled columns; the right side negated
    int lo_prev = __shfl_up(lo, 1, 64), hi_prev = __shfl_up(hi, 1, 64);
    if (lane == 0) { lo_prev = LINF; hi_prev = LINF; }
    // profile values (k_convex): between two rows the outer one of both, on a row its extreme pixel moved out by half a pixel
    const bool has0 = lane <= h, has1 = lane < h;
    const int l0 = lo_prev < lo ? lo_prev : lo, l1 = last >= 0 ? lo - 1 : LINF;
    const int u0 = hi_prev < hi ? hi_prev : hi, u1 = last >= 0 ? hi - 1 : LINF;
    const int L = hull_wave_bound(l0, l1, has0, has1, lane, h);
    const int U = -hull_wave_bound(u0, u1, has0, has1, lane, h);
    unsigned long long cnt = (lane < h && U >= L) ? (unsigned long long)(U - L + 1) : 0ull;
    cnt = wave_sum_u64(cnt);
    if (lane == 0) sums[(size_t)s * PSEG_NSUMS + 14] = (long long)cnt;
}
__global__ void k_init_sums(long long *sums, int n_saved)
{
    size_t g = GTID;
    if (g >= (size_t)n_saved * PSEG_NSUMS) return;
    sums[g] = (g % PSEG_NSUMS == 6) ? 255 : 0;             // min intensity starts at 255
}

