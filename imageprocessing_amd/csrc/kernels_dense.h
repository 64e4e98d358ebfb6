// kernels_dense.h -- part of libpiscoseg.so, included by piscoseg.hip (one translation unit).
// Dense, HBM-bound kernels: background correction (k_bg_stream / k_bg_generic), threshold, the per-frame reduction of the block sums, disk mask, synthetic frames; wave-level helpers (DPP sums).
#pragma once

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// four u32 block sums -> out[0..3] (thread 0 writes); blockDim.x == 256
__device__ __forceinline__ void block_sum4_store(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t *out)
{
    __shared__ uint32_t red[4][4];
    a = wave_sum_u32(a); b = wave_sum_u32(b); c = wave_sum_u32(c); d = wave_sum_u32(d);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { red[wv][0] = a; red[wv][1] = b; red[wv][2] = c; red[wv][3] = d; }
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__device__ __forceinline__ uint32_t bmin4(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 8) { uint32_t x = (a >> k) & 255, y = (b >> k) & 255; r |= (x < y ? x : y) << k; }
    return r;
}
__device__ __forceinline__ uint32_t bmax4(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 8) { uint32_t x = (a >> k) & 255, y = (b >> k) & 255; r |= (x > y ? x : y) << k; }
    return r;
}
__device__ __forceinline__ uint32_t babsdiff4(uint32_t a, uint32_t b)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 8) { uint32_t x = (a >> k) & 255, y = (b >> k) & 255; r |= (x > y ? x - y : y - x) << k; }
    return r;
}
// per-dword tail shared by the dense kernels: corr = |img-bg| (already computed), inmask bytes
// (0xff where mask == 255); accumulates the four sums and returns the 4 threshold bits
__device__ __forceinline__ uint32_t tail4(uint32_t img, uint32_t corr, uint32_t inm, int thr, uint32_t &s0, uint32_t &s1,
                                          uint32_t &s2, uint32_t &s3)
{
    uint32_t cm = corr & inm, bitsv = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        uint32_t x = (img >> (8 * k)) & 255, c = (cm >> (8 * k)) & 255;
        s0 += x; s1 += x * x; s2 += c; s3 += c * c;
        bitsv |= (uint32_t)((int)c > thr) << k;
    }
    return bitsv;
}
__device__ __forceinline__ uint32_t eq255_4(uint32_t m)
{
    uint32_t r = 0;
#pragma unroll
    for (int k = 0; k < 32; k += 8) r |= (((m >> k) & 255) == 255 ? 255u : 0u) << k;
    return r;
}

// ------------------------------------------------------------------------------------------------
// dense kernels.  Thread = one 8-pixel strip (uint2); block = 256 threads = 2048 pixels;
// grid = (ceil(P/2048), frames).  partials[(f*nblk+blk)*4 + {0,1,2,3}] = sum x, sum x^2 (raw frame),
// sum c, sum c^2 (corrected inside the mask).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_bg_generic(const uint8_t *__restrict__ frames, int n, int NW, size_t P,
                                                    uint8_t *__restrict__ cleaned, const uint8_t *__restrict__ mask,
                                                    int thr, uint8_t *__restrict__ bits8, uint32_t *__restrict__ partials)
{
    size_t strip = (size_t)blockIdx.x * 256 + threadIdx.x;
    int f = blockIdx.y;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (strip * 8 < P) {
        int start = (f + NW < n) ? f : n - NW;             // bg_correction.py:72 (may be negative: Python wrap)
        auto ld = [&](int idx) { if (idx < 0) idx += n; return *(const uint2 *)(frames + (size_t)idx * P + strip * 8); };
        uint2 prev = ld(start), bg = make_uint2(0, 0);
        for (int j = 1; j < NW; j++) {                     // :74-78
            uint2 cur = ld(start + j);
            bg.x = bmax4(bg.x, bmin4(prev.x, cur.x));
            bg.y = bmax4(bg.y, bmin4(prev.y, cur.y));
            prev = cur;
        }
        uint2 img = ld(f);
        uint2 corr = make_uint2(babsdiff4(img.x, bg.x), babsdiff4(img.y, bg.y));   // :80
        *(uint2 *)(cleaned + (size_t)f * P + strip * 8) = make_uint2(~corr.x, ~corr.y);   // :81 (255-c == ~c)
        uint2 inm = make_uint2(0xffffffffu, 0xffffffffu);
        if (mask) { uint2 m = *(const uint2 *)(mask + strip * 8); inm = make_uint2(eq255_4(m.x), eq255_4(m.y)); }
        uint32_t b = tail4(img.x, corr.x, inm.x, thr, s0, s1, s2, s3) | (tail4(img.y, corr.y, inm.y, thr, s0, s1, s2, s3) << 4);
        bits8[(size_t)f * (P / 8) + strip] = (uint8_t)b;
    }
    block_sum4_store(s0, s1, s2, s3, partials + ((size_t)f * gridDim.x + blockIdx.x) * 4);
}

// ------------------------------------------------------------------------------------------------
// streaming background kernel (window NW = 2*n_bg_imgs frames; the notes below are written for the reference's NW = 12,
// n_bg_imgs = 6 -- the other windows differ in the last combination of the sparse table, see BgWin)
//
// A thread owns one 8-pixel strip and walks the frames of its time chunk [a, b) in order, so every
// raw byte is read from HBM once (plus an 11-frame halo per chunk) instead of 12 times.
//   m[j]   = min(I[j], I[j+1])                          (pair minima, bg_correction.py:76)
//   bg[s]  = max(m[s] .. m[s+10])                       (bg_correction.py:78) as a trailing sparse table:
//            T2[j] = max(m[j], m[j-1]); T4[j] = max(T2[j], T2[j-2]); T8[j] = max(T4[j], T4[j-4]);
//            T11[j] = max(T8[j], T8[j-3]) = max(m[j-10..j]) = bg[j-10]
//   frame s = j-10 is emitted when frame j+1 arrives; the last 12 frames share bg[n-12] (:72) and are
//   emitted together when frame n-1 arrives.
// Pixel values live as u16 pairs (even / odd bytes of a dword) so that v_pk_min/max/sub_u16 apply.
// All rings are indexed with compile-time phases (loop unrolled by 16); 4 loads stay in flight.
// No LDS, no barriers: each wave reduces its four sums with DPP and lane 63 stores them.
// ------------------------------------------------------------------------------------------------
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pkmax(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b))); }
__device__ __forceinline__ uint32_t pkmin(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b))); }
__device__ __forceinline__ uint32_t pksub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) - __builtin_bit_cast(u16x2_t, b))); }
template <int CTRL, int RM> __device__ __forceinline__ uint32_t dpp_add(uint32_t v) { return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, RM, 0xf, false); }
// sum over the wave, valid in lane 63
__device__ __forceinline__ uint32_t wave_sum63(uint32_t v)
{
    v = dpp_add<0x111, 0xf>(v); v = dpp_add<0x112, 0xf>(v); v = dpp_add<0x114, 0xf>(v); v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);     // row_bcast:15 into rows 1,3
    v = dpp_add<0x143, 0xc>(v);     // row_bcast:31 into rows 2,3
    return v;
}
struct Px8 { uint32_t e0, o0, e1, o1; };      // 8 pixels as four registers of two u16 lanes
__device__ __forceinline__ Px8 px_split(uint2 v)
{
    Px8 r;
    r.e0 = __builtin_amdgcn_perm(0, v.x, 0x0c020c00); r.o0 = __builtin_amdgcn_perm(0, v.x, 0x0c030c01);
    r.e1 = __builtin_amdgcn_perm(0, v.y, 0x0c020c00); r.o1 = __builtin_amdgcn_perm(0, v.y, 0x0c030c01);
    return r;
}
__device__ __forceinline__ Px8 px_max(Px8 a, Px8 b) { return {pkmax(a.e0, b.e0), pkmax(a.o0, b.o0), pkmax(a.e1, b.e1), pkmax(a.o1, b.o1)}; }
__device__ __forceinline__ Px8 px_min(Px8 a, Px8 b) { return {pkmin(a.e0, b.e0), pkmin(a.o0, b.o0), pkmin(a.e1, b.e1), pkmin(a.o1, b.o1)}; }

typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
// A wave-uniform base pointer pinned to scalar registers: the 32-bit lane offset added to it becomes the instruction's
// offset operand (global_load / global_store with an SGPR base) instead of 64-bit vector arithmetic per access.
#define PSEG_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ T PSEG_GLOBAL *uniform_base(const void *p)
{
    const unsigned long long v = (unsigned long long)p;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (T PSEG_GLOBAL *)(((unsigned long long)hi << 32) | lo);
}
// raw buffer resource over a wave-uniform base: buffer_load / buffer_store take the 32-bit lane offset as it is ("offen"),
// no vector arithmetic for the address; aux 2 = non-temporal
__device__ __forceinline__ __amdgpu_buffer_rsrc_t raw_rsrc(const void *base)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)uniform_base<uint8_t>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ uint2 ld_stream_at(const uint8_t *base, uint32_t off)      // base wave-uniform
{
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(raw_rsrc(base), (int)off, 0, 2);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint2 ld_stream(const uint8_t *p)
{
    u32x2_t v = __builtin_nontemporal_load((const u32x2_t *)p);
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ uint32_t pkadd(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, (u16x2_t)(__builtin_bit_cast(u16x2_t, a) + __builtin_bit_cast(u16x2_t, b))); }
struct BgEmit {
    uint8_t *cleaned, *bits8; uint4 *partials;              // real outputs
    uint8_t *t_cleaned, *t_bits8; uint4 *t_partials;        // trash rows: where a disabled step stores (keeps the loop branch-free)
    size_t P; uint32_t off8, strip32; uint2 inm, m80; uint32_t kadd, lowm; uint32_t nwaves, wave;
    uint32_t act;          // 0xffffffff for strips inside the frame, 0 for the padding lanes of the last wave
};
__device__ __forceinline__ uint32_t gather4(uint32_t r)     // bits 7,15,23,31 -> bits 0..3
{
    r >>= 7; r |= r >> 7; r |= r >> 14;
    return r & 0xfu;
}
// frame s: cleaned = 255 - |img - bg| (bg_correction.py:80-81), the four sums and the threshold bits
// (detection.py:178-202).  c = px_split(img).  en is wave-uniform; a disabled call stores into the trash rows.
// The four sums of the frame are left per lane in (op, os1, os3) -- sum x | sum c << 16, sum x^2, sum c^2 --; bg_flush
// reduces them over the wave for four frames at a time.
template <bool PAD>
__device__ __forceinline__ void bg_emit(const BgEmit &E, uint2 img, Px8 c, Px8 bg, int s, bool en, uint32_t &op, uint32_t &os1, uint32_t &os3)
{
    uint32_t de0 = pksub(pkmax(c.e0, bg.e0), pkmin(c.e0, bg.e0)), do0 = pksub(pkmax(c.o0, bg.o0), pkmin(c.o0, bg.o0));
    uint32_t de1 = pksub(pkmax(c.e1, bg.e1), pkmin(c.e1, bg.e1)), do1 = pksub(pkmax(c.o1, bg.o1), pkmin(c.o1, bg.o1));
    uint2 corr = make_uint2(__builtin_amdgcn_perm(do0, de0, 0x06020400), __builtin_amdgcn_perm(do1, de1, 0x06020400));
    // every store is unconditional (disabled steps and padding lanes write to the trash rows): with no
    // branch in the loop body the compiler counts the outstanding loads exactly (vmcnt counts stores too)
    const bool real = PAD ? (en && E.act) : en;           // without padding lanes the bases stay wave-uniform (scalar registers)
    uint8_t *pc = real ? E.cleaned + (size_t)s * E.P : E.t_cleaned;          // wave-uniform bases, 32-bit lane offsets
    uint8_t *pb = real ? E.bits8 + (size_t)s * (E.P / 8) : E.t_bits8;

    // streamed out: a plain store would leave 1.3 GB of dirty lines to drain from L2 under the kernels that follow
    {
        u32x2_t cv; cv.x = ~corr.x; cv.y = ~corr.y;
        if (PAD) __builtin_nontemporal_store(cv, (u32x2_t *)(pc + E.off8));
        else __builtin_amdgcn_raw_buffer_store_b64(cv, raw_rsrc(pc), (int)E.off8, 0, 2);
    }
    uint2 cm = make_uint2(corr.x & E.inm.x, corr.y & E.inm.y);
    if (PAD) { img.x &= E.act; img.y &= E.act; }
    uint32_t s0 = __builtin_amdgcn_sad_u8(img.x, 0u, __builtin_amdgcn_sad_u8(img.y, 0u, 0u));
    uint32_t s1 = __builtin_amdgcn_udot4(img.x, img.x, __builtin_amdgcn_udot4(img.y, img.y, 0u, false), false);
    uint32_t s2 = __builtin_amdgcn_sad_u8(cm.x, 0u, __builtin_amdgcn_sad_u8(cm.y, 0u, 0u));
    uint32_t s3 = __builtin_amdgcn_udot4(cm.x, cm.x, __builtin_amdgcn_udot4(cm.y, cm.y, 0u, false), false);
    // threshold bits on the packed difference bytes: (d & 0x7f) + (128 - t) carries into bit 7 iff the low seven bits reach
    // t (t <= 128; together with "d >= 128" by OR) resp. t - 128 (t > 128; AND) -- no carry crosses a byte.  The four flag
    // bits of a register are gathered by one dot product with the weights 1, 2, 4, 8 (16 .. 128 for the upper register).
    // OR / AND of the two conditions = majority with a constant third input (all ones: OR, zero: AND), one v_bitop3;
    // m80 = bit 7 of the bytes inside the mask (0 when no pixel can pass): isolates the flags and applies the mask at once
    const uint32_t t0 = (corr.x & 0x7f7f7f7fu) + E.kadd, t1 = (corr.y & 0x7f7f7f7fu) + E.kadd;
    const uint32_t f0 = __builtin_amdgcn_bitop3_b32(t0, corr.x, E.lowm, 0xE8) & E.m80.x, f1 = __builtin_amdgcn_bitop3_b32(t1, corr.y, E.lowm, 0xE8) & E.m80.y;
    const uint32_t b = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false) >> 7;
    if (PAD) pb[E.strip32] = (uint8_t)b;
    else __builtin_amdgcn_raw_buffer_store_b8((uint8_t)b, raw_rsrc(pb), (int)E.strip32, 0, 0);
    op = s0 | (s2 << 16);                                   // sum x and sum c (<= 2040 per lane) travel packed in one register
    os1 = s1; os3 = s3;
}
// Wave sums of four frames at once.  Two swap levels put frame r into row r of the wave (v_permlane32_swap exchanges the
// upper half of one register with the lower half of another, v_permlane16_swap the odd rows with the even rows: swap + add
// halves two registers into one, each lane then holds the sum over the four rows at its position), four row_shr levels sum
// the 16 lanes of a row: lanes 15 / 31 / 47 / 63 hold the totals of frames s0 .. s0+3 and store them.  30 vector
// instructions for four frames and three registers instead of 19 per frame.
// Frame s0 + r is stored for r < cnt and lo <= s0 + r < hi; the others go to the trash rows.
typedef unsigned int u32pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rows_of_frames(uint32_t f0, uint32_t f1, uint32_t f2, uint32_t f3)
{
    const u32pair_t a = __builtin_amdgcn_permlane32_swap(f0, f2, false, false);      // [f0 lo, f2 lo], [f0 hi, f2 hi]
    const u32pair_t b = __builtin_amdgcn_permlane32_swap(f1, f3, false, false);
    const uint32_t x = a[0] + a[1], y = b[0] + b[1];        // rows 0,1: f0 (f1); rows 2,3: f2 (f3)
    const u32pair_t c = __builtin_amdgcn_permlane16_swap(x, y, false, false);         // [x0, y0, x2, y2], [x1, y1, x3, y3]
    return c[0] + c[1];                                      // row r: frame r, summed over the four rows
}
struct BgSums { uint32_t p[4], s1[4], s3[4]; };
__device__ __forceinline__ void bg_flush(const BgEmit &E, const BgSums &A, int s0, int lo, int hi, int cnt)
{
    uint32_t p = rows_of_frames(A.p[0], A.p[1], A.p[2], A.p[3]);
    uint32_t s1 = rows_of_frames(A.s1[0], A.s1[1], A.s1[2], A.s1[3]);
    uint32_t s3 = rows_of_frames(A.s3[0], A.s3[1], A.s3[2], A.s3[3]);
    p = dpp_add<0x111, 0xf>(p); p = dpp_add<0x112, 0xf>(p); p = dpp_add<0x114, 0xf>(p);     // <= 8 * 4 * 2040 per half: no carry yet
    const uint32_t q = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0x118, 0xf, 0xf, false);
    s1 = dpp_add<0x111, 0xf>(s1); s1 = dpp_add<0x112, 0xf>(s1); s1 = dpp_add<0x114, 0xf>(s1); s1 = dpp_add<0x118, 0xf>(s1);
    s3 = dpp_add<0x111, 0xf>(s3); s3 = dpp_add<0x112, 0xf>(s3); s3 = dpp_add<0x114, 0xf>(s3); s3 = dpp_add<0x118, 0xf>(s3);
    const int lane = threadIdx.x & 63, r = lane >> 4, fr = s0 + r;
    if ((lane & 15) == 15) {
        const bool ok = r < cnt && fr >= lo && fr < hi;
        uint4 *pp = ok ? E.partials + (size_t)fr * E.nwaves : E.t_partials;
        pp[E.wave] = make_uint4((p & 0xffffu) + (q & 0xffffu), s1, (p >> 16) + (q >> 16), s3);
    }
}

// Per-thread state.  The history of raw pixels (NW frames) lives in an LDS ring (dynamic slot index, private to the thread:
// no barriers); the sparse table and 4 prefetched frames live in registers with compile-time phases (loop unrolled by 4,
// by 8 for windows whose T8 history is deeper than 4 steps).
// Window NW = 2 * n_bg_imgs frames (bg_correction.py:65) -> M = NW - 1 pair minima per background:
//   M in 8 .. 15:  TM[j] = max(T8[j], T8[j - (M - 8)])      (NW = 10: T8[j-1], 12: T8[j-3], 14: T8[j-5], 16: T8[j-7])
//   M in 4 .. 7:   TM[j] = max(T4[j], T4[j - (M - 4)])      (NW = 6: T4[j-1], 8: T4[j-3])
//   M = 3:         TM[j] = max(T2[j], T2[j - 1])            (NW = 4)
template <int NW> struct BgWin {
    static constexpr int M = NW - 1;
    static constexpr int D8 = M >= 8 ? M - 8 : 0;           // steps back into the T8 history
    static constexpr int T8N = D8 > 4 ? 8 : 4;              // its ring
    static constexpr int U = D8 > 4 ? 8 : 4;                // loop unroll (phases)
    static constexpr int BLOCKS = D8 > 4 ? 3 : 4;           // blocks per CU: the ring of eight T8 entries takes 16 registers more
                                                            // than 4 blocks per CU leave a thread (128; 20 - 26 spilled)
    static_assert(NW >= 4 && NW <= 16 && NW % 2 == 0, "window of 4 .. 16 frames");
};
template <int NW>
struct BgRegs {
    uint2 pf[4];           // frames k .. k+3 (pf[PH & 3] is the frame of the current step)
    Px8 prev;              // frame k-1, split
    Px8 mp, t2[2], t4[4], t8[BgWin<NW>::T8N], last;
    BgSums acc;            // per-lane sums of the frames emitted by the last (up to) four steps
};
#define BG_SLOTS 16
template <int PH, int NW, bool PAD>
__device__ __forceinline__ void bg_step(BgRegs<NW> &R, uint2 *ring, const BgEmit &E, const uint8_t *__restrict__ src, int k, int a,
                                        int kend, int n)
{
    constexpr int M = BgWin<NW>::M, T8M = BgWin<NW>::T8N - 1;
    const uint2 cur = R.pf[PH & 3];
    int kn = k + 4 <= kend ? k + 4 : kend;                  // clamped prefetch: always a valid frame
    R.pf[PH & 3] = ld_stream_at(src + (size_t)kn * E.P, E.off8);   // src is wave-uniform
    ring[(k & (BG_SLOTS - 1)) * 256] = cur;
    const uint2 old = ring[((k - (NW - 1)) & (BG_SLOTS - 1)) * 256];   // frame k-(NW-1) (zeros during warm-up)
    const Px8 cs = px_split(cur);
    Px8 m = px_min(R.prev, cs);                             // pair minimum m[k-1]
    if (k == a) m = Px8{0, 0, 0, 0};                         // no pair yet at the first frame of the chunk
    R.prev = cs;
    Px8 t2 = px_max(m, R.mp); R.mp = m;
    Px8 tm;
    if constexpr (M < 4) {
        tm = px_max(t2, R.t2[(PH + 1) & 1]); R.t2[PH & 1] = t2;
    } else {
        Px8 t4 = px_max(t2, R.t2[PH & 1]); R.t2[PH & 1] = t2;            // (slot written two steps ago: read, then replaced)
        if constexpr (M < 8) {
            tm = M == 4 ? t4 : px_max(t4, R.t4[(PH + 8 - (M - 4)) & 3]); R.t4[PH & 3] = t4;
        } else {
            Px8 t8 = px_max(t4, R.t4[PH & 3]); R.t4[PH & 3] = t4;        // (written four steps ago)
            tm = M == 8 ? t8 : px_max(t8, R.t8[(PH + 16 - (M - 8)) & T8M]); R.t8[PH & T8M] = t8;
        }
    }
    R.last = tm;
    bg_emit<PAD>(E, old, px_split(old), tm, k - (NW - 1), k - (NW - 1) >= a && k < n - 1, R.acc.p[PH & 3], R.acc.s1[PH & 3], R.acc.s3[PH & 3]);
}

// grid = (ceil(P/2048), chunks); chunk c emits frames [bounds[c], bounds[c+1]) of the reference batch that ends at frame
// bounds[chunks + 1 + c] (several batches in one launch: every batch is at least one chunk).  PAD: the last block has lanes
// beyond the frame (P % 2048 != 0).  NW: the window (2 * n_bg_imgs frames).
template <int NW, bool PAD>
__global__ __launch_bounds__(256, BgWin<NW>::BLOCKS) void k_bg_stream(const uint8_t *__restrict__ frames, size_t P, uint8_t *__restrict__ cleaned,
                                                      const uint8_t *__restrict__ mask, int thr, uint8_t *__restrict__ bits8,
                                                      uint4 *__restrict__ partials, const int *__restrict__ bounds,
                                                      uint8_t *__restrict__ trash, uint32_t *__restrict__ maskpart,
                                                      int nfull, int tailk)
{
    __shared__ uint2 lds[BG_SLOTS * 256];
    // strips [0, nfull) walk their whole chunk; each strip behind them is shared by tailk blocks, which walk the tail bounds
    // (stored behind the chunk tables; tailk > 1 only with one chunk) -- see run_bg
    int col = blockIdx.x, bi = blockIdx.y;
    if (tailk > 1 && col >= nfull) {
        const int t = col - nfull;
        col = nfull + t / tailk;
        bi = 2 * gridDim.y + 1 + (t - (col - nfull) * tailk);
    }
    const int ncols = nfull + (gridDim.x - nfull) / tailk;
    size_t strip = (size_t)col * 256 + threadIdx.x;
    const bool act = strip * 8 < P;                         // padding lanes stay in the wave (DPP sums), load strip 0, store nothing
    if (!act) strip = 0;
    const int a = bounds[bi], b = bounds[bi + 1];
    const int n = bounds[gridDim.y + 1 + blockIdx.y];        // end of the reference batch this chunk belongs to
    const int kend = (b == n) ? n - 1 : b + NW - 2;
    BgEmit E;
    E.cleaned = cleaned; E.bits8 = bits8; E.partials = partials; E.P = P;
    E.strip32 = (uint32_t)strip; E.off8 = (uint32_t)strip * 8u;             // P <= 2^29: lane offsets fit 32 bits
    E.t_cleaned = trash; E.t_bits8 = trash + P; E.t_partials = (uint4 *)(trash + P + P / 8);
    E.inm = make_uint2(0xffffffffu, 0xffffffffu);
    if (mask) { uint2 m = *(const uint2 *)(mask + strip * 8); E.inm = make_uint2(eq255_4(m.x), eq255_4(m.y)); }
    E.act = act ? 0xffffffffu : 0u;
    E.inm.x &= E.act; E.inm.y &= E.act;
    const int t1 = thr + 1 > 0 ? thr + 1 : 0;               // pixel > thr  <=>  pixel >= thr + 1 (negative thr: every pixel)
    const bool thr_ok = t1 <= 255;                          // thr >= 255: no pixel passes
    E.kadd = (uint32_t)(t1 <= 128 ? 128 - t1 : 256 - (t1 > 255 ? 255 : t1)) * 0x01010101u;
    E.lowm = t1 <= 128 ? 0xffffffffu : 0u;
    E.m80 = thr_ok ? make_uint2(E.inm.x & 0x80808080u, E.inm.y & 0x80808080u) : make_uint2(0u, 0u);
    E.nwaves = ncols * 4; E.wave = col * 4 + (threadIdx.x >> 6);
    if (mask && a == 0) {                          // pixels with mask == 255, per wave (summed by k_reduce_partials)
        const uint32_t mc = wave_sum63((uint32_t)__popc(gather4(E.inm.x & 0x80808080u) | (gather4(E.inm.y & 0x80808080u) << 4)));
        if ((threadIdx.x & 63) == 63) maskpart[E.wave] = mc;
    }
    const uint8_t *src = frames;
    uint2 *ring = lds + threadIdx.x;
#pragma unroll
    for (int q = 0; q < BG_SLOTS; q++) ring[q * 256] = make_uint2(0, 0);
    BgRegs<NW> R;
    const Px8 z = {0, 0, 0, 0};
    R.mp = z; R.last = z; R.prev = z;
    for (int i = 0; i < 2; i++) R.t2[i] = z;
    for (int i = 0; i < 4; i++) R.t4[i] = z;
    for (int i = 0; i < BgWin<NW>::T8N; i++) R.t8[i] = z;
#pragma unroll
    for (int i = 0; i < 4; i++) R.pf[i] = ld_stream(src + (size_t)(a + i <= kend ? a + i : kend) * P + E.off8);
    constexpr int U = BgWin<NW>::U;
    int k = a;
    // branch-free body: counted waits keep 4 loads in flight.  Step k emits frame k - (NW-1), stored while a <= k - (NW-1)
    // and k < n - 1 (the frames from n - NW on share one background and follow the loop).
    if constexpr (U == 8) {
        for (; k + 7 <= kend; k += 8) {
            bg_step<0, NW, PAD>(R, ring, E, src, k, a, kend, n);
            bg_step<1, NW, PAD>(R, ring, E, src, k + 1, a, kend, n);
            bg_step<2, NW, PAD>(R, ring, E, src, k + 2, a, kend, n);
            bg_step<3, NW, PAD>(R, ring, E, src, k + 3, a, kend, n);
            bg_flush(E, R.acc, k - (NW - 1), a, n - NW, 4);
            bg_step<4, NW, PAD>(R, ring, E, src, k + 4, a, kend, n);
            bg_step<5, NW, PAD>(R, ring, E, src, k + 5, a, kend, n);
            bg_step<6, NW, PAD>(R, ring, E, src, k + 6, a, kend, n);
            bg_step<7, NW, PAD>(R, ring, E, src, k + 7, a, kend, n);
            bg_flush(E, R.acc, k + 4 - (NW - 1), a, n - NW, 4);
        }
        if (k + 3 <= kend) {
            bg_step<0, NW, PAD>(R, ring, E, src, k, a, kend, n);
            bg_step<1, NW, PAD>(R, ring, E, src, k + 1, a, kend, n);
            bg_step<2, NW, PAD>(R, ring, E, src, k + 2, a, kend, n);
            bg_step<3, NW, PAD>(R, ring, E, src, k + 3, a, kend, n);
            bg_flush(E, R.acc, k - (NW - 1), a, n - NW, 4);
            k += 4;
            if (k <= kend) {
                const int left = kend - k + 1;              // 1 .. 3 steps
                bg_step<4, NW, PAD>(R, ring, E, src, k, a, kend, n);
                if (left > 1) bg_step<5, NW, PAD>(R, ring, E, src, k + 1, a, kend, n);
                if (left > 2) bg_step<6, NW, PAD>(R, ring, E, src, k + 2, a, kend, n);
                bg_flush(E, R.acc, k - (NW - 1), a, n - NW, left);
            }
        } else if (k <= kend) {
            const int left = kend - k + 1;
            bg_step<0, NW, PAD>(R, ring, E, src, k, a, kend, n);
            if (left > 1) bg_step<1, NW, PAD>(R, ring, E, src, k + 1, a, kend, n);
            if (left > 2) bg_step<2, NW, PAD>(R, ring, E, src, k + 2, a, kend, n);
            bg_flush(E, R.acc, k - (NW - 1), a, n - NW, left);
        }
    } else {
        for (; k + 3 <= kend; k += 4) {
            bg_step<0, NW, PAD>(R, ring, E, src, k, a, kend, n);
            bg_step<1, NW, PAD>(R, ring, E, src, k + 1, a, kend, n);
            bg_step<2, NW, PAD>(R, ring, E, src, k + 2, a, kend, n);
            bg_step<3, NW, PAD>(R, ring, E, src, k + 3, a, kend, n);
            bg_flush(E, R.acc, k - (NW - 1), a, n - NW, 4);
        }
        if (k <= kend) {
            const int left = kend - k + 1;                  // 1 .. 3 steps
            bg_step<0, NW, PAD>(R, ring, E, src, k, a, kend, n);
            if (left > 1) bg_step<1, NW, PAD>(R, ring, E, src, k + 1, a, kend, n);
            if (left > 2) bg_step<2, NW, PAD>(R, ring, E, src, k + 2, a, kend, n);
            bg_flush(E, R.acc, k - (NW - 1), a, n - NW, left);
        }
    }
    if (b == n) {                                           // the last NW frames share bg[n-NW] = TM of the final step;
#pragma unroll 1
        for (int i = n - NW; i < n; i += 4) {                // their raw pixels are still in the ring (a <= n - NW)
            const int cnt = n - i < 4 ? n - i : 4;           // (NW = 10, 14: the last group holds two frames)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint2 img = ring[((i + j) & (BG_SLOTS - 1)) * 256];
                bg_emit<PAD>(E, img, px_split(img), R.last, i + j, j < cnt, R.acc.p[j], R.acc.s1[j], R.acc.s3[j]);
            }
            bg_flush(E, R.acc, i, n - NW, n, cnt);
        }
    }
}

// detection.py:178-202 for a cleaned image that did not come from k_bg_* (the reference's default path: the deconvolved
// frame, segmenter.py:243-254): corrected = 255 - cleaned, masked sum / sum of squares, corrected[outside] = 0, the
// threshold bit plane.  Streaming form, like k_bg_stream: a thread owns one 8-pixel strip (mask bytes, threshold constants
// and addresses stay in registers) and walks the frames [a, b) of its chunk with four non-temporal loads in flight; the
// packed byte forms of bg_emit (v_sad_u8 / v_dot4 sums, carry-free byte compare + v_bitop3 for the flags, two dot products
// to gather them); the per-lane sums of four frames are reduced over the wave at once (rows_of_frames + four row_shr
// levels) and lanes 15 / 31 / 47 / 63 store them.  No LDS, no barriers.  partials[f][wave] = (0, 0, sum c, sum c^2).
// Algorithmic traffic: P read + P / 8 written per frame.
#define THR_FCHUNK 40                                       // frames per block (a multiple of 4)
template <bool PAD>
__global__ __launch_bounds__(256) void k_threshold(const uint8_t *__restrict__ cleaned, size_t P, const uint8_t *__restrict__ mask,
                                                   int thr, uint8_t *__restrict__ bits8, uint4 *__restrict__ partials, int n,
                                                   uint32_t *__restrict__ maskpart)
{
    size_t strip = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool act = !PAD || strip * 8 < P;
    if (!act) strip = 0;
    const int a = blockIdx.y * THR_FCHUNK, b = a + THR_FCHUNK < n ? a + THR_FCHUNK : n;
    const uint32_t off8 = (uint32_t)strip * 8u, strip32 = (uint32_t)strip;
    uint2 inm = make_uint2(0xffffffffu, 0xffffffffu);
    if (mask) { const uint2 m = *(const uint2 *)(mask + strip * 8); inm = make_uint2(eq255_4(m.x), eq255_4(m.y)); }
    if (!act) inm = make_uint2(0u, 0u);
    const int t1 = thr + 1 > 0 ? thr + 1 : 0;               // pixel > thr  <=>  pixel >= thr + 1 (see bg_emit)
    const bool thr_ok = t1 <= 255;
    const uint32_t kadd = (uint32_t)(t1 <= 128 ? 128 - t1 : 256 - (t1 > 255 ? 255 : t1)) * 0x01010101u;
    const uint32_t lowm = t1 <= 128 ? 0xffffffffu : 0u;
    const uint2 m80 = thr_ok ? make_uint2(inm.x & 0x80808080u, inm.y & 0x80808080u) : make_uint2(0u, 0u);
    const uint32_t nwaves = gridDim.x * 4, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (mask && blockIdx.y == 0) {                           // pixels with mask == 255, per wave (summed by k_reduce_partials)
        const uint32_t mc = wave_sum63((uint32_t)__popc(gather4(inm.x & 0x80808080u) | (gather4(inm.y & 0x80808080u) << 4)));
        if (lane == 63) maskpart[wave] = mc;
    }
    uint2 pf[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pf[i] = ld_stream_at(cleaned + (size_t)(a + i < b ? a + i : b - 1) * P, off8);
    for (int k = a; k < b; k += 4) {
        uint32_t ps[4], qs[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint2 cl = pf[j];
            const int kn = k + 4 + j < b ? k + 4 + j : b - 1;        // clamped prefetch: always a valid frame
            pf[j] = ld_stream_at(cleaned + (size_t)kn * P, off8);
            const uint2 corr = make_uint2(~cl.x, ~cl.y);             // detection.py:178
            const uint2 cm = make_uint2(corr.x & inm.x, corr.y & inm.y);
            ps[j] = __builtin_amdgcn_sad_u8(cm.x, 0u, __builtin_amdgcn_sad_u8(cm.y, 0u, 0u));
            qs[j] = __builtin_amdgcn_udot4(cm.x, cm.x, __builtin_amdgcn_udot4(cm.y, cm.y, 0u, false), false);
            const uint32_t t0 = (corr.x & 0x7f7f7f7fu) + kadd, t1v = (corr.y & 0x7f7f7f7fu) + kadd;
            const uint32_t f0 = __builtin_amdgcn_bitop3_b32(t0, corr.x, lowm, 0xE8) & m80.x, f1 = __builtin_amdgcn_bitop3_b32(t1v, corr.y, lowm, 0xE8) & m80.y;
            const uint32_t bb = __builtin_amdgcn_udot4(f1, 0x80402010u, __builtin_amdgcn_udot4(f0, 0x08040201u, 0u, false), false) >> 7;
            if (k + j < b && act)                                    // (wave-uniform but for the padding lanes of the last block)
                __builtin_amdgcn_raw_buffer_store_b8((uint8_t)bb, raw_rsrc(bits8 + (size_t)(k + j) * (P / 8)), (int)strip32, 0, 0);
        }
        uint32_t p = rows_of_frames(ps[0], ps[1], ps[2], ps[3]);     // row r of the wave: frame k + r
        uint32_t q = rows_of_frames(qs[0], qs[1], qs[2], qs[3]);
        p = dpp_add<0x111, 0xf>(p); p = dpp_add<0x112, 0xf>(p); p = dpp_add<0x114, 0xf>(p); p = dpp_add<0x118, 0xf>(p);
        q = dpp_add<0x111, 0xf>(q); q = dpp_add<0x112, 0xf>(q); q = dpp_add<0x114, 0xf>(q); q = dpp_add<0x118, 0xf>(q);
        const int fr = k + (lane >> 4);
        if ((lane & 15) == 15 && fr < b) partials[(size_t)fr * nwaves + wave] = make_uint4(0u, 0u, p, q);
    }
}

// per-frame reduction of the block partials; fsums[f*4+k] u64.  If setvalid: valid[f] = std > 2,
// decided exactly in integers: P*sum(x^2) - sum(x)^2 > 4*P^2   (bg_correction.py:71)
__global__ __launch_bounds__(1024) void k_reduce_partials(const uint32_t *__restrict__ partials, int nblk, unsigned long long P,
                                                          unsigned long long *__restrict__ fsums, uint8_t *__restrict__ valid,
                                                          int setvalid, int first, int count,
                                                          unsigned long long *__restrict__ cnt_totals, uint32_t *__restrict__ bandmax,
                                                          const uint32_t *__restrict__ maskpart, int nmaskpart,
                                                          unsigned long long *__restrict__ scan, int nscan)
{
    __shared__ unsigned long long red[16][4];
    int f = blockIdx.x;
    // every block clears its share of the look-back scan's state (ticket, error flag, one word per band: k_runs_band)
    for (int i = f * 1024 + threadIdx.x; i < nscan; i += gridDim.x * 1024) scan[i] = 0;
    // block 0 also clears the counters of the detection that follows (mask count | 4 totals, largest band) and, when the
    // background kernel left per-wave counts of mask pixels, sums them
    if (f == 0 && cnt_totals) {
        if (threadIdx.x >= 1 && threadIdx.x < 5) cnt_totals[threadIdx.x] = 0;
        if (threadIdx.x == 5) { bandmax[0] = 0; bandmax[1] = 0; }
        unsigned long long m = 0;
        for (int b = threadIdx.x; b < nmaskpart; b += 1024) m += maskpart[b];
        for (int o = 32; o > 0; o >>= 1) m += __shfl_down(m, o, 64);
        __shared__ unsigned long long mred[16];
        if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) { unsigned long long t = 0; for (int w = 0; w < 16; w++) t += mred[w]; cnt_totals[0] = t; }
    }
    unsigned long long a[4] = {0, 0, 0, 0};
    const uint4 *p4 = (const uint4 *)partials + (size_t)f * nblk;
    for (int b = threadIdx.x; b < nblk; b += 1024) {
        const uint4 v = p4[b];
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; k++)
        for (int o = 32; o > 0; o >>= 1) a[k] += __shfl_down(a[k], o, 64);
    int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) for (int k = 0; k < 4; k++) red[wv][k] = a[k];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long s[4] = {0, 0, 0, 0};
        for (int w = 0; w < 16; w++) for (int k = 0; k < 4; k++) s[k] += red[w][k];
        for (int k = first; k < first + count; k++) fsums[(size_t)f * 4 + k] = s[k];
        // (128-bit: P * sum x^2 reaches 2^64 from frames of 2^24 pixels on)
        if (setvalid) valid[f] = ((unsigned __int128)P * s[1] - (unsigned __int128)s[0] * s[0]) > (unsigned __int128)4 * P * P;
    }
}

__global__ __launch_bounds__(256) void k_mask_count(const uint8_t *__restrict__ mask, size_t P, unsigned long long *cnt)
{
    uint32_t c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * 8 < P; i += (size_t)gridDim.x * blockDim.x) {
        uint2 m = *(const uint2 *)(mask + i * 8);
        c += __popc(eq255_4(m.x) & 0x01010101u) + __popc(eq255_4(m.y) & 0x01010101u);
    }
    __shared__ uint32_t red[4];
    c = wave_sum_u32(c);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0 && (red[0] | red[1] | red[2] | red[3])) atomicAdd(cnt, (unsigned long long)red[0] + red[1] + red[2] + red[3]);
}

__global__ void k_fill_spans(uint8_t *mask, int H, int W, const int32_t *spans)
{
    int y = blockIdx.x;
    int xa = spans[2 * y], xb = spans[2 * y + 1];
    for (int x = threadIdx.x; x < W; x += blockDim.x) mask[(size_t)y * W + x] = (x >= xa && x <= xb) ? 255 : 0;
}

__global__ void k_synth(uint8_t *out, int H, int W, int frame0, unsigned long long seed, const int32_t *objs, const int32_t *foff)
{
    int f = blockIdx.y;
    size_t P = (size_t)H * W;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    int y = (int)(i / W), x = (int)(i % W);
    int v = synth_background(x, y, H, W, seed, frame0 + f);
    for (int k = foff[f]; k < foff[f + 1]; k++) v -= synth_object_hit(objs + 8 * (size_t)k, x, y);
    out[(size_t)f * P + i] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

