// piscoseg.hip -- gfx950 kernels and the C ABI of libpiscoseg.so (see include/piscoseg.h).
//
// Hot path of fannz00/ImageProcessing restated for MI355X:
//   dense, HBM-bound:  k_bg_stream / k_bg_generic  (bg_correction.py:65-86 + detection.py:178-202)
//                      k_threshold                  (detection.py:178-202 when cleaned comes from elsewhere)
//                      k_label_fill / k_label_fill_rows   (label map)
//   sparse, on the 1-bit threshold plane: run extraction, union-find over runs, enclosure tree,
//                      quad-count contour area, bbox, ordering (pseg_core.h; detection.py:204-216)
//   per saved object:  cumulative filled mask window, three crops, integer sums behind the 32
//                      region properties (detection.py:243-258, 75-159)
// No MFMA: the path is byte/integer work bounded by HBM traffic and by latency on the sparse chain.
#include <hip/hip_runtime.h>
#include <cfloat>
#include <cmath>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <mutex>
#include <condition_variable>
#include <thread>

#include "../../include/piscoseg.h"
#include "pseg_core.h"
#include "synth_core.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};
struct HostBuf {
    std::vector<uint8_t> v;
    template <class T> T *as(size_t n) { if (v.size() < n * sizeof(T)) v.resize(n * sizeof(T)); return (T *)v.data(); }
};

struct PinBuf {             // page-locked host memory: D2H copies into it are single DMA transfers, no staging
    void *p = nullptr;
    size_t bytes = 0;
};

struct RowsDev {           // device SoA of the kept-object rows
    int32_t *frame, *c, *label, *area2, *x, *y, *w, *h, *saved, *root;
};

struct pseg_ctx {
    int device = 0, H = 0, W = 0, WW = 0, maxF = 0;
    size_t P = 0;
    hipStream_t stream = nullptr; // the context's own stream
    hipStream_t cur = nullptr;    // where work is queued right now: the device's shared streaming stream while the context holds
                                  // the GPU's turn (GpuTurn), else `stream`
    hipEvent_t ev_turn = nullptr; // recorded behind the streaming part on the shared stream; the own stream waits for it
    bool own_stream = false;
    std::string err;
    hipEvent_t ev[9] = {};         // 0..7 time the kernel groups, 8 marks the run count on its way to the host
    float ms[6] = {};
    // dense workspaces
    DevBuf bits, partials, maskpart, rowspans;
    // head arena: every small per-frame table the host needs, contiguous so that ONE copy fetches it.
    // fsums / valid_dev / maskcnt / totals / ftot are views into it (head_setup), not allocations.
    DevBuf head, fsums, valid_dev, maskcnt, totals, ftot;
    size_t head_copy_off = 0, head_copy_bytes = 0;
    int head_n = -1;
    PinBuf pin_head, pin_tail, pin_nfg;
    DevBuf outpack;               // sums | crop_off | rows (10 columns) | saved_row, packed for ONE copy to pin_tail
    bool head_on_host = false;    // pin_head holds this call's head (pseg_process reuses it for the raw statistics)
    // sparse workspaces
    DevBuf rowcnt, rowoff, framebase, scan_state;
    DevBuf run_x, run_fy, fg_parent, ext, gapl, runlabel, bg_parent, bg_left, minx, maxx, maxy, area2, extrank,
        keptrank, savedrank, objlabel, objflag;
    DevBuf rows, saved_row, saved_geo, crop_off, bits_off, maskbits, crops_raw, crops_masked, crops_mask;
    bool mask_counted = false;    // maskcnt already holds the number of mask == 255 pixels (fused background kernel)
    bool have_bits = false;       // threshold plane + masked sums already produced by the bg kernel
    DevBuf bounds, band_off, band_obj /* one 64-byte descriptor per crop band: k_band_map */, trash, rp_small, rp_scratch, resize_tab, synth_objs, synth_off;
    int bounds_n = -1, bounds_emit = -1, bounds_chunks = -1, bounds_batch = -1;
    bool runs_old = false;        // PSEG_RUNS_OLD=1: two-pass run extraction (k_runs) + k_band_label instead of the fused k_runs_band
    bool exclusive = true;        // the device part of a call holds the GPU's turn (PSEG_EXCLUSIVE=0: batches interleave freely)
    bool shared_stream = false;   // PSEG_SHARED_STREAM=1: the streaming parts of all contexts go through one stream per device (GpuTurn)
    bool early_release = true;    // hand the turn on behind the label fill instead of behind the last kernel (PSEG_EARLY_RELEASE=0)
    struct GpuTurn *turn = nullptr;
    int bg_chunks = 1;            // time chunks of the streaming background kernel (PSEG_BG_CHUNKS)
    int bg_resident = 0;          // blocks of the streaming background kernel the device holds at once (0: no tail split)
    int bounds_tailk = -1;
    bool force_generic = false;   // PSEG_BG_GENERIC=1: use the 12x-read reference kernel (debug / cross-check)
    bool no_bands = false;        // PSEG_NO_BANDS=1: label with the global union-find only (the path of very dense frames)
    int fill_rows = -1;           // PSEG_FILL_ROWS: 1 = label fill with one wave per row (k_label_fill_rows), 0 = one wave per
                                  // 256-pixel span, -1 (default) = by the density of the previous batch
    bool dense_rows = false;      // the last batch had FILL_ROWS_MIN or more runs per row on average
    // host results
    HostBuf h_frame_row_offset, h_frame_n_contours, h_frame_valid, h_frame_mean, h_frame_std, h_area, h_saved8, h_props;
    int64_t crop_bytes = 0;
    int n_saved = 0;
    // host views of the last table (page-locked tail buffer): rows [10][t_hr], saved_row, crop_off
    const int32_t *t_rows = nullptr, *t_saved_row = nullptr; const int64_t *t_crop_off = nullptr; size_t t_hr = 1;
    int band_cap = 1024, band_calm = 0;   // LDS size class of k_runs_band (BAND_CAP / _MID / _DENSE) and calm calls in a row
    int spec_hold = 0;            // calls for which the fast path stays off (after a batch too dense for the LDS labelling)
    size_t runcap = 0;            // runs the tables are sized for: the labelling chain is queued against it without a host round trip
    struct AsyncWorker *worker = nullptr;   // pseg_submit / pseg_wait: the context's own host thread (started on first use)
};

static int fail(pseg_ctx *c, int code, const std::string &msg)
{
    if (c) c->err = msg;
    return code;
}
#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(ctx, PSEG_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));         \
    } while (0)

static int ensure(pseg_ctx *ctx, DevBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return 0;
    if (b.p) HIPCHK(hipFree(b.p));
    b.p = nullptr; b.bytes = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIPCHK(hipMalloc(&b.p, want));
    b.bytes = want;
    return 0;
}
#define ENSURE(buf, bytes) do { int r_ = ensure(ctx, (buf), (bytes)); if (r_) return r_; } while (0)

static int ensure_pin(pseg_ctx *ctx, PinBuf &b, size_t bytes)
{
    if (bytes <= b.bytes) return 0;
    if (b.p) HIPCHK(hipHostFree(b.p));
    b.p = nullptr; b.bytes = 0;
    size_t want = bytes + bytes / 4 + 256;
    HIPCHK(hipHostMalloc(&b.p, want, hipHostMallocDefault));
    b.bytes = want;
    return 0;
}
#define ENSURE_PIN(buf, bytes) do { int r_ = ensure_pin(ctx, (buf), (bytes)); if (r_) return r_; } while (0)

static inline size_t sz_stride(size_t cap) { return (cap + 8 + 1) & ~(size_t)1; }   // a size column of k_emit / k_scan_sizes: even, 8 entries of slack
static inline int scan_words(const pseg_ctx *ctx, int n) { return 2 + n * ((ctx->H + 63) / 64); }   // (64 = BAND_H)

// head arena for a batch of n frames (u32 words unless noted):
//   ftot [n] | ftot3 [3n] rowbase [n+1] savedbase [n+1] | pad | fsums [4n] u64 | maskcnt u64 | totals [8] i64 | valid [n] u8
// everything from ftot3 on is what the host reads back (one copy of head_copy_bytes from head_copy_off)
static int head_setup(pseg_ctx *ctx, int n)
{
    if (ctx->head_n == n) return 0;
    size_t o_ftot = 0, o_ftot3 = (size_t)n * 4, o_end32 = o_ftot3 + ((size_t)5 * n + 2) * 4;
    size_t o_fsums = (o_end32 + 7) & ~(size_t)7, o_maskcnt = o_fsums + (size_t)n * 32, o_totals = o_maskcnt + 8;
    size_t o_valid = o_totals + 64, o_end = o_valid + (size_t)n;
    ENSURE(ctx->head, o_end + 64);
    ENSURE(ctx->framebase, (size_t)(n + 3) * 4);             // [n] = total runs, [n+1] = largest band, [n+2] = scan error flag
    ENSURE(ctx->scan_state, (size_t)scan_words(ctx, n) * 8); // look-back scan of k_runs_band: ticket, error flag, one word per band
    ENSURE_PIN(ctx->pin_head, o_end - o_ftot3 + 64);
    ENSURE_PIN(ctx->pin_nfg, 64);
    char *b = (char *)ctx->head.p;
    ctx->ftot.p = b + o_ftot; ctx->fsums.p = b + o_fsums; ctx->maskcnt.p = b + o_maskcnt; ctx->totals.p = b + o_totals;
    ctx->valid_dev.p = b + o_valid;
    ctx->head_copy_off = o_ftot3; ctx->head_copy_bytes = o_end - o_ftot3;
    ctx->head_n = n;
    return 0;
}
// host view of the copied head
struct HeadHost {
    const uint32_t *ftot3, *rowbase, *savedbase;
    const uint64_t *fsums;
    uint64_t maskcnt;
    const long long *totals;
    const uint8_t *valid;
};
static HeadHost head_host(const pseg_ctx *ctx, int n)
{
    const char *b = (const char *)ctx->pin_head.p;            // b[0] corresponds to head offset head_copy_off
    size_t o_end32 = ((size_t)5 * n + 2) * 4 + (size_t)n * 4;
    size_t o_fsums = ((o_end32 + 7) & ~(size_t)7) - ctx->head_copy_off;
    HeadHost h;
    h.ftot3 = (const uint32_t *)b; h.rowbase = h.ftot3 + 3 * (size_t)n; h.savedbase = h.rowbase + (n + 1);
    h.fsums = (const uint64_t *)(b + o_fsums);
    h.maskcnt = *(const uint64_t *)(b + o_fsums + (size_t)n * 32);
    h.totals = (const long long *)(b + o_fsums + (size_t)n * 32 + 8);
    h.valid = (const uint8_t *)(b + o_fsums + (size_t)n * 32 + 72);
    return h;
}

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
// streaming background kernel (window of 12 frames = the reference's 2*n_bg_imgs with n_bg_imgs = 6)
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

// Per-thread state.  The 12-frame history of raw pixels lives in an LDS ring (dynamic slot index,
// private to the thread: no barriers); the sparse table and 4 prefetched frames live in registers
// with compile-time phases (loop unrolled by 4).
struct BgRegs {
    uint2 pf[4];           // frames k .. k+3 (pf[PH] is the frame of the current step)
    Px8 prev;              // frame k-1, split
    Px8 mp, t2[2], t4[4], t8[4], last;
    BgSums acc;            // per-lane sums of the frames emitted by the last (up to) four steps
};
#define BG_SLOTS 16
template <int PH, bool PAD>
__device__ __forceinline__ void bg_step(BgRegs &R, uint2 *ring, const BgEmit &E, const uint8_t *__restrict__ src, int k, int a,
                                        int kend, int n)
{
    const uint2 cur = R.pf[PH];
    int kn = k + 4 <= kend ? k + 4 : kend;                  // clamped prefetch: always a valid frame
    R.pf[PH] = ld_stream_at(src + (size_t)kn * E.P, E.off8);   // src is wave-uniform
    ring[(k & (BG_SLOTS - 1)) * 256] = cur;
    const uint2 old = ring[((k - 11) & (BG_SLOTS - 1)) * 256];   // frame k-11 (zeros during warm-up)
    const Px8 cs = px_split(cur);
    Px8 m = px_min(R.prev, cs);                             // pair minimum m[k-1]
    if (k == a) m = Px8{0, 0, 0, 0};                         // no pair yet at the first frame of the chunk
    R.prev = cs;
    Px8 t2 = px_max(m, R.mp); R.mp = m;
    Px8 t4 = px_max(t2, R.t2[(PH + 1) & 1]); R.t2[(PH + 1) & 1] = t2;
    Px8 t8 = px_max(t4, R.t4[(PH + 3) & 3]); R.t4[(PH + 3) & 3] = t4;
    Px8 t11 = px_max(t8, R.t8[PH & 3]); R.t8[(PH + 3) & 3] = t8;
    R.last = t11;
    bg_emit<PAD>(E, old, px_split(old), t11, k - 11, k - 11 >= a && k < n - 1, R.acc.p[PH], R.acc.s1[PH], R.acc.s3[PH]);
}

// grid = (ceil(P/2048), chunks); chunk c emits frames [bounds[c], bounds[c+1]) of the reference batch that ends at frame
// bounds[chunks + 1 + c] (several batches in one launch: every batch is at least one chunk).  PAD: the last block has lanes
// beyond the frame (P % 2048 != 0).
template <bool PAD>
__global__ __launch_bounds__(256, 4) void k_bg_stream(const uint8_t *__restrict__ frames, size_t P, uint8_t *__restrict__ cleaned,
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
    const int kend = (b == n) ? n - 1 : b + 10;
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
    BgRegs R;
    const Px8 z = {0, 0, 0, 0};
    R.mp = z; R.last = z; R.prev = z;
    for (int i = 0; i < 2; i++) R.t2[i] = z;
    for (int i = 0; i < 4; i++) { R.t4[i] = z; R.t8[i] = z; }
#pragma unroll
    for (int i = 0; i < 4; i++) R.pf[i] = ld_stream(src + (size_t)(a + i <= kend ? a + i : kend) * P + E.off8);
    int k = a;
    for (; k + 3 <= kend; k += 4) {                         // branch-free body: counted waits keep 4 loads in flight
        bg_step<0, PAD>(R, ring, E, src, k, a, kend, n);
        bg_step<1, PAD>(R, ring, E, src, k + 1, a, kend, n);
        bg_step<2, PAD>(R, ring, E, src, k + 2, a, kend, n);
        bg_step<3, PAD>(R, ring, E, src, k + 3, a, kend, n);
        bg_flush(E, R.acc, k - 11, a, n - 12, 4);           // (step k emits frame k - 11, stored while a <= k - 11 and k < n - 1)
    }
    if (k <= kend) {
        const int left = kend - k + 1;                      // 1 .. 3 steps
        bg_step<0, PAD>(R, ring, E, src, k, a, kend, n);
        if (left > 1) bg_step<1, PAD>(R, ring, E, src, k + 1, a, kend, n);
        if (left > 2) bg_step<2, PAD>(R, ring, E, src, k + 2, a, kend, n);
        bg_flush(E, R.acc, k - 11, a, n - 12, left);
    }
    if (b == n) {                                           // the last 12 frames share bg[n-12] = T11 of the final step;
#pragma unroll 1
        for (int i = n - 12; i < n; i += 4) {                // their raw pixels are still in the ring (a <= n - 12)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint2 img = ring[((i + j) & (BG_SLOTS - 1)) * 256];
                bg_emit<PAD>(E, img, px_split(img), R.last, i + j, true, R.acc.p[j], R.acc.s1[j], R.acc.s3[j]);
            }
            bg_flush(E, R.acc, i, n - 12, n, 4);
        }
    }
}

// detection.py:178-202 for a cleaned image that did not come from k_bg_* (e.g. after deconvolution)
__global__ __launch_bounds__(256) void k_threshold(const uint8_t *__restrict__ cleaned, size_t P, const uint8_t *__restrict__ mask,
                                                   int thr, uint8_t *__restrict__ bits8, uint32_t *__restrict__ partials)
{
    size_t strip = (size_t)blockIdx.x * 256 + threadIdx.x;
    int f = blockIdx.y;
    uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (strip * 8 < P) {
        uint2 cl = *(const uint2 *)(cleaned + (size_t)f * P + strip * 8);
        uint2 corr = make_uint2(~cl.x, ~cl.y);
        uint2 inm = make_uint2(0xffffffffu, 0xffffffffu);
        if (mask) { uint2 m = *(const uint2 *)(mask + strip * 8); inm = make_uint2(eq255_4(m.x), eq255_4(m.y)); }
        uint32_t b = tail4(cl.x, corr.x, inm.x, thr, s0, s1, s2, s3) | (tail4(cl.y, corr.y, inm.y, thr, s0, s1, s2, s3) << 4);
        bits8[(size_t)f * (P / 8) + strip] = (uint8_t)b;
    }
    block_sum4_store(s0, s1, s2, s3, partials + ((size_t)f * gridDim.x + blockIdx.x) * 4);
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
    return S.framebase[S.F] <= cap && S.framebase[S.F + 1] <= bcap;
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
#define SCAN_AGG (1ull << 62)
#define SCAN_INC (2ull << 62)
#define SCAN_SPIN_MAX (1 << 22)
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
                                                   uint32_t *__restrict__ bandmax)
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
    if (threadIdx.x == 0) s_ticket = (uint32_t)atomicAdd(&scan[0], 1ull);
    __syncthreads();
    BAND_STAMP(0);                                        // ticket
    const int t = (int)s_ticket;
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
                while ((sv[k] >> 62) == 0 && ++spins < SCAN_SPIN_MAX) {
                    __builtin_amdgcn_s_sleep(2);
                    sv[k] = __hip_atomic_load(&scan[2 + idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if ((sv[k] >> 62) == 0) { timeout = true; sv[k] = SCAN_INC; }   // never expected; the host reports it
                const unsigned long long incm = __ballot((sv[k] >> 62) == 2);
                const int L = incm ? (int)__builtin_ctzll(incm) : 64;
                const uint32_t part_sum = wave_sum63(lane <= L ? (uint32_t)sv[k] : 0u);
                if (lane == 63) { lb_sum[4 * k + wv] = part_sum; lb_hit[4 * k + wv] = incm != 0; }
            }
            if (timeout) bandmax[1] = 1;
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

// ranks of external roots / kept / saved objects inside each frame (block per frame; a thread takes RANK_IPT consecutive
// runs per round, the three counts of a thread travel through the wave scan packed in one register).
// ftot3[f*3+{0,1,2}] = number of external contours, kept, saved.
#define RANK_IPT 4
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
#ifndef CROP_LDS_WORDS
#define CROP_LDS_WORDS 2048          // mask words of a band + 2 halo rows on each side (frames up to 8192 px wide: 8 rows of 256 words)
#endif
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
                       int32_t *saved_row, int32_t *runlabel, int32_t *geo, unsigned long long *sz, size_t szstride)
{
    if (!chain_ok(S, cap, bcap)) return;
    const int nfg = (int)S.framebase[S.F];
    size_t g = GTID;
    if (g >= (size_t)nfg) return;
    int r = (int)g;
    objflag[r] = 0;
    objlabel[r] = 0;
    int f = (int)(S.run_fy[r] >> 16);
    {   // label-map value of this run: the label of its external component, 0 for components nested in holes
        const int E = S.ext[r];
        runlabel[r] = (S.fg_parent[r] == E) ? (int)ftot3[f * 3] - S.extrank[E] : 0;
    }
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
        sz[2 * szstride + sidx] = (unsigned long long)((h + rpb - 1) / rpb);
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
    if (!chain_ok(S, cap, bcap)) return;                    // (queued with the chain: see chain_ok)
    const int lane = threadIdx.x & 63;
    const int f = blockIdx.z, y = blockIdx.y;
    const int x0 = blockIdx.x * 1024 + (threadIdx.x >> 6) * 256, x = x0 + 4 * lane;
    if (x0 >= S.W) return;                                   // wave beyond the row end
    const bool in = x < S.W;
    const size_t rowpix = ((size_t)f * S.H + y) * S.W;
    i32x4_t out = {0, 0, 0, 0};
    // the row's run list (L2-resident tables) decides, not the threshold plane: a read stream next to the write stream
    // costs the stores more than the lookups do
    const int rs = seg_rowstart(S, f, y), n = seg_rowcount(S, f, y);
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

// exclusive scans of the three size columns by one block of 1024 threads; out[k][0..n].  A thread takes 8 consecutive
// objects (one pass of wave / block scans per 8192 objects), read and written as 16-byte vectors: the columns are
// stored one after the other (stride `szstride`, even, with 8 entries of slack behind each column)
#define SCAN_IPT 8
typedef unsigned long long u64x2_t __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(1024) void k_scan_sizes(SegState S, uint32_t cap, uint32_t bcap, const unsigned long long *sz, size_t szstride,
                                                      const uint32_t *n_p, long long *crop_off, long long *bits_off,
                                                      long long *band_off, long long *totals)
{
    __shared__ unsigned long long wtot[16][3];
    __shared__ unsigned long long carry[3];
    // totals[4..6]: runs of the batch, largest band, scan error flag -- what the host needs to accept or repeat the chain
    if (threadIdx.x < 3) totals[4 + threadIdx.x] = (long long)S.framebase[S.F + threadIdx.x];
    if (!chain_ok(S, cap, bcap)) { if (threadIdx.x < 4) totals[threadIdx.x] = 0; return; }
    const int n = (int)*n_p, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 3) carry[threadIdx.x] = 0;
    __syncthreads();
    long long *outp[3] = {crop_off, bits_off, band_off};
    unsigned long long hmax = 0;                             // tallest saved object (LDS size of k_convex): totals[3]
    for (int base = 0; base < n; base += 1024 * SCAN_IPT) {
        const int i0 = base + threadIdx.x * SCAN_IPT;
        unsigned long long v[3][SCAN_IPT], tot[3] = {0, 0, 0}, inc[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (i0 < n) {
                const u64x2_t *p = (const u64x2_t *)(sz + k * szstride + i0);
#pragma unroll
                for (int q = 0; q < SCAN_IPT / 2; q++) { const u64x2_t t = p[q]; v[k][2 * q] = t.x; v[k][2 * q + 1] = t.y; }
            }
#pragma unroll
            for (int j = 0; j < SCAN_IPT; j++) { if (i0 + j >= n) v[k][j] = 0; tot[k] += v[k][j]; }
        }
#pragma unroll
        for (int j = 0; j < SCAN_IPT; j++) hmax = v[1][j] > hmax ? v[1][j] : hmax;   // column 1 = rows of the object
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
            if (i0 + SCAN_IPT <= n) {
                u64x2_t *o = (u64x2_t *)(outp[k] + i0);
#pragma unroll
                for (int q = 0; q < SCAN_IPT / 2; q++) {
                    u64x2_t t; t.x = run; run += v[k][2 * q]; t.y = run; run += v[k][2 * q + 1];
                    o[q] = t;
                }
            } else {
#pragma unroll
                for (int j = 0; j < SCAN_IPT; j++) { if (i0 + j < n) outp[k][i0 + j] = (long long)run; run += v[k][j]; }
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) for (int k = 0; k < 3; k++) carry[k] = pre[k] + inc[k];
        __syncthreads();
    }
    if (threadIdx.x < 3) { outp[threadIdx.x][n] = (long long)carry[threadIdx.x]; totals[threadIdx.x] = (long long)carry[threadIdx.x]; }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long t = __shfl_xor(hmax, o, 64); hmax = t > hmax ? t : hmax; }
    __syncthreads();
    if (lane == 0) wtot[wv][0] = hmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) hmax = wtot[w][0] > hmax ? wtot[w][0] : hmax;
        totals[3] = (long long)hmax;
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
    const long long b0 = band_off[s], b1 = band_off[s + 1], co = crop_off[s], ro = row_off[s];
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
template <bool DENSE>
__global__ __launch_bounds__(CROP_THREADS) void k_crops_sums(SegState S, int n_saved, const int32_t *__restrict__ geo,
                                                    const long long *__restrict__ crop_off, const long long *__restrict__ band_off,
                                                    const int32_t *__restrict__ objflag, const uint8_t *__restrict__ cleaned,
                                                    const uint8_t *__restrict__ raw, uint8_t *__restrict__ c_raw,
                                                    uint8_t *__restrict__ c_masked, uint8_t *__restrict__ c_mask, long long *sums,
                                                    const long long *__restrict__ row_off, int32_t *__restrict__ rowext,
                                                    const int32_t *__restrict__ band_obj, const uint8_t *__restrict__ dense_mask,
                                                    const int32_t *__restrict__ band_desc)
{
    __shared__ uint32_t mbuf[CROP_LDS_WORDS];             // mask bits of rows r0-2 .. r1+1 (zero outside the window)
    __shared__ uint32_t bd[CROP_LDS_WORDS];               // 4-neighbour border bits of rows r0-1 .. r1
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
    const int rpb = crop_rows_per_band(w);
    const int r1 = (r0 + rpb < h) ? r0 + rpb : h;
    const int nrows = r1 - r0;
    const float rcp_wpr = 1.0f / (float)wpr, rcp_wpr1 = 1.0f / (float)(wpr + 1);
    for (int wi = threadIdx.x; wi < (nrows + 4) * wpr; wi += CROP_THREADS) mbuf[wi] = 0;
    if (threadIdx.x < 32) acc[threadIdx.x] = threadIdx.x == 15 ? 255u : 0u;    // acc[15] = running minimum
    __syncthreads();
    if (DENSE) {
        // pseg_regionprops: the mask is a dense uint8 image (nonzero = set) of row stride S.W instead of the run tables
        for (int wi = threadIdx.x; wi < (nrows + 4) * wpr; wi += CROP_THREADS) {
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
        // 16 threads per row walk the row's runs; a run paints itself and the gap on its left when the object that fills
        // them (ext / gapl) is saved and was iterated at or before this one.  Two dependent global loads per row.
        const int sub = threadIdx.x & (CROP_TPR - 1), x_lo = bx, x_hi = bx + w - 1;
        auto paint = [&](uint32_t *mrow, int xa, int xb) {
            int la = (xa > x_lo ? xa : x_lo) - bx, lb = (xb < x_hi ? xb : x_hi) - bx;
            for (int wd = la >> 5; wd <= (lb >> 5); wd++) {
                int ba = la > 32 * wd ? la - 32 * wd : 0, bb = lb < 32 * wd + 31 ? lb - 32 * wd : 31;
                uint32_t m = (bb - ba == 31) ? 0xffffffffu : (((1u << (bb - ba + 1)) - 1u) << ba);
                atomicOr(&mrow[wd], m);
            }
        };
        for (int yl = threadIdx.x / CROP_TPR; yl < nrows + 4; yl += CROP_THREADS / CROP_TPR) {
            const int yy = r0 - 2 + yl;
            if (yy < 0 || yy >= h) continue;
            const int rs = seg_rowstart(S, f, by + yy), n = seg_rowcount(S, f, by + yy);
            uint32_t *mrow = mbuf + yl * wpr;
            for (int j = sub; j < n; j += CROP_TPR) {
                const int r = rs + j;
                const uint32_t rx = S.run_x[r];
                const int xs = (int)(rx & 0xffffu), xe = (int)(rx >> 16);
                const int pe = j > 0 ? run_xe(S, r - 1) : x_hi;      // j == 0: the leading gap belongs to no object
                const int idr = S.ext[r], idg = S.gapl[r];
                // (the object itself is saved by definition: the flag is only looked up for other objects inside the window)
                if (xe >= x_lo && xs <= x_hi && (idr == root || (idr > root && (objflag[idr] & 2)))) paint(mrow, xs, xe);
                if (xs - 1 >= x_lo && pe + 1 <= x_hi && (idg == root || (idg > root && (objflag[idg] & 2)))) paint(mrow, pe + 1, xs - 1);
            }
        }
    }
    __syncthreads();
    // border = mask minus its erosion by the 4-neighbour cross, zero outside the window
    // (skimage perimeter(image, 4): binary_erosion(..., border_value=0), _regionprops_utils.py:225-231)
    for (int wi = threadIdx.x; wi < (nrows + 2) * wpr; wi += CROP_THREADS) {
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
    if (w >= 16) {
        // thread = 16 consecutive pixels of a row (the 8-pixel form below with 128-bit accesses)
        const int cpr = (w + 15) >> 4, nitem = nrows * cpr;
        const float rcp = 1.0f / (float)cpr;
#pragma unroll CROP_UNROLL
        for (int p = threadIdx.x; p < nitem; p += CROP_THREADS) {
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
        for (int p = threadIdx.x; p < nitem; p += CROP_THREADS) {
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
        for (int p = threadIdx.x; p < nitem; p += CROP_THREADS) {
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
        for (int p = threadIdx.x; p < nrows * w; p += CROP_THREADS) {
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
    for (int wi = threadIdx.x; wi < nrows * wpr; wi += CROP_THREADS) {
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
        for (int wi = threadIdx.x; wi < ylast * (wpr + 1); wi += CROP_THREADS) {
            int yl, wx;
            divmod_small(wi, wpr + 1, rcp_wpr1, yl, wx);
            const uint32_t *B = mbuf + __umul24((uint32_t)(yl + 2), (uint32_t)wpr), *T = B - wpr;     // bottom row r0+yl, top row r0+yl-1
            uint32_t tr = wx < wpr ? T[wx] : 0u, br = wx < wpr ? B[wx] : 0u;
            uint32_t tl = (tr << 1) | (wx > 0 ? T[wx - 1] >> 31 : 0u), bl = (br << 1) | (wx > 0 ? B[wx - 1] >> 31 : 0u);
            eul += __popc(tl & ~tr & ~bl & ~br) - __popc(~tl & tr & bl & ~br) - __popc(tl & tr & bl & ~br);
        }
    }
    // first / last set column of every row of the band (input of the convex hull kernel)
    for (int yl = threadIdx.x; yl < nrows; yl += CROP_THREADS) {
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
    if (threadIdx.x == 0) {
        unsigned long long *q = (unsigned long long *)(sums + (size_t)s * PSEG_NSUMS);
        const unsigned long long N = acc[0], R0 = (unsigned long long)r0;
        if (N) {
            const unsigned long long sx = 32ull * acc[1] + acc[3];                                    // sum c
            const unsigned long long sx2 = 1024ull * acc[2] + 64ull * acc[4] + acc[5];                // sum c^2
            const unsigned long long sy = acc[6], sy2 = acc[7], syx = 32ull * acc[8] + acc[9];        // band-local rows
            atomicAdd(&q[0], N); atomicAdd(&q[1], R0 * N + sy); atomicAdd(&q[2], sx);
            atomicAdd(&q[3], R0 * R0 * N + 2ull * R0 * sy + sy2); atomicAdd(&q[4], sx2); atomicAdd(&q[5], R0 * sx + syx);
        }
        if (acc[10]) atomicAdd(&q[8], (unsigned long long)acc[10]);
        if (acc[15] < 255u) atomicMin((long long *)&q[6], (long long)acc[15]);
        if (acc[16] > 0u) atomicMax((long long *)&q[7], (long long)acc[16]);
        if (acc[11]) atomicAdd(&q[10], (unsigned long long)acc[11]);
        if (acc[12]) atomicAdd(&q[11], (unsigned long long)acc[12]);
        if (acc[13]) atomicAdd(&q[12], (unsigned long long)acc[13]);
        if (acc[14]) atomicAdd(&q[13], (unsigned long long)(long long)(int)acc[14]);
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
    for (int s = 1; s < 64; s <<= 1) {                      // merge the hulls of lanes [l, l+s) and [l+s, l+2s)
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
    if (threadIdx.x < 64) hull_chain<true>(pvl, np, nxl, prl, fl, h, threadIdx.x);
    else hull_chain<false>(pvu, np, nxu, pru, fl, h, threadIdx.x - 64);
    __syncthreads();
    unsigned long long cnt = 0;
    for (int r = threadIdx.x; r < h; r += 128)
        if (fl[2 * r + 1] >= fl[2 * r]) cnt += (unsigned long long)(fl[2 * r + 1] - fl[2 * r] + 1);
    cnt = wave_sum_u64(cnt);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd((unsigned long long *)&q[14], cnt);
}
__global__ void k_init_sums(long long *sums, int n_saved)
{
    size_t g = GTID;
    if (g >= (size_t)n_saved * PSEG_NSUMS) return;
    sums[g] = (g % PSEG_NSUMS == 6) ? 255 : 0;             // min intensity starts at 255
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline unsigned nblocks(size_t n, unsigned bs = 256) { return (unsigned)((n + bs - 1) / bs); }
static hipError_t rp_static_lookup(int device);              // device address of the no-alloc entry point's scratch array
static void async_stop(pseg_ctx *ctx);                       // ends the context's worker thread (pseg_submit / pseg_wait)

extern "C" int pseg_create(int device, int H, int W, int max_frames, pseg_ctx **out)
{
    if (!out) return PSEG_ERR_ARG;
    *out = nullptr;
    pseg_ctx *ctx = new pseg_ctx();
    // (row counts of a batch are 32-bit: max_frames * (H + 1) must stay below 2^31)
    if (H < 2 || W < 64 || (W % 64) || H > 65535 || W > 8192 || max_frames < 1 || max_frames > 65535 ||
        (long long)max_frames * (H + 1) >= (1LL << 31)) {
        delete ctx;
        return PSEG_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device || device < 0) { delete ctx; return PSEG_ERR_NODEVICE; }
    ctx->device = device; ctx->H = H; ctx->W = W; ctx->WW = W / 64; ctx->maxF = max_frames; ctx->P = (size_t)H * W;
    if (hipSetDevice(device) != hipSuccess) { delete ctx; return PSEG_ERR_HIP; }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return PSEG_ERR_HIP; }
    // (non-blocking streams: a caller that synchronises the legacy default stream -- torch's current stream, before it hands
    // over its tensors -- must not wait for the batches other contexts have in flight on the library's streams)
    ctx->own_stream = true;
    ctx->cur = ctx->stream;
    if (hipEventCreateWithFlags(&ctx->ev_turn, hipEventDisableTiming) != hipSuccess) ctx->ev_turn = nullptr;
    if (rp_static_lookup(device) != hipSuccess) { hipStreamDestroy(ctx->stream); delete ctx; return PSEG_ERR_HIP; }
    if (const char *e = getenv("PSEG_BG_CHUNKS")) ctx->bg_chunks = atoi(e) > 0 ? atoi(e) : 1;
    if (const char *e = getenv("PSEG_BG_GENERIC")) ctx->force_generic = atoi(e) != 0;
    {
        hipDeviceProp_t prop;                                // 4 blocks per CU (the kernel's launch bounds): see run_bg
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->bg_resident = 4 * prop.multiProcessorCount;
        if (const char *e = getenv("PSEG_BG_TAIL")) ctx->bg_resident = atoi(e) > 0 ? atoi(e) : 0;   // 0: off, N: pretend N slots
    }
    if (const char *e = getenv("PSEG_EXCLUSIVE")) ctx->exclusive = atoi(e) != 0;
    if (const char *e = getenv("PSEG_EARLY_RELEASE")) ctx->early_release = atoi(e) != 0;
    if (const char *e = getenv("PSEG_SHARED_STREAM")) ctx->shared_stream = atoi(e) != 0;
    if (const char *e = getenv("PSEG_RUNS_OLD")) ctx->runs_old = atoi(e) != 0;
    if (const char *e = getenv("PSEG_NO_BANDS")) ctx->no_bands = atoi(e) != 0;
    if (const char *e = getenv("PSEG_FILL_ROWS")) ctx->fill_rows = atoi(e);
    for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { pseg_destroy(ctx); return PSEG_ERR_HIP; }   // frees what exists so far
    *out = ctx;
    return PSEG_OK;
}

extern "C" int pseg_destroy(pseg_ctx *ctx)
{
    if (!ctx) return PSEG_ERR_ARG;
    async_stop(ctx);
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
#ifdef BAND_PROF
    {
        unsigned long long p[16] = {0};
        if (hipMemcpyFromSymbol(p, HIP_SYMBOL(g_band_prof), sizeof p) == hipSuccess && p[15]) {
            fprintf(stderr, "k_runs_band phases, us per block over %llu blocks:", p[15]);
            const char *nm[6] = {"ticket", "load+count", "runs->LDS", "label", "look-back", "publish"};
            for (int k = 0; k < 6; k++) fprintf(stderr, " %s %.2f", nm[k], 0.01 * (double)p[k] / (double)p[15]);
            fprintf(stderr, "\n");
        }
    }
#endif
    DevBuf *bufs[] = {&ctx->bits, &ctx->partials, &ctx->maskpart, &ctx->head, &ctx->outpack, &ctx->rowspans, &ctx->rowcnt,
                      &ctx->rowoff, &ctx->framebase, &ctx->scan_state, &ctx->run_x, &ctx->run_fy, &ctx->fg_parent, &ctx->ext, &ctx->gapl, &ctx->runlabel,
                      &ctx->bg_parent, &ctx->bg_left, &ctx->minx, &ctx->maxx, &ctx->maxy, &ctx->area2,
                      &ctx->extrank, &ctx->keptrank, &ctx->savedrank, &ctx->objlabel, &ctx->objflag, &ctx->rows,
                      &ctx->saved_row, &ctx->saved_geo, &ctx->crop_off, &ctx->bits_off, &ctx->maskbits,
                      &ctx->crops_raw, &ctx->crops_masked, &ctx->crops_mask, &ctx->bounds, &ctx->band_off, &ctx->band_obj, &ctx->trash, &ctx->rp_small, &ctx->rp_scratch, &ctx->resize_tab,
                      &ctx->synth_objs, &ctx->synth_off};
    for (DevBuf *b : bufs) if (b->p) hipFree(b->p);
    for (PinBuf *b : {&ctx->pin_head, &ctx->pin_tail, &ctx->pin_nfg}) if (b->p) hipHostFree(b->p);
    for (auto &e : ctx->ev) if (e) hipEventDestroy(e);
    if (ctx->ev_turn) hipEventDestroy(ctx->ev_turn);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return PSEG_OK;
}

extern "C" const char *pseg_last_error(const pseg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" float pseg_last_ms(const pseg_ctx *ctx, int which) { return (ctx && which >= 0 && which < 6) ? ctx->ms[which] : -1.f; }
extern "C" long long pseg_tuning_state(const pseg_ctx *ctx, int which)
{
    if (!ctx) return -1;
    switch (which) {
    case 0: return (long long)ctx->runcap;
    case 1: return ctx->band_cap;
    case 2: return ctx->spec_hold;
    case 3: return ctx->fill_rows >= 0 ? (ctx->fill_rows > 0) : (ctx->dense_rows ? 1 : 0);
    default: return -1;
    }
}

extern "C" int pseg_set_stream(pseg_ctx *ctx, void *hip_stream)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    ctx->cur = ctx->stream;
    ctx->own_stream = false;
    return PSEG_OK;
}

static int check_batch(pseg_ctx *ctx, int n)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (n < 1 || n > ctx->maxF) return fail(ctx, PSEG_ERR_ARG, "n out of range for this context");
    hipError_t e = hipSetDevice(ctx->device);
    if (e != hipSuccess) return fail(ctx, PSEG_ERR_HIP, hipGetErrorString(e));
    return 0;
}

// ---- one batch on the GPU at a time ---------------------------------------------------------------------------------
// Several contexts (host threads) share a GPU so that the host part of one batch (table formatting, 0.4 ms) hides under the
// kernels of another.  Letting the kernels of two batches interleave as well is a loss on this path: the streaming kernels
// need every wave slot and the whole memory system (two label fills side by side take 2.9 ms instead of 2 x 0.8 ms; a fill
// next to the other batch's labelling kernels 1.4 ms, measured with rocprofv3 kernel traces, profiles/README.md).  The
// streaming part of a call (background kernel, labelling, label fill) therefore runs under a per-device turn.  The turn is
// handed on behind the label fill: the crop / property / hull kernels that follow wait on memory latency, and the next
// batch's background kernel runs beside them at a small loss (0.92 ms for both against 0.65 + 0.38 ms one after the other).
// Other pairings were traced and dropped: the crop chain beside the next batch's labelling kernels (both stretch: 0.64 ms
// against 0.40 + 0.38 ms, and the background kernel alone again) comes out level.
#define PSEG_MAX_DEVICES 64
// first come, first served: with a plain mutex a context that has just finished can overtake one that has been waiting, and a
// caller that collects its results in submission order (pseg_submit / pseg_wait) then sits on the overtaken one while the
// others have nothing queued -- the pipeline runs dry (80 us of idle GPU per batch in the kernel trace)
struct TurnLock {
    std::mutex m;
    std::condition_variable cv;
    unsigned long long next = 0, serving = 0;
    void lock()
    {
        std::unique_lock<std::mutex> lk(m);
        const unsigned long long my = next++;
        cv.wait(lk, [&] { return serving == my; });
    }
    void unlock()
    {
        { std::lock_guard<std::mutex> lk(m); serving++; }
        cv.notify_all();
    }
};
static TurnLock g_turn_mu[PSEG_MAX_DEVICES];
static hipEvent_t g_turn_ev[PSEG_MAX_DEVICES];              // recorded behind the last streaming kernel of the previous holder
static bool g_turn_ev_set[PSEG_MAX_DEVICES];
static hipStream_t g_turn_last[PSEG_MAX_DEVICES];            // the stream that event was recorded on
static hipStream_t g_main_stream[PSEG_MAX_DEVICES];          // the device's shared streaming stream
static bool g_main_tried[PSEG_MAX_DEVICES];
static hipStream_t g_tail_stream[PSEG_MAX_DEVICES][2];       // where the tails of consecutive calls go, alternating
static unsigned g_tail_next[PSEG_MAX_DEVICES];
// The streaming parts of consecutive batches (background kernel ... label fill) are queued on ONE stream per device, in
// turn order: the next batch's background kernel follows the label fill in the same hardware queue with no cross-queue
// dependency in between (a wait on another queue's event cost about 70 us of idle device per batch in the kernel trace).
// The latency-bound tail of a call (crops, properties, hulls, copies) goes to one of two tail streams of the device (taken
// in turns), behind an event recorded at the hand-over, and runs beside the next batch's background kernel.  Three streams
// per device whatever the number of contexts: a process has few hardware queues (GPU_MAX_HW_QUEUES, 4 by default), and
// streams that share one run in order -- with a stream per context the tail of one batch and the background kernel of the
// next landed in the same queue and ran one after the other.  A context on a caller's stream
// (pseg_set_stream) keeps everything on that stream and is ordered against the others through g_turn_ev.
struct GpuTurn {
    pseg_ctx *ctx; bool held = false;
    explicit GpuTurn(pseg_ctx *c) : ctx(c)
    {
        if (!c || !c->exclusive || c->turn || c->device < 0 || c->device >= PSEG_MAX_DEVICES) return;
        const int d = c->device;
        g_turn_mu[d].lock();
        held = true; c->turn = this;
        if (c->own_stream && c->ev_turn && c->shared_stream) {
            if (!g_main_tried[d]) {
                g_main_tried[d] = true;
                if (hipStreamCreateWithFlags(&g_main_stream[d], hipStreamNonBlocking) != hipSuccess) g_main_stream[d] = nullptr;
                for (auto &t : g_tail_stream[d]) if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) t = nullptr;
            }
            if (g_main_stream[d]) c->cur = g_main_stream[d];
        }
        if (g_turn_ev_set[d] && g_turn_last[d] != c->cur) (void)hipStreamWaitEvent(c->cur, g_turn_ev[d], 0);
    }
    void release()                                           // call after the last kernel of the streaming part is queued
    {
        if (!held) return;
        const int d = ctx->device;
        if (!g_turn_ev_set[d]) g_turn_ev_set[d] = hipEventCreateWithFlags(&g_turn_ev[d], hipEventDisableTiming) == hipSuccess;
        if (g_turn_ev_set[d]) { (void)hipEventRecord(g_turn_ev[d], ctx->cur); g_turn_last[d] = ctx->cur; }
        else (void)hipStreamSynchronize(ctx->cur);           // no event: hand over on the host
        if (ctx->cur != ctx->stream) {                       // the rest of the call runs on a tail stream, behind what is queued
            hipStream_t t = g_tail_stream[d][g_tail_next[d]++ & 1];
            if (!t) t = ctx->stream;
            (void)hipEventRecord(ctx->ev_turn, ctx->cur);
            (void)hipStreamWaitEvent(t, ctx->ev_turn, 0);
            ctx->cur = t;
        }
        held = false; ctx->turn = nullptr;
        g_turn_mu[d].unlock();
    }
    ~GpuTurn() { release(); if (ctx) ctx->cur = ctx->stream; }
};

// background kernel launch (+ reduction of the raw sums; masked sums too when fused).
// n_avail frames are readable, the first n_emit are emitted.  n_emit == n_avail: a whole reference batch (the last 12
// frames share one window, bg_correction.py:72).  n_emit < n_avail: the leading part of a batch cut for another GPU;
// the rest is the forward halo (>= window - 1 frames), so no emitted frame reaches the clamp -- the kernels are simply
// told the batch is one frame longer than what they may read.
// batch_len > 0: the n_avail = n_emit frames are consecutive reference batches of batch_len frames (the last one may be
// shorter); every batch gets its own end-of-batch window (segmenter.py:209-213 cuts the file list, bg_correction.py:72 clamps
// inside a batch).
static int run_bg(pseg_ctx *ctx, const uint8_t *frames, int n_avail, int n_emit, int n_bg_imgs, const uint8_t *mask, int thr,
                  uint8_t *cleaned, int batch_len = 0)
{
    int NW = 2 * n_bg_imgs;                                   // bg_correction.py:65
    const bool part = n_emit < n_avail;
    const int n = n_emit;
    if (batch_len <= 0 || batch_len >= n_avail) batch_len = 0;
    const int nbatch = batch_len ? (n_avail + batch_len - 1) / batch_len : 1;
    const int last_len = batch_len ? n_avail - (nbatch - 1) * batch_len : n_avail;
    if (batch_len && part) return fail(ctx, PSEG_ERR_ARG, "a part of a batch cannot be split into batches");
    if (n_bg_imgs < 1 || 2 * last_len < NW || (batch_len && 2 * batch_len < NW))
        return fail(ctx, PSEG_ERR_ARG, "need n >= n_bg_imgs (reference raises IndexError below that)");
    if (part && n_avail - n_emit < NW - 1) return fail(ctx, PSEG_ERR_ARG, "a part of a batch needs a forward halo of 2 * n_bg_imgs - 1 frames");
    const int n_kernel = part ? n_avail + 1 : n_avail;       // batch length as the kernels see it
    size_t P = ctx->P;
    unsigned nblk = nblocks(P / 8);
    const bool stream = (NW == 12 && last_len >= 12 && (!batch_len || batch_len >= 12) && !ctx->force_generic);
    unsigned npart = stream ? nblk * 4 : nblk;               // partial sums per frame: per wave / per block
    ENSURE(ctx->bits, (size_t)n * P / 8);
    ENSURE(ctx->partials, (size_t)n * npart * 4 * sizeof(uint32_t));
    ENSURE(ctx->maskpart, (size_t)npart * 4 + 16);
    { int r_ = head_setup(ctx, n); if (r_) return r_; }
    int chunks = 1, tailk = 1;
    unsigned nfull = nblk;                                   // strips whose block walks its whole chunk
    if (stream) {
        ENSURE(ctx->trash, P + P / 8 + (size_t)npart * 16 + 256);
        // chunk bounds [chunks + 1] followed by the end of the batch each chunk lies in [chunks]
        std::vector<int> bnd;
        if (batch_len) {
            chunks = nbatch;
            bnd.resize(2 * chunks + 1);
            for (int c = 0; c < chunks; c++) { bnd[c] = c * batch_len; bnd[chunks + 1 + c] = (c + 1) * batch_len < n_avail ? (c + 1) * batch_len : n_avail; }
            bnd[chunks] = n_avail;
        } else {
            chunks = ctx->bg_chunks;
            const int span = part ? n_emit : n_avail - 12;    // chunk starts lie in [0, span)
            if (chunks > span) chunks = span;
            if (chunks < 1) chunks = 1;
            bnd.resize(2 * chunks + 1);
            for (int c = 0; c < chunks; c++) { bnd[c] = (int)((long long)c * span / chunks); bnd[chunks + 1 + c] = n_kernel; }
            bnd[chunks] = n_emit;
            // All blocks take the same time, so they run in rounds of bg_resident blocks and the last round is only partly
            // filled (2560^2: 3200 strips on 1024 slots = 3 rounds + 128 strips).  The strips of that round are shared by
            // tailk blocks each, which walk tailk shorter time ranges (11 halo frames more per range), so that the device
            // stays filled to the end: 0.611 -> 0.585 ms.  Measured level (0.584 - 0.592 ms): 2944 - 3072 whole strips with
            // the rest shared by 3 - 7 blocks; 2560 whole strips and 2 - 7 blocks for the rest: 0.591 - 0.605 ms.
            const int res = ctx->bg_resident;
            if (chunks == 1 && res > 0 && nblk > (unsigned)res && nblk % res) {
                const unsigned rem = nblk % res;
                tailk = (int)(res / rem);
                if (tailk > 4) tailk = 4;
                if (tailk > span / 24) tailk = span / 24;
                if (tailk > 1) {
                    nfull = nblk - rem;
                    for (int c = 0; c < tailk; c++) bnd.push_back((int)((long long)c * span / tailk));
                    bnd.push_back(n_emit);
                } else tailk = 1;
            }
        }
        if (ctx->bounds_n != n_avail || ctx->bounds_emit != n_emit || ctx->bounds_chunks != chunks || ctx->bounds_batch != batch_len ||
            ctx->bounds_tailk != tailk) {
            ENSURE(ctx->bounds, bnd.size() * 4);
            HIPCHK(hipMemcpyAsync(ctx->bounds.p, bnd.data(), bnd.size() * 4, hipMemcpyHostToDevice, ctx->cur));
            HIPCHK(hipStreamSynchronize(ctx->cur));
            ctx->bounds_n = n_avail; ctx->bounds_emit = n_emit; ctx->bounds_chunks = chunks; ctx->bounds_batch = batch_len;
            ctx->bounds_tailk = tailk;
        }
    }
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->cur));
    if (stream && P % 2048 == 0)
        hipLaunchKernelGGL(k_bg_stream<false>, dim3(nfull + (nblk - nfull) * tailk, chunks), dim3(256), 0, ctx->cur, frames, P, cleaned,
                           mask, thr, (uint8_t *)ctx->bits.p, (uint4 *)ctx->partials.p, (const int *)ctx->bounds.p, (uint8_t *)ctx->trash.p,
                           (uint32_t *)ctx->maskpart.p, (int)nfull, tailk);
    else if (stream)
        hipLaunchKernelGGL(k_bg_stream<true>, dim3(nfull + (nblk - nfull) * tailk, chunks), dim3(256), 0, ctx->cur, frames, P, cleaned,
                           mask, thr, (uint8_t *)ctx->bits.p, (uint4 *)ctx->partials.p, (const int *)ctx->bounds.p, (uint8_t *)ctx->trash.p,
                           (uint32_t *)ctx->maskpart.p, (int)nfull, tailk);
    else
        for (int c = 0; c < nbatch; c++) {                   // the re-reading kernel takes one batch per launch
            const size_t f0 = (size_t)c * (batch_len ? batch_len : 0);
            const int nc = batch_len ? ((c + 1) * batch_len < n_avail ? batch_len : last_len) : n;
            hipLaunchKernelGGL(k_bg_generic, dim3(nblk, nc), dim3(256), 0, ctx->cur, frames + f0 * P, batch_len ? nc : n_kernel, NW, P,
                               cleaned + f0 * P, mask, thr, (uint8_t *)ctx->bits.p + f0 * (P / 8), (uint32_t *)ctx->partials.p + f0 * npart * 4);
        }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->cur));
    hipLaunchKernelGGL(k_reduce_partials, dim3(n), dim3(1024), 0, ctx->cur, (const uint32_t *)ctx->partials.p, (int)npart,
                       (unsigned long long)P, (unsigned long long *)ctx->fsums.p, (uint8_t *)ctx->valid_dev.p, 1, 0, 4,
                       (unsigned long long *)ctx->maskcnt.p, (uint32_t *)ctx->framebase.p + n + 1,
                       (const uint32_t *)ctx->maskpart.p, (stream && mask) ? (int)npart : 0,
                       (unsigned long long *)ctx->scan_state.p, scan_words(ctx, n));
    ctx->mask_counted = stream && mask;
    HIPCHK(hipGetLastError());
    return 0;
}

static void finish_meanstd(const uint64_t *s, double cnt, double *mean, double *std_)
{
    // exact integers: var = (cnt*sum2 - sum^2) / cnt^2
    double m = (double)s[0] / cnt;
    long double num = (long double)cnt * (long double)s[1] - (long double)s[0] * (long double)s[0];
    double var = (double)(num / ((long double)cnt * (long double)cnt));
    *mean = m;
    *std_ = sqrt(var < 0 ? 0 : var);
}

static int fetch_head(pseg_ctx *ctx)
{
    HIPCHK(hipMemcpyAsync(ctx->pin_head.p, (const char *)ctx->head.p + ctx->head_copy_off, ctx->head_copy_bytes,
                          hipMemcpyDeviceToHost, ctx->cur));
    return 0;
}

static int fetch_raw_stats(pseg_ctx *ctx, int n, double *mean_std, uint8_t *valid)
{
    if (!ctx->head_on_host) {
        int r = fetch_head(ctx);
        if (r) return r;
        HIPCHK(hipStreamSynchronize(ctx->cur));
    }
    HeadHost hh = head_host(ctx, n);
    for (int f = 0; f < n; f++) {
        if (mean_std) finish_meanstd(hh.fsums + 4 * f, (double)ctx->P, &mean_std[2 * f], &mean_std[2 * f + 1]);
        if (valid) valid[f] = hh.valid[f];
    }
    return 0;
}

extern "C" int pseg_bgcorrect(pseg_ctx *ctx, const uint8_t *frames, int n, int n_bg_imgs, uint8_t *cleaned, double *mean_std,
                              uint8_t *valid)
{
    int r = check_batch(ctx, n);
    if (r) return r;
    if (!frames || !cleaned) return fail(ctx, PSEG_ERR_ARG, "null frame pointer");
    ctx->have_bits = false;
    ctx->head_on_host = false;
    GpuTurn turn(ctx);
    r = run_bg(ctx, frames, n, n, n_bg_imgs, nullptr, 255, cleaned);
    if (r) return r;
    r = fetch_raw_stats(ctx, n, mean_std, valid);
    if (r) return r;
    HIPCHK(hipEventElapsedTime(&ctx->ms[0], ctx->ev[0], ctx->ev[1]));
    return PSEG_OK;
}

// the 32 columns of regionprop2zooprocess (detection.py:84-152, scikit-image definitions per SURVEY.md 8(a)) from the exact
// integer sums of one window; q[9] is set to the filled area
static void props_from_sums(int64_t *q, double w, double h, double *p)
{
    q[9] = q[0];                                         // filled area (see FA below)
    const double N = (double)q[0];
    const double cr = q[1] / N, cc = q[2] / N;
    const double mu20 = q[3] / N - cr * cr, mu02 = q[4] / N - cc * cc, mu11 = q[5] / N - cr * cc;
    const double a = mu02, b = -mu11, c = mu20;       // inertia tensor [[a,b],[b,c]]
    const double tr = 0.5 * (a + c), dt = sqrt(0.25 * (a - c) * (a - c) + b * b);
    double l1 = tr + dt, l2 = tr - dt;
    if (l1 < 0) l1 = 0;
    if (l2 < 0) l2 = 0;
    const double major = 4 * sqrt(l1), minor = 4 * sqrt(l2);
    double orient = (a - c == 0) ? (b < 0 ? -M_PI / 4 : M_PI / 4) : 0.5 * atan2(-2 * b, c - a);
    // detection.py:84-152 in emitted order; scikit-image definitions per SURVEY.md 8(a) R9-table
    const double FA = N;            // filled_area: a window of hole-filled objects has no 8-connected holes
    const double perim = (double)q[10] + (double)q[11] * sqrt(2.0) + (double)q[12] * ((1.0 + sqrt(2.0)) / 2.0);
    const double meanI = q[8] / N;
    const double conv = q[14] >= 0 ? (double)q[14] : NAN;
    p[0] = w; p[1] = h; p[2] = 0; p[3] = 0;
    p[4] = (4 * M_PI * FA) / (perim * perim);          // circ.
    p[5] = N;                                            // area_exc
    p[6] = FA;                                           // area_rprops
    p[7] = 1 - N / FA;                                   // %area
    p[8] = major; p[9] = minor; p[10] = cr; p[11] = cc;
    p[12] = conv;                                        // convex_area
    p[13] = (double)q[6]; p[14] = (double)q[7]; p[15] = meanI;
    p[16] = FA * meanI;                                  // intden
    p[17] = perim;                                       // perim.
    p[18] = major / minor;                               // elongation (inf / nan like np.divide)
    p[19] = (double)(uint8_t)(q[7] - q[6]);              // range (uint8 arithmetic)
    p[20] = perim / N;                                   // perimareaexc
    p[21] = perim / major;                               // perimmajor
    p[22] = (4 * M_PI * N) / (perim * perim);            // circex
    p[23] = orient / M_PI * 180 + 90;                    // angle
    p[24] = w * h;                                       // bounding_box_area
    p[25] = l1 == 0 ? 0 : sqrt(1 - l2 / l1);             // eccentricity
    p[26] = sqrt(4 * N / M_PI);                          // equivalent_diameter
    p[27] = (double)q[13];                               // euler_number
    p[28] = N / (w * h);                                 // extent
    p[29] = cc; p[30] = cr;                              // local_centroid_col / _row
    p[31] = N / conv;                                    // solidity
}

// rows (10 columns of stride rcap -> stride n_rows), saved_row and crop_off into the packed output arena
__global__ __launch_bounds__(256) void k_pack_out(const int32_t *__restrict__ rows, size_t rcap, int n_rows,
                                                  const int32_t *__restrict__ saved_row, const long long *__restrict__ crop_off,
                                                  int n_saved, int32_t *__restrict__ o_rows, int32_t *__restrict__ o_saved_row,
                                                  long long *__restrict__ o_crop_off)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_rows) {
#pragma unroll
        for (int k = 0; k < 10; k++) o_rows[(size_t)k * n_rows + i] = rows[(size_t)k * rcap + i];
    }
    if (i < n_saved) o_saved_row[i] = saved_row[i];
    if (i <= n_saved) o_crop_off[i] = crop_off[i];
}

// everything after the threshold plane exists.  valid flags are in ctx->valid_dev.
static int run_detect(pseg_ctx *ctx, const uint8_t *cleaned, const uint8_t *raw, const uint8_t *mask, int n,
                      double min_seg, double min_save, int32_t *labels, pseg_table *out)
{
    const int H = ctx->H, W = ctx->W, WW = ctx->WW;
    const size_t P = ctx->P;
    SegState S;
    memset(&S, 0, sizeof(S));
    S.F = n; S.H = H; S.W = W; S.WW = WW;
    S.bits = (const uint64_t *)ctx->bits.p;
    S.valid = (const uint8_t *)ctx->valid_dev.p;
    ENSURE(ctx->rowcnt, (size_t)n * H * 4);
    ENSURE(ctx->rowoff, (size_t)n * (H + 1) * 4);
    { int r_ = head_setup(ctx, n); if (r_) return r_; }
    S.rowcnt = (uint32_t *)ctx->rowcnt.p; S.rowoff = (uint32_t *)ctx->rowoff.p; S.framebase = (uint32_t *)ctx->framebase.p;
    uint32_t *ftot = (uint32_t *)ctx->ftot.p;            // [n] run totals | [3n] rank totals | [n+1] rowbase | [n+1] savedbase
    uint32_t *ftot3 = ftot + n, *rowbase = ftot3 + 3 * (size_t)n, *savedbase = rowbase + (n + 1);

    HIPCHK(hipEventRecord(ctx->ev[2], ctx->cur));
    // (the counters were cleared by k_reduce_partials; the mask pixels were counted by the background kernel when it ran fused)
    if (mask && !ctx->mask_counted)
        hipLaunchKernelGGL(k_mask_count, dim3(1024), dim3(256), 0, ctx->cur, mask, P, (unsigned long long *)ctx->maskcnt.p);
    const int runs_wpl = (S.WW + 3) / 4;                     // words per lane of k_runs_band
    const bool fused = !(ctx->runs_old || runs_wpl > 16);    // rows wider than 4096 px take the two-pass extraction (k_runs)
    const int nbpf = (H + BAND_H - 1) / BAND_H;
    uint32_t *bandmax_d = S.framebase + n + 1;
    unsigned long long *scan = (unsigned long long *)ctx->scan_state.p;
    const dim3 runs_grid(nblocks((size_t)n * H, 4 * RUNS_RPW));
    RowsDev R;
    int32_t *objlabel = nullptr, *objflag = nullptr, *saved_row = nullptr, *geo = nullptr;
    unsigned long long *sz = nullptr;
    long long *totals = (long long *)ctx->totals.p;
    size_t rcap = 1;
    // every table indexed by run (or by object: there are at most as many objects as runs) for `cap` runs
    auto ensure_tables = [&](size_t cap) -> int {
        const size_t nbgc = cap + (size_t)n * H;
        ENSURE(ctx->run_x, cap * 4); ENSURE(ctx->run_fy, cap * 4); ENSURE(ctx->fg_parent, cap * 4); ENSURE(ctx->ext, cap * 4); ENSURE(ctx->gapl, cap * 4);
        ENSURE(ctx->bg_parent, nbgc * 4); ENSURE(ctx->bg_left, nbgc * 4);
        ENSURE(ctx->minx, cap * 4); ENSURE(ctx->maxx, cap * 4); ENSURE(ctx->maxy, cap * 4);
        ENSURE(ctx->area2, cap * 4); ENSURE(ctx->extrank, cap * 4); ENSURE(ctx->keptrank, cap * 4); ENSURE(ctx->savedrank, cap * 4);
        ENSURE(ctx->objlabel, cap * 4); ENSURE(ctx->objflag, cap * 4); ENSURE(ctx->runlabel, cap * 4);
        ENSURE(ctx->rows, cap * 10 * 4);                     // rows: one per kept object; n_rows <= external roots <= runs
        ENSURE(ctx->saved_row, cap * 4);
        ENSURE(ctx->saved_geo, cap * 8 * 4 + 3 * sz_stride(cap) * 8);
        ENSURE(ctx->crop_off, (cap + 1) * 8); ENSURE(ctx->bits_off, (cap + 1) * 8); ENSURE(ctx->band_off, (cap + 1) * 8);
        S.run_x = (uint32_t *)ctx->run_x.p; S.run_fy = (uint32_t *)ctx->run_fy.p;
        S.fg_parent = (int32_t *)ctx->fg_parent.p; S.ext = (int32_t *)ctx->ext.p; S.gapl = (int32_t *)ctx->gapl.p;
        S.bg_parent = (int32_t *)ctx->bg_parent.p; S.bg_left = (int32_t *)ctx->bg_left.p;
        S.minx = (int32_t *)ctx->minx.p; S.maxx = (int32_t *)ctx->maxx.p; S.maxy = (int32_t *)ctx->maxy.p;
        S.area2 = (int32_t *)ctx->area2.p;
        S.extrank = (int32_t *)ctx->extrank.p; S.keptrank = (int32_t *)ctx->keptrank.p; S.savedrank = (int32_t *)ctx->savedrank.p;
        objlabel = (int32_t *)ctx->objlabel.p; objflag = (int32_t *)ctx->objflag.p;
        rcap = cap;
        int32_t *rb_ = (int32_t *)ctx->rows.p;
        R.frame = rb_; R.c = rb_ + rcap; R.label = rb_ + 2 * rcap; R.area2 = rb_ + 3 * rcap; R.x = rb_ + 4 * rcap; R.y = rb_ + 5 * rcap;
        R.w = rb_ + 6 * rcap; R.h = rb_ + 7 * rcap; R.saved = rb_ + 8 * rcap; R.root = rb_ + 9 * rcap;
        saved_row = (int32_t *)ctx->saved_row.p;
        geo = (int32_t *)ctx->saved_geo.p;
        sz = (unsigned long long *)(geo + 8 * rcap);
        return 0;
    };
    auto launch_fused = [&](uint32_t capv) {
        const dim3 g(nbpf * n), b(256);
#define BAND_LAUNCH(WPL_, CAP_) hipLaunchKernelGGL((k_runs_band<WPL_, CAP_>), g, b, 0, ctx->cur, S, runs_wpl, capv, scan, bandmax_d)
        if (runs_wpl > 10) {
            if (ctx->band_cap > BAND_CAP_MID) BAND_LAUNCH(16, BAND_CAP_DENSE);
            else if (ctx->band_cap > BAND_CAP) BAND_LAUNCH(16, BAND_CAP_MID);
            else BAND_LAUNCH(16, BAND_CAP);
        } else {
            if (ctx->band_cap > BAND_CAP_MID) BAND_LAUNCH(10, BAND_CAP_DENSE);
            else if (ctx->band_cap > BAND_CAP) BAND_LAUNCH(10, BAND_CAP_MID);
            else BAND_LAUNCH(10, BAND_CAP);
        }
#undef BAND_LAUNCH
        if (H > BAND_H) hipLaunchKernelGGL(k_unions_rows, dim3((H - 1) / BAND_H, n), dim3(64), 0, ctx->cur, S, BAND_H, capv, (uint32_t)ctx->band_cap);
    };
    // from the flattening to the size scans; gridded for `runs` runs, guarded by (cap, bcap) on the device
    auto launch_chain = [&](size_t runs, uint32_t capv, uint32_t bcap) {
        const size_t nbgr = runs + (size_t)n * H;
        hipLaunchKernelGGL(k_flatten_resolve, dim3(nblocks(nbgr)), dim3(256), 0, ctx->cur, S, capv, bcap);
        if (runs > 0) {
            hipLaunchKernelGGL(k_bbox, dim3(nblocks(runs)), dim3(256), 0, ctx->cur, S, capv, bcap);
            hipLaunchKernelGGL(k_area, dim3(nblocks(runs)), dim3(256), 0, ctx->cur, S, capv, bcap);
        }
        hipLaunchKernelGGL(k_rank, dim3(n), dim3(1024), 0, ctx->cur, S, min_seg, min_save, ftot3, capv, bcap);
        hipLaunchKernelGGL(k_scan_small, dim3(2), dim3(256), 0, ctx->cur, (const uint32_t *)(ftot3 + 1), rowbase, n, 3, 1);   // rowbase | savedbase
        if (runs > 0)
            hipLaunchKernelGGL(k_emit, dim3(nblocks(runs)), dim3(256), 0, ctx->cur, S, capv, bcap, min_seg, min_save, (const uint32_t *)ftot3,
                               (const uint32_t *)rowbase, (const uint32_t *)savedbase, R, objlabel, objflag, saved_row,
                               (int32_t *)ctx->runlabel.p, geo, sz, sz_stride(rcap));
        hipLaunchKernelGGL(k_scan_sizes, dim3(1), dim3(1024), 0, ctx->cur, S, capv, bcap, (const unsigned long long *)sz, sz_stride(rcap),
                           (const uint32_t *)(savedbase + n), (long long *)ctx->crop_off.p, (long long *)ctx->bits_off.p,
                           (long long *)ctx->band_off.p, totals);
    };
    auto launch_fill = [&](bool spec_, int nfg_) -> int {
        // one wave per row where rows hold many runs (FILL_ROWS_MIN per row on average: 1.0 ms against 1.7 ms at 19 runs per
        // row), one wave per 256-pixel span otherwise (0.86 ms against 0.96 ms at 3 runs per row); the speculative chain
        // does not know this batch's count yet and goes by the previous batch
        const bool by_rows = ctx->fill_rows >= 0 ? ctx->fill_rows > 0
                                                 : (spec_ ? ctx->dense_rows : (size_t)nfg_ >= (size_t)n * H * FILL_ROWS_MIN);
        const uint32_t c0 = spec_ ? (uint32_t)ctx->runcap : 0xffffffffu, c1 = spec_ ? (uint32_t)ctx->band_cap : 0xffffffffu;
        if (by_rows)
            hipLaunchKernelGGL(k_label_fill_rows, dim3(1, (H + 3) / 4, n), dim3(256), 0, ctx->cur, S, (const int32_t *)ctx->runlabel.p, labels, c0, c1);
        else
            hipLaunchKernelGGL(k_label_fill, dim3((W + 1023) / 1024, H, n), dim3(256), 0, ctx->cur, S, (const int32_t *)ctx->runlabel.p, labels, c0, c1);
        HIPCHK(hipGetLastError());
        return 0;
    };
    // Fast path: the whole chain is queued at once, sized by the tables' capacity (what the largest batch so far needed plus a
    // quarter); the host looks at the totals only when the head arrives, behind the label fill.  Slow path (first batch, a
    // denser batch, bands beyond the LDS labelling, very wide rows): totals first, then the chain.
    bool spec = fused && !ctx->no_bands && ctx->runcap > 0 && ctx->spec_hold == 0;
    if (ctx->spec_hold > 0) ctx->spec_hold--;
    bool turn_released = false;
    int nfg = 0;
    uint32_t band_seen = 0;
    for (;;) {
        if (spec) {
            { int r_ = ensure_tables(ctx->runcap); if (r_) return r_; }
            launch_fused((uint32_t)ctx->runcap);
            launch_chain(ctx->runcap, (uint32_t)ctx->runcap, (uint32_t)ctx->band_cap);
        } else {
            { int r_ = ensure_tables(ctx->runcap > 0 ? ctx->runcap : 1); if (r_) return r_; }
            const uint32_t cap0 = fused && ctx->runcap > 0 ? (uint32_t)ctx->runcap : 0u;
            if (fused) {
                launch_fused(cap0);
            } else {
                hipLaunchKernelGGL(k_runs<false>, runs_grid, dim3(256), 0, ctx->cur, S, n * H);
                hipLaunchKernelGGL(k_scan_rows, dim3(n), dim3(256), 0, ctx->cur, S, ftot, bandmax_d);
                hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(256), 0, ctx->cur, (const uint32_t *)ftot, S.framebase, n, 1, 1);
                hipLaunchKernelGGL(k_rowoff_global, dim3(nblocks((size_t)n * (H + 1))), dim3(256), 0, ctx->cur, S);
            }
            HIPCHK(hipGetLastError());
            HIPCHK(hipMemcpyAsync(ctx->pin_nfg.p, S.framebase + n, 12, hipMemcpyDeviceToHost, ctx->cur));
            HIPCHK(hipEventRecord(ctx->ev[8], ctx->cur));
            HIPCHK(hipEventSynchronize(ctx->ev[8]));
            const uint32_t nfg_u = ((const volatile uint32_t *)ctx->pin_nfg.p)[0], bandmax = ((const volatile uint32_t *)ctx->pin_nfg.p)[1];
            if (fused && ((const volatile uint32_t *)ctx->pin_nfg.p)[2] != 0)
                return fail(ctx, PSEG_ERR_HIP, "k_runs_band: a block waited for its predecessors' totals for too long");
            if (nfg_u > 0x3fffffffu) return fail(ctx, PSEG_ERR_CAPACITY, "too many foreground runs");
            nfg = (int)nfg_u;
            const bool fits = fused && cap0 > 0 && nfg_u <= cap0;
            if ((size_t)nfg > ctx->runcap || ctx->runcap == 0) ctx->runcap = (size_t)nfg + (size_t)nfg / 4 + 1024;
            { int r_ = ensure_tables(ctx->runcap); if (r_) return r_; }
            // (the two-pass path labels with k_band_label, whose tables are the small class)
            const bool bands_ok = bandmax <= (uint32_t)(fused ? ctx->band_cap : BAND_CAP) && !ctx->no_bands;
            band_seen = bandmax;
            if (fused && !fits) {                            // the tables were too small: once more
                HIPCHK(hipMemsetAsync(scan, 0, (size_t)scan_words(ctx, n) * 8, ctx->cur));
                launch_fused(0xffffffffu);
            }
            if (!fused) {
                if (nfg > 0) hipLaunchKernelGGL(k_runs<true>, runs_grid, dim3(256), 0, ctx->cur, S, n * H);
                if (bands_ok) {
                    // bands of BAND_H rows are labelled in LDS; the global kernels only unite across the band boundaries
                    hipLaunchKernelGGL(k_band_label, dim3(nbpf, n), dim3(256), 0, ctx->cur, S);
                    if (H > BAND_H) hipLaunchKernelGGL(k_unions_rows, dim3((H - 1) / BAND_H, n), dim3(64), 0, ctx->cur, S, BAND_H, 0xffffffffu, 0xffffffffu);
                }
            }
            if (!bands_ok) {                                 // frames too dense for the LDS tables (or PSEG_NO_BANDS): global union-find
                const int nbg = nfg + n * H;
                hipLaunchKernelGGL(k_init, dim3(nblocks(nbg)), dim3(256), 0, ctx->cur, S, nfg, nbg);
                hipLaunchKernelGGL(k_unions, dim3(nblocks(nbg)), dim3(256), 0, ctx->cur, S, nfg, nbg, 0);
            }
            launch_chain((size_t)nfg, 0xffffffffu, 0xffffffffu);
        }
        HIPCHK(hipGetLastError());
        // per-frame tables and totals to the host (one copy of the head arena); the label map is written while the host waits
        { int r_ = fetch_head(ctx); if (r_) return r_; }
        HIPCHK(hipEventRecord(ctx->ev[3], ctx->cur));
        // (running the label fill on a second stream next to the crop / property kernels was measured: the fill takes 2.0 ms
        // instead of 0.93 ms when it shares the CUs, a net loss)
        if (labels) { int r_ = launch_fill(spec, nfg); if (r_) return r_; }
        HIPCHK(hipEventRecord(ctx->ev[4], ctx->cur));
        // The streaming kernels of this batch are queued.  What follows (crops, properties, hulls) waits on memory latency and
        // fills neither the vector units nor the memory system: the next batch's background kernel may run beside it.
        if (ctx->turn && ctx->early_release && !turn_released) { ctx->turn->release(); turn_released = true; }
        HIPCHK(hipEventSynchronize(ctx->ev[3]));
        if (!spec) break;
        const HeadHost h0 = head_host(ctx, n);
        if (h0.totals[6] != 0) return fail(ctx, PSEG_ERR_HIP, "k_runs_band: a block waited for its predecessors' totals for too long");
        if ((unsigned long long)h0.totals[4] > 0x3fffffffull) return fail(ctx, PSEG_ERR_CAPACITY, "too many foreground runs");
        nfg = (int)h0.totals[4];
        band_seen = (uint32_t)h0.totals[5];
        if ((size_t)nfg <= ctx->runcap && h0.totals[5] <= ctx->band_cap) break;   // the chain ran
        // it did not: more runs than the tables hold, or a band beyond the LDS labelling -- repeat, totals first
        if (h0.totals[5] > BAND_CAP_DENSE) ctx->spec_hold = 16;   // frames beyond even the large tables come in stretches: straight
                                                                  // to the slow path (global union-find) for a while
        spec = false;
        HIPCHK(hipMemsetAsync(scan, 0, (size_t)scan_words(ctx, n) * 8, ctx->cur));
    }
    // size class of the band kernel for the next call: the large tables as soon as a band asked for them, the small ones
    // again after 8 calls that would have fitted with room to spare
    ctx->dense_rows = (size_t)nfg >= (size_t)n * H * FILL_ROWS_MIN;
    {
        auto class_for = [](uint32_t b) { return b <= BAND_CAP ? BAND_CAP : (b <= BAND_CAP_MID ? BAND_CAP_MID : BAND_CAP_DENSE); };
        const int need = class_for(band_seen), roomy = class_for(band_seen + band_seen / 3);
        if (need > ctx->band_cap) { ctx->band_cap = need; ctx->band_calm = 0; }
        else if (roomy < ctx->band_cap) { if (++ctx->band_calm >= 8) { ctx->band_cap = roomy; ctx->band_calm = 0; } }
        else ctx->band_calm = 0;
    }
    ctx->head_on_host = true;
    const HeadHost hh = head_host(ctx, n);
    const uint32_t *h_ftot3 = hh.ftot3, *h_rowbase = hh.rowbase, *h_savedbase = hh.savedbase;
    const long long *h_tot = hh.totals;
    const uint64_t h_maskcnt = mask ? hh.maskcnt : (uint64_t)P;
    const int n_rows = (int)h_rowbase[n], n_saved = (int)h_savedbase[n];
    // packed output arena: sums [n_saved][24] i64 | crop_off [n_saved+1] i64 | rows [10][n_rows] i32 | saved_row [n_saved] i32
    const size_t hr = (size_t)(n_rows > 0 ? n_rows : 1), hs = (size_t)(n_saved > 0 ? n_saved : 1);
    const size_t o_sums = 0, o_coff = o_sums + hs * PSEG_NSUMS * 8, o_rows = o_coff + (hs + 1) * 8, o_srow = o_rows + hr * 10 * 4;
    const size_t o_bytes = o_srow + hs * 4;
    ENSURE(ctx->outpack, o_bytes);
    ENSURE_PIN(ctx->pin_tail, o_bytes);
    long long *d_sums = (long long *)((char *)ctx->outpack.p + o_sums);
    if (n_saved > 0) {
        if (h_tot[2] > 0x7fffffffLL) return fail(ctx, PSEG_ERR_CAPACITY, "too many crop bands");
        ENSURE(ctx->crops_raw, (size_t)h_tot[0]); ENSURE(ctx->crops_masked, (size_t)h_tot[0]); ENSURE(ctx->crops_mask, (size_t)h_tot[0]);
        ENSURE(ctx->maskbits, (size_t)h_tot[1] * 2 * 4);     // first / last column of every window row (convex hull input)
        ENSURE(ctx->band_obj, (size_t)h_tot[2] * BAND_DESC_INTS * 4);
        hipLaunchKernelGGL(k_band_map, dim3(nblocks((size_t)n_saved, 4)), dim3(256), 0, ctx->cur, n_saved, (const long long *)ctx->band_off.p,
                           (const int32_t *)geo, (const long long *)ctx->crop_off.p, (const long long *)ctx->bits_off.p,
                           (int32_t *)ctx->band_obj.p, d_sums);
        hipLaunchKernelGGL(k_crops_sums<false>, dim3((unsigned)h_tot[2]), dim3(CROP_THREADS), 0, ctx->cur, S, n_saved, (const int32_t *)geo,
                           (const long long *)ctx->crop_off.p, (const long long *)ctx->band_off.p, (const int32_t *)objflag,
                           cleaned, raw, (uint8_t *)ctx->crops_raw.p, (uint8_t *)ctx->crops_masked.p,
                           (uint8_t *)ctx->crops_mask.p, d_sums, (const long long *)ctx->bits_off.p,
                           (int32_t *)ctx->maskbits.p, (const int32_t *)nullptr, (const uint8_t *)nullptr, (const int32_t *)ctx->band_obj.p);
        const int hmax = (int)h_tot[3];
        // two size classes when tall objects are present: objects up to 256 rows keep 14 KB of LDS per block, the rest is sized by
        // the tallest one
        // (a tallest object of up to 448 rows = 25 KB still leaves 6 blocks per CU: one launch, one kernel latency less)
        const int hsmall = hmax <= 448 ? hmax : 256;
        const size_t lds_small = (size_t)hsmall * HULL_LDS_PER_ROW + 64, lds_big = (size_t)hmax * HULL_LDS_PER_ROW + 64;
        if (hmax > 2700) return fail(ctx, PSEG_ERR_CAPACITY, "saved object too tall for the convex-hull kernel (limit 2700 rows)");
        if (lds_big > 48 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)k_convex, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
        hipLaunchKernelGGL(k_convex, dim3(n_saved), dim3(128), lds_small, ctx->cur, n_saved, (const int32_t *)geo,
                           (const long long *)ctx->bits_off.p, (const int32_t *)ctx->maskbits.p, d_sums, 0, hsmall);
        if (hmax > hsmall)
            hipLaunchKernelGGL(k_convex, dim3(n_saved), dim3(128), lds_big, ctx->cur, n_saved, (const int32_t *)geo,
                               (const long long *)ctx->bits_off.p, (const int32_t *)ctx->maskbits.p, d_sums, hsmall, hmax);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(ctx->ev[5], ctx->cur));
    ctx->crop_bytes = h_tot[0];
    ctx->n_saved = n_saved;

    // ---- results to the host: pack on the device, ONE copy into page-locked memory ----
    char *hp = (char *)ctx->pin_tail.p;
    int64_t *h_sums = (int64_t *)(hp + o_sums), *h_crop_off = (int64_t *)(hp + o_coff);
    int32_t *hrows = (int32_t *)(hp + o_rows), *h_saved_row = (int32_t *)(hp + o_srow);
    h_crop_off[0] = 0;
    if (n_rows > 0) {
        hipLaunchKernelGGL(k_pack_out, dim3(nblocks((size_t)n_rows + 1)), dim3(256), 0, ctx->cur, (const int32_t *)ctx->rows.p, rcap, n_rows,
                           (const int32_t *)saved_row, (const long long *)ctx->crop_off.p, n_saved,
                           (int32_t *)((char *)ctx->outpack.p + o_rows), (int32_t *)((char *)ctx->outpack.p + o_srow),
                           (long long *)((char *)ctx->outpack.p + o_coff));
        HIPCHK(hipGetLastError());
    }
    if (ctx->turn) ctx->turn->release();                    // last kernel queued: the next batch may follow it on the GPU
    if (n_rows > 0) HIPCHK(hipMemcpyAsync(hp, ctx->outpack.p, o_bytes, hipMemcpyDeviceToHost, ctx->cur));
    HIPCHK(hipEventRecord(ctx->ev[8], ctx->cur));           // (an event, not the stream: other contexts' tails share it)
    HIPCHK(hipEventSynchronize(ctx->ev[8]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[2], ctx->ev[2], ctx->ev[3]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[3], ctx->ev[3], ctx->ev[4]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[4], ctx->ev[4], ctx->ev[5]));

    int32_t *h_fro = ctx->h_frame_row_offset.as<int32_t>(n + 1);
    int32_t *h_fnc = ctx->h_frame_n_contours.as<int32_t>(n);
    uint8_t *h_fv = ctx->h_frame_valid.as<uint8_t>(n);
    double *h_mean = ctx->h_frame_mean.as<double>(n);
    double *h_std = ctx->h_frame_std.as<double>(n);
    for (int f = 0; f <= n; f++) h_fro[f] = (int32_t)h_rowbase[f];
    for (int f = 0; f < n; f++) {
        h_fnc[f] = (int32_t)h_ftot3[3 * f];
        h_fv[f] = hh.valid[f];
        finish_meanstd(hh.fsums + 4 * f + 2, (double)h_maskcnt, &h_mean[f], &h_std[f]);
    }
    double *h_area = ctx->h_area.as<double>(hr);
    uint8_t *h_s8 = ctx->h_saved8.as<uint8_t>(hr);
    double *h_props = ctx->h_props.as<double>(hs * PSEG_NPROPS);       // [n_saved][32]
    for (int i = 0; i < n_rows; i++) {
        h_area[i] = 0.5 * (double)hrows[3 * hr + i];
        h_s8[i] = (uint8_t)hrows[8 * hr + i];
    }
    for (int s = 0; s < n_saved; s++) {
        const int i = h_saved_row[s];
        props_from_sums(h_sums + (size_t)s * PSEG_NSUMS, hrows[6 * hr + i], hrows[7 * hr + i], h_props + (size_t)s * PSEG_NPROPS);
    }
    ctx->t_rows = hrows; ctx->t_saved_row = h_saved_row; ctx->t_crop_off = h_crop_off; ctx->t_hr = hr;
    if (out) {
        memset(out, 0, sizeof(*out));
        out->n_frames = n; out->n_rows = n_rows; out->n_saved = n_saved; out->crop_bytes = ctx->crop_bytes;
        out->frame_row_offset = h_fro; out->frame_n_contours = h_fnc; out->frame_valid = h_fv;
        out->frame_mean = h_mean; out->frame_std = h_std;
        out->row_frame = hrows; out->row_c = hrows + hr; out->row_label = hrows + 2 * hr; out->row_area = h_area;
        out->row_x = hrows + 4 * hr; out->row_y = hrows + 5 * hr; out->row_w = hrows + 6 * hr; out->row_h = hrows + 7 * hr;
        out->row_saved = h_s8; out->saved_props = h_props; out->saved_row = h_saved_row; out->crop_offset = h_crop_off;
        out->sums = h_sums;
        out->crops_raw_dev = (const uint8_t *)ctx->crops_raw.p; out->crops_masked_dev = (const uint8_t *)ctx->crops_masked.p;
        out->crops_mask_dev = (const uint8_t *)ctx->crops_mask.p;
    }
    return PSEG_OK;
}

extern "C" int pseg_detect(pseg_ctx *ctx, const uint8_t *cleaned, const uint8_t *raw, const uint8_t *mask, const uint8_t *valid,
                           int n, int threshold, double min_area_to_segment, double min_area_to_save, int32_t *labels,
                           pseg_table *out)
{
    int r = check_batch(ctx, n);
    if (r) return r;
    if (!cleaned) return fail(ctx, PSEG_ERR_ARG, "null cleaned pointer");
    if (!raw) raw = cleaned;
    size_t P = ctx->P;
    unsigned nblk = nblocks(P / 8);
    ENSURE(ctx->bits, (size_t)n * P / 8);
    ENSURE(ctx->partials, (size_t)n * nblk * 4 * sizeof(uint32_t));
    { int r_ = head_setup(ctx, n); if (r_) return r_; }
    GpuTurn turn(ctx);
    HIPCHK(hipEventRecord(ctx->ev[6], ctx->cur));
    if (valid) HIPCHK(hipMemcpyAsync(ctx->valid_dev.p, valid, n, hipMemcpyHostToDevice, ctx->cur));
    else HIPCHK(hipMemsetAsync(ctx->valid_dev.p, 1, n, ctx->cur));
    HIPCHK(hipEventRecord(ctx->ev[0], ctx->cur));
    hipLaunchKernelGGL(k_threshold, dim3(nblk, n), dim3(256), 0, ctx->cur, cleaned, P, mask, threshold, (uint8_t *)ctx->bits.p,
                       (uint32_t *)ctx->partials.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(ctx->ev[1], ctx->cur));
    hipLaunchKernelGGL(k_reduce_partials, dim3(n), dim3(1024), 0, ctx->cur, (const uint32_t *)ctx->partials.p, (int)nblk,
                       (unsigned long long)P, (unsigned long long *)ctx->fsums.p, (uint8_t *)ctx->valid_dev.p, 0, 2, 2,
                       (unsigned long long *)ctx->maskcnt.p, (uint32_t *)ctx->framebase.p + n + 1, (const uint32_t *)nullptr, 0,
                       (unsigned long long *)ctx->scan_state.p, scan_words(ctx, n));
    ctx->mask_counted = false;
    HIPCHK(hipGetLastError());
    r = run_detect(ctx, cleaned, raw, mask, n, min_area_to_segment, min_area_to_save, labels, out);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[7], ctx->cur));
    HIPCHK(hipEventSynchronize(ctx->ev[7]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[1], ctx->ev[0], ctx->ev[1]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[5], ctx->ev[6], ctx->ev[7]));
    ctx->ms[0] = 0;
    return PSEG_OK;
}

static int process_impl(pseg_ctx *ctx, const uint8_t *frames, int n_avail, int n_emit, int batch_len, int n_bg_imgs, const uint8_t *mask,
                        int threshold, double min_area_to_segment, double min_area_to_save, uint8_t *cleaned,
                        double *mean_std, uint8_t *valid, int32_t *labels, pseg_table *out)
{
    int r = check_batch(ctx, n_avail);
    if (r) return r;
    if (!frames || !cleaned) return fail(ctx, PSEG_ERR_ARG, "null frame pointer");
    if (n_emit < 1 || n_emit > n_avail) return fail(ctx, PSEG_ERR_ARG, "n_emit out of range");
    const int n = n_emit;
    GpuTurn turn(ctx);
    HIPCHK(hipEventRecord(ctx->ev[6], ctx->cur));
    r = run_bg(ctx, frames, n_avail, n_emit, n_bg_imgs, mask, threshold, cleaned, batch_len);
    if (r) return r;
    r = run_detect(ctx, cleaned, cleaned, mask, n, min_area_to_segment, min_area_to_save, labels, out);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[7], ctx->cur));
    r = fetch_raw_stats(ctx, n, mean_std, valid);
    if (r) return r;
    HIPCHK(hipEventSynchronize(ctx->ev[7]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[0], ctx->ev[0], ctx->ev[1]));
    HIPCHK(hipEventElapsedTime(&ctx->ms[5], ctx->ev[6], ctx->ev[7]));
    ctx->ms[1] = 0;
    return PSEG_OK;
}

extern "C" int pseg_process_part(pseg_ctx *ctx, const uint8_t *frames, int n_avail, int n_emit, int n_bg_imgs, const uint8_t *mask,
                                 int threshold, double min_area_to_segment, double min_area_to_save, uint8_t *cleaned,
                                 double *mean_std, uint8_t *valid, int32_t *labels, pseg_table *out)
{
    return process_impl(ctx, frames, n_avail, n_emit, 0, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save, cleaned,
                        mean_std, valid, labels, out);
}

extern "C" int pseg_process(pseg_ctx *ctx, const uint8_t *frames, int n, int n_bg_imgs, const uint8_t *mask, int threshold,
                            double min_area_to_segment, double min_area_to_save, uint8_t *cleaned, double *mean_std,
                            uint8_t *valid, int32_t *labels, pseg_table *out)
{
    return process_impl(ctx, frames, n, n, 0, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save, cleaned, mean_std,
                        valid, labels, out);
}

extern "C" int pseg_process_batches(pseg_ctx *ctx, const uint8_t *frames, int n, int batch_len, int n_bg_imgs, const uint8_t *mask,
                                    int threshold, double min_area_to_segment, double min_area_to_save, uint8_t *cleaned,
                                    double *mean_std, uint8_t *valid, int32_t *labels, pseg_table *out)
{
    if (batch_len < 1) return ctx ? fail(ctx, PSEG_ERR_ARG, "batch_len must be positive") : PSEG_ERR_ARG;
    return process_impl(ctx, frames, n, n, batch_len, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save, cleaned,
                        mean_std, valid, labels, out);
}

// ---- asynchronous form: pseg_submit / pseg_wait ---------------------------------------------------------------------------
// One call of the fused path makes three short host round trips (run count, table sizes, results) and ends with the float64
// epilogue of the table.  A caller that wants the GPU busy meanwhile had to bring one host thread per context.  Here every
// context owns a worker thread inside the library: pseg_submit hands it the arguments and returns at once, pseg_wait blocks
// until the call has finished and delivers its table.  One caller thread can so keep several contexts in flight
// (submit A, submit B, wait A, submit A, wait B, ...); on the device the contexts take turns exactly as with caller threads.
struct AsyncWorker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    bool have_job = false, done = false, quit = false;
    int rc = 0;
    pseg_table table;
    struct { const uint8_t *frames; int n_avail, n_emit, batch_len, n_bg_imgs; const uint8_t *mask; int threshold; double min_seg, min_save;
             uint8_t *cleaned; double *mean_std; uint8_t *valid; int32_t *labels; } a;
};
static void async_main(pseg_ctx *ctx)
{
    AsyncWorker *w = ctx->worker;
    std::unique_lock<std::mutex> lk(w->mu);
    for (;;) {
        w->cv.wait(lk, [&] { return w->have_job || w->quit; });
        if (w->quit) return;
        auto a = w->a;
        lk.unlock();
        const int rc = process_impl(ctx, a.frames, a.n_avail, a.n_emit, a.batch_len, a.n_bg_imgs, a.mask, a.threshold, a.min_seg, a.min_save,
                                    a.cleaned, a.mean_std, a.valid, a.labels, &w->table);
        lk.lock();
        w->rc = rc; w->have_job = false; w->done = true;
        w->cv.notify_all();
    }
}
extern "C" int pseg_submit(pseg_ctx *ctx, const uint8_t *frames, int n_avail, int n_emit, int batch_len, int n_bg_imgs, const uint8_t *mask,
                           int threshold, double min_area_to_segment, double min_area_to_save, uint8_t *cleaned, double *mean_std,
                           uint8_t *valid, int32_t *labels)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (!ctx->worker) {
        ctx->worker = new AsyncWorker();
        ctx->worker->th = std::thread(async_main, ctx);
    }
    AsyncWorker *w = ctx->worker;
    std::lock_guard<std::mutex> lk(w->mu);
    if (w->have_job || w->done) return fail(ctx, PSEG_ERR_ARG, "pseg_submit: the previous submission has not been waited for");
    w->a = {frames, n_avail, n_emit, batch_len, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save, cleaned, mean_std, valid, labels};
    w->have_job = true;
    w->cv.notify_all();
    return PSEG_OK;
}
extern "C" int pseg_wait(pseg_ctx *ctx, pseg_table *out)
{
    if (!ctx) return PSEG_ERR_ARG;
    AsyncWorker *w = ctx->worker;
    if (!w) return fail(ctx, PSEG_ERR_ARG, "pseg_wait: nothing was submitted");
    std::unique_lock<std::mutex> lk(w->mu);
    if (!w->have_job && !w->done) return fail(ctx, PSEG_ERR_ARG, "pseg_wait: nothing was submitted");
    w->cv.wait(lk, [&] { return w->done; });
    w->done = false;
    if (out && w->rc == PSEG_OK) *out = w->table;
    return w->rc;
}
static void async_stop(pseg_ctx *ctx)
{
    AsyncWorker *w = ctx->worker;
    if (!w) return;
    {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return !w->have_job; });       // a submitted call finishes first
        w->quit = true;
        w->cv.notify_all();
    }
    w->th.join();
    delete w;
    ctx->worker = nullptr;
}

extern "C" int pseg_fetch_crops(pseg_ctx *ctx, uint8_t *raw, uint8_t *masked, uint8_t *mask)
{
    if (!ctx) return PSEG_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    size_t nb = (size_t)ctx->crop_bytes;
    if (nb == 0) return PSEG_OK;
    if (raw) HIPCHK(hipMemcpyAsync(raw, ctx->crops_raw.p, nb, hipMemcpyDeviceToHost, ctx->stream));
    if (masked) HIPCHK(hipMemcpyAsync(masked, ctx->crops_masked.p, nb, hipMemcpyDeviceToHost, ctx->stream));
    if (mask) HIPCHK(hipMemcpyAsync(mask, ctx->crops_mask.p, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PSEG_OK;
}

// ---- classifier input tensors straight from the packed crops ---------------------------------------
// Classify_ViT.py:92-145 without the random rotation: the crop (grey, replicated to RGB) is resized so that its longer edge
// becomes 224 (the other edge int-truncated), padded with 255 to 224 x 224 (centred, the odd pixel right / bottom), scaled
// to [0,1] and normalised with mean = std = 0.5.  torchvision's resize of a PIL image is PIL's Image.resize(BILINEAR); its
// algorithm (Resample.c) is restated exactly: per output sample the filter support widens with the scale, the weights
// are normalised in double precision (no fused multiply-adds: the C library is compiled without them), converted to
// 22-bit fixed point, and the two passes (horizontal, then vertical) round to uint8 in between.
#define VIT_SIZE 224
__device__ __forceinline__ int pil_resample_1d(const uint8_t *src, size_t stride, int in_size, int out_size, int xx)
{
    const double scale = __ddiv_rn((double)in_size, (double)out_size);
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = filterscale;                     // bilinear: support 1.0 * filterscale
    const double center = __dmul_rn(__dadd_rn((double)xx, 0.5), scale);
    const double ss = __ddiv_rn(1.0, filterscale);
    int xmin = (int)__dadd_rn(__dsub_rn(center, support), 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)__dadd_rn(__dadd_rn(center, support), 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    auto weight = [&](int x) {
        double t = __dmul_rn(__dadd_rn(__dsub_rn((double)(x + xmin), center), 0.5), ss);
        if (t < 0.0) t = -t;
        return t < 1.0 ? __dsub_rn(1.0, t) : 0.0;
    };
    double ww = 0.0;
    for (int x = 0; x < xmax; x++) ww = __dadd_rn(ww, weight(x));
    int acc = 1 << 21;                                      // PRECISION_BITS = 22
    for (int x = 0; x < xmax; x++) {
        double k = weight(x);
        if (ww != 0.0) k = __ddiv_rn(k, ww);
        const int kk = (int)__dadd_rn(0.5, __dmul_rn(k, 4194304.0));
        acc += (int)src[(size_t)(x + xmin) * stride] * kk;
    }
    acc >>= 22;
    return acc < 0 ? 0 : (acc > 255 ? 255 : acc);
}
// par[k*8] = crop offset lo, hi, w, h, new_w, new_h, tmp offset lo, hi
__global__ __launch_bounds__(256) void k_vit_horizontal(const uint8_t *__restrict__ plane, const int32_t *__restrict__ par, uint8_t *__restrict__ tmp)
{
    const int32_t *q = par + 8 * blockIdx.y;
    const long long off = ((long long)q[1] << 32) | (uint32_t)q[0], toff = ((long long)q[7] << 32) | (uint32_t)q[6];
    const int w = q[2], h = q[3], nw = q[4];
    const int total = h * nw;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < total; p += gridDim.x * 256) {
        const int yy = p / nw, xx = p - yy * nw;
        tmp[toff + p] = (uint8_t)pil_resample_1d(plane + off + (size_t)yy * w, 1, w, nw, xx);
    }
}
__global__ __launch_bounds__(256) void k_vit_vertical(const int32_t *__restrict__ par, const uint8_t *__restrict__ tmp, float *__restrict__ out)
{
    const int32_t *q = par + 8 * blockIdx.y;
    const long long toff = ((long long)q[7] << 32) | (uint32_t)q[6];
    const int h = q[3], nw = q[4], nh = q[5];
    const int left = (VIT_SIZE - nw) / 2, top = (VIT_SIZE - nh) / 2;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= VIT_SIZE * VIT_SIZE) return;
    const int oy = p / VIT_SIZE, ox = p - oy * VIT_SIZE;
    int v = 255;                                            // padding colour
    if (nw > 0 && nh > 0 && ox >= left && ox < left + nw && oy >= top && oy < top + nh)
        v = pil_resample_1d(tmp + toff + (ox - left), (size_t)nw, h, nh, oy - top);
    const float t = ((float)v / 255.0f - 0.5f) / 0.5f;      // ToTensor, Normalize(0.5, 0.5)
    float *o = out + (size_t)blockIdx.y * 3 * VIT_SIZE * VIT_SIZE + p;
    o[0] = t; o[VIT_SIZE * VIT_SIZE] = t; o[2 * VIT_SIZE * VIT_SIZE] = t;
}
extern "C" int pseg_vit_tensors(pseg_ctx *ctx, int plane, int first, int count, float *out, uint8_t *valid)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (!out || plane < 0 || plane > 2 || first < 0 || count < 0 || first + count > ctx->n_saved)
        return fail(ctx, PSEG_ERR_ARG, "pseg_vit_tensors: bad plane or object range (objects are the saved ones of the last detection)");
    if (count == 0) return PSEG_OK;
    HIPCHK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::vector<int32_t> par((size_t)count * 8);
    long long tmp_total = 0;
    int max_pix = 1;
    for (int k = 0; k < count; k++) {
        const int s = first + k, i = ctx->t_saved_row[s];
        const int w = ctx->t_rows[6 * ctx->t_hr + i], h = ctx->t_rows[7 * ctx->t_hr + i];
        const long long off = ctx->t_crop_off[s];
        const double scale_factor = 224.0 / (double)(w > h ? w : h);           // Classify_ViT.py:92-106, Python float arithmetic
        const int nw = (int)((double)w * scale_factor), nh = (int)((double)h * scale_factor);
        int32_t *q = &par[(size_t)k * 8];
        q[0] = (int32_t)(off & 0xffffffffLL); q[1] = (int32_t)(off >> 32); q[2] = w; q[3] = h; q[4] = nw; q[5] = nh;
        q[6] = (int32_t)(tmp_total & 0xffffffffLL); q[7] = (int32_t)(tmp_total >> 32);
        if (valid) valid[k] = (nw > 0 && nh > 0) ? 1 : 0;                       // F.resize raises for an empty size: the reference skips the image
        if (nw > 0 && nh > 0) { tmp_total += (long long)h * nw; if (h * nw > max_pix) max_pix = h * nw; }
        else { q[4] = 0; q[5] = 0; }
    }
    ENSURE(ctx->rp_small, par.size() * 4);
    ENSURE(ctx->rp_scratch, (size_t)tmp_total + 64);
    HIPCHK(hipMemcpyAsync(ctx->rp_small.p, par.data(), par.size() * 4, hipMemcpyHostToDevice, st));
    const uint8_t *pl = (const uint8_t *)(plane == 0 ? ctx->crops_raw.p : plane == 1 ? ctx->crops_masked.p : ctx->crops_mask.p);
    unsigned gx = nblocks((size_t)max_pix);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(k_vit_horizontal, dim3(gx, count), dim3(256), 0, st, pl, (const int32_t *)ctx->rp_small.p, (uint8_t *)ctx->rp_scratch.p);
    hipLaunchKernelGGL(k_vit_vertical, dim3(nblocks((size_t)VIT_SIZE * VIT_SIZE), count), dim3(256), 0, st, (const int32_t *)ctx->rp_small.p,
                       (const uint8_t *)ctx->rp_scratch.p, out);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return PSEG_OK;
}

// calculate_regionprops(mask, img) (detection.py:75-159) for one window given as dense device images: the same band
// kernel as the fused path with the mask read from memory instead of painted from the run tables.
extern "C" int pseg_regionprops(pseg_ctx *ctx, const uint8_t *mask, const uint8_t *img, int h, int w, int stride, double *props,
                                int64_t *sums_out)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (!mask || !img || !props || h < 1 || w < 1 || w > 8192 || h > 65535 || stride < w) return fail(ctx, PSEG_ERR_ARG, "bad window");
    HIPCHK(hipSetDevice(ctx->device));
    const size_t lds = (size_t)h * HULL_LDS_PER_ROW + 64;
    if (h > 2700) return fail(ctx, PSEG_ERR_CAPACITY, "window too tall for the convex-hull kernel (limit 2700 rows)");
    hipStream_t st = ctx->stream;
    const int rpb = crop_rows_per_band(w), nbands = (h + rpb - 1) / rpb;
    // small tables: geo[8] | crop_off[2] | band_off[2] | row_off[2] (long long) | band_obj[nbands] | sums[24] | rowext[2h]
    const size_t o_geo = 0, o_ll = 64, o_bobj = o_ll + 6 * 8, o_sums = (o_bobj + (size_t)nbands * 4 + 7) & ~(size_t)7;
    const size_t o_ext = o_sums + PSEG_NSUMS * 8, o_end = o_ext + (size_t)h * 2 * 4;
    ENSURE(ctx->rp_small, o_end);
    ENSURE(ctx->rp_scratch, (size_t)3 * w * h + 64);
    std::vector<char> hb(o_sums, 0);
    int32_t *g = (int32_t *)(hb.data() + o_geo);
    g[0] = 0; g[1] = 0; g[2] = 0; g[3] = w; g[4] = h; g[5] = 0; g[6] = (w + 31) >> 5; g[7] = 0;
    long long *ll = (long long *)(hb.data() + o_ll);
    ll[0] = 0; ll[1] = (long long)w * h; ll[2] = 0; ll[3] = nbands; ll[4] = 0; ll[5] = h;
    char *d = (char *)ctx->rp_small.p;
    HIPCHK(hipMemcpyAsync(d, hb.data(), o_sums, hipMemcpyHostToDevice, st));
    SegState S;
    memset(&S, 0, sizeof(S));
    S.F = 1; S.H = h; S.W = stride; S.WW = (stride + 63) / 64;
    long long *d_sums = (long long *)(d + o_sums);
    uint8_t *c0 = (uint8_t *)ctx->rp_scratch.p;
    hipLaunchKernelGGL(k_init_sums, dim3(1), dim3(256), 0, st, d_sums, 1);
    hipLaunchKernelGGL(k_crops_sums<true>, dim3(nbands), dim3(CROP_THREADS), 0, st, S, 1, (const int32_t *)(d + o_geo), (const long long *)(d + o_ll),
                       (const long long *)(d + o_ll + 16), (const int32_t *)nullptr, img, img, c0, c0 + (size_t)w * h, c0 + (size_t)2 * w * h,
                       d_sums, (const long long *)(d + o_ll + 32), (int32_t *)(d + o_ext), (const int32_t *)(d + o_bobj), mask, (const int32_t *)nullptr);
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void *)k_convex, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_convex, dim3(1), dim3(128), lds, st, 1, (const int32_t *)(d + o_geo), (const long long *)(d + o_ll + 32),
                       (const int32_t *)(d + o_ext), d_sums, 0, h);
    HIPCHK(hipGetLastError());
    int64_t q[PSEG_NSUMS];
    HIPCHK(hipMemcpyAsync(q, d_sums, sizeof(q), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (q[0] == 0) {                                        // empty mask: scikit-image finds no region, the reference would raise
        for (int k = 0; k < PSEG_NPROPS; k++) props[k] = NAN;
        return fail(ctx, PSEG_ERR_ARG, "mask has no set pixel");
    }
    props_from_sums(q, (double)w, (double)h, props);
    if (sums_out) memcpy(sums_out, q, sizeof(q));
    return PSEG_OK;
}

// ---- pseg_regionprops for callers that cannot allocate ----------------------------------------------------------------
// Same kernels as pseg_regionprops; the scratch they need (small tables, sums, row extremes, the three crop planes the
// band kernel writes) lives in one file-scope device array instead of the context's growing buffers.  Every GPU holds its
// own copy of the array; its device address is looked up once per device in pseg_create (hipGetSymbolAddress) and kept in
// g_rp_static_ptr -- host code uses that pointer, never the array's name.  All calls on one GPU share the array: the call
// holds a per-device mutex from its first launch to the end of its copy-back, so calls from several contexts of one process
// queue up instead of racing (callers in other processes have their own copy of the array).
#define RP_NA_MAX_PIXELS PSEG_NOALLOC_MAX_PIXELS
#define RP_NA_MAX_ROWS PSEG_NOALLOC_MAX_ROWS
#define RP_NA_MAX_BANDS 64
#define RP_NA_O_GEO 0                                       // int32 [8]
#define RP_NA_O_LL 64                                       // int64 [6]: crop_off[2] | band_off[2] | row_off[2]
#define RP_NA_O_BOBJ 128                                    // int32 [RP_NA_MAX_BANDS]
#define RP_NA_O_SUMS (RP_NA_O_BOBJ + RP_NA_MAX_BANDS * 4)   // int64 [PSEG_NSUMS]
#define RP_NA_O_EXT (RP_NA_O_SUMS + PSEG_NSUMS * 8)         // int32 [2 * RP_NA_MAX_ROWS]
#define RP_NA_O_CROPS ((RP_NA_O_EXT + 2 * RP_NA_MAX_ROWS * 4 + 255) & ~255)
#define RP_NA_BYTES (RP_NA_O_CROPS + 3 * RP_NA_MAX_PIXELS + 256)
static __device__ __align__(256) unsigned char g_rp_static[RP_NA_BYTES];
static void *g_rp_static_ptr[PSEG_MAX_DEVICES];             // device address of g_rp_static on each GPU (set in pseg_create)
static std::mutex g_rp_static_mu[PSEG_MAX_DEVICES];
static std::mutex g_rp_static_init_mu;

// once per device, with that device current
static hipError_t rp_static_lookup(int device)
{
    if (device < 0 || device >= PSEG_MAX_DEVICES) return hipSuccess;     // such a device simply has no no-alloc entry point
    std::lock_guard<std::mutex> lk(g_rp_static_init_mu);
    if (g_rp_static_ptr[device]) return hipSuccess;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_rp_static));
    if (e == hipSuccess) g_rp_static_ptr[device] = p;
    return e;
}

// the small tables of one window, written on the device (no host staging buffer, no copy): geo, offsets, band map, sums
__global__ void k_rp_setup(unsigned char *d, int w, int h, int nbands)
{
    int32_t *g = (int32_t *)(d + RP_NA_O_GEO);
    long long *ll = (long long *)(d + RP_NA_O_LL);
    int32_t *bobj = (int32_t *)(d + RP_NA_O_BOBJ);
    long long *sums = (long long *)(d + RP_NA_O_SUMS);
    const int t = threadIdx.x;
    if (t == 0) {
        g[0] = 0; g[1] = 0; g[2] = 0; g[3] = w; g[4] = h; g[5] = 0; g[6] = (w + 31) >> 5; g[7] = 0;
        ll[0] = 0; ll[1] = (long long)w * h; ll[2] = 0; ll[3] = nbands; ll[4] = 0; ll[5] = h;
    }
    if (t < RP_NA_MAX_BANDS) bobj[t] = 0;                    // every band belongs to object 0
    if (t < PSEG_NSUMS) sums[t] = t == 6 ? 255 : 0;          // the sums accumulate: they start from zero on every call (min at 255)
}

extern "C" int pseg_regionprops_noalloc(pseg_ctx *ctx, const uint8_t *mask, const uint8_t *img, int h, int w, int stride,
                                        double *props, int64_t *sums_out)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (!mask || !img || !props || h < 1 || w < 1 || stride < w) return fail(ctx, PSEG_ERR_ARG, "bad window");
    if (w > 8192 || h > RP_NA_MAX_ROWS || (long long)w * h > RP_NA_MAX_PIXELS)
        return fail(ctx, PSEG_ERR_CAPACITY, "pseg_regionprops_noalloc: window beyond the static scratch (at most " +
                    std::to_string(RP_NA_MAX_ROWS) + " rows, 8192 columns and " + std::to_string(RP_NA_MAX_PIXELS) +
                    " pixels); use pseg_regionprops");
    const int rpb = crop_rows_per_band(w), nbands = (h + rpb - 1) / rpb;
    if (nbands > RP_NA_MAX_BANDS) return fail(ctx, PSEG_ERR_CAPACITY, "pseg_regionprops_noalloc: too many row bands");
    if (ctx->device < 0 || ctx->device >= PSEG_MAX_DEVICES || !g_rp_static_ptr[ctx->device])
        return fail(ctx, PSEG_ERR_HIP, "pseg_regionprops_noalloc: no static scratch on this device");
    HIPCHK(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lk(g_rp_static_mu[ctx->device]);
    unsigned char *d = (unsigned char *)g_rp_static_ptr[ctx->device];
    hipStream_t st = ctx->stream;
    const size_t lds = (size_t)h * HULL_LDS_PER_ROW + 64;    // <= 28.7 KB for RP_NA_MAX_ROWS: inside the default limit
    SegState S;
    memset(&S, 0, sizeof(S));
    S.F = 1; S.H = h; S.W = stride; S.WW = (stride + 63) / 64;
    long long *d_sums = (long long *)(d + RP_NA_O_SUMS);
    uint8_t *c0 = d + RP_NA_O_CROPS;
    hipLaunchKernelGGL(k_rp_setup, dim3(1), dim3(64), 0, st, d, w, h, nbands);
    hipLaunchKernelGGL(k_crops_sums<true>, dim3(nbands), dim3(CROP_THREADS), 0, st, S, 1, (const int32_t *)(d + RP_NA_O_GEO),
                       (const long long *)(d + RP_NA_O_LL), (const long long *)(d + RP_NA_O_LL + 16), (const int32_t *)nullptr, img, img,
                       c0, c0 + (size_t)w * h, c0 + (size_t)2 * w * h, d_sums, (const long long *)(d + RP_NA_O_LL + 32),
                       (int32_t *)(d + RP_NA_O_EXT), (const int32_t *)(d + RP_NA_O_BOBJ), mask, (const int32_t *)nullptr);
    hipLaunchKernelGGL(k_convex, dim3(1), dim3(128), lds, st, 1, (const int32_t *)(d + RP_NA_O_GEO), (const long long *)(d + RP_NA_O_LL + 32),
                       (const int32_t *)(d + RP_NA_O_EXT), d_sums, 0, h);
    HIPCHK(hipGetLastError());
    int64_t q[PSEG_NSUMS];
    HIPCHK(hipMemcpyAsync(q, d_sums, sizeof(q), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (q[0] == 0) {
        for (int k = 0; k < PSEG_NPROPS; k++) props[k] = NAN;
        return fail(ctx, PSEG_ERR_ARG, "mask has no set pixel");
    }
    props_from_sums(q, (double)w, (double)h, props);
    if (sums_out) memcpy(sums_out, q, sizeof(q));
    return PSEG_OK;
}

// cv.circle(mask, (W/2, H/2), radius, 255, -1): midpoint circle, horizontal spans (detection.py:320-326)
extern "C" int pseg_disk_mask(pseg_ctx *ctx, int radius, uint8_t *mask)
{
    if (!ctx || !mask || radius < 0) return PSEG_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    const int H = ctx->H, W = ctx->W, cx = H / 2, cy = W / 2;   // reference passes (shape[0]/2, shape[1]/2) as (x, y)
    std::vector<int32_t> sp((size_t)2 * H);
    for (int y = 0; y < H; y++) { sp[2 * y] = W; sp[2 * y + 1] = -1; }
    auto span = [&](int y, int xa, int xb) {
        if (y < 0 || y >= H) return;
        if (xa < sp[2 * y]) sp[2 * y] = xa;
        if (xb > sp[2 * y + 1]) sp[2 * y + 1] = xb;
    };
    int err = 0, dx = radius, dy = 0, plus = 1, minus = (radius << 1) - 1;
    while (dx >= dy) {
        span(cy - dy, cx - dx, cx + dx); span(cy + dy, cx - dx, cx + dx);
        span(cy - dx, cx - dy, cx + dy); span(cy + dx, cx - dy, cx + dy);
        dy++; err += plus; plus += 2;
        int m = (err <= 0) - 1;
        err -= minus & m; dx += m; minus -= m & 2;
    }
    ENSURE(ctx->rowspans, (size_t)2 * H * 4);
    HIPCHK(hipMemcpyAsync(ctx->rowspans.p, sp.data(), (size_t)2 * H * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fill_spans, dim3(H), dim3(256), 0, ctx->stream, mask, H, W, (const int32_t *)ctx->rowspans.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PSEG_OK;
}

// ---- cv.resize(img, (W, H)) of the reader (reader.py:79) -------------------------------------------
// OpenCV's default interpolation (INTER_LINEAR) on 8-bit images is fixed-point arithmetic (imgproc resize.cpp): per output
// column a source column sx and two 11-bit weights, per output row the same; a horizontal pass into 32-bit integers, a
// vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.  A reduction by exactly 2 x 2 is switched
// to the area average (s00 + s01 + s10 + s11 + 2) >> 2 by the library.  The tables are made on the host with the library's
// float / double operations; the kernel is integer only.  tab = xofs [W] | yofs [H] | alpha [2W] | beta [2H].
__global__ void k_resize_linear(const uint8_t *__restrict__ src, int sh, int sw, uint8_t *__restrict__ dst, int H, int W,
                                const int32_t *__restrict__ tab)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t *s = src + (size_t)blockIdx.z * sh * sw;
    const int sx = tab[x], sy = tab[W + y];
    const int a0 = tab[W + H + 2 * x], a1 = tab[W + H + 2 * x + 1], b0 = tab[3 * W + H + 2 * y], b1 = tab[3 * W + H + 2 * y + 1];
    const int x1 = sx + 1 < sw ? sx + 1 : sw - 1;                                   // weight 0 there
    const int y0 = sy < 0 ? 0 : (sy < sh ? sy : sh - 1), y1 = sy + 1 < 0 ? 0 : (sy + 1 < sh ? sy + 1 : sh - 1);
    const int S0 = s[(size_t)y0 * sw + sx] * a0 + s[(size_t)y0 * sw + x1] * a1;
    const int S1 = s[(size_t)y1 * sw + sx] * a0 + s[(size_t)y1 * sw + x1] * a1;
    dst[((size_t)blockIdx.z * H + y) * W + x] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}
__global__ void k_resize_half(const uint8_t *__restrict__ src, int sw, uint8_t *__restrict__ dst, int H, int W)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const uint8_t *s = src + ((size_t)blockIdx.z * 2 * H + 2 * y) * sw + 2 * x;
    dst[((size_t)blockIdx.z * H + y) * W + x] = (uint8_t)((s[0] + s[1] + s[sw] + s[sw + 1] + 2) >> 2);
}
static short resize_coef(float c)                            // saturate_cast<short>(c * INTER_RESIZE_COEF_SCALE): round to nearest even
{
    const long v = lrint((double)(c * 2048.f));
    return (short)(v < -32768 ? -32768 : v > 32767 ? 32767 : v);
}
extern "C" int pseg_resize_frames(pseg_ctx *ctx, const uint8_t *src, int n, int sh, int sw, uint8_t *dst)
{
    if (!ctx) return PSEG_ERR_ARG;
    if (!src || !dst || n < 1 || sh < 1 || sw < 1 || sh > 32767 || sw > 32767) return fail(ctx, PSEG_ERR_ARG, "resize: bad source shape");
    HIPCHK(hipSetDevice(ctx->device));
    const int H = ctx->H, W = ctx->W;
    hipStream_t st = ctx->stream;
    if (sh == H && sw == W) {                                // same size: cv.resize returns a copy
        HIPCHK(hipMemcpyAsync(dst, src, (size_t)n * H * W, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
        return PSEG_OK;
    }
    const double inv_x = (double)W / sw, inv_y = (double)H / sh, scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    const dim3 grid(nblocks((size_t)W), H, n);
    if (std::fabs(scale_x - 2.) < DBL_EPSILON && std::fabs(scale_y - 2.) < DBL_EPSILON) {
        hipLaunchKernelGGL(k_resize_half, grid, dim3(256), 0, st, src, sw, dst, H, W);
    } else {
        std::vector<int32_t> tab((size_t)3 * W + 3 * H);
        for (int dx = 0; dx < W; dx++) {
            float fx = (float)((dx + 0.5) * scale_x - 0.5);
            int sx = (int)std::floor(fx);
            fx -= sx;
            if (sx < 0) { fx = 0; sx = 0; }
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
            tab[dx] = sx;
            tab[W + H + 2 * dx] = resize_coef(1.f - fx); tab[W + H + 2 * dx + 1] = resize_coef(fx);
        }
        for (int dy = 0; dy < H; dy++) {                     // rows are clamped when they are read, their weights are not touched
            float fy = (float)((dy + 0.5) * scale_y - 0.5);
            const int sy = (int)std::floor(fy);
            fy -= sy;
            tab[W + dy] = sy;
            tab[3 * W + H + 2 * dy] = resize_coef(1.f - fy); tab[3 * W + H + 2 * dy + 1] = resize_coef(fy);
        }
        ENSURE(ctx->resize_tab, tab.size() * 4);
        HIPCHK(hipMemcpyAsync(ctx->resize_tab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));                    // tab leaves scope
        hipLaunchKernelGGL(k_resize_linear, grid, dim3(256), 0, st, src, sh, sw, dst, H, W, (const int32_t *)ctx->resize_tab.p);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));
    return PSEG_OK;
}

extern "C" int pseg_synth_frames(pseg_ctx *ctx, int n, int frame0, uint64_t seed, const int32_t *objs, const int32_t *frame_obj_off,
                                 uint8_t *out)
{
    if (!ctx || !out || !frame_obj_off || n < 1) return PSEG_ERR_ARG;
    HIPCHK(hipSetDevice(ctx->device));
    int nobj = frame_obj_off[n];
    DevBuf &dobj = ctx->synth_objs, &doff = ctx->synth_off;   // owned by the context: nothing to free on an error return
    ENSURE(dobj, (size_t)(nobj > 0 ? nobj : 1) * 8 * 4);
    ENSURE(doff, (size_t)(n + 1) * 4);
    if (nobj > 0) HIPCHK(hipMemcpy(dobj.p, objs, (size_t)nobj * 8 * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(doff.p, frame_obj_off, (size_t)(n + 1) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_synth, dim3(nblocks(ctx->P), n), dim3(256), 0, ctx->stream, out, ctx->H, ctx->W, frame0,
                       (unsigned long long)seed, (const int32_t *)dobj.p, (const int32_t *)doff.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return PSEG_OK;
}
