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
#include <chrono>
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
    hipEvent_t ev[12] = {};        // 0..7 time the kernel groups, 8 marks the run count / the head on its way to the host,
                                   // 9..10 time the table kernels on the side stream, 11 the start of a deferred label fill
    float ms[7] = {};
    long long n_gaveup = 0;       // deferred fills that were queued because nobody came (pseg_tuning_state 9)
    long long n_deferred = 0, n_on_queue = 0;   // calls whose label fill was deferred / whose streaming kernels went to the device's streaming queue (pseg_tuning_state 7 / 8)
    bool counted_live = false;    // in g_live_ctx[device]
    bool side_tried = false;
    hipStream_t side = nullptr;   // the table kernels (boxes, areas, ranks, rows, size scans) run here, beside the label fill
    hipEvent_t ev_a = nullptr;    // recorded on the main stream behind the enclosure tree: the side stream waits for it
    int table_mode = -1;          // where the table kernels run: 0 = before the label fill (the order of rounds 1 and 2), 1 = beside it on
                                  // the side stream, 2 = behind it, beside the next batch's background kernel; -1 (default) = 1 for
                                  // ordinary batches, 2 for dense ones (PSEG_TABLE_MODE; a caller's stream or the shared stream: 0 for 1)
    // dense workspaces
    DevBuf bits, partials, maskpart, rowspans;
    // head arena: every small per-frame table the host needs, contiguous so that ONE copy fetches it.
    // fsums / valid_dev / maskcnt / totals / ftot are views into it (head_setup), not allocations.
    DevBuf head, fsums, valid_dev, maskcnt, totals, ftot;
    size_t head_copy_off = 0, head_copy_bytes = 0;
    int head_n = -1;
    PinBuf pin_head, pin_tail, pin_nfg;
    PinBuf pin_rp;                // 256 B for the sums pseg_regionprops_noalloc copies back (allocated in pseg_create: that
                                  // entry point allocates nothing, not even through the runtime's staging of a pageable copy)
    DevBuf outpack;               // sums | crop_off | rows (10 columns) | saved_row, packed for ONE copy to pin_tail
    bool head_on_host = false;    // pin_head holds this call's head (pseg_process reuses it for the raw statistics)
    // sparse workspaces
    DevBuf rowcnt, rowoff, framebase, scan_state;
    DevBuf run_x, run_fy, fg_parent, ext, gapl, runlabel, bg_parent, bg_left, minx, maxx, maxy, area2, extrank,
        keptrank, savedrank, objlabel, objflag;
    DevBuf rows, saved_row, saved_geo, crop_off, bits_off, maskbits, crops_raw, crops_masked, crops_mask;
    bool mask_counted = false;    // maskcnt already holds the number of mask == 255 pixels (fused background kernel)
    bool have_bits = false;       // threshold plane + masked sums already produced by the bg kernel
    DevBuf scan_part;             // chunk totals of the size scan (k_scan_sizes_part)
    DevBuf bounds, band_off, band_obj /* one 64-byte descriptor per crop band: k_band_map */, trash, rp_small, rp_scratch, resize_tab, synth_objs, synth_off;
    int bounds_n = -1, bounds_emit = -1, bounds_chunks = -1, bounds_batch = -1;
    int band_order = 1;           // k_runs_band's scan order: 1 = block index (BAND_ORDER_INDEX; falls back to 0 = tickets for good when
                                  // a scan times out), 2 = raise the timeout flag once (test of the fallback); PSEG_BAND_ORDER
    int band_test_call = -1, detect_calls = 0;   // PSEG_BAND_ORDER=3: index order, and mode 2 on the context's third batch (a queued chain)
    int band_fallbacks = 0;       // times the index order was given up (pseg_tuning_state)
    int scan_multi_min = 16384;   // (SCAN_MULTI_MIN of kernels_output.h) PSEG_SCAN_MULTI_MIN: saved objects of the previous batch from which the size scan takes two launches
    bool crop_classes = true;     // PSEG_CROP_CLASSES=0: one size class of k_crops_sums for every band
    bool convex_small = true;     // PSEG_CONVEX_SMALL=0: every convex hull through k_convex (no wave-per-window kernel for small windows)
    bool timing = true;           // PSEG_TIMING=0: no timing events in the stream
    bool runs_old = false;        // PSEG_RUNS_OLD=1: two-pass run extraction (k_runs) + k_band_label instead of the fused k_runs_band
    bool exclusive = true;        // the device part of a call holds the GPU's turn (PSEG_EXCLUSIVE=0: batches interleave freely)
    bool shared_stream = false;   // PSEG_SHARED_STREAM=1: the streaming parts of all contexts go through one stream per device (GpuTurn)
    bool early_release = true;    // hand the turn on behind the label fill instead of behind the last kernel (PSEG_EARLY_RELEASE=0)
    int fill_refresh = 2;         // PSEG_FILL_REFRESH=0: a deferred label fill reads its tables as the caches hold them (k_refresh_tables);
                                  // 1: refreshed in front of the fill; 2: beside its start when the fill is on the streaming queue
    bool defer_test_giveup = false;   // PSEG_DEFER_FILL=2 (tests): a deferred fill does not wait and counts as given up
    bool stream_queue = true;     // PSEG_STREAM_QUEUE=0: the streaming kernels of a call that plans to defer its fill stay on the context's stream
    hipEvent_t ev_b = nullptr;    // hand-overs between the context's stream and the device's streaming queue
    bool defer_fill = true;       // PSEG_DEFER_FILL=0: a call's label fill never waits for the next call's background kernel (DevSched)
    struct GpuTurn *turn = nullptr;
    int bg_chunks = 1;            // time chunks of the streaming background kernel (PSEG_BG_CHUNKS)
    int bg_resident = 0;          // blocks of the streaming background kernel the device holds at once (0: no tail split)
    bool bg_resident_forced = false;
    int last_bg_stream = -1;      // the last background call took the streaming kernel (1) / the re-reading one (0): pseg_tuning_state 6
    int bounds_tailk = -1;
    bool force_generic = false;   // PSEG_BG_GENERIC=1: use the 12x-read reference kernel (debug / cross-check)
    bool no_bands = false;        // PSEG_NO_BANDS=1: label with the global union-find only (the path of very dense frames)
    int fill_rows = -1;           // PSEG_FILL_ROWS: 1 = label fill with one wave per row (k_label_fill_rows), 0 = one wave per
                                  // 256-pixel span, -1 (default) = by the density of the previous batch
    bool heavy_tables = false;    // the last batch kept more than 500 objects per frame
    bool dense_rows = false;      // the last batch had FILL_ROWS_MIN or more runs per row on average
    // host results
    HostBuf h_frame_row_offset, h_frame_n_contours, h_frame_valid, h_frame_mean, h_frame_std, h_area, h_saved8;
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

// events that only time the kernel groups (pseg_last_ms); PSEG_TIMING=0 leaves them out (each is a few microseconds of idle
// device between two kernels) and pseg_last_ms returns 0
#define TIMED_RECORD(k) do { if (ctx->timing) HIPCHK(hipEventRecord(ctx->ev[k], ctx->cur)); } while (0)
#define TIMED_ELAPSED(m, a, b) do { if (ctx->timing) HIPCHK(hipEventElapsedTime(&ctx->ms[m], ctx->ev[a], ctx->ev[b])); else ctx->ms[m] = 0; } while (0)
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

#include "kernels_dense.h"
#include "kernels_labelling.h"
#include "kernels_output.h"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#define PSEG_MAX_DEVICES 64
// (the schedule of one device: see "one batch on the GPU at a time" below)
struct DevSched {
    std::mutex m;
    std::condition_variable cv;
    unsigned long long next = 0, serving = 0;                // tickets of phase 1
    unsigned long long p1_done = 0;                          // phase-1 sections finished so far
    bool busy = false;                                       // a section is being queued
    int fills_pending = 0;                                   // calls between the end of their phase 1 and their fill
    int in_flight = 0;                                       // calls that hold a GpuTurn object
    // A deferred fill that ends by giving up (the streaming kernels drained and the next call had not come) has gained nothing
    // and meets the crop tail of the call before it.  Callers that resubmit slowly (bench.py --gpus N packs tables between a
    // wait and the next submit) produce mostly those.  A window of 4 deferred fills fails when 2 or more end that way; after
    // two failed windows in a row (one alone is what the first calls of any caller look like) the device keeps the plain order for
    // the next 64 calls and then tries one window again -- 128, 256, ... 8192 calls while the trials keep failing.
    int window_deferred = 0, window_gaveup = 0, fail_streak = 0, hold = 0, hold_len = 64;
    // (a sign that looked cheaper -- a call that arrives when the streaming kernels queued before it have drained found the
    // device waiting for the host -- does not tell the two kinds of callers apart: in the plain order a call of the pipelined
    // bench arrives beside the previous call's crop kernels, with nothing streaming queued; measured and removed)
    unsigned long long total_gaveup = 0;
};
static DevSched g_sched[PSEG_MAX_DEVICES];
static int g_live_ctx[PSEG_MAX_DEVICES];                     // contexts alive per device (under g_sched[d].m)
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
    // (a failed lookup only means "no no-alloc entry point on this device": pseg_regionprops_noalloc says so by itself)
    (void)rp_static_lookup(device);
    if (hipHostMalloc(&ctx->pin_rp.p, 256, hipHostMallocDefault) == hipSuccess) ctx->pin_rp.bytes = 256; else ctx->pin_rp.p = nullptr;
    {
        // (the side stream itself is created when a call first wants it: a process has four hardware queues)
        if (hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming) != hipSuccess) ctx->ev_a = nullptr;
        if (const char *e = getenv("PSEG_TABLE_MODE")) ctx->table_mode = atoi(e) >= 0 && atoi(e) <= 2 ? atoi(e) : -1;
    }
    if (const char *e = getenv("PSEG_BG_CHUNKS")) ctx->bg_chunks = atoi(e) > 0 ? atoi(e) : 1;
    if (const char *e = getenv("PSEG_BG_GENERIC")) ctx->force_generic = atoi(e) != 0;
    {
        hipDeviceProp_t prop;                                // 4 blocks per CU (the kernel's launch bounds): see run_bg
        if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->bg_resident = 4 * prop.multiProcessorCount;
        if (const char *e = getenv("PSEG_BG_TAIL")) { ctx->bg_resident = atoi(e) > 0 ? atoi(e) : 0; ctx->bg_resident_forced = true; }   // 0: off, N: pretend N slots
    }
    if (const char *e = getenv("PSEG_EXCLUSIVE")) ctx->exclusive = atoi(e) != 0;
    if (const char *e = getenv("PSEG_EARLY_RELEASE")) ctx->early_release = atoi(e) != 0;
    if (const char *e = getenv("PSEG_DEFER_FILL")) { ctx->defer_fill = atoi(e) != 0; ctx->defer_test_giveup = atoi(e) == 2; }
    if (const char *e = getenv("PSEG_STREAM_QUEUE")) ctx->stream_queue = atoi(e) != 0;
    if (hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming) != hipSuccess) ctx->ev_b = nullptr;
    if (const char *e = getenv("PSEG_FILL_REFRESH")) ctx->fill_refresh = atoi(e);
    if (const char *e = getenv("PSEG_SHARED_STREAM")) ctx->shared_stream = atoi(e) != 0;
    if (const char *e = getenv("PSEG_RUNS_OLD")) ctx->runs_old = atoi(e) != 0;
    if (const char *e = getenv("PSEG_TIMING")) ctx->timing = atoi(e) != 0;
    if (const char *e = getenv("PSEG_CROP_CLASSES")) ctx->crop_classes = atoi(e) != 0;
    if (const char *e = getenv("PSEG_CONVEX_SMALL")) ctx->convex_small = atoi(e) != 0;
    if (const char *e = getenv("PSEG_SCAN_MULTI_MIN")) ctx->scan_multi_min = atoi(e);
    if (!ctx->exclusive) ctx->band_order = 0;                // no turns: two band kernels may share the device (see BAND_ORDER_INDEX)
    if (const char *e = getenv("PSEG_BAND_ORDER")) {
        const int v = atoi(e);
        ctx->band_order = v >= 0 && v <= 2 ? v : (v == 3 ? 1 : 0);
        if (v == 3) ctx->band_test_call = 2;
    }
    if (const char *e = getenv("PSEG_NO_BANDS")) ctx->no_bands = atoi(e) != 0;
    if (const char *e = getenv("PSEG_FILL_ROWS")) ctx->fill_rows = atoi(e);
    for (auto &e : ctx->ev) if (hipEventCreate(&e) != hipSuccess) { pseg_destroy(ctx); return PSEG_ERR_HIP; }   // frees what exists so far
    if (device < PSEG_MAX_DEVICES) { std::lock_guard<std::mutex> lk(g_sched[device].m); g_live_ctx[device]++; ctx->counted_live = true; }
    *out = ctx;
    return PSEG_OK;
}

extern "C" int pseg_destroy(pseg_ctx *ctx)
{
    if (!ctx) return PSEG_ERR_ARG;
    async_stop(ctx);
    if (ctx->counted_live) { std::lock_guard<std::mutex> lk(g_sched[ctx->device].m); g_live_ctx[ctx->device]--; }
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
#ifdef BAND_PROF
    {
        unsigned long long p[16] = {0};
        if (hipMemcpyFromSymbol(p, HIP_SYMBOL(g_band_prof), sizeof p) == hipSuccess && p[15]) {
            fprintf(stderr, "k_runs_band phases, us per block over %llu blocks:", p[15]);
            const char *nm[6] = {"ticket", "load+count", "runs->LDS", "label", "look-back", "publish"};
            for (int k = 0; k < 6; k++) fprintf(stderr, " %s %.2f", nm[k], 0.01 * (double)p[k] / (double)p[15]);
            fprintf(stderr, "; look-back rounds per block %.2f, threads that polled per block %.2f, polls per such thread %.1f\n",
                    (double)p[8] / (double)p[15], (double)p[10] / (double)p[15], p[10] ? (double)p[9] / (double)p[10] : 0.0);
        }
    }
#endif
    DevBuf *bufs[] = {&ctx->bits, &ctx->partials, &ctx->maskpart, &ctx->head, &ctx->outpack, &ctx->rowspans, &ctx->rowcnt,
                      &ctx->rowoff, &ctx->framebase, &ctx->scan_state, &ctx->run_x, &ctx->run_fy, &ctx->fg_parent, &ctx->ext, &ctx->gapl, &ctx->runlabel,
                      &ctx->bg_parent, &ctx->bg_left, &ctx->minx, &ctx->maxx, &ctx->maxy, &ctx->area2,
                      &ctx->extrank, &ctx->keptrank, &ctx->savedrank, &ctx->objlabel, &ctx->objflag, &ctx->rows,
                      &ctx->saved_row, &ctx->saved_geo, &ctx->crop_off, &ctx->bits_off, &ctx->maskbits,
                      &ctx->crops_raw, &ctx->crops_masked, &ctx->crops_mask, &ctx->bounds, &ctx->band_off, &ctx->band_obj, &ctx->scan_part, &ctx->trash, &ctx->rp_small, &ctx->rp_scratch, &ctx->resize_tab,
                      &ctx->synth_objs, &ctx->synth_off};
    for (DevBuf *b : bufs) if (b->p) hipFree(b->p);
    for (PinBuf *b : {&ctx->pin_head, &ctx->pin_tail, &ctx->pin_nfg, &ctx->pin_rp}) if (b->p) hipHostFree(b->p);
    for (auto &e : ctx->ev) if (e) hipEventDestroy(e);
    if (ctx->ev_turn) hipEventDestroy(ctx->ev_turn);
    if (ctx->ev_a) hipEventDestroy(ctx->ev_a);
    if (ctx->ev_b) hipEventDestroy(ctx->ev_b);
    if (ctx->side) hipStreamDestroy(ctx->side);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return PSEG_OK;
}

extern "C" const char *pseg_last_error(const pseg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" float pseg_last_ms(const pseg_ctx *ctx, int which) { return (ctx && which >= 0 && which < 7) ? ctx->ms[which] : -1.f; }
extern "C" long long pseg_tuning_state(const pseg_ctx *ctx, int which)
{
    if (!ctx) return -1;
    switch (which) {
    case 0: return (long long)ctx->runcap;
    case 1: return ctx->band_cap;
    case 2: return ctx->spec_hold;
    case 3: return ctx->fill_rows >= 0 ? (ctx->fill_rows > 0) : (ctx->dense_rows ? 1 : 0);
    case 4: return ctx->band_order;
    case 5: return ctx->band_fallbacks;
    case 6: return ctx->last_bg_stream;
    case 7: return ctx->n_deferred;
    case 8: return ctx->n_on_queue;
    case 9: return ctx->n_gaveup;
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
// next to the other batch's labelling kernels 1.4 ms -- with the fill's tables cold, as round 3 found --, measured with
// rocprofv3 kernel traces, profiles/README.md).  Rounds 1 and 2 therefore ran the streaming part of a call (background kernel,
// labelling, label fill) under a per-device turn, handed on behind the label fill: the crop / property / hull kernels that
// follow wait on memory latency, and the next batch's background kernel runs beside them at a small loss (0.92 ms for both
// against 0.65 + 0.38 ms one after the other).  With one or two calls in flight that is still the order of things; round 3
// split the turn into two sections (below).
// Schedule of one device.  The STREAMING kernels of all contexts (background kernel, label fill) run one at a time, in the order
// in which the host queued them: each waits for the event recorded behind the one queued before it (g_turn_ev).  A call queues
// them in two sections:
//   phase 1: background kernel, then -- not part of the streaming order -- the labelling chain and the table kernels;
//   phase 2: the label fill.
// Phase 1 is first come, first served (with a plain mutex a context that has just finished could overtake one that has been
// waiting, and a caller that collects results in submission order then sits on the overtaken one while the others have nothing
// queued).  When another call is on its way (it waits for phase 1 already, or three calls are in flight on the device), a
// call DEFERS its label fill until that call has queued its phase 1, so that the device sees
//   bg(t), fill(t-1) | chain(t), bg(t+1), fill(t) | chain(t+1), ...
// and the latency-bound labelling chain of a batch (0.16 ms during which the memory system idled) runs beside the previous
// batch's label fill instead of alone.  No more than two fills are ever outstanding, a deferred fill is queued at the latest
// when the streaming kernels queued so far have drained and nobody has come (a poll of g_turn_ev), and the band kernels of
// consecutive calls are ordered among themselves by an event of their own (g_band_ev: the index-ordered look-back of
// k_runs_band wants to be the only one of its kind on the device).
static hipEvent_t g_turn_ev[PSEG_MAX_DEVICES];              // recorded behind the streaming kernel that was queued last
static bool g_turn_ev_set[PSEG_MAX_DEVICES];
static hipStream_t g_turn_last[PSEG_MAX_DEVICES];            // the stream that event was recorded on
static hipEvent_t g_band_ev[PSEG_MAX_DEVICES];              // recorded behind the band kernel that was queued last
static bool g_band_ev_set[PSEG_MAX_DEVICES];
static hipStream_t g_band_last[PSEG_MAX_DEVICES];
static hipStream_t g_sq[PSEG_MAX_DEVICES];                   // the device's streaming queue (see GpuTurn::sq)
static bool g_sq_tried[PSEG_MAX_DEVICES];
static hipStream_t g_main_stream[PSEG_MAX_DEVICES];          // the device's shared streaming stream
static bool g_main_tried[PSEG_MAX_DEVICES];
static hipStream_t g_tail_stream[PSEG_MAX_DEVICES][2];       // where the tails of consecutive calls go, alternating
static unsigned g_tail_next[PSEG_MAX_DEVICES];
// PSEG_SHARED_STREAM=1 (a layout that was measured and left off): the streaming parts of consecutive batches are queued on
// ONE stream per device, the latency-bound tail of a call (crops, properties, hulls, copies) on one of two tail streams of the
// device, behind an event recorded at the hand-over.  A context on a caller's stream (pseg_set_stream) keeps everything on
// that stream and is ordered against the others through g_turn_ev.
struct GpuTurn {
    pseg_ctx *ctx;
    enum State { IDLE, P1, PENDING, P2 } st = IDLE;
    bool counted = false;                                    // in g_sched[d].in_flight
    bool ev_fresh = false;                                   // g_turn_ev was recorded behind the last streaming kernel this call queued
    // A call that expects to defer its fill (the next call is waiting, or three are in flight) queues its two streaming kernels
    // on the device's streaming queue instead of its own stream: consecutive streaming kernels of different calls then follow
    // each other inside one hardware queue (a wait for another queue's event is 20 - 30 us of idle streaming path, twice per
    // batch), and the hand-overs between that queue and the context's stream lie beside a streaming kernel.
    hipStream_t sq = nullptr;
    bool may_defer = false;                                  // a call that writes a label map, on a context of its own, of light frames
    // fill_expected: the call writes a label map (only then is there a fill to defer)
    explicit GpuTurn(pseg_ctx *c, bool fill_expected = false) : ctx(c)
    {
        if (!c || !c->exclusive || c->turn || c->device < 0 || c->device >= PSEG_MAX_DEVICES) return;
        const int d = c->device;
        DevSched &s = g_sched[d];
        int live;
        bool company;
        {
            std::unique_lock<std::mutex> lk(s.m);
            const unsigned long long my = s.next++;
            s.in_flight++; counted = true;
            s.cv.wait(lk, [&] { return s.serving == my && !s.busy && s.fills_pending <= 1; });
            s.busy = true;
            // The next call is on its way when it waits for its phase 1 already, or when three calls are in flight (measured with
            // three contexts: the next one then arrives before the previous call's fill has drained).  Frames with many runs or
            // many objects keep the plain order: their chain and table kernels are too heavy a neighbour for a fill
            // (uvp5_dense: 1.65 ms against 1.08 ms).
            may_defer = fill_expected && c->defer_fill && c->own_stream && !c->shared_stream && !c->heavy_tables && !c->dense_rows;
            if (may_defer && s.hold > 0) { s.hold--; may_defer = false; }
            company = s.next - s.serving > 1 || s.in_flight >= 3;
            live = g_live_ctx[d];
        }
        st = P1; c->turn = this;
        // (more than three contexts: the streaming queue would share one of the process's four hardware queues with a context)
        if (may_defer && company && c->stream_queue && live <= 3 && c->ev_b && c->ev_turn) {
            if (!g_sq_tried[d]) {                            // (inside a section: one thread at a time)
                g_sq_tried[d] = true;
                if (hipStreamCreateWithFlags(&g_sq[d], hipStreamNonBlocking) != hipSuccess) g_sq[d] = nullptr;
            }
            sq = g_sq[d];
            if (sq) c->n_on_queue++;
        }
        if (c->own_stream && c->ev_turn && c->shared_stream) {
            if (!g_main_tried[d]) {
                g_main_tried[d] = true;
                if (hipStreamCreateWithFlags(&g_main_stream[d], hipStreamNonBlocking) != hipSuccess) g_main_stream[d] = nullptr;
                for (auto &t : g_tail_stream[d]) if (hipStreamCreateWithFlags(&t, hipStreamNonBlocking) != hipSuccess) t = nullptr;
            }
            if (g_main_stream[d]) c->cur = g_main_stream[d];
        }
        wait_streaming();
    }
    bool held() const { return st == P1 || st == P2; }
    // the stream waits for the streaming kernel that was queued last on the device (only inside a section)
    hipStream_t target() const { return sq ? sq : ctx->cur; }  // where this call's streaming kernels go
    void wait_streaming()
    {
        const int d = ctx->device;
        if (g_turn_ev_set[d] && g_turn_last[d] != target()) (void)hipStreamWaitEvent(target(), g_turn_ev[d], 0);
    }
    // the stream for a streaming kernel; with_ctx: the kernel reads what the context's stream has queued so far
    hipStream_t stream_begin(bool with_ctx)
    {
        if (!sq || !held()) return ctx->cur;
        if (with_ctx) { (void)hipEventRecord(ctx->ev_b, ctx->cur); (void)hipStreamWaitEvent(sq, ctx->ev_b, 0); }
        return sq;
    }
    // the kernel is queued: the context's stream goes on behind it
    // (call streamed() first: the event it records behind the kernel serves this wait too -- every event between two kernels
    // of a queue is 6 us of idle queue)
    void stream_end()
    {
        if (!sq || !held()) return;
        const int d = ctx->device;
        if (ev_fresh && g_turn_ev_set[d] && g_turn_last[d] == sq) { (void)hipStreamWaitEvent(ctx->cur, g_turn_ev[d], 0); return; }
        (void)hipEventRecord(ctx->ev_turn, sq);
        (void)hipStreamWaitEvent(ctx->cur, ctx->ev_turn, 0);
    }
    // the context's stream waits for the streaming kernel queued last on the device (for work that should run beside the next one)
    void ctx_waits_streaming()
    {
        const int d = ctx->device;
        if (held() && g_turn_ev_set[d] && g_turn_last[d] != ctx->cur) (void)hipStreamWaitEvent(ctx->cur, g_turn_ev[d], 0);
    }
    // a streaming kernel of this call has just been queued: the next one, whoever queues it, follows it
    void streamed()
    {
        if (!held()) return;
        const int d = ctx->device;
        std::lock_guard<std::mutex> lk(g_sched[d].m);        // (the deferred fill of another call may be polling the event)
        if (!g_turn_ev_set[d]) g_turn_ev_set[d] = hipEventCreateWithFlags(&g_turn_ev[d], hipEventDisableTiming) == hipSuccess;
        if (g_turn_ev_set[d]) { (void)hipEventRecord(g_turn_ev[d], target()); g_turn_last[d] = target(); ev_fresh = true; }
    }
    void unstreamed() { ev_fresh = false; }                  // another streaming kernel of this call follows
    // the band kernels of consecutive calls, one at a time
    void band_begin()
    {
        const int d = ctx->device;
        if (held() && g_band_ev_set[d] && g_band_last[d] != ctx->cur) (void)hipStreamWaitEvent(ctx->cur, g_band_ev[d], 0);
    }
    void band_end()
    {
        if (!held()) return;
        const int d = ctx->device;
        if (!g_band_ev_set[d]) g_band_ev_set[d] = hipEventCreateWithFlags(&g_band_ev[d], hipEventDisableTiming) == hipSuccess;
        if (g_band_ev_set[d]) { (void)hipEventRecord(g_band_ev[d], ctx->cur); g_band_last[d] = ctx->cur; }
    }
    // should this call's label fill wait for the next call's phase 1?
    bool want_defer()
    {
        if (st != P1 || !may_defer || ctx->cur != ctx->stream || !ev_fresh) return false;
        DevSched &s = g_sched[ctx->device];
        std::lock_guard<std::mutex> lk(s.m);
        return s.next - s.serving > 1 || s.in_flight >= 3;
    }
    // ends phase 1 and returns inside phase 2, with the stream waiting for the streaming kernel queued last
    void defer_fill()
    {
        const int d = ctx->device;
        DevSched &s = g_sched[d];
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.busy = false; s.serving++;
            const unsigned long long mine = ++s.p1_done;
            s.fills_pending++;
            st = PENDING;
            s.cv.notify_all();
            bool gaveup = false;
            for (;;) {
                if (!s.busy) {
                    if (s.p1_done > mine) break;             // the next call has queued its background kernel
                    // nobody on the way and nothing streaming left on the device: waiting longer would leave it idle
                    if (s.next == s.serving && (!g_turn_ev_set[d] || hipEventQuery(g_turn_ev[d]) != hipErrorNotReady)) { gaveup = true; break; }
                    if (ctx->defer_test_giveup) { gaveup = true; break; }   // (test of the hold: every deferred fill counts as given up)
                }
                s.cv.wait_for(lk, std::chrono::microseconds(40));
            }
            s.window_deferred++;
            if (gaveup) { s.window_gaveup++; s.total_gaveup++; ctx->n_gaveup++; }
            if (s.window_deferred >= 4) {
                if (s.window_gaveup >= 2) {
                    if (++s.fail_streak >= 2) { s.hold = s.hold_len; if (s.hold_len < 8192) s.hold_len *= 2; }
                } else { s.fail_streak = 0; s.hold_len = 64; }
                s.window_deferred = s.window_gaveup = 0;
            }
            s.busy = true;
            st = P2;
        }
        wait_streaming();
        ev_fresh = false;
    }
    void release()                                           // call after the last streaming kernel of the section is queued
    {
        if (st == IDLE) return;
        const int d = ctx->device;
        DevSched &s = g_sched[d];
        if (held() && !ev_fresh) {
            streamed();
            if (!ev_fresh) (void)hipStreamSynchronize(target());   // no event: hand over on the host
        }
        if (held() && ctx->cur != ctx->stream) {             // the rest of the call runs on a tail stream, behind what is queued
            hipStream_t t = g_tail_stream[d][g_tail_next[d]++ & 1];
            if (!t) t = ctx->stream;
            (void)hipEventRecord(ctx->ev_turn, ctx->cur);
            (void)hipStreamWaitEvent(t, ctx->ev_turn, 0);
            ctx->cur = t;
        }
        {
            std::lock_guard<std::mutex> lk(s.m);
            if (st == P1) { s.busy = false; s.serving++; s.p1_done++; }
            else if (st == P2) { s.busy = false; s.fills_pending--; }
            else if (st == PENDING) s.fills_pending--;
            st = IDLE;
        }
        ctx->turn = nullptr;
        s.cv.notify_all();
    }
    ~GpuTurn()
    {
        release();
        if (counted) { std::lock_guard<std::mutex> lk(g_sched[ctx->device].m); g_sched[ctx->device].in_flight--; }
        if (ctx) ctx->cur = ctx->stream;
    }
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
    // streaming kernel: instantiated for windows of 8, 10, 12 and 16 frames (n_bg_imgs = 4, 5, 6 -- the reference's -- and 8);
    // other windows, and batches shorter than one window, take the kernel that re-reads the window per frame
    const bool stream = ((NW == 8 || NW == 10 || NW == 12 || NW == 16) && last_len >= NW && (!batch_len || batch_len >= NW) &&
                         !ctx->force_generic);
    ctx->last_bg_stream = stream ? 1 : 0;
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
            const int span = part ? n_emit : n_avail - NW;    // chunk starts lie in [0, span)
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
            const int res = NW > 12 && !ctx->bg_resident_forced ? ctx->bg_resident / 4 * 3 : ctx->bg_resident;   // (BgWin<NW>::BLOCKS per CU)
            if (chunks == 1 && res > 0 && nblk > (unsigned)res && nblk % res) {
                const unsigned rem = nblk % res;
                tailk = (int)(res / rem);
                if (tailk > 4) tailk = 4;
                if (tailk > span / (2 * NW)) tailk = span / (2 * NW);
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
    const hipStream_t bst = ctx->turn ? ctx->turn->stream_begin(false) : ctx->cur;   // (the device's streaming queue, or the context's stream)
    if (ctx->timing) HIPCHK(hipEventRecord(ctx->ev[0], bst));
    if (stream) {
        const dim3 grid(nfull + (nblk - nfull) * tailk, chunks);
#define BG_LAUNCH(NW_, PAD_) hipLaunchKernelGGL((k_bg_stream<NW_, PAD_>), grid, dim3(256), 0, bst, frames, P, cleaned, mask, thr,       \
                                                (uint8_t *)ctx->bits.p, (uint4 *)ctx->partials.p, (const int *)ctx->bounds.p,                \
                                                (uint8_t *)ctx->trash.p, (uint32_t *)ctx->maskpart.p, (int)nfull, tailk)
        const bool pad = P % 2048 != 0;
        switch (NW) {
        case 8: if (pad) BG_LAUNCH(8, true); else BG_LAUNCH(8, false); break;
        case 10: if (pad) BG_LAUNCH(10, true); else BG_LAUNCH(10, false); break;
        case 16: if (pad) BG_LAUNCH(16, true); else BG_LAUNCH(16, false); break;
        default: if (pad) BG_LAUNCH(12, true); else BG_LAUNCH(12, false); break;
        }
#undef BG_LAUNCH
    }
    else
        for (int c = 0; c < nbatch; c++) {                   // the re-reading kernel takes one batch per launch
            const size_t f0 = (size_t)c * (batch_len ? batch_len : 0);
            const int nc = batch_len ? ((c + 1) * batch_len < n_avail ? batch_len : last_len) : n;
            hipLaunchKernelGGL(k_bg_generic, dim3(nblk, nc), dim3(256), 0, bst, frames + f0 * P, batch_len ? nc : n_kernel, NW, P,
                               cleaned + f0 * P, mask, thr, (uint8_t *)ctx->bits.p + f0 * (P / 8), (uint32_t *)ctx->partials.p + f0 * npart * 4);
        }
    HIPCHK(hipGetLastError());
    // the next streaming kernel on the device follows this one (on the streaming queue: at once; on the context's stream the
    // event is recorded behind the reduction, where the group's timing event lies anyway)
    const bool on_sq = bst != ctx->cur;
    if (ctx->turn && on_sq) { ctx->turn->streamed(); ctx->turn->stream_end(); }
    hipLaunchKernelGGL(k_reduce_partials, dim3(n), dim3(1024), 0, ctx->cur, (const uint32_t *)ctx->partials.p, (int)npart,
                       (unsigned long long)P, (unsigned long long *)ctx->fsums.p, (uint8_t *)ctx->valid_dev.p, 1, 0, 4,
                       (unsigned long long *)ctx->maskcnt.p, (uint32_t *)ctx->framebase.p + n + 1,
                       (const uint32_t *)ctx->maskpart.p, (stream && mask) ? (int)npart : 0,
                       (unsigned long long *)ctx->scan_state.p, scan_words(ctx, n));
    ctx->mask_counted = stream && mask;
    HIPCHK(hipGetLastError());
    // one event ends the background group (the reduction of the raw sums belongs to it) and starts the labelling group: every
    // event between two kernels of a stream showed as 6.5 us of idle device in the kernel trace
    TIMED_RECORD(1);
    if (ctx->turn && !on_sq) ctx->turn->streamed();
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

static int fetch_head(pseg_ctx *ctx, hipStream_t st = nullptr)
{
    HIPCHK(hipMemcpyAsync(ctx->pin_head.p, (const char *)ctx->head.p + ctx->head_copy_off, ctx->head_copy_bytes,
                          hipMemcpyDeviceToHost, st ? st : ctx->cur));
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
    TIMED_ELAPSED(0, 0, 1);
    return PSEG_OK;
}

// the 32 columns of regionprop2zooprocess (detection.py:84-152, scikit-image definitions per SURVEY.md 8(a)) from the exact
// integer sums of one window; q[9] is set to the filled area
__host__ __device__ static void props_from_sums(int64_t *q, double w, double h, double *p)
{
    // (no fused multiply-adds: mu11 of a symmetric window must come out as the exact zero it is in unfused arithmetic -- the
    // sign of atan2's first argument decides between an angle of 0 and of 180 degrees -- and host and device then agree bit for
    // bit in everything but the last place of atan2)
#pragma clang fp contract(off)
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

// the 32 property columns of every saved object on the device, a thread per object, into the packed output arena (the
// host's loop over 5700 objects was 0.18 ms between two calls of a context, over the 100 000 of a dense batch 3 ms)
__global__ __launch_bounds__(256) void k_props(int n_saved, long long *__restrict__ sums, const int32_t *__restrict__ geo, double *__restrict__ props)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n_saved) return;
    double p[PSEG_NPROPS];
    props_from_sums((int64_t *)(sums + (size_t)s * PSEG_NSUMS), (double)geo[8 * (size_t)s + 3], (double)geo[8 * (size_t)s + 4], p);
#pragma unroll
    for (int k = 0; k < PSEG_NPROPS; k++) props[(size_t)s * PSEG_NPROPS + k] = p[k];
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

// What the label fill reads (row table, runs, their labels: 14 MB for 200 frames) written once more in place.  A fill
// that was deferred behind the next call's background kernel finds those tables evicted from the caches -- 2.9 GB went through
// them since the chain wrote them -- and its waves, which live for a few microseconds, then wait for reads from memory under
// its own store stream: 1.53 ms instead of 1.0 ms (profiles/r03/21_deferred_fill.txt).  Reading the tables ahead of the fill did
// not help (1.43 ms), writing them does: lines that were just stored are served on-die, as they are when the fill follows
// its own chain.
__global__ __launch_bounds__(256) void k_refresh_tables(uint4 *a, size_t na, uint4 *b, size_t nb, uint4 *c, size_t nc)
{
    const size_t g = GTID, step = (size_t)gridDim.x * 256;
    for (size_t i = g; i < na; i += step) { uint4 v = a[i]; asm volatile("" : "+v"(v.x)); a[i] = v; }
    for (size_t i = g; i < nb; i += step) { uint4 v = b[i]; asm volatile("" : "+v"(v.x)); b[i] = v; }
    for (size_t i = g; i < nc; i += step) { uint4 v = c[i]; asm volatile("" : "+v"(v.x)); c[i] = v; }
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

    // (the labelling group is timed from ev[1], which run_bg / pseg_detect recorded behind the reduction kernel)
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
        ENSURE(ctx->scan_part, (size_t)SCAN_BLOCKS * 4 * 8);
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
    bool turn_released = false, fill_deferred = false, fill_ev11 = false;
    auto launch_fused = [&](uint32_t capv) {
        // (a batch that is repeated after its turn was handed on may meet another context's band kernel on the device: tickets)
        const int band_order = (turn_released && ctx->band_order == 1) ? 0 : ctx->band_order;
        const dim3 g(nbpf * n), b(256);
        if (ctx->turn) ctx->turn->band_begin();              // (behind the band kernel of the call before this one)
#define BAND_LAUNCH(WPL_, CAP_) hipLaunchKernelGGL((k_runs_band<WPL_, CAP_>), g, b, 0, ctx->cur, S, runs_wpl, capv, scan, bandmax_d, band_order)
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
        if (ctx->turn) ctx->turn->band_end();
        if (H > BAND_H) hipLaunchKernelGGL(k_unions_rows, dim3((H - 1) / BAND_H, n), dim3(64), 0, ctx->cur, S, BAND_H, capv, (uint32_t)ctx->band_cap);
    };
    // The chain behind the band labelling comes in two parts.  Part A is what the label map needs: the enclosure tree and the
    // labels of the runs (k_rank_ext).  Part B -- boxes, areas, ranks, rows, size scans: the table -- is queued BEHIND the label
    // fill, after the GPU's turn has been handed on, and runs beside the next batch's background kernel.  Both are gridded for
    // `runs` runs and guarded by (cap, bcap) on the device.
    auto launch_chain_a = [&](size_t runs, uint32_t capv, uint32_t bcap) {
        const size_t nbgr = runs + (size_t)n * H;
        hipLaunchKernelGGL(k_flatten_resolve, dim3(nblocks(nbgr)), dim3(256), 0, ctx->cur, S, capv, bcap);
        // (the run labels only feed the label fill; k_rank counts the external contours again for the table)
        if (labels) hipLaunchKernelGGL(k_rank_ext, dim3(n), dim3(1024), 0, ctx->cur, S, ftot3, (int32_t *)ctx->runlabel.p, capv, bcap);
    };
    auto launch_chain_b = [&](hipStream_t st, size_t runs, uint32_t capv, uint32_t bcap) {
        if (runs > 0) {
            hipLaunchKernelGGL(k_bbox, dim3(nblocks(runs)), dim3(256), 0, st, S, capv, bcap);
            hipLaunchKernelGGL(k_area, dim3(nblocks(runs)), dim3(256), 0, st, S, capv, bcap);
        }
        hipLaunchKernelGGL(k_rank, dim3(n), dim3(1024), 0, st, S, min_seg, min_save, ftot3, capv, bcap);
        hipLaunchKernelGGL(k_scan_small, dim3(2), dim3(256), 0, st, (const uint32_t *)(ftot3 + 1), rowbase, n, 3, 1);   // rowbase | savedbase
        if (runs > 0)
            hipLaunchKernelGGL(k_emit, dim3(nblocks(runs)), dim3(256), 0, st, S, capv, bcap, min_seg, min_save, (const uint32_t *)ftot3,
                               (const uint32_t *)rowbase, (const uint32_t *)savedbase, R, objlabel, objflag, saved_row,
                               geo, sz, sz_stride(rcap));
        const int scan_blocks = ctx->n_saved > ctx->scan_multi_min ? SCAN_BLOCKS : 1;      // (by the previous batch)
        if (scan_blocks > 1)
            hipLaunchKernelGGL(k_scan_sizes_part, dim3(scan_blocks), dim3(1024), 0, st, S, capv, bcap, (const unsigned long long *)sz, sz_stride(rcap),
                               (const uint32_t *)(savedbase + n), (unsigned long long *)ctx->scan_part.p);
        hipLaunchKernelGGL(k_scan_sizes, dim3(scan_blocks), dim3(1024), 0, st, S, capv, bcap, (const unsigned long long *)sz, sz_stride(rcap),
                           (const uint32_t *)(savedbase + n), (long long *)ctx->crop_off.p, (long long *)ctx->bits_off.p,
                           (long long *)ctx->band_off.p, totals, (const unsigned long long *)ctx->scan_part.p);
    };
    auto launch_fill = [&](hipStream_t fst, bool spec_, int nfg_) -> int {
        // one wave per row where rows hold many runs (FILL_ROWS_MIN per row on average: 1.0 ms against 1.7 ms at 19 runs per
        // row), one wave per 256-pixel span otherwise (0.86 ms against 0.96 ms at 3 runs per row); the speculative chain
        // does not know this batch's count yet and goes by the previous batch
        const bool by_rows = ctx->fill_rows >= 0 ? ctx->fill_rows > 0
                                                 : (spec_ ? ctx->dense_rows : (size_t)nfg_ >= (size_t)n * H * FILL_ROWS_MIN);
        const uint32_t c0 = spec_ ? (uint32_t)ctx->runcap : 0xffffffffu, c1 = spec_ ? (uint32_t)ctx->band_cap : 0xffffffffu;
        if (by_rows)
            hipLaunchKernelGGL(k_label_fill_rows, dim3(1, (H + 3) / 4, n), dim3(256), 0, fst, S, (const int32_t *)ctx->runlabel.p, labels, c0, c1);
        else
            hipLaunchKernelGGL(k_label_fill, dim3((W + 1023) / 1024, H, n), dim3(256), 0, fst, S, (const int32_t *)ctx->runlabel.p, labels, c0, c1);
        HIPCHK(hipGetLastError());
        return 0;
    };
    // a look-back scan in block-index order timed out: tickets from here on (this batch again, and every later one)
    auto band_order_fallback = [&]() -> int {
        ctx->band_order = 0;
        ctx->band_fallbacks++;
        HIPCHK(hipMemsetAsync(scan, 0, (size_t)scan_words(ctx, n) * 8, ctx->cur));
        HIPCHK(hipMemsetAsync(bandmax_d + 1, 0, 4, ctx->cur));
        return 0;
    };
    // Fast path: the whole chain is queued at once, sized by the tables' capacity (what the largest batch so far needed plus a
    // quarter); the host looks at the totals only when the head arrives, behind the label fill.  Slow path (first batch, a
    // denser batch, bands beyond the LDS labelling, very wide rows): totals first, then the chain.
    if (ctx->detect_calls++ == ctx->band_test_call && ctx->band_order == 1) ctx->band_order = 2;
    bool spec = fused && !ctx->no_bands && ctx->runcap > 0 && ctx->spec_hold == 0;
    if (ctx->spec_hold > 0) ctx->spec_hold--;
    int nfg = 0;
    uint32_t band_seen = 0;
    for (;;) {
        size_t chain_runs; uint32_t chain_cap, chain_bcap;   // part B is queued with what part A was queued with
        if (spec) {
            { int r_ = ensure_tables(ctx->runcap); if (r_) return r_; }
            launch_fused((uint32_t)ctx->runcap);
            chain_runs = ctx->runcap; chain_cap = (uint32_t)ctx->runcap; chain_bcap = (uint32_t)ctx->band_cap;
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
            if (fused && ((const volatile uint32_t *)ctx->pin_nfg.p)[2] != 0) {
                if (ctx->band_order == 0)
                    return fail(ctx, PSEG_ERR_HIP, "k_runs_band: a block waited for its predecessors' totals for too long");
                { int r_ = band_order_fallback(); if (r_) return r_; }
                continue;
            }
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
            chain_runs = (size_t)nfg; chain_cap = 0xffffffffu; chain_bcap = 0xffffffffu;
        }
        launch_chain_a(chain_runs, chain_cap, chain_bcap);
        HIPCHK(hipGetLastError());
        // where the table kernels go (ctx->table_mode): beside the label fill needs the context's own streams
        // Measured on one box (profiles/r03/10_table_modes.txt), frames/s of the 2000-frame stacks with 3 contexts:
        //   ordinary frames (3 runs per row):  before 98 300 - 99 600, beside 105 400 - 106 000, behind 101 200
        //   2000 objects per frame (19 runs per row, 440 us of table kernels):  before 58 700, beside 51 300, behind 61 700
        // Beside the fill the table kernels are some 20 000 short blocks of dependent loads among the fill's 512 000 blocks of
        // stores and the head reaches the host long before the fill ends; on dense frames they take the fill's wave slots for
        // 0.4 ms (fill 1.03 -> 1.30 ms) and go behind it instead, where they meet the next batch's background kernel.
        // A call that knows the next one is on its way lets that call's background kernel go first (DevSched): the labelling
        // chain above then runs beside the PREVIOUS call's label fill, and this fill beside the next call's chain.
        const bool defer = labels && ctx->turn && ctx->turn->want_defer();
        int tmode = ctx->table_mode >= 0 ? ctx->table_mode : (ctx->dense_rows ? 2 : 1);      // (density: by the previous batch)
        // (a deferred fill: the table kernels follow the chain on the main stream -- a side stream shares its hardware queue, of
        // which a process has four, with another context's main stream, whose label fill then waited behind these kernels)
        // (frames with hundreds of objects: 0.4 ms of table kernels, which the previous call's fill would have beside it together
        // with this call's chain -- 1.67 ms for that fill against 1.27 ms; they go behind this call's own fill instead)
        if (defer && tmode == 1) tmode = ctx->heavy_tables ? 2 : 0;
        if (tmode == 1 && ctx->turn && ctx->turn->sq) tmode = 0;   // (a call on the streaming queue that does not defer after all)
        if (tmode == 1 && !labels) tmode = 2;                // no label map: nothing to run beside; the turn is handed on first
        if (tmode == 1 && ctx->own_stream && ctx->cur == ctx->stream && !ctx->side && !ctx->side_tried) {
            ctx->side_tried = true;
            // (normal priority: with a high-priority side stream a call hung in hipEventSynchronize as soon as a handful of
            // contexts -- two streams each on the process's four hardware queues -- were alive; the same sequence at normal
            // priority runs, profiles/tools/r3_hang_probe.sh)
            if (hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking) != hipSuccess) ctx->side = nullptr;
        }
        if (tmode == 1 && !(ctx->own_stream && ctx->cur == ctx->stream && ctx->side && ctx->ev_a)) tmode = 0;
        ctx->ms[6] = 0;
        if (tmode == 0) {                                    // before the fill: the head travels to the host while the map is written
            launch_chain_b(ctx->cur, chain_runs, chain_cap, chain_bcap);
            { int r_ = fetch_head(ctx); if (r_) return r_; }
            HIPCHK(hipEventRecord(ctx->ev[8], ctx->cur));
        }
        TIMED_RECORD(3);                                     // the labelling group ends here, the label fill begins
        if (tmode == 1) {
            // beside the fill, on the side stream: about 20 000 short-lived blocks of dependent loads next to the fill's 512 000
            // blocks of stores; the head is on the host long before the fill ends, as in mode 0
            HIPCHK(hipEventRecord(ctx->ev_a, ctx->cur));
            HIPCHK(hipStreamWaitEvent(ctx->side, ctx->ev_a, 0));
            if (ctx->timing) HIPCHK(hipEventRecord(ctx->ev[9], ctx->side));
            launch_chain_b(ctx->side, chain_runs, chain_cap, chain_bcap);
            if (ctx->timing) HIPCHK(hipEventRecord(ctx->ev[10], ctx->side));
            { int r_ = fetch_head(ctx, ctx->side); if (r_) return r_; }
            HIPCHK(hipEventRecord(ctx->ev[8], ctx->side));
        }
        // (running the label fill on a second stream next to the crop / property kernels was measured: the fill takes 2.0 ms
        // instead of 0.93 ms when it shares the CUs, a net loss)
        fill_deferred = false; fill_ev11 = false;
        if (defer) { ctx->turn->defer_fill(); fill_deferred = true; ctx->n_deferred++; }
        if (labels) {
            // (the fill goes to the device's streaming queue when the call's background kernel went there; that queue then
            // waits for what the context's stream has queued so far -- the chain)
            hipStream_t fst = ctx->cur;
            if (ctx->turn) { ctx->turn->unstreamed(); fst = ctx->turn->stream_begin(true); }
            if (fill_deferred && ctx->fill_refresh) {        // (whole 16-byte pieces: the tables' capacities are padded)
                // On the streaming queue the fill does not wait for it: the refresh runs on the context's stream from the moment
                // the next call's background kernel ends, beside the fill's first 40 us (it stores the values that are there: a
                // wave of the fill reads the same bytes before and after).
                const bool beside = fst != ctx->cur && ctx->fill_refresh == 2;
                if (beside) ctx->turn->ctx_waits_streaming();
                const size_t cap_ = spec ? ctx->runcap : (size_t)nfg;
                hipLaunchKernelGGL(k_refresh_tables, dim3(4096), dim3(256), 0, beside ? ctx->cur : fst, (uint4 *)S.rowoff, (size_t)n * (H + 1) / 4,
                                   (uint4 *)S.run_x, cap_ / 4, (uint4 *)ctx->runlabel.p, cap_ / 4);
            }
            if ((fill_deferred || fst != ctx->cur) && ctx->timing) { HIPCHK(hipEventRecord(ctx->ev[11], fst)); fill_ev11 = true; }
            int r_ = launch_fill(fst, spec, nfg);
            if (r_) return r_;
            if (ctx->timing) HIPCHK(hipEventRecord(ctx->ev[4], fst));
            if (ctx->turn && fst != ctx->cur) { ctx->turn->streamed(); ctx->turn->stream_end(); }
        } else
            TIMED_RECORD(4);
        // The streaming kernels of this batch are queued.  What follows (crops, properties, hulls) waits on memory latency and
        // fills neither the vector units nor the memory system: the next batch's background kernel may run beside it.
        if (ctx->turn && ctx->early_release && !turn_released) { ctx->turn->release(); turn_released = true; }
        if (tmode == 2) {                                    // behind the fill, beside the next batch's background kernel (measured:
            launch_chain_b(ctx->cur, chain_runs, chain_cap, chain_bcap);   // the tail then outlasts that kernel and meets the next chain)
            { int r_ = fetch_head(ctx); if (r_) return r_; }
            HIPCHK(hipEventRecord(ctx->ev[8], ctx->cur));
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventSynchronize(ctx->ev[8]));             // (in mode 1 the table kernels have finished: the crop kernels may follow)
        if (tmode == 1 && ctx->timing) HIPCHK(hipEventElapsedTime(&ctx->ms[6], ctx->ev[9], ctx->ev[10]));
        if (!spec) break;
        const HeadHost h0 = head_host(ctx, n);
        if (h0.totals[6] != 0) {
            if (ctx->band_order == 0)
                return fail(ctx, PSEG_ERR_HIP, "k_runs_band: a block waited for its predecessors' totals for too long");
            { int r_ = band_order_fallback(); if (r_) return r_; }
            spec = false;
            continue;
        }
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
    ctx->heavy_tables = (size_t)head_host(ctx, n).rowbase[n] > (size_t)n * 500;
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
    const size_t o_props = (o_srow + hs * 4 + 15) & ~(size_t)15;        // float64 [n_saved][32], written by k_props
    const size_t o_bytes = o_props + hs * PSEG_NPROPS * 8;
    ENSURE(ctx->outpack, o_bytes);
    ENSURE_PIN(ctx->pin_tail, o_bytes);
    long long *d_sums = (long long *)((char *)ctx->outpack.p + o_sums);
    if (n_saved > 0) {
        // (band column of the size scan: bands | small windows << 40, see k_emit)
        const long long n_bands = (long long)((unsigned long long)h_tot[2] & (CROP_SMALL_ONE - 1)), n_small = (long long)((unsigned long long)h_tot[2] >> 40);
        if (n_bands > 0x7fffffffLL) return fail(ctx, PSEG_ERR_CAPACITY, "too many crop bands");
        ENSURE(ctx->crops_raw, (size_t)h_tot[0]); ENSURE(ctx->crops_masked, (size_t)h_tot[0]); ENSURE(ctx->crops_mask, (size_t)h_tot[0]);
        ENSURE(ctx->maskbits, (size_t)h_tot[1] * 2 * 4);     // first / last column of every window row (convex hull input)
        ENSURE(ctx->band_obj, (size_t)n_bands * BAND_DESC_INTS * 4);
        hipLaunchKernelGGL(k_band_map, dim3(nblocks((size_t)n_saved, 4)), dim3(256), 0, ctx->cur, n_saved, (const long long *)ctx->band_off.p,
                           (const int32_t *)geo, (const long long *)ctx->crop_off.p, (const long long *)ctx->bits_off.p,
                           (int32_t *)ctx->band_obj.p, d_sums);
#define CROPS_ARGS(first) S, n_saved, (const int32_t *)geo, (const long long *)ctx->crop_off.p, (const long long *)ctx->band_off.p, (const int32_t *)objflag, \
                   cleaned, raw, (uint8_t *)ctx->crops_raw.p, (uint8_t *)ctx->crops_masked.p, (uint8_t *)ctx->crops_mask.p, d_sums,            \
                   (const long long *)ctx->bits_off.p, (int32_t *)ctx->maskbits.p, (const int32_t *)nullptr, (const uint8_t *)nullptr,          \
                   (const int32_t *)ctx->band_obj.p + BAND_DESC_INTS * (size_t)(first)
        // two size classes: the descriptors of the small windows come first (k_band_map); PSEG_CROP_CLASSES=0: one class for all
        // (the small class only where it carries most of the batch: with a quarter of small windows among large ones -- the
        // headline workload -- the two launches take 0.195 against 0.173 ms alone, and beside the next batch's background kernel
        // the large class then ran 0.31 instead of 0.22 ms)
        if (ctx->crop_classes && n_small >= 2048 && 2 * n_small >= n_bands) {
            if (n_small > 0)
                hipLaunchKernelGGL((k_crops_sums<false, CROP_SMALL_THREADS, CROP_SMALL_TPR, CROP_SMALL_WORDS>), dim3((unsigned)n_small), dim3(CROP_SMALL_THREADS), 0, ctx->cur, CROPS_ARGS(0), 1);
            if (n_bands > n_small)
                hipLaunchKernelGGL((k_crops_sums<false, CROP_THREADS, CROP_TPR, CROP_LDS_WORDS>), dim3((unsigned)(n_bands - n_small)), dim3(CROP_THREADS), 0, ctx->cur, CROPS_ARGS(n_small), 2);
        } else
            hipLaunchKernelGGL((k_crops_sums<false, CROP_THREADS, CROP_TPR, CROP_LDS_WORDS>), dim3((unsigned)n_bands), dim3(CROP_THREADS), 0, ctx->cur, CROPS_ARGS(0), 0);
#undef CROPS_ARGS
        const int hmax = (int)h_tot[3];
        // two size classes when tall objects are present: objects up to 256 rows keep 14 KB of LDS per block, the rest is sized by
        // the tallest one
        // (a tallest object of up to 448 rows = 25 KB still leaves 6 blocks per CU: one launch, one kernel latency less)
        const int hsmall = hmax <= 448 ? hmax : 256;
        const size_t lds_small = (size_t)hsmall * HULL_LDS_PER_ROW + 64, lds_big = (size_t)hmax * HULL_LDS_PER_ROW + 64;
        if (hmax > 2700) return fail(ctx, PSEG_ERR_CAPACITY, "saved object too tall for the convex-hull kernel (limit 2700 rows)");
        if (lds_big > 48 * 1024)
            HIPCHK(hipFuncSetAttribute((const void *)k_convex, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
        // windows of up to CONVEX_SMALL_H rows: a wave per window, no LDS (PSEG_CONVEX_SMALL=0: every window through k_convex)
        const int hlo = ctx->convex_small ? CONVEX_SMALL_H : 0;
        if (hlo > 0)
            hipLaunchKernelGGL(k_convex_small, dim3(nblocks((size_t)n_saved, 4)), dim3(256), 0, ctx->cur, n_saved, (const int32_t *)geo,
                               (const long long *)ctx->bits_off.p, (const int32_t *)ctx->maskbits.p, d_sums);
        if (hmax > hlo)
        hipLaunchKernelGGL(k_convex, dim3(n_saved), dim3(128), lds_small, ctx->cur, n_saved, (const int32_t *)geo,
                           (const long long *)ctx->bits_off.p, (const int32_t *)ctx->maskbits.p, d_sums, hlo, hsmall);
        if (hmax > hsmall)
            hipLaunchKernelGGL(k_convex, dim3(n_saved), dim3(128), lds_big, ctx->cur, n_saved, (const int32_t *)geo,
                               (const long long *)ctx->bits_off.p, (const int32_t *)ctx->maskbits.p, d_sums, hsmall, hmax);
        HIPCHK(hipGetLastError());
    }
    TIMED_RECORD(5);
    ctx->crop_bytes = h_tot[0];
    ctx->n_saved = n_saved;

    // ---- results to the host: pack on the device, ONE copy into page-locked memory ----
    char *hp = (char *)ctx->pin_tail.p;
    int64_t *h_sums = (int64_t *)(hp + o_sums), *h_crop_off = (int64_t *)(hp + o_coff);
    int32_t *hrows = (int32_t *)(hp + o_rows), *h_saved_row = (int32_t *)(hp + o_srow);
    h_crop_off[0] = 0;
    if (n_saved > 0)
        hipLaunchKernelGGL(k_props, dim3(nblocks((size_t)n_saved)), dim3(256), 0, ctx->cur, n_saved, d_sums, (const int32_t *)geo,
                           (double *)((char *)ctx->outpack.p + o_props));
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
#ifdef HOST_PROF
    const auto hp_t0 = std::chrono::steady_clock::now();
#endif
    TIMED_ELAPSED(2, 1, 3);
    TIMED_ELAPSED(3, fill_ev11 ? 11 : 3, 4);
    TIMED_ELAPSED(4, 4, 5);

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
    double *h_props = (double *)(hp + o_props);                       // [n_saved][32]
    for (int i = 0; i < n_rows; i++) {
        h_area[i] = 0.5 * (double)hrows[3 * hr + i];
        h_s8[i] = (uint8_t)hrows[8 * hr + i];
    }
#ifdef HOST_PROF
    fprintf(stderr, "host epilogue %.1f us (%d rows, %d saved)\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hp_t0).count(), n_rows, n_saved);
#endif
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
    const unsigned npart = nblk * 4;                         // partial sums per frame: one uint4 per wave
    ENSURE(ctx->bits, (size_t)n * P / 8);
    ENSURE(ctx->partials, (size_t)n * npart * 4 * sizeof(uint32_t));
    ENSURE(ctx->maskpart, (size_t)npart * 4 + 16);
    { int r_ = head_setup(ctx, n); if (r_) return r_; }
    // (the two-call path never defers its fill: between its calls runs the caller's network, a streaming kernel the library does
    // not order -- with deferred fills the stand-in bench lost 4 %)
    GpuTurn turn(ctx, false);
    TIMED_RECORD(6);
    if (valid) HIPCHK(hipMemcpyAsync(ctx->valid_dev.p, valid, n, hipMemcpyHostToDevice, ctx->cur));
    else HIPCHK(hipMemsetAsync(ctx->valid_dev.p, 1, n, ctx->cur));
    const hipStream_t tst = ctx->turn ? ctx->turn->stream_begin(true) : ctx->cur;
    if (ctx->timing) HIPCHK(hipEventRecord(ctx->ev[0], tst));
    {
        const dim3 grid(nblk, (n + THR_FCHUNK - 1) / THR_FCHUNK);
        if (P % 2048 == 0)
            hipLaunchKernelGGL(k_threshold<false>, grid, dim3(256), 0, tst, cleaned, P, mask, threshold, (uint8_t *)ctx->bits.p, (uint4 *)ctx->partials.p, n, (uint32_t *)ctx->maskpart.p);
        else
            hipLaunchKernelGGL(k_threshold<true>, grid, dim3(256), 0, tst, cleaned, P, mask, threshold, (uint8_t *)ctx->bits.p, (uint4 *)ctx->partials.p, n, (uint32_t *)ctx->maskpart.p);
    }
    HIPCHK(hipGetLastError());
    if (ctx->turn && tst != ctx->cur) { ctx->turn->streamed(); ctx->turn->stream_end(); }
    hipLaunchKernelGGL(k_reduce_partials, dim3(n), dim3(1024), 0, ctx->cur, (const uint32_t *)ctx->partials.p, (int)npart,
                       (unsigned long long)P, (unsigned long long *)ctx->fsums.p, (uint8_t *)ctx->valid_dev.p, 0, 2, 2,
                       (unsigned long long *)ctx->maskcnt.p, (uint32_t *)ctx->framebase.p + n + 1, (const uint32_t *)ctx->maskpart.p,
                       mask ? (int)npart : 0, (unsigned long long *)ctx->scan_state.p, scan_words(ctx, n));
    ctx->mask_counted = mask != nullptr;                     // (k_threshold left per-wave counts of the mask pixels)
    HIPCHK(hipGetLastError());
    TIMED_RECORD(1);
    if (ctx->turn && tst == ctx->cur) ctx->turn->streamed();
    r = run_detect(ctx, cleaned, raw, mask, n, min_area_to_segment, min_area_to_save, labels, out);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[7], ctx->cur));
    HIPCHK(hipEventSynchronize(ctx->ev[7]));
    TIMED_ELAPSED(1, 0, 1);
    TIMED_ELAPSED(5, 6, 7);
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
    GpuTurn turn(ctx, labels != nullptr);
    TIMED_RECORD(6);
    r = run_bg(ctx, frames, n_avail, n_emit, n_bg_imgs, mask, threshold, cleaned, batch_len);
    if (r) return r;
    r = run_detect(ctx, cleaned, cleaned, mask, n, min_area_to_segment, min_area_to_save, labels, out);
    if (r) return r;
    HIPCHK(hipEventRecord(ctx->ev[7], ctx->cur));
    r = fetch_raw_stats(ctx, n, mean_std, valid);
    if (r) return r;
    HIPCHK(hipEventSynchronize(ctx->ev[7]));
    TIMED_ELAPSED(0, 0, 1);
    TIMED_ELAPSED(5, 6, 7);
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
    // (no text for pseg_last_error here: the worker may be writing ctx->err inside the call that is still in flight)
    if (w->have_job || w->done) return PSEG_ERR_ARG;
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
    hipLaunchKernelGGL((k_crops_sums<true, CROP_THREADS, CROP_TPR, CROP_LDS_WORDS>), dim3(nbands), dim3(CROP_THREADS), 0, st, S, 1, (const int32_t *)(d + o_geo), (const long long *)(d + o_ll),
                       (const long long *)(d + o_ll + 16), (const int32_t *)nullptr, img, img, c0, c0 + (size_t)w * h, c0 + (size_t)2 * w * h,
                       d_sums, (const long long *)(d + o_ll + 32), (int32_t *)(d + o_ext), (const int32_t *)(d + o_bobj), mask, (const int32_t *)nullptr, 0);
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
    if (ctx->device < 0 || ctx->device >= PSEG_MAX_DEVICES || !g_rp_static_ptr[ctx->device] || !ctx->pin_rp.p)
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
    hipLaunchKernelGGL((k_crops_sums<true, CROP_THREADS, CROP_TPR, CROP_LDS_WORDS>), dim3(nbands), dim3(CROP_THREADS), 0, st, S, 1, (const int32_t *)(d + RP_NA_O_GEO),
                       (const long long *)(d + RP_NA_O_LL), (const long long *)(d + RP_NA_O_LL + 16), (const int32_t *)nullptr, img, img,
                       c0, c0 + (size_t)w * h, c0 + (size_t)2 * w * h, d_sums, (const long long *)(d + RP_NA_O_LL + 32),
                       (int32_t *)(d + RP_NA_O_EXT), (const int32_t *)(d + RP_NA_O_BOBJ), mask, (const int32_t *)nullptr, 0);
    hipLaunchKernelGGL(k_convex, dim3(1), dim3(128), lds, st, 1, (const int32_t *)(d + RP_NA_O_GEO), (const long long *)(d + RP_NA_O_LL + 32),
                       (const int32_t *)(d + RP_NA_O_EXT), d_sums, 0, h);
    HIPCHK(hipGetLastError());
    // the sums come back into the context's page-locked buffer (a single DMA write; a pageable destination would go through a
    // staging buffer of the runtime's).  The call ends with a stream synchronisation: it is not for streams under capture.
    int64_t q[PSEG_NSUMS];
    HIPCHK(hipMemcpyAsync(ctx->pin_rp.p, d_sums, sizeof(q), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    memcpy(q, ctx->pin_rp.p, sizeof(q));
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
