/*
 * piscoseg.h -- C ABI of libpiscoseg.so, the MI355X (gfx950) implementation of the hot path of
 * fannz00/ImageProcessing: background correction (bg_correction.py) -> threshold -> external
 * objects -> per-object area / bbox / crops / region-property sums (detection.py).
 *
 * The reference is pure Python and has no FFI; the entry points below are what a ctypes binding
 * placed inside the reference's own functions would call (INTEGRATION.md shows the stubs).  Each
 * declaration cites the reference code it replaces (paths are /root/reference/...).
 *
 * Conventions: every function returns 0 on success and a negative code on failure (never
 * aborts; the reference's convention is "print and carry on").  pseg_last_error() gives the text.
 * Pointers marked [dev] are device pointers on the context's GPU (e.g. torch.Tensor.data_ptr()),
 * [host] are host pointers.  Frames are row-major uint8 [n][H][W]; W % 64 == 0, W <= 8192, H <= 65535,
 * n <= max_frames <= 65535 and max_frames * (H + 1) < 2^31 (pseg_create refuses anything else with PSEG_ERR_ARG);
 * a saved object may be at most 2700 rows tall (convex-hull kernel).
 * A context is not re-entrant: one call at a time per context (one context per Python thread/GPU).
 */
#ifndef PISCOSEG_H
#define PISCOSEG_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pseg_ctx pseg_ctx;

#define PSEG_OK 0
#define PSEG_ERR_ARG (-1)
#define PSEG_ERR_HIP (-2)
#define PSEG_ERR_CAPACITY (-3)
#define PSEG_ERR_NODEVICE (-4)

#define PSEG_NPROPS 32 /* columns 13..44 of the reference's CSV row, detection.py:84-152 */
#define PSEG_NSUMS 24  /* integer sums per saved object, see pseg_table.sums */

/* Result of one detection call.  All arrays are host memory owned by the context and stay valid
 * until the next call on that context.  Row i describes one kept object (contourArea >=
 * min_area_to_segment); rows are ordered by (frame, c) exactly like the reference's per-frame CSV
 * (detection.py:222-273).  Replaces: the crop_data list of detect_on_img (detection.py:220,265,271). */
typedef struct pseg_table {
    int32_t n_frames;
    int32_t n_rows;                   /* kept objects over all frames                               */
    int32_t n_saved;                  /* rows with contourArea > min_area_to_save                   */
    int64_t crop_bytes;               /* total bytes of one packed crop plane (sum of w*h, saved)   */
    const int32_t *frame_row_offset;  /* [n_frames+1] rows of frame f = [off[f], off[f+1])          */
    const int32_t *frame_n_contours;  /* [n_frames]  external contours before the area filters      */
    const uint8_t *frame_valid;       /* [n_frames]  0 = corrupt frame (raw std <= 2), no rows      */
    const double *frame_mean;         /* [n_frames]  mean of corrected inside the mask (CSV col 5)  */
    const double *frame_std;          /* [n_frames]  std  of corrected inside the mask (CSV col 6)  */
    const int32_t *row_frame;         /* [n_rows]                                                   */
    const int32_t *row_c;             /* [n_rows] object index c (CSV col 1, file name suffix)      */
    const int32_t *row_label;         /* [n_rows] 1-based OpenCV contour index = label-map value    */
    const double *row_area;           /* [n_rows] cv.contourArea (CSV col 7)                        */
    const int32_t *row_x, *row_y, *row_w, *row_h; /* cv.boundingRect, top-left + size              */
    const uint8_t *row_saved;         /* [n_rows] 1 if contourArea > min_area_to_save (CSV col 12)  */
    const int32_t *saved_row;         /* [n_saved] row index of saved object s                      */
    const int64_t *crop_offset;       /* [n_saved+1] byte offset of object s in each crop plane     */
    const double *saved_props;        /* [n_saved*32] regionprops columns of saved object s (the reference writes
                                         32 x nan for unsaved rows, detection.py:269-271)              */
    const int64_t *sums;              /* [n_saved*PSEG_NSUMS] exact integer sums behind saved_props */
    /* packed crops on the device, row-major h*w per saved object at crop_offset[s]:              */
    const uint8_t *crops_raw_dev;     /* raw_bg_corr[y:y+h, x:x+w]          (Crops/,  detection.py:244) */
    const uint8_t *crops_masked_dev;  /* where(c_mask==255, crop, 255)      (Deconv_crops/, :247)      */
    const uint8_t *crops_mask_dev;    /* cumulative filled mask window      (Masks/,  :245-246)        */
} pseg_table;

/* Create / destroy a context bound to one GPU.  max_frames bounds n of every later call. */
int pseg_create(int device, int H, int W, int max_frames, pseg_ctx **out);
int pseg_destroy(pseg_ctx *ctx);
const char *pseg_last_error(const pseg_ctx *ctx);
/* Run the context's work on an existing HIP stream (hipStream_t) instead of its own. */
int pseg_set_stream(pseg_ctx *ctx, void *hip_stream);

/* Background correction of one batch.  Replaces correct_img for every index of the batch
 * (bg_correction.py:41-90; driven by run_bg_correction :93-127):
 *   mean/std of the raw frame (:69-70), gate std > 2 (:71), window start (:72),
 *   bg = max_k min(I[s+k], I[s+k+1]) over 2*n_bg_imgs frames (:74-78), cleaned = 255-|img-bg| (:80-81).
 * frames [dev] u8[n][H][W]; cleaned [dev] u8[n][H][W] (contents of corrupt frames unspecified);
 * mean_std [host] double[n][2]; valid [host] u8[n].  6 <= n_bg_imgs*... : needs n >= n_bg_imgs
 * (the reference raises IndexError below that). */
int pseg_bgcorrect(pseg_ctx *ctx, const uint8_t *frames, int n, int n_bg_imgs, uint8_t *cleaned,
                   double *mean_std, uint8_t *valid);

/* Detection on one batch of background-corrected frames.  Replaces detect_on_img for every
 * frame (detection.py:162-297) minus PNG/CSV I/O:
 *   corrected = 255 - cleaned (:178); mean/std inside mask == 255 and zero outside (:185-188);
 *   thresh = corrected > threshold (:195-202); external contours in OpenCV order (:204);
 *   contourArea / boundingRect (:208-216); area filters (:223,235); cumulative FILLED mask,
 *   three crops (:243-247); sums for the 32 region properties (:75-159).
 * cleaned [dev] u8[n][H][W]; raw [dev] the bg_corr image the Crops/ plane is cut from (may equal
 * cleaned); mask [dev] u8[H][W] or NULL (settings.mask_img False); valid [host] u8[n] or NULL;
 * labels [dev] int32[n][H][W] or NULL: k on the foreground pixels of the component of contour k
 * (1-based, OpenCV order), 0 elsewhere. */
int pseg_detect(pseg_ctx *ctx, const uint8_t *cleaned, const uint8_t *raw, const uint8_t *mask,
                const uint8_t *valid, int n, int threshold, double min_area_to_segment,
                double min_area_to_save, int32_t *labels, pseg_table *out);

/* Fused hot path: pseg_bgcorrect + pseg_detect with the threshold plane and the masked sums
 * produced inside the background kernel (no second pass over cleaned).  Same outputs as the two
 * calls in sequence with raw == cleaned (the reference without deconvolution, segmenter.py:248-249). */
int pseg_process(pseg_ctx *ctx, const uint8_t *frames, int n, int n_bg_imgs, const uint8_t *mask,
                 int threshold, double min_area_to_segment, double min_area_to_save,
                 uint8_t *cleaned, double *mean_std, uint8_t *valid, int32_t *labels, pseg_table *out);

/* The leading part of a reference batch, for stacks cut into frame ranges over several GPUs (the
 * reference's unit is the whole 200-frame batch, segmenter.py:209-213).  frames [dev] u8[n_avail][H][W] holds the
 * part's frames followed by its forward halo; the first n_emit frames are processed, each with the window
 * frames[i .. i + 2*n_bg_imgs) (bg_correction.py:72 without the clamp), so n_avail - n_emit >= 2*n_bg_imgs - 1 when
 * n_emit < n_avail.  n_emit == n_avail is pseg_process (the part that ends the batch: the last 2*n_bg_imgs frames share
 * one window).  cleaned / labels / mean_std / valid / out cover the n_emit frames.  A part must not begin inside the
 * last 2*n_bg_imgs frames of its batch (imageprocessing_amd.sharding.plan_ranges snaps the cuts). */
int pseg_process_part(pseg_ctx *ctx, const uint8_t *frames, int n_avail, int n_emit, int n_bg_imgs,
                      const uint8_t *mask, int threshold, double min_area_to_segment, double min_area_to_save,
                      uint8_t *cleaned, double *mean_std, uint8_t *valid, int32_t *labels, pseg_table *out);

/* Several consecutive reference batches in one call: frames [dev] u8[n][H][W] = the batches run_segmenter would cut
 * (segmenter.py:209-213: batch_len = 200 files, the last batch takes the remainder), each background-corrected with its own
 * window clamp (bg_correction.py:72) exactly as separate pseg_process calls would, then detected together (frames are
 * independent from there on).  One table, frame indices 0 .. n-1 across the batches.  Every batch needs >= n_bg_imgs
 * frames (as the reference).  Fewer, larger launches: the labelling kernels are latency-bound per launch. */
int pseg_process_batches(pseg_ctx *ctx, const uint8_t *frames, int n, int batch_len, int n_bg_imgs, const uint8_t *mask,
                         int threshold, double min_area_to_segment, double min_area_to_save, uint8_t *cleaned,
                         double *mean_std, uint8_t *valid, int32_t *labels, pseg_table *out);

/* Asynchronous form of the fused path, for ONE caller thread that keeps several batches in flight (the reference overlaps
 * its stages with process pools and bounded queues, segmenter.py:221-254; here the overlap is between batches).  The
 * context owns a worker thread inside the library: pseg_submit hands it the arguments of pseg_process_part (batch_len == 0)
 * or pseg_process_batches (batch_len > 0, n_emit == n_avail) and returns at once; pseg_wait blocks until that call has
 * finished, returns its status and fills *out.  All pointers must stay valid until pseg_wait returns; one submission in flight
 * per context (a second pseg_submit before pseg_wait is PSEG_ERR_ARG, and leaves pseg_last_error alone: the call in flight owns it).  With two or three contexts on one GPU --
 * submit(A), submit(B), wait(A), submit(A), wait(B), ... -- the host round trips and the table epilogue of one batch hide
 * under the kernels of the next, without any thread on the caller's side.  Results are identical to the synchronous calls. */
int pseg_submit(pseg_ctx *ctx, const uint8_t *frames, int n_avail, int n_emit, int batch_len, int n_bg_imgs,
                const uint8_t *mask, int threshold, double min_area_to_segment, double min_area_to_save,
                uint8_t *cleaned, double *mean_std, uint8_t *valid, int32_t *labels);
int pseg_wait(pseg_ctx *ctx, pseg_table *out);

/* Copy the three packed crop planes of the last detection to host buffers of crop_bytes each
 * (any may be NULL).  Replaces the arrays handed to cv.imwrite (detection.py:251-253). */
int pseg_fetch_crops(pseg_ctx *ctx, uint8_t *raw, uint8_t *masked, uint8_t *mask);

/* calculate_regionprops(mask, img) (detection.py:75-159) for one window: the 32 columns of regionprop2zooprocess in the
 * reference's order.  mask [dev] u8 (nonzero = region pixel), img [dev] u8, both h rows of `stride` bytes, w <= 8192,
 * h <= 2700; props [host] double[32]; sums_out [host] int64[PSEG_NSUMS] or NULL.  The window must be the region's
 * bounding box (as in the reference, where bx = by = 0 and width / height are the window's) and hold no background
 * hole (8-connected) -- true for every window detect_on_img passes, whose masks are drawn FILLED (detection.py:245):
 * filled_area is taken equal to area.  An empty mask is an error (the reference indexes regionprops(...)[0]). */
int pseg_regionprops(pseg_ctx *ctx, const uint8_t *mask, const uint8_t *img, int h, int w, int stride,
                     double *props, int64_t *sums_out);

/* pseg_regionprops for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers that must not
 * reach hipMalloc / hipFree -- tight serving loops, harnesses that forbid allocation after start-up.  Same arguments, same
 * kernels and bit-identical results as pseg_regionprops; the scratch (tables, sums, row extremes and the three crop planes
 * the band kernel writes) is a file-scope device array of about 0.8 MB that every GPU the library runs on holds until the
 * process ends, so the window is limited: h <= PSEG_NOALLOC_MAX_ROWS, w <= 8192 and w * h <= PSEG_NOALLOC_MAX_PIXELS
 * (PSEG_ERR_CAPACITY beyond that).  The call allocates nothing: no hipMalloc / hipFree / hipHostMalloc, and the 192 bytes of
 * sums come back into a page-locked buffer that pseg_create made (no pageable staging inside the runtime).  It does END WITH A
 * STREAM SYNCHRONISATION (props is host memory), so it cannot run on a stream that is being captured into a HIP graph.
 * All calls on one GPU share the array: calls made through this library in one process are serialised by a per-device
 * lock; anything else that overlaps them on the same array would race. */
#define PSEG_NOALLOC_MAX_PIXELS 262144
#define PSEG_NOALLOC_MAX_ROWS 1024
int pseg_regionprops_noalloc(pseg_ctx *ctx, const uint8_t *mask, const uint8_t *img, int h, int w, int stride,
                             double *props, int64_t *sums_out);

/* SURVEY.md 8(f) N3 -- classifier inputs of saved objects [first, first+count) of the context's last detection, read from
 * the packed device crops (plane 0 = raw, 1 = masked: the Deconv_crops/ images the classifier is pointed at, 2 = mask).
 * Replaces, for crops that never left the GPU, Image.open(...).convert('RGB') + get_pad_transform's ResizeAndPad((224, 224))
 * + ToTensor + Normalize(0.5, 0.5) (Classify_ViT.py:92-145; the RandomRotation(180, fill=255) of that pipeline is random
 * and left to the caller): longer edge -> 224 with PIL's bilinear resize (int-truncated other edge), centred white
 * padding (odd pixel right / bottom), (p/255 - 0.5)/0.5 in float32, grey replicated to 3 channels.
 * out [dev] float[count][3][224][224]; valid [host] u8[count] or NULL: 0 where the resized short edge would be 0 px
 * (the reference's load raises and the file is skipped, Classify_ViT.py:170-176; the tensor is then all padding). */
int pseg_vit_tensors(pseg_ctx *ctx, int plane, int first, int count, float *out, uint8_t *valid);

/* cv.circle(mask, (W/2, H/2), radius, 255, -1) on the device (detection.py:315-326). mask [dev] u8[H][W]. */
int pseg_disk_mask(pseg_ctx *ctx, int radius, uint8_t *mask);

/* cv.resize(img, (2560, 2560)) of the reader (reader.py:79), i.e. OpenCV's INTER_LINEAR on 8-bit images, for n frames
 * of sh x sw pixels into the context's H x W: the library's 11-bit fixed-point weights and two-pass rounding; the area
 * average it substitutes for a reduction by exactly 2 x 2; a plain copy for equal sizes.  src [dev] u8[n][sh][sw];
 * dst [dev] u8[n][H][W].  SURVEY.md 8(f) N4a: decoding stays on the host, frames are uploaded at their native size. */
int pseg_resize_frames(pseg_ctx *ctx, const uint8_t *src, int n, int sh, int sw, uint8_t *dst);

/* Synthetic PISCO-like frames rendered on the device (bench / tests; no counterpart in the
 * reference).  objs [host] int32[n_obj][8] = cx,cy,a,b,cosQ12,sinQ12,contrast,kind;
 * frame_obj_off [host] int32[n+1]; out [dev] u8[n][H][W]. */
int pseg_synth_frames(pseg_ctx *ctx, int n, int frame0, uint64_t seed, const int32_t *objs,
                      const int32_t *frame_obj_off, uint8_t *out);

/* Duration in milliseconds of the named kernel group of the last call, measured with HIP events on the
 * context's stream: 0 = background kernel, 1 = threshold, 2 = sparse labelling chain,
 * 3 = label-map write, 4 = masks + crops + sums, 5 = whole call, 6 = the table kernels (boxes, areas, ranks, rows, size scans)
 * when they ran on the side stream beside the label-map write (0 when they ran inside group 2 or 4). */
float pseg_last_ms(const pseg_ctx *ctx, int which);

/* What the context has adapted to over the batches it has seen (no counterpart in the reference; for tests and
 * diagnosis): 0 = capacity of the run tables (runs; the labelling chain is queued without a host round trip while a
 * batch stays below it), 1 = size class of the band kernel (runs a 64-row band may hold for the labelling in LDS:
 * 1024, 2048 or 4096), 2 = calls for which the chain is queued only after the run count is known (after a batch too dense
 * even for the largest class), 3 = 1 when the next label map is written row-wise (dense rows), 0 span-wise,
 * 4 = order of the band kernel's look-back scan (1 = block index, the default; 0 = tickets from an atomic counter, which
 * a context moves to for good after a scan in index order timed out -- the batch is repeated inside the call; environment
 * PSEG_BAND_ORDER=0 starts there, and so do contexts that take no turns, PSEG_EXCLUSIVE=0; several processes sharing one GPU
 * should set it: their band kernels can block each other in index order until the bound, about 0.1 s, ends it),
 * 5 = times that has happened, 6 = 1 when the last background correction ran the streaming kernel (windows of 8, 10, 12, 16
 * frames in batches of at least one window), 0 the kernel that re-reads the window per frame, -1 before the first call,
 * 7 = calls of this context whose label-map write was deferred behind the next call's background kernel (several contexts
 * in flight on one GPU: the labelling of a batch then runs beside the previous batch's label-map write; DESIGN.md section 3;
 * PSEG_DEFER_FILL=0 turns it off), 8 = calls whose background kernel and label-map write were queued on the device's shared
 * streaming queue instead of the context's own stream (PSEG_STREAM_QUEUE=0 turns that off), 9 = deferred writes of this
 * context that were queued because the next call had not come by the time the device ran out of streaming work (callers
 * that resubmit slowly: when 2 of 4 deferred writes on a device end that way twice in a row, the device keeps the plain order
 * for its next 64 calls, then tries again, backing off to 128, 256 ... 8192 calls while the trials keep failing).
 * Returns -1 for a bad argument. */
long long pseg_tuning_state(const pseg_ctx *ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
