"""Thin Python face of the C ABI: device tensors in, object tables out.  torch is used for
device memory only (``tensor.data_ptr()``); all arithmetic happens in libpiscoseg.so."""
import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

PROP_KEYS = [
    "width", "height", "bx", "by", "circ.", "area_exc", "area_rprops", "%area", "major", "minor",
    "centroid_y", "centroid_x", "convex_area", "min_intensity", "max_intensity", "mean_intensity",
    "intden", "perim.", "elongation", "range", "perimareaexc", "perimmajor", "circex", "angle",
    "bounding_box_area", "eccentricity", "equivalent_diameter", "euler_number", "extent",
    "local_centroid_col", "local_centroid_row", "solidity",
]


# every column is produced on the device (exact integer sums) and finished in float64 by the library
PENDING_PROPS = set()


class PsegError(RuntimeError):
    pass


def _np(ptr, n, dtype):
    if n <= 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


@dataclass
class ObjectTable:
    """Host copy of pseg_table (include/piscoseg.h); rows ordered (frame, c) like the reference's
    per-frame CSV (detection.py:265,271)."""
    n_frames: int
    frame_row_offset: np.ndarray
    frame_n_contours: np.ndarray
    frame_valid: np.ndarray
    frame_mean: np.ndarray
    frame_std: np.ndarray
    frame: np.ndarray
    c: np.ndarray
    label: np.ndarray
    area: np.ndarray
    x: np.ndarray
    y: np.ndarray
    w: np.ndarray
    h: np.ndarray
    saved: np.ndarray
    saved_props: np.ndarray                       # [n_saved, 32] region properties of saved object s
    saved_row: np.ndarray
    crop_offset: np.ndarray
    sums: np.ndarray
    crop_bytes: int
    crops: dict = field(default_factory=dict)     # 'raw' / 'masked' / 'mask' packed uint8 planes (host), optional

    @property
    def n_rows(self):
        return len(self.frame)

    @property
    def props(self):
        """[n_rows, 32]: the reference's columns 13..44, nan for unsaved rows (detection.py:269-271)."""
        full = np.full((self.n_rows, len(PROP_KEYS)), np.nan)
        full[self.saved_row] = self.saved_props
        return full

    def crop(self, s, plane="masked"):
        """h x w uint8 crop of saved object s from a fetched plane."""
        i = self.saved_row[s]
        a = self.crop_offset[s]
        return self.crops[plane][a:a + self.h[i] * self.w[i]].reshape(self.h[i], self.w[i])


class Segmenter:
    """One context per GPU / Python thread.  Mirrors the stage split of the reference:
    ``bgcorrect`` = bg_correction.correct_img over a batch, ``detect`` = detection.detect_on_img over a
    batch, ``process`` = both fused (segmenter.py:241,254 without deconvolution)."""

    def __init__(self, H=2560, W=2560, max_frames=400, device=0):
        import threading
        self.L = _lib.load()
        self.H, self.W, self.max_frames, self.device = H, W, max_frames, device
        self.lock = threading.RLock()          # for callers that share a context between threads (one call at a time)
        h = C.c_void_p()
        rc = self.L.pseg_create(device, H, W, max_frames, C.byref(h))
        if rc != 0:
            why = {-1: "bad geometry: 2 <= H <= 65535, W a multiple of 64 up to 8192, 1 <= max_frames <= 65535, "
                       "max_frames * (H + 1) < 2^31", -2: "a HIP call failed", -4: "no usable HIP device"}.get(rc, "")
            raise PsegError(f"pseg_create({device}, {H}, {W}, {max_frames}) failed with code {rc} ({why})")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.pseg_destroy(self.h)                    # (waits for a submitted call first)
            self.h = None
            self._pending = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PsegError(f"libpiscoseg error {rc}: {self.L.pseg_last_error(self.h).decode()}")

    def _check_frames(self, t, n=None, dtype=None):
        import torch
        dtype = dtype or torch.uint8
        if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
            raise PsegError("expected a contiguous CUDA tensor of dtype %s" % dtype)
        if t.shape[-2:] != (self.H, self.W) or (n is not None and t.shape[0] != n):
            raise PsegError(f"bad tensor shape {tuple(t.shape)}")

    def tuning_state(self):
        """what the context has adapted to (pseg_tuning_state): run-table capacity, band size class, slow-path calls left,
        row-wise label fill for the next batch"""
        return dict(run_capacity=int(self.L.pseg_tuning_state(self.h, 0)), band_class=int(self.L.pseg_tuning_state(self.h, 1)),
                    slow_path_calls=int(self.L.pseg_tuning_state(self.h, 2)), fill_by_rows=bool(self.L.pseg_tuning_state(self.h, 3)),
                    band_order=int(self.L.pseg_tuning_state(self.h, 4)), band_order_fallbacks=int(self.L.pseg_tuning_state(self.h, 5)),
                    bg_streaming=int(self.L.pseg_tuning_state(self.h, 6)), fills_deferred=int(self.L.pseg_tuning_state(self.h, 7)),
                    calls_on_streaming_queue=int(self.L.pseg_tuning_state(self.h, 8)),
                    deferred_fills_given_up=int(self.L.pseg_tuning_state(self.h, 9)))

    def last_ms(self, which):
        return float(self.L.pseg_last_ms(self.h, which))

    def disk_mask(self, radius):
        import torch
        m = torch.empty((self.H, self.W), dtype=torch.uint8, device=f"cuda:{self.device}")
        self._chk(self.L.pseg_disk_mask(self.h, int(radius), m.data_ptr()))
        return m

    def resize_frames(self, src, out=None):
        """``cv.resize(img, (W, H))`` of the reader (reader.py:79: OpenCV's default INTER_LINEAR on uint8) for a stack of
        frames ``src`` uint8 ``[n, sh, sw]`` on this device -> ``[n, H, W]``.  Decoded frames are uploaded at their native
        size and resized here instead of on the host."""
        import torch
        if src.dim() == 2:
            src = src[None]
        if src.dtype != torch.uint8 or src.dim() != 3 or not src.is_cuda or not src.is_contiguous():
            raise PsegError("resize_frames: need a contiguous uint8 CUDA tensor [n, sh, sw]")
        n, sh, sw = src.shape
        if out is None:
            out = torch.empty((n, self.H, self.W), dtype=torch.uint8, device=src.device)
        self._chk(self.L.pseg_resize_frames(self.h, src.data_ptr(), int(n), int(sh), int(sw), out.data_ptr()))
        return out

    def bgcorrect(self, frames, n_bg_imgs=6, cleaned=None):
        import torch
        n = frames.shape[0]
        self._check_frames(frames)
        if cleaned is None:
            cleaned = torch.empty_like(frames)
        self._check_frames(cleaned, n)
        ms = np.zeros((n, 2), np.float64)
        valid = np.zeros(n, np.uint8)
        torch.cuda.current_stream(frames.device).synchronize()
        self._chk(self.L.pseg_bgcorrect(self.h, frames.data_ptr(), n, n_bg_imgs, cleaned.data_ptr(), ms.ctypes.data, valid.ctypes.data))
        return cleaned, ms, valid

    def _table(self, t, fetch_crops):
        n, nr, ns = t.n_frames, t.n_rows, t.n_saved
        self._n_saved = ns
        tab = ObjectTable(
            n_frames=n,
            frame_row_offset=_np(t.frame_row_offset, n + 1, np.int32), frame_n_contours=_np(t.frame_n_contours, n, np.int32),
            frame_valid=_np(t.frame_valid, n, np.uint8), frame_mean=_np(t.frame_mean, n, np.float64), frame_std=_np(t.frame_std, n, np.float64),
            frame=_np(t.row_frame, nr, np.int32), c=_np(t.row_c, nr, np.int32), label=_np(t.row_label, nr, np.int32),
            area=_np(t.row_area, nr, np.float64), x=_np(t.row_x, nr, np.int32), y=_np(t.row_y, nr, np.int32),
            w=_np(t.row_w, nr, np.int32), h=_np(t.row_h, nr, np.int32), saved=_np(t.row_saved, nr, np.uint8),
            saved_props=_np(t.saved_props, ns * _lib.NPROPS, np.float64).reshape(ns, _lib.NPROPS),
            saved_row=_np(t.saved_row, ns, np.int32), crop_offset=_np(t.crop_offset, ns + 1, np.int64),
            sums=_np(t.sums, ns * _lib.NSUMS, np.int64).reshape(ns, _lib.NSUMS), crop_bytes=int(t.crop_bytes))
        if fetch_crops:
            self.fetch_crops(tab, fetch_crops)
        return tab

    def fetch_crops(self, tab, mode=True):
        """the three crop planes of the context's last table into host memory (pseg_fetch_crops): ``tab.crops``"""
        planes = self._crop_buffers(max(tab.crop_bytes, 1), mode)
        self._chk(self.L.pseg_fetch_crops(self.h, planes["raw"].ctypes.data, planes["masked"].ctypes.data, planes["mask"].ctypes.data))
        tab.crops = {k: v[:tab.crop_bytes] for k, v in planes.items()}
        return tab

    def _crop_buffers(self, nbytes, mode):
        """host buffers of the three crop planes.  mode "pinned": page-locked memory owned by the context and reused by its
        next fetch (the copies are single DMA transfers at PCIe rate; pageable memory goes through a staging buffer at a
        fraction of it) -- for consumers that are done with the crops before the context's next call.  Otherwise fresh
        NumPy arrays that the caller may keep."""
        if mode != "pinned":
            return {k: np.empty(nbytes, np.uint8) for k in ("raw", "masked", "mask")}
        import torch
        if getattr(self, "_pin_crops", None) is None or self._pin_crops.shape[1] < nbytes:
            self._pin_crops = torch.empty((3, nbytes + nbytes // 4 + 4096), dtype=torch.uint8, pin_memory=True)
        a = self._pin_crops.numpy()
        return {"raw": a[0], "masked": a[1], "mask": a[2]}

    def detect(self, cleaned, raw=None, mask=None, valid=None, threshold=10, min_area_to_segment=10.0,
               min_area_to_save=400.0, labels=None, fetch_crops=False):
        import torch
        n = cleaned.shape[0]
        self._check_frames(cleaned)
        if raw is not None:
            self._check_frames(raw, n)
        if labels is not None:
            self._check_frames(labels, n, torch.int32)
        v = None if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
        t = _lib.Table()
        torch.cuda.current_stream(cleaned.device).synchronize()
        self._chk(self.L.pseg_detect(self.h, cleaned.data_ptr(), None if raw is None else raw.data_ptr(),
                                     None if mask is None else mask.data_ptr(), None if v is None else v.ctypes.data,
                                     n, int(threshold), float(min_area_to_segment), float(min_area_to_save),
                                     None if labels is None else labels.data_ptr(), C.byref(t)))
        return self._table(t, fetch_crops)

    def _process_args(self, frames, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save, cleaned, labels, n_emit,
                      batch_len):
        import torch
        n_avail = frames.shape[0]
        n = n_avail if n_emit is None else int(n_emit)
        self._check_frames(frames)
        if cleaned is None:
            cleaned = torch.empty((n,) + tuple(frames.shape[1:]), dtype=torch.uint8, device=frames.device)
        self._check_frames(cleaned, n)
        if labels is not None:
            self._check_frames(labels, n, torch.int32)
        if batch_len is not None and batch_len < n_avail:
            if n != n_avail:
                raise PsegError("n_emit and batch_len exclude each other")
            bl = int(batch_len)
        else:
            bl = 0
        ms = np.zeros((n, 2), np.float64)
        valid = np.zeros(n, np.uint8)
        torch.cuda.current_stream(frames.device).synchronize()
        args = (frames.data_ptr(), n_avail, n, bl, n_bg_imgs, None if mask is None else mask.data_ptr(), int(threshold),
                float(min_area_to_segment), float(min_area_to_save), cleaned.data_ptr(), ms.ctypes.data, valid.ctypes.data,
                None if labels is None else labels.data_ptr())
        return args, cleaned, ms, valid

    def process(self, frames, n_bg_imgs=6, mask=None, threshold=10, min_area_to_segment=10.0, min_area_to_save=400.0,
                cleaned=None, labels=None, fetch_crops=False, want_table=True, n_emit=None, batch_len=None):
        """Fused hot path over one batch.  ``n_emit`` < len(frames): ``frames`` is the leading part of a batch plus its
        forward halo (sharding.plan_ranges); only the first n_emit frames are processed and returned.
        ``batch_len``: ``frames`` holds several consecutive reference batches of that many frames (the last may be
        shorter), each background-corrected inside its own batch; one table over all frames."""
        args, cleaned, ms, valid = self._process_args(frames, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save,
                                                      cleaned, labels, n_emit, batch_len)
        t = _lib.Table()
        fr, n_avail, n, bl = args[:4]
        if bl:
            self._chk(self.L.pseg_process_batches(self.h, fr, n_avail, bl, *args[4:], C.byref(t)))
        else:
            self._chk(self.L.pseg_process_part(self.h, fr, n_avail, n, *args[4:], C.byref(t)))
        tab = self._table(t, fetch_crops) if want_table else None
        return cleaned, ms, valid, tab

    def submit(self, frames, n_bg_imgs=6, mask=None, threshold=10, min_area_to_segment=10.0, min_area_to_save=400.0,
               cleaned=None, labels=None, n_emit=None, batch_len=None):
        """``process`` without waiting (pseg_submit): the call runs on the context's worker thread inside the library and
        this returns at once; ``wait`` delivers what ``process`` returns.  One submission in flight per context; a caller
        thread pipelines batches by alternating two or three contexts."""
        if getattr(self, "_pending", None) is not None:
            raise PsegError("submit: the previous submission has not been waited for")
        args, cleaned, ms, valid = self._process_args(frames, n_bg_imgs, mask, threshold, min_area_to_segment, min_area_to_save,
                                                      cleaned, labels, n_emit, batch_len)
        self._chk(self.L.pseg_submit(self.h, *args))
        self._pending = (cleaned, ms, valid, (frames, mask, labels))          # keeps every buffer alive until wait()
        return self

    def wait(self, fetch_crops=False, want_table=True):
        """(cleaned, mean_std, valid, table) of the submitted call (pseg_wait)."""
        if getattr(self, "_pending", None) is None:
            raise PsegError("wait: nothing was submitted")
        cleaned, ms, valid, _keep = self._pending
        self._pending = None
        t = _lib.Table()
        self._chk(self.L.pseg_wait(self.h, C.byref(t)))
        tab = self._table(t, fetch_crops) if want_table else None
        return cleaned, ms, valid, tab

    def regionprops(self, mask, img, no_alloc=False):
        """calculate_regionprops (detection.py:75) of one window: mask / img are 2-D uint8 CUDA tensors of equal shape
        (rows may be strided).  Returns float64[32] in PROP_KEYS order.  ``no_alloc``: the entry point that works in a
        static device array (windows up to 1024 rows / 262144 pixels) and allocates nothing."""
        h, w = mask.shape
        if img.shape != mask.shape or mask.stride(1) != 1 or img.stride(1) != 1 or mask.stride(0) != img.stride(0):
            raise PsegError("mask and img must be 2-D uint8 windows of the same shape and row stride")
        import torch
        torch.cuda.current_stream(mask.device).synchronize()
        props = np.zeros(_lib.NPROPS, np.float64)
        fn = self.L.pseg_regionprops_noalloc if no_alloc else self.L.pseg_regionprops
        self._chk(fn(self.h, mask.data_ptr(), img.data_ptr(), int(h), int(w), int(mask.stride(0)), props.ctypes.data, None))
        return props

    def vit_tensors(self, first=0, count=None, plane="masked", out=None):
        """Classifier inputs of saved objects [first, first+count) of the last detection, straight from the packed device
        crops: what Classify_ViT.py:92-145 (resize the longer edge to 224 with PIL's bilinear filter, pad with white,
        ToTensor, Normalize(0.5, 0.5)) makes of the PNG the reference would have written, minus the random rotation.
        Returns (float32 CUDA tensor [count, 3, 224, 224], valid uint8[count]); valid 0 = a crop so thin that its resized
        short edge is 0 px (the reference logs the error and skips the file)."""
        import torch
        ns = self.n_saved_last()
        count = ns - first if count is None else int(count)
        if out is None:
            out = torch.empty((count, 3, 224, 224), dtype=torch.float32, device=f"cuda:{self.device}")
        if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != count * 3 * 224 * 224 or out.device.index != self.device:
            raise PsegError("out must be a contiguous float32 tensor [count, 3, 224, 224] on the segmenter's GPU")
        valid = np.zeros(max(count, 1), np.uint8)
        torch.cuda.current_stream(out.device).synchronize()
        self._chk(self.L.pseg_vit_tensors(self.h, {"raw": 0, "masked": 1, "mask": 2}[plane], int(first), count, out.data_ptr(),
                                          valid.ctypes.data))
        return out, valid[:count]

    def n_saved_last(self):
        return self._n_saved

    def synth_frames(self, n, seed, objs, frame_obj_off, frame0=0, out=None):
        import torch
        if out is None:
            out = torch.empty((n, self.H, self.W), dtype=torch.uint8, device=f"cuda:{self.device}")
        objs = np.ascontiguousarray(objs, dtype=np.int32).reshape(-1, 8)
        off = np.ascontiguousarray(frame_obj_off, dtype=np.int32)
        assert off.shape[0] == n + 1 and off[-1] == objs.shape[0]
        self._chk(self.L.pseg_synth_frames(self.h, n, frame0, int(seed), objs.ctypes.data, off.ctypes.data, out.data_ptr()))
        return out


class HostPipeline:
    """Batches that start in host memory (decoded files, a camera buffer): upload, fused hot path and the way back of the
    table and crops, overlapped across batches from ONE caller thread.  ``depth`` contexts (pseg_submit / pseg_wait) with a
    page-locked staging buffer and device buffers each: while batch b computes, batch b+1 crosses PCIe (22.8 ms per
    200 frames of 2560 x 2560 at 57 GB/s against 2 ms of kernels -- the link is the bound of this form of the path,
    DESIGN.md section 6).  No counterpart in the reference, which moves every frame through pickled queue items
    (segmenter.py:221-254)."""

    def __init__(self, H=2560, W=2560, max_frames=200, device=0, depth=2, labels=False):
        import torch
        dev = f"cuda:{device}"
        self.H, self.W, self.max_frames, self.depth = H, W, max_frames, depth
        self.segs = [Segmenter(H, W, max_frames=max_frames, device=device) for _ in range(depth)]
        self.host = [torch.empty((max_frames, H, W), dtype=torch.uint8, pin_memory=True) for _ in range(depth)]
        self.raw = [torch.empty((max_frames, H, W), dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.cleaned = [torch.empty((max_frames, H, W), dtype=torch.uint8, device=dev) for _ in range(depth)]
        self.labels = [torch.empty((max_frames, H, W), dtype=torch.int32, device=dev) if labels else None for _ in range(depth)]
        self.copy_stream = torch.cuda.Stream(device=dev)
        self.uploaded = [torch.cuda.Event() for _ in range(depth)]

    def staging(self, k):
        """uint8 [max_frames, H, W] NumPy view of slot k's page-locked buffer: decode / copy the next batch straight into it"""
        return self.host[k].numpy()

    def run(self, batches, mask=None, fetch_crops="pinned", **kw):
        """``batches`` yields ``(n, source)``: source is None (the producer has filled ``staging(b % depth)[:n]`` already),
        a uint8 NumPy array / CPU tensor [n, H, W] to copy into the staging buffer, or a page-locked CPU tensor that is
        uploaded as it is.  Yields ``(b, cleaned_dev[:n], mean_std, valid, table)`` in batch order; the device tensors and
        pinned crops of slot b % depth are reused by batch b + depth."""
        import torch
        inflight = []

        def finished(b0, k0, n0):                           # the slot's kernels are done: its device buffers are free again
            c, ms, v, tab = self.segs[k0].wait(fetch_crops=False)
            return b0, k0, c, ms, v, tab

        def result(b0, k0, c, ms, v, tab):
            if fetch_crops:
                self.segs[k0].fetch_crops(tab, fetch_crops)
            return b0, c, ms, v, tab
        for b, (n, src) in enumerate(batches):
            k = b % self.depth
            done = finished(*inflight.pop(0)) if len(inflight) == self.depth else None
            if src is None:
                h = self.host[k][:n]
            elif torch.is_tensor(src) and src.is_pinned():
                h = src[:n]
            else:
                h = self.host[k][:n]
                h.numpy()[...] = src.numpy() if torch.is_tensor(src) else src
            with torch.cuda.stream(self.copy_stream):
                self.raw[k][:n].copy_(h, non_blocking=True)
                self.uploaded[k].record(self.copy_stream)
            # the crops of the batch that has just left this slot travel to the host while the next batch travels to the
            # device (the link is full duplex; one after the other they cost 30 ms per batch instead of 23)
            if done is not None:
                yield result(*done)
            self.uploaded[k].synchronize()                  # earlier batches compute on their contexts' threads meanwhile
            self.segs[k].submit(self.raw[k][:n], mask=mask, cleaned=self.cleaned[k][:n],
                                labels=None if self.labels[k] is None else self.labels[k][:n], **kw)
            inflight.append((b, k, n))
        for it in inflight:
            yield result(*finished(*it))

    def close(self):
        for s in self.segs:
            s.close()
