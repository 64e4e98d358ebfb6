"""Background-correction stage with the reference's call signatures (/root/reference/bg_correction.py),
computed by libpiscoseg.so on the GPU.  No CPU fallback: without the library / a GPU these raise."""
import threading
import time

import numpy as np

from .engine import Segmenter

_segs = {}
_lock = threading.Lock()


def _segmenter(H, W, n, device=0, role="bg"):
    """One long-lived context per (stage, device, frame size), shared by whatever threads run that stage: callers hold
    ``seg.lock`` for the duration of a call (a context takes one call at a time).  The two stages have contexts of their
    own so that the background correction of one batch runs beside the detection of the previous one."""
    key = (role, device, H, W)
    with _lock:
        s = _segs.get(key)
        if s is None or s.max_frames < n:
            if s is not None:
                with s.lock:
                    s.close()
            s = Segmenter(H, W, max_frames=max(n, 400), device=device)
            _segs[key] = s
    return s


def close_contexts():
    """free the cached contexts (device workspaces, page-locked buffers)"""
    with _lock:
        for s in _segs.values():
            with s.lock:
                s.close()
        _segs.clear()


def is_ready(img_index, input, n_bg_imgs):
    """bg_correction.py:12-38."""
    start = img_index if img_index + n_bg_imgs < len(input.images) else len(input.images) - n_bg_imgs
    for i in range(start, start + n_bg_imgs):
        if input.images[i] is None:
            return False
    return True


def correct_img(img_index, input, output, n_bg_imgs, index=0):
    """bg_correction.py:41-90 for one index: the 2*n_bg_imgs frames of its window are corrected as one small
    batch (every frame of a batch of exactly 2*n_bg_imgs frames uses the same window) and the frame's tuple
    ``(bg_corr_img, cleaned_img, [mean, stdev], fn)`` -- ``([], [], [mean, stdev], fn)`` for a corrupt frame -- is put
    on ``output``."""
    import torch
    nw = 2 * n_bg_imgs
    while not is_ready(img_index, input, nw):
        time.sleep(0.1)
    n = len(input.images)
    start = img_index if img_index + nw < n else n - nw
    idx = [(start + k) % n if start + k < 0 else start + k for k in range(nw)]      # Python's negative-index wrap
    img, fn = input.images[img_index]
    stack = np.stack([input.images[i][0] for i in idx])
    H, W = img.shape
    seg = _segmenter(H, W, nw)
    with seg.lock:
        cleaned, ms, valid = seg.bgcorrect(torch.from_numpy(stack).to(f"cuda:{seg.device}"), n_bg_imgs=n_bg_imgs)
    k = idx.index(img_index)
    mean, stdev = float(ms[k, 0]), float(ms[k, 1])
    if valid[k]:
        c = cleaned[k].cpu().numpy()
        output.put((c, c, [mean, stdev], fn))
    else:
        print("found corrupt image: ", fn)
        output.put(([], [], [mean, stdev], fn))


def _upload(images, device, resize_to, staging):
    """the frames of a batch as one uint8 device tensor [n, H, W].  Frames decoded into a page-locked staging buffer go
    up in one DMA transfer; frames of another size than ``resize_to`` are uploaded at their own size and resized on the
    device with OpenCV's INTER_LINEAR arithmetic (reader.py:79; Segmenter.resize_frames)."""
    import torch
    dev = f"cuda:{device}"
    n = len(images)
    shapes = {im.shape for im, _ in images}
    target = tuple(resize_to) if resize_to is not None else None
    if target is None or shapes == {target}:
        if len(shapes) != 1:
            raise ValueError(f"frames of different sizes in one batch: {sorted(shapes)}")
        if staging is not None and n <= staging.shape[0] and all(im.ctypes.data == staging[i].ctypes.data for i, (im, _) in enumerate(images)):
            return torch.from_numpy(staging[:n]).to(dev, non_blocking=True)
        return torch.from_numpy(np.stack([im for im, _ in images])).to(dev)
    out = torch.empty((n,) + target, dtype=torch.uint8, device=dev)
    seg = _segmenter(target[0], target[1], 1, device, role="bg")
    for shp in shapes:
        idx = [i for i, (im, _) in enumerate(images) if im.shape == shp]
        src = torch.from_numpy(np.stack([images[i][0] for i in idx])).to(dev)
        with seg.lock:
            res = seg.resize_frames(src)
        out[torch.as_tensor(idx, device=dev)] = res
    return out


def correct_batch(images, n_bg_imgs, device=0, keep_on_device=False, resize_to=None, staging=None):
    """All frames of one batch in one library call (the fast path behind run_bg_correction).
    Returns the list of reference tuples in frame order; with ``keep_on_device`` the images are rows of one uint8 CUDA
    tensor (the detection stage takes them without another upload)."""
    n = len(images)
    stack = _upload(images, device, resize_to, staging)
    H, W = stack.shape[-2:]
    seg = _segmenter(H, W, n, device, role="bg")
    with seg.lock:
        cleaned, ms, valid = seg.bgcorrect(stack, n_bg_imgs=n_bg_imgs)
    host = None if keep_on_device else cleaned.cpu().numpy()
    out = []
    for i, (_, fn) in enumerate(images):
        if valid[i]:
            c = cleaned[i] if keep_on_device else host[i]
            out.append((c, c, [float(ms[i, 0]), float(ms[i, 1])], fn))
        else:
            print("found corrupt image: ", fn)
            out.append(([], [], [float(ms[i, 0]), float(ms[i, 1])], fn))
    return out


def run_bg_correction(input, output, n_bg_imgs, running, keep_on_device=False, resize_to=None, device=0):
    """bg_correction.py:93-127: returns at once; a worker thread waits for the reader to fill the batch, corrects it on
    the GPU and puts one tuple per frame on ``output`` (frame order; the reference's order is arbitrary).

    Exactly ``len(input.images)`` tuples are put whatever happens, so that ``run_detection`` -- which counts them -- always
    returns: a frame whose file the reader could not decode (``input.failed``; the reference would wait for it forever,
    bg_correction.py:66-67) is left out of the batch and reported as the corrupt tuple ``([], [], [0, 0], fn)``; if the
    correction itself fails, the error is printed (the reference's convention) and every frame is reported corrupt."""
    n = len(input.images)

    def work():
        emitted = 0
        names = {}
        try:
            done = getattr(input, "done", None)
            while True:
                imgs = list(input.images)
                failed = dict(getattr(input, "failed", {}))
                missing = [i for i, im in enumerate(imgs) if im is None and i not in failed]
                if not missing:
                    break
                if done is not None and done.is_set():
                    imgs = list(input.images)              # the reader has finished: what is still empty will never come
                    failed = dict(getattr(input, "failed", {}))
                    for i in range(n):
                        if imgs[i] is None and i not in failed:
                            failed[i] = f"frame_{i:05d}.png"
                    break
                if running is not None and getattr(running, "value", 1) != 1:
                    return
                time.sleep(0.01)
            good = [i for i in range(n) if imgs[i] is not None and i not in failed]
            names = {i: (failed[i] if i in failed else imgs[i][1]) for i in range(n)}
            items = {}
            if good:
                res = correct_batch([imgs[i] for i in good], n_bg_imgs, device=device, keep_on_device=keep_on_device,
                                    resize_to=resize_to, staging=getattr(input, "staging", None) if len(good) == n else None)
                items = dict(zip(good, res))
            for i in range(n):
                if i in items:
                    output.put(items[i])
                else:
                    print("found unreadable image: ", names[i])
                    output.put(([], [], [0.0, 0.0], names[i]))
                emitted += 1
        except Exception as e:                   # reference convention: print and carry on
            print(f"Exception in bg correction: {e}")
            for i in range(emitted, n):
                im = input.images[i]
                output.put(([], [], [0.0, 0.0], names.get(i) or (im[1] if im is not None else f"frame_{i:05d}.png")))
    t = threading.Thread(target=work, name="bg_corr", daemon=True)
    t.start()
    return t
