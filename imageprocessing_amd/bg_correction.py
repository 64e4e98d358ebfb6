"""Background-correction stage with the reference's call signatures (/root/reference/bg_correction.py),
computed by libpiscoseg.so on the GPU.  No CPU fallback: without the library / a GPU these raise."""
import threading
import time

import numpy as np

from .engine import Segmenter

_segs = {}
_lock = threading.Lock()


def _segmenter(H, W, n, device=0):
    """one cached context per (thread, frame size)"""
    key = (threading.get_ident(), H, W, device)
    with _lock:
        s = _segs.get(key)
        if s is None or s.max_frames < n:
            if s is not None:
                s.close()
            s = Segmenter(H, W, max_frames=max(n, 400), device=device)
            _segs[key] = s
    return s


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
    cleaned, ms, valid = seg.bgcorrect(torch.from_numpy(stack).to(f"cuda:{seg.device}"), n_bg_imgs=n_bg_imgs)
    k = idx.index(img_index)
    mean, stdev = float(ms[k, 0]), float(ms[k, 1])
    if valid[k]:
        c = cleaned[k].cpu().numpy()
        output.put((c, c, [mean, stdev], fn))
    else:
        print("found corrupt image: ", fn)
        output.put(([], [], [mean, stdev], fn))


def correct_batch(images, n_bg_imgs, device=0, keep_on_device=False):
    """All frames of one batch in one library call (the fast path behind run_bg_correction).
    Returns the list of reference tuples in frame order."""
    import torch
    n = len(images)
    H, W = images[0][0].shape
    seg = _segmenter(H, W, n, device)
    stack = torch.from_numpy(np.stack([im for im, _ in images])).to(f"cuda:{device}")
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


def run_bg_correction(input, output, n_bg_imgs, running):
    """bg_correction.py:93-127: returns at once; a worker thread waits for the reader to fill the batch, corrects
    it on the GPU and puts one tuple per frame on ``output`` (frame order; the reference's order is arbitrary)."""
    def work():
        try:
            while any(im is None for im in input.images):
                if running is not None and getattr(running, "value", 1) != 1:
                    return
                time.sleep(0.01)
            for item in correct_batch(list(input.images), n_bg_imgs):
                output.put(item)
        except Exception as e:                   # reference convention: print and carry on
            print(f"Exception in bg correction: {e}")
    t = threading.Thread(target=work, name="bg_corr", daemon=True)
    t.start()
    return t
