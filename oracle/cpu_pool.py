"""CPU baseline with the reference's process topology (test / bench infrastructure only, never the product path).

The reference runs background correction in a pool of 3 processes (bg_correction.py:120, segmenter.py:235) and
detection in a pool of 8 (segmenter.py:252, process_pool.py) over one batch of frames.  This module does the same with
the CPU restatement in ``oracle.pipeline_ref``: frames travel through /dev/shm memory maps instead of pickled queue
items (kinder to the baseline than the reference's queues), PNG/CSV writing is off.
"""
import multiprocessing as mp
import os
import tempfile
import time

import numpy as np

_state = {}


def _init(raw_path, cleaned_path, shape, mask_path):
    from oracle import cv_ref
    cv_ref.lib()                                           # loads oracle/liboracle.so (built beforehand by build())
    _state["raw"] = np.lib.format.open_memmap(raw_path, mode="r")
    _state["cleaned"] = np.lib.format.open_memmap(cleaned_path, mode="r+")
    _state["mask"] = np.load(mask_path)


def _warm(_):
    return os.getpid()


def _bg(i):
    from oracle import pipeline_ref as ref
    raw = _state["raw"]
    images = [(raw[k], f"img_{k:05d}.png") for k in range(raw.shape[0])]
    it = ref.correct_img(i, images, 6)
    ok = not isinstance(it[1], list)
    if ok:
        _state["cleaned"][i] = it[1]
    return i, ok, it[2]


def _det(args):
    from oracle import pipeline_ref as ref
    i, ok, ms = args
    fn = f"img_{i:05d}.png"
    item = (np.array(_state["cleaned"][i]), np.array(_state["cleaned"][i]), ms, fn) if ok else ([], [], ms, fn)
    d = ref.detect_on_img(item, _state["mask"], want_labels=False)
    return len(d["rows"])


def run(stack, mask, n_bg_procs=3, n_det_procs=8):
    """stack: uint8 [n, H, W] (one reference batch); returns dict(seconds_bg, seconds_det, rows, procs)."""
    tmp = tempfile.mkdtemp(prefix="pseg_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    raw_path, cl_path, mask_path = (os.path.join(tmp, k) for k in ("raw.npy", "cleaned.npy", "mask.npy"))
    try:
        np.save(raw_path, stack)
        np.lib.format.open_memmap(cl_path, mode="w+", dtype=np.uint8, shape=stack.shape).flush()
        np.save(mask_path, mask)
        ctx = mp.get_context("spawn")                      # never fork a process that has initialised HIP
        init = (raw_path, cl_path, stack.shape, mask_path)
        n = stack.shape[0]
        with ctx.Pool(n_bg_procs, initializer=_init, initargs=init) as pb, ctx.Pool(n_det_procs, initializer=_init, initargs=init) as pd:
            pb.map(_warm, range(4 * n_bg_procs)); pd.map(_warm, range(4 * n_det_procs))   # workers up, oracle library loaded
            t0 = time.perf_counter()
            items = pb.map(_bg, range(n), chunksize=1)
            t1 = time.perf_counter()
            rows = pd.map(_det, items, chunksize=1)
            t2 = time.perf_counter()
        return dict(seconds_bg=t1 - t0, seconds_det=t2 - t1, rows=int(sum(rows)), procs=(n_bg_procs, n_det_procs))
    finally:
        for f in (raw_path, cl_path, mask_path):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)
