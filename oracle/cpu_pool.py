"""CPU baseline with the reference's process topology (test / bench infrastructure only, never the product path).

The reference runs background correction in a pool of 3 processes (bg_correction.py:120, segmenter.py:241) and detection
in a pool of 8 (segmenter.py:252, process_pool.py) AT THE SAME TIME: the background workers put their tuples on a
``Queue(10)`` (segmenter.py:222), ``run_detection`` takes them off one by one (detection.py:333-336) and hands them to the
detection pool through its own bounded task queue (``ProcessPool(..., max_tasks)``, process_pool.py).  This module keeps that
shape with the CPU restatement in ``oracle.pipeline_ref``: both pools run concurrently, at most 10 finished background
items wait between them, at most 10 detection tasks are queued.  Frames travel through /dev/shm memory maps instead of
pickled queue items (kinder to the baseline than the reference's queues).  ``write_dir``: the PNG / CSV files of
detection.py:251-253,293 are written too (PIL's PNG encoder stands in for cv.imwrite).
"""
import multiprocessing as mp
import os
import queue
import tempfile
import threading
import time

import numpy as np

_state = {}


def _init(raw_path, cleaned_path, shape, mask_path, write_dir):
    from oracle import cv_ref
    cv_ref.lib()                                           # loads oracle/liboracle.so (built beforehand by build())
    _state["raw"] = np.lib.format.open_memmap(raw_path, mode="r")
    _state["cleaned"] = np.lib.format.open_memmap(cleaned_path, mode="r+")
    _state["mask"] = np.load(mask_path)
    _state["write_dir"] = write_dir


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
    wd = _state["write_dir"]
    if wd:                                                 # detection.py:251-253, 293-297
        import csv
        from PIL import Image
        for c, masked, raw, cmask in d["crops"]:
            name = f"{fn[:-4]}_{c}.png"
            Image.fromarray(masked).save(os.path.join(wd, "Deconv_crops", name))
            Image.fromarray(raw).save(os.path.join(wd, "Crops", name))
            Image.fromarray(cmask).save(os.path.join(wd, "Masks", name))
        with open(os.path.join(wd, "Data", fn[:-4] + ".csv"), "w", newline="") as f:
            csv.writer(f, delimiter=",").writerows(d["rows"])
    return len(d["rows"])


def run(stack, mask, n_bg_procs=3, n_det_procs=8, write_dir=None, queue_len=10):
    """stack: uint8 [n, H, W] (one reference batch).  Returns dict(seconds, rows, procs, seconds_bg): wall clock of the two
    pools running concurrently, and the time until the background pool had finished."""
    tmp = tempfile.mkdtemp(prefix="pseg_cpu_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    raw_path, cl_path, mask_path = (os.path.join(tmp, k) for k in ("raw.npy", "cleaned.npy", "mask.npy"))
    if write_dir:
        for k in ("Data", "Crops", "Deconv_crops", "Masks"):
            os.makedirs(os.path.join(write_dir, k), exist_ok=True)
    try:
        np.save(raw_path, stack)
        np.lib.format.open_memmap(cl_path, mode="w+", dtype=np.uint8, shape=stack.shape).flush()
        np.save(mask_path, mask)
        ctx = mp.get_context("spawn")                      # never fork a process that has initialised HIP
        init = (raw_path, cl_path, stack.shape, mask_path, write_dir)
        n = stack.shape[0]
        with ctx.Pool(n_bg_procs, initializer=_init, initargs=init) as pb, ctx.Pool(n_det_procs, initializer=_init, initargs=init) as pd:
            pb.map(_warm, range(4 * n_bg_procs)); pd.map(_warm, range(4 * n_det_procs))   # workers up, oracle library loaded
            between = queue.Queue(queue_len)               # bg_output_queue = Queue(10), segmenter.py:222
            slots = threading.Semaphore(queue_len)         # the detection pool's bounded task queue
            t_bg = [0.0]
            t0 = time.perf_counter()

            def feed():
                for item in pb.imap_unordered(_bg, range(n), chunksize=1):       # completion order, as in the reference
                    between.put(item)
                t_bg[0] = time.perf_counter() - t0
            th = threading.Thread(target=feed)
            th.start()
            res = []
            for _ in range(n):                             # detection.py:333-336
                item = between.get()
                slots.acquire()
                res.append(pd.apply_async(_det, (item,), callback=lambda _r: slots.release(),
                                          error_callback=lambda _e: slots.release()))
            rows = [r.get() for r in res]
            t1 = time.perf_counter()
            th.join()
        return dict(seconds=t1 - t0, seconds_bg=t_bg[0], rows=int(sum(rows)), procs=(n_bg_procs, n_det_procs))
    finally:
        for f in (raw_path, cl_path, mask_path):
            if os.path.exists(f):
                os.remove(f)
        os.rmdir(tmp)
