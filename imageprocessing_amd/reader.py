"""Input container of the hot path -- mirrors /root/reference/reader.py.

``ReaderOutput.images[i] = (img, fn)`` is the contract the background stage consumes (reader.py:16-51).  Decoding is I/O
outside the GPU path: ``read_img`` uses PIL instead of cv2.  With ``native=True`` (what
``run_segmenter`` uses) a decoded frame keeps its own size and the resize of reader.py:79 (``cv.resize`` to 2560 x 2560,
OpenCV's INTER_LINEAR) is done by the library on the device after the upload (``pseg_resize_frames``); with
``native=False`` PIL's bilinear filter resizes on the host (not OpenCV's arithmetic: only for callers without a GPU stage
behind them).

Two additions to the reference's container, both invisible to code that only uses ``images`` / ``add_output``:
``failed`` / ``done`` let the background stage see that a file could not be decoded (the reference leaves the slot at None
and ``is_ready`` then spins forever, bg_correction.py:66-67), and ``staging`` is an optional page-locked buffer the frames
are decoded into, so that the upload of a batch is one DMA transfer."""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class ReaderOutput:
    """reader.py:16-51.  ``manager`` may be a multiprocessing.Manager (as in the reference) or None (plain list: the GPU
    stages run in threads of one process, no pickling of frames).  ``staging``: a uint8 NumPy view ``[n, H, W]`` (of
    page-locked memory) -- frames of that shape are stored in it and ``images[i][0]`` is the view ``staging[i]``."""

    def __init__(self, n_images, manager=None, staging=None):
        self.lock = threading.Lock()
        self.images = manager.list([None for _ in range(n_images)]) if manager is not None else [None] * n_images
        self.failed = {}                      # index -> file name of the frames that could not be read
        self.done = threading.Event()         # set by run_reader when every file of the batch has been tried
        self.staging = staging

    def add_output(self, img, fn, index):
        st = self.staging
        if st is not None and index < st.shape[0] and img.shape == st.shape[1:] and img.dtype == st.dtype:
            st[index] = img
            img = st[index]
        with self.lock:
            self.images[index] = (img, fn)

    def mark_failed(self, fn, index):
        with self.lock:
            self.failed[index] = fn


def decode_gray(file):
    """cv.imread(file, cv.IMREAD_GRAYSCALE) of reader.py:75 with PIL: uint8 [h, w]."""
    from PIL import Image
    with Image.open(file) as im:
        return np.ascontiguousarray(np.asarray(im.convert("L"), dtype=np.uint8))


def read_img(output, input, thread_index=0, native=False):
    """reader.py:54-82.  native: no resize here (the frames go to ``Segmenter.resize_frames`` on the device).  A file that
    is empty or cannot be decoded is reported with ``output.mark_failed`` (the reference only prints)."""
    file, img_index = input
    fn = os.path.basename(file)
    try:
        if os.path.getsize(file) == 0:
            print(f"Thread {thread_index}: File {file}is an empty image file")
            raise ValueError("empty file")
        img = decode_gray(file)
        if not native and img.shape != (2560, 2560):
            from PIL import Image
            img = np.ascontiguousarray(np.asarray(Image.fromarray(img).resize((2560, 2560), Image.BILINEAR), dtype=np.uint8))
        output.add_output(img, fn, img_index)
    except Exception as e:                       # reference: print and carry on (reader.py:81-82)
        print("...exception occurred:", e)
        if hasattr(output, "mark_failed"):
            output.mark_failed(fn, img_index)


def run_reader(files, output, n_threads, resize, native=False):
    """reader.py:86-110: ``files`` is a list of (path, index).  (The reference ignores ``resize`` and always resizes.)"""
    try:
        with ThreadPoolExecutor(max_workers=max(1, n_threads)) as ex:
            list(ex.map(lambda f: read_img(output, f, native=native), files))
        print("all files in batch added to reader pool")
    finally:
        if hasattr(output, "done"):
            output.done.set()
