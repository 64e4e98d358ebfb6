"""Input container of the hot path -- mirrors /root/reference/reader.py.

``ReaderOutput.images[i] = (img, fn)`` is the contract the background stage consumes
(reader.py:16-51).  Decoding is I/O outside the GPU path: ``read_img`` uses
PIL instead of cv2 (grayscale, bilinear resize to 2560x2560 only when the size differs), so decoded
pixels of JPEGs / resized frames are not claimed bit-identical to OpenCV's.  ``run_reader(..., native=True)``
leaves the frames at their decoded size; ``Segmenter.resize_frames`` then applies OpenCV's resize arithmetic
(INTER_LINEAR, reader.py:79) on the device after the upload."""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np


class ReaderOutput:
    """reader.py:16-51.  ``manager`` may be a multiprocessing.Manager (as in the reference) or None
    (plain list: the GPU stages run in threads of one process, no pickling of frames)."""

    def __init__(self, n_images, manager=None):
        self.lock = threading.Lock()
        self.images = manager.list([None for _ in range(n_images)]) if manager is not None else [None] * n_images

    def add_output(self, img, fn, index):
        with self.lock:
            self.images[index] = (img, fn)


def read_img(output, input, thread_index=0, native=False):
    """reader.py:54-82.  native: no resize here (frames go to ``Segmenter.resize_frames`` on the device)."""
    from PIL import Image
    file, img_index = input
    fn = os.path.basename(file)
    if os.path.getsize(file) == 0:
        print(f"Thread {thread_index}: File {file}is an empty image file")
        return
    try:
        im = Image.open(file).convert("L")
        if not native and im.size != (2560, 2560):
            im = im.resize((2560, 2560), Image.BILINEAR)
        output.add_output(np.ascontiguousarray(np.asarray(im, dtype=np.uint8)), fn, img_index)
    except Exception as e:                       # reference: print and carry on (reader.py:81-82)
        print("...exception occurred:", e)


def run_reader(files, output, n_threads, resize, native=False):
    """reader.py:86-110: ``files`` is a list of (path, index).  (The reference ignores ``resize`` and always resizes.)"""
    with ThreadPoolExecutor(max_workers=max(1, n_threads)) as ex:
        list(ex.map(lambda f: read_img(output, f, native=native), files))
    print("all files in batch added to reader pool")
