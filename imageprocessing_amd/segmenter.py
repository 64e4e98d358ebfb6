"""Stage driver with the reference's signature (/root/reference/segmenter.py:108-265): same directories,
file filter / sort, radius, settings.csv and 200-frame batches; the two hot stages run on the GPU."""
import csv
import os
import time
from queue import Queue
from threading import Thread

import numpy as np

from .bg_correction import run_bg_correction
from .detection import DetectionSettings, run_detection
from .reader import ReaderOutput, run_reader
from .sharding import batches_of


def _disk(shape, radius):
    """Host copy of cv.circle(mask, (1280,1280), radius, 255, -1) (midpoint circle, horizontal spans)."""
    H, W = shape
    mask = np.zeros(shape, np.uint8)
    cx, cy = 1280, 1280
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    def span(y, xa, xb):
        if 0 <= y < H:
            mask[y, max(xa, 0):min(xb, W - 1) + 1] = 255
    while dx >= dy:
        span(cy - dy, cx - dx, cx + dx); span(cy + dy, cx - dx, cx + dx)
        span(cy - dx, cx - dy, cx + dy); span(cy + dx, cx - dy, cx + dy)
        dy += 1; err += plus; plus += 2
        if err > 0:
            err -= minus; dx -= 1; minus -= 2
    return mask


def compute_radius(files):
    """segmenter.py:28-63 (host side, once per folder)."""
    from PIL import Image
    imgs = []
    for file in files[:10]:
        im = Image.open(file).convert("L")
        if im.size != (2560, 2560):
            im = im.resize((2560, 2560), Image.BILINEAR)
        imgs.append(np.asarray(im, dtype=np.uint8))
    bg = np.max(imgs, axis=0)
    bg = np.where((255 - bg) > 180, 255, 0).astype(np.uint8)      # threshold(bitwise_not(bg), 180, 255, BINARY)
    bg = 255 - bg
    bg[bg == 0] = 10
    count_bg_px, radius = 0, 1000
    while count_bg_px < 10000:
        radius += 50
        masked = bg & _disk(bg.shape, radius)
        count_bg_px = int((masked == 10).sum())
    return radius - 50


def save_detection_settings_to_csv(settings, file_path, src_path):
    """segmenter.py:65-84."""
    with open(file_path, mode="w", newline="") as file:
        writer = csv.writer(file)
        writer.writerow(["Field Name", "Value"])
        writer.writerow(["data source", src_path])
        for field_name in settings.__annotations__:
            writer.writerow([field_name, getattr(settings, field_name)])


def run_segmenter(src_path: str, save_path: str, deconvolution: bool, model=None):
    """segmenter.py:108-265.  With ``deconvolution=True`` the network between the two stages (LUCYD in the reference, a
    module-level global of deconvolution.py) is passed as ``model``: any torch callable float32 [b,1,H,W] -> same shape
    (or a tuple starting with it); the network itself is outside this library."""
    if deconvolution and model is None:
        raise ValueError("deconvolution=True needs model= (the reference's LUCYD module and weights live in its repository)")
    os.makedirs(save_path, exist_ok=True)
    paths = {k: os.path.join(save_path, k) for k in ("Crops", "Deconv_crops", "Masks", "Data", "Images")}
    for p in paths.values():
        os.makedirs(p, exist_ok=True)
    segmented = {os.path.splitext(name)[0] for name in os.listdir(paths["Data"])}
    files = []
    print("checking files...")
    for name in os.listdir(src_path):
        file = os.path.join(src_path, name)
        if os.path.splitext(name)[0] in segmented:
            continue                                               # resume: already segmented (:156-161)
        if (file.endswith(".png") or file.endswith(".jpg")) and os.path.getsize(file) > 0:
            files.append(file)
    print(f"{len(segmented)} of {len(files)} images already segmented")
    try:
        files.sort(key=lambda x: float(os.path.basename(x).split("_")[0].split("-")[1]))      # :168-172
    except Exception:
        print("Key-based sort failed")
        files.sort()
    print("start segmentation...")
    radius = compute_radius(files)
    print("Radius: ", radius)
    settings = DetectionSettings(paths["Data"], paths["Crops"], paths["Deconv_crops"], paths["Masks"], paths["Images"],
                                 400, 10, 1, False, True, True, False, True, True, radius)        # :185-201
    save_detection_settings_to_csv(settings, os.path.join(save_path, "settings.csv"), src_path)
    batches = [files[a:a + m] for a, m in batches_of(len(files))]                               # :209-213
    for batch_n, batch in enumerate(batches, 1):
        print(f"Batch {batch_n} of {len(batches)}")
        start = time.perf_counter()
        reader_output = ReaderOutput(len(batch))
        bg_output_queue = Queue()
        Thread(target=run_reader, args=([(f, i) for i, f in enumerate(batch)], reader_output, 8, settings.resize)).start()
        run_bg_correction(reader_output, bg_output_queue, 6, None)                              # :239-241
        if deconvolution:                                                                       # :243-249
            from .deconv_bridge import run_deconvolution
            deconv_output_queue = Queue()
            Thread(target=run_deconvolution, args=(bg_output_queue, deconv_output_queue, len(batch), 1, model)).start()
        else:
            deconv_output_queue = bg_output_queue
        run_detection(deconv_output_queue, settings, 8, len(batch), None)                       # :254
        print(f"Time per image: {(time.perf_counter() - start) / len(batch)}")                  # :256-258
    print("finished segmentation...")
