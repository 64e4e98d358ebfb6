"""Stage driver with the reference's signature (/root/reference/segmenter.py:108-265): same directories,
file filter / sort, radius, settings.csv and 200-frame batches; the two hot stages run on the GPU."""
import csv
import os
import time
from queue import Queue
from threading import Thread

import numpy as np

from .bg_correction import _segmenter, run_bg_correction
from .detection import DetectionSettings, run_detection
from .reader import ReaderOutput, decode_gray, run_reader
from .sharding import batches_of

FRAME = (2560, 2560)                 # reader.py:79 / detection.py:315: every frame is brought to this size


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


def image_kind(file):
    """imghdr.what(file) of segmenter.py:162 for the formats a '.png' / '.jpg' file can plausibly hold (imghdr's own header
    tests, restated: the module is gone from Python 3.13).  None = not an image: the file is left out of the batch."""
    try:
        with open(file, "rb") as f:
            h = f.read(32)
    except OSError:
        return None
    if h.startswith(b"\x89PNG\r\n\x1a\n"):
        return "png"
    if h[6:10] in (b"JFIF", b"Exif") or h[:4] == b"\xff\xd8\xff\xdb":
        return "jpeg"
    if h[:6] in (b"GIF87a", b"GIF89a"):
        return "gif"
    if h[:2] in (b"MM", b"II"):
        return "tiff"
    if h.startswith(b"BM"):
        return "bmp"
    if h.startswith(b"RIFF") and h[8:12] == b"WEBP":
        return "webp"
    if len(h) >= 3 and h[0] == ord(b"P") and h[1] in b"123456" and h[2] in b" \t\n\r":
        return {1: "pbm", 4: "pbm", 2: "pgm", 5: "pgm", 3: "ppm", 6: "ppm"}[h[1] - ord(b"0")]
    return None


def read_frames_resized(files, device=0):
    """[cv.resize(cv.imread(f, GRAYSCALE), (2560, 2560)) for f in files] (segmenter.py:45 / reader.py:75-79) as a uint8
    NumPy stack: PIL decodes, frames of another size are resized on the device with OpenCV's INTER_LINEAR arithmetic."""
    import torch
    imgs = [decode_gray(f) for f in files]
    out = np.empty((len(imgs),) + FRAME, np.uint8)
    seg = None
    for i, im in enumerate(imgs):
        if im.shape == FRAME:
            out[i] = im
        else:
            seg = seg or _segmenter(FRAME[0], FRAME[1], 1, device, role="bg")
            with seg.lock:
                out[i] = seg.resize_frames(torch.from_numpy(im).to(f"cuda:{device}"))[0].cpu().numpy()
    return out


def radius_from_frames(imgs):
    """segmenter.py:47-63 on the (resized) first frames."""
    bg = np.max(imgs, axis=0)                                             # :47
    bg = np.where((255 - bg) > 180, 255, 0).astype(np.uint8)              # :49 threshold(bitwise_not(bg), 180, 255, BINARY)
    bg = 255 - bg                                                         # :50
    bg[bg == 0] = 10                                                      # :51
    count_bg_px, radius = 0, 1000
    while count_bg_px < 10000:                                            # :56-60 (the reference never leaves this loop when the
        radius += 50                                                      #  frames have no dark surround)
        if radius > 1850:                                                 # the disk covers the whole frame: nothing more can come
            raise RuntimeError("compute_radius: fewer than 10000 dark pixels (gray < 75) in the first frames; the reference "
                               "never returns on such a folder (segmenter.py:56-60)")
        masked = bg & _disk(bg.shape, radius)
        count_bg_px = int((masked == 10).sum())
    return radius - 50                                                    # :62


def compute_radius(files):
    """segmenter.py:28-63 (host side, once per folder)."""
    return radius_from_frames(read_frames_resized(files[:10]))


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
    (or a tuple starting with it); the network itself is outside this library.

    The stages of a batch are the reference's (reader thread -> run_bg_correction -> [run_deconvolution] -> run_detection);
    frames stay on the device between them, and the reader + upload + background correction of the next batch are started
    before this batch's detection blocks, so that decoding and PCIe transfers of one batch hide under the file writing of
    the previous one."""
    import torch
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
        if (file.endswith(".png") or file.endswith(".jpg")) and image_kind(file) is not None:       # :162
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
    staging = {}

    def start(k):
        """reader thread + background stage of batch k (both return at once)"""
        batch = batches[k]
        slot = k % 2
        if slot not in staging or staging[slot].shape[0] < len(batch):
            staging[slot] = torch.empty((len(batch),) + FRAME, dtype=torch.uint8, pin_memory=True)
        reader_output = ReaderOutput(len(batch), staging=staging[slot].numpy())
        q = Queue()
        Thread(target=run_reader, args=([(f, i) for i, f in enumerate(batch)], reader_output, 8, settings.resize),
               kwargs=dict(native=True)).start()                                                # :233-237
        run_bg_correction(reader_output, q, 6, None, keep_on_device=True, resize_to=FRAME)      # :239-241
        return q

    nxt = start(0) if batches and batches[0] else None
    for batch_n, batch in enumerate(batches, 1):
        print(f"Batch {batch_n} of {len(batches)}")
        t0 = time.perf_counter()
        bg_output_queue = nxt
        if deconvolution:                                                                       # :243-249
            from .deconv_bridge import run_deconvolution
            deconv_output_queue = Queue()
            Thread(target=run_deconvolution, args=(bg_output_queue, deconv_output_queue, len(batch), 1, model)).start()
        else:
            deconv_output_queue = bg_output_queue
        nxt = start(batch_n) if batch_n < len(batches) else None      # the next batch decodes / uploads meanwhile
        run_detection(deconv_output_queue, settings, 8, len(batch), None)                       # :254
        print(f"Time per image: {(time.perf_counter() - t0) / len(batch)}")                     # :256-258
    print("finished segmentation...")
