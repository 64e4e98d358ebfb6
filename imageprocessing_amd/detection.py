"""Detection stage with the reference's call signatures (/root/reference/detection.py), computed by
libpiscoseg.so on the GPU.  PNG / CSV writing stays on the host (I/O, detection.py:251-253,293)."""
import csv
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass

import numpy as np

from .bg_correction import _segmenter
from .engine import PROP_KEYS

# columns of regionprop2zooprocess that the reference emits as integers (ints / numpy ints / uint8)
_INT_PROPS = {"width", "height", "bx", "by", "area_exc", "area_rprops", "convex_area", "min_intensity", "max_intensity",
              "range", "bounding_box_area", "euler_number"}


@dataclass
class DetectionSettings:
    """detection.py:22-58 (same fields, same order)."""
    data_path: str
    raw_crop_path: str
    deconv_crop_path: str
    mask_path: str
    img_path: str
    min_area_to_save: float
    min_area_to_segment: float
    n_sigma: float
    save_bb_image: bool
    save_crops: bool
    equalize_hist: bool
    resize: bool
    clear_save_path: bool
    mask_img: bool
    mask_radius: int


def calculate_regionprops(mask, img, device=0):
    """detection.py:75-159: the 32 zooprocess-style properties of ``measure.regionprops(mask, intensity_image=img)[0]`` as
    the dict the reference builds, computed on the GPU (``pseg_regionprops``).  mask / img: 2-D uint8 arrays.

    ``regionprops(...)[0]`` is the region of the SMALLEST nonzero value of ``mask`` (detect_on_img only ever passes
    0 / 255).  The library works on the region's bounding box (``pseg_regionprops`` takes bx = by = 0 and width / height
    from its window, and assumes no background hole), so a general mask is cut to that box here, the box offset is put
    back into bx / by / centroid_x / centroid_y, and a region with holes gets its filled area -- and the four columns
    derived from it -- from ``scipy.ndimage.binary_fill_holes`` exactly as scikit-image computes it (_regionprops.py:352-359).
    An empty mask raises IndexError, as ``regionprops(...)[0]`` does."""
    import torch
    mask = np.asarray(mask)
    img = np.asarray(img)
    if mask.ndim != 2 or img.shape != mask.shape:
        raise ValueError("mask and img must be 2-D arrays of the same shape")
    nz = mask[mask > 0]
    if nz.size == 0:
        raise IndexError("list index out of range")       # what props[0] raises in the reference
    region = mask == nz.min()
    rows, cols = np.nonzero(region.any(axis=1))[0], np.nonzero(region.any(axis=0))[0]
    y0, y1, x0, x1 = int(rows[0]), int(rows[-1]) + 1, int(cols[0]), int(cols[-1]) + 1
    win = np.ascontiguousarray(region[y0:y1, x0:x1]).astype(np.uint8) * 255
    gray = np.ascontiguousarray(img[y0:y1, x0:x1], dtype=np.uint8)
    h, w = win.shape
    seg = _segmenter(64, 64, 1, device, role="props")              # (the window's size travels with the call, not the context)
    with seg.lock:
        vals = seg.regionprops(torch.from_numpy(win).to(f"cuda:{device}"), torch.from_numpy(gray).to(f"cuda:{device}"))
    out = {k: (int(v) if k in _INT_PROPS else float(v)) for k, v in zip(PROP_KEYS, vals)}
    out["bx"], out["by"] = x0, y0                                   # prop.bbox[1], prop.bbox[0]
    out["centroid_y"] += y0                                          # prop.centroid is in image coordinates,
    out["centroid_x"] += x0                                          # prop.local_centroid in the box's
    from scipy import ndimage as ndi
    filled = int(ndi.binary_fill_holes(win > 0, structure=np.ones((3, 3))).sum())
    if filled != out["area_exc"]:                                   # holes: filled_area and the columns computed from it
        perim = out["perim."]
        out["area_rprops"] = filled
        out["circ."] = (4 * np.pi * filled) / perim ** 2 if perim else float("inf")
        out["%area"] = 1 - out["area_exc"] / filled
        out["intden"] = filled * out["mean_intensity"]
    return out


def save_crop_data(path, data):
    """detection.py:61-73."""
    with open(path, "w", newline="") as f:
        csv.writer(f, delimiter=",").writerows(data)


def disk_mask(settings, shape=(2560, 2560), device=0):
    """The mask run_detection builds once per batch (detection.py:315-326); a device tensor."""
    import torch
    seg = _segmenter(shape[0], shape[1], 1, device, role="detect")
    if settings.mask_img:
        with seg.lock:
            return seg.disk_mask(settings.mask_radius)
    return torch.zeros(shape, dtype=torch.uint8, device=f"cuda:{device}")


def rows_of_frame(tab, f, fn, mean_raw):
    """The crop_data list of detect_on_img for frame f (detection.py:265,271): 44 entries per kept object with the
    reference's value types (ints for saved boxes, centre floats for unsaved ones, nan padding)."""
    rows = []
    a, b = int(tab.frame_row_offset[f]), int(tab.frame_row_offset[f + 1])
    s_of_row = getattr(tab, "_s_of_row", None)
    if s_of_row is None:
        s_of_row = {int(r): s for s, r in enumerate(tab.saved_row)}
        tab._s_of_row = s_of_row
    for i in range(a, b):
        c = int(tab.c[i])
        name = f"{fn[:-4]}_{c}.png"
        x, y, w, h = int(tab.x[i]), int(tab.y[i]), int(tab.w[i]), int(tab.h[i])
        head = [c, name, mean_raw[0], mean_raw[1], float(tab.frame_mean[f]), float(tab.frame_std[f]), float(tab.area[i])]
        if tab.saved[i]:
            props = [int(v) if k in _INT_PROPS else float(v) for k, v in zip(PROP_KEYS, tab.saved_props[s_of_row[i]])]
            rows.append(head + [x, y, w, h, 1] + props)
        else:
            rows.append(head + [x + w / 2, y + h / 2, float(w), float(h), 0] + [np.nan] * 32)
    return rows


def _write_frame(tab, f, fn, mean_raw, settings, saved_of_row):
    from PIL import Image
    if not tab.frame_valid[f]:                                           # detection.py:294-297
        save_crop_data(os.path.join(settings.data_path, fn[:-4] + ".csv"), ["corrupt image!"])
        return
    rows = rows_of_frame(tab, f, fn, mean_raw)
    if settings.save_crops:
        a = int(tab.frame_row_offset[f])
        for k, row in enumerate(rows):
            if row[11] != 1:
                continue
            s = saved_of_row[a + k]
            Image.fromarray(tab.crop(s, "masked")).save(os.path.join(settings.deconv_crop_path, row[1]))   # :251
            Image.fromarray(tab.crop(s, "raw")).save(os.path.join(settings.raw_crop_path, row[1]))         # :252
            Image.fromarray(tab.crop(s, "mask")).save(os.path.join(settings.mask_path, row[1]))            # :253
    save_crop_data(os.path.join(settings.data_path, fn[:-4] + ".csv"), rows)                               # :293


def check_settings(settings):
    """Settings the reference acts on and this path does not are refused, never ignored.  ``resize`` (detection.py:209-218:
    areas x 4, boxes x 2 for frames that were halved upstream, then crops cut from the unscaled frame at the doubled
    coordinates) is off in every settings literal of the reference (segmenter.py:185-201) and has no device form here."""
    if settings.resize:
        raise NotImplementedError("DetectionSettings.resize=True (detection.py:209-218) is not supported by the GPU path; "
                                  "the reference's own callers pass False (segmenter.py:196)")


def _write_overview_images(cleaned, good, tab, settings, device):
    """detection.py:275-277 (save_bb_image): the cleaned frame as a 3-channel image, cv.resize'd to 1280 x 1280, written to
    img_path under the frame's own file name (the contour drawing above it is commented out in the reference, :230-231).
    The resize runs on the device (pseg_resize_frames = OpenCV's INTER_LINEAR / exact 2 x 2 averaging restated)."""
    from PIL import Image
    n = cleaned.shape[0]
    half = _segmenter(1280, 1280, n, device, role="overview")
    with half.lock:
        small = half.resize_frames(cleaned.contiguous()).cpu().numpy()
    for f in range(n):
        if tab.frame_valid[f]:                               # (corrupt frames take the else branch: no image)
            Image.fromarray(np.repeat(small[f][:, :, None], 3, axis=2)).save(os.path.join(settings.img_path, good[f][3]))


def detect_batch(items, settings, mask, n_writers=8, device=0):
    """All frames of a batch in one library call + host-side file writing.  ``items`` are the tuples of the
    background stage; ``mask`` is the uint8 disk mask (numpy array or device tensor)."""
    import torch
    dev = f"cuda:{device}"
    check_settings(settings)
    good = [it for it in items if not isinstance(it[1], list)]
    for it in items:
        if isinstance(it[1], list):                                      # corrupt frame (detection.py:294-297)
            save_crop_data(os.path.join(settings.data_path, it[3][:-4] + ".csv"), ["corrupt image!"])
    if not good:
        return None
    def dev_stack(k):
        first = good[0][k]
        if torch.is_tensor(first):
            # rows of one contiguous batch tensor, in order (what run_bg_correction(keep_on_device=True) hands over): a view
            base, step = first._base, first.numel()
            if base is not None and base.is_contiguous() and all(
                    torch.is_tensor(it[k]) and it[k]._base is base and it[k].is_contiguous()
                    and it[k].storage_offset() == first.storage_offset() + j * step for j, it in enumerate(good)):
                j0 = (first.storage_offset() - base.storage_offset()) // step
                return base.view(-1, *first.shape)[j0:j0 + len(good)]
            return torch.stack([it[k] for it in good]).contiguous()
        return torch.from_numpy(np.stack([it[k] for it in good])).to(dev)
    cleaned = dev_stack(1)
    raw = cleaned if all(it[0] is it[1] for it in good) else dev_stack(0)
    H, W = cleaned.shape[-2:]
    seg = _segmenter(H, W, len(good), device, role="detect")
    m = None
    if settings.mask_img and mask is not None:
        m = mask if torch.is_tensor(mask) else torch.from_numpy(np.ascontiguousarray(mask)).to(dev)
    valid = np.array([1 if it[2][1] > 2 else 0 for it in good], np.uint8)          # detection.py:179
    with seg.lock:                                                       # the crops sit in the context's page-locked buffer until
        tab = seg.detect(cleaned, raw=raw, mask=m, valid=valid, threshold=10,    # the files are written
                         min_area_to_segment=settings.min_area_to_segment, min_area_to_save=settings.min_area_to_save,
                         fetch_crops="pinned" if settings.save_crops else False)
        saved_of_row = {int(r): s for s, r in enumerate(tab.saved_row)}
        if not settings.save_crops and len(tab.saved_row):
            # detection.py:237-258: with save_crops False the crops and c_mask are never made, and the first object above
            # min_area_to_save reaches calculate_regionprops(c_mask, ...) with c_mask undefined -- the reference dies there
            raise NameError("name 'c_mask' is not defined (detection.py:257): save_crops=False with an object above "
                            "min_area_to_save; no file of this batch was written")
        if settings.save_bb_image:
            _write_overview_images(cleaned, good, tab, settings, device)
        with ThreadPoolExecutor(max_workers=max(1, n_writers)) as ex:
            list(ex.map(lambda f: _write_frame(tab, f, good[f][3], good[f][2], settings, saved_of_row), range(len(good))))
        if tab.crops:
            tab.crops = {k: v.copy() for k, v in tab.crops.items()} if tab.crop_bytes < (64 << 20) else {}
    return tab


def detect_on_img(input, settings, mask, index=0):
    """detection.py:162-297 for one frame."""
    detect_batch([input], settings, mask, n_writers=1)


def run_detection(input, settings, n_cores, n_imgs, running):
    """detection.py:300-344: takes n_imgs items from the queue and blocks until every file is written."""
    mask = disk_mask(settings) if settings.mask_img else None
    items = []
    for i in range(n_imgs):
        task = input.get()
        items.append(task)
        print(i, task[-1])
    H = next((it[1].shape[0] for it in items if not isinstance(it[1], list)), 2560)
    if mask is not None and mask.shape[0] != H:
        mask = disk_mask(settings, shape=(H, H))
    detect_batch(items, settings, mask, n_writers=n_cores)
    print("detection finished")
