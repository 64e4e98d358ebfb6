"""Oracle for the two hot-path stages -- TEST INFRASTRUCTURE ONLY (oracle/__init__.py).

``correct_img`` restates /root/reference/bg_correction.py:41-90 and ``detect_on_img`` restates
/root/reference/detection.py:162-297, statement for statement, with the cv2 calls routed to
oracle/cv_ref.py and regionprops to oracle/regionprops_ref.py.  Instead of pushing to a queue /
writing PNG+CSV they return the same payloads (tuple for the queue; rows + crops for the files).
"""
import os

import numpy as np

from . import cv_ref as cv
from .regionprops_ref import calculate_regionprops


def correct_img(img_index, images, n_bg_imgs):
    """bg_correction.py:65-90.  ``images`` is the list behind ReaderOutput.images: (img, fn)
    tuples.  Returns the tuple the reference puts on the output queue."""
    n_bg_imgs *= 2                                                            # :65
    img, fn = images[img_index]                                               # :68
    mean = np.mean(img)                                                       # :69
    stdev = np.std(img)                                                       # :70
    if stdev > 2:                                                             # :71
        start = (img_index if img_index + n_bg_imgs < len(images) else len(images) - n_bg_imgs)  # :72
        bg_imgs = []
        for i in range(1, n_bg_imgs):                                         # :74-77
            bg_imgs.append(np.min([images[start + i - 1][0], images[start + i][0]], axis=0))
        bg = np.max(bg_imgs, axis=0)                                          # :78
        cleaned_img = cv.bitwise_not(cv.absdiff(img, bg))                     # :80-81
        return (cleaned_img, cleaned_img, [mean, stdev], fn)                  # :86
    return ([], [], [mean, stdev], fn)                                        # :90


def disk_mask(radius, shape=(2560, 2560)):
    """detection.py:315-326."""
    mask = np.zeros(shape, dtype=np.uint8)
    cv.circle(mask, (int(mask.shape[0] / 2), int(mask.shape[1] / 2)), radius, 255, -1)
    return mask


def detect_on_img(input, mask, min_area_to_segment=10, min_area_to_save=400, mask_img=True,
                  want_labels=False):
    """detection.py:176-297 with save_crops=True, resize=False, save_bb_image=False.

    Returns dict(corrupt, rows, crops, labels):
      rows  -- list of 44-entry lists exactly as handed to csv.writer (detection.py:265,271)
      crops -- list of (c, crop_masked, raw_crop, c_mask) for saved objects (detection.py:251-253)
      labels-- int32 label map in contour order, if want_labels
    """
    corrected, cleaned, mean_raw, fn = input                                  # :176
    raw_bg_corr = corrected
    if not mean_raw[1] > 2:                                                   # :179 / :294-297
        return dict(corrupt=True, rows=["corrupt image!"], crops=[], labels=None)
    corrected = cv.bitwise_not(cleaned)                                       # :178
    if mask_img:                                                              # :185-188
        mean = np.mean(corrected[np.where(mask == 255)])
        std = np.std(corrected[np.where(mask == 255)])
        corrected[np.where(mask < 255)] = 0
    else:
        mean = np.mean(corrected)
        std = np.std(corrected)
    thresh = cv.threshold(corrected, 10, 255, cv.THRESH_BINARY)[1].astype(np.uint8)   # :195-202
    cnts, _ = cv.findContours(thresh, cv.RETR_EXTERNAL, cv.CHAIN_APPROX_SIMPLE)       # :204
    labels = cv.label_map(thresh, cnts) if want_labels else None
    mask = np.zeros_like(thresh, dtype=np.uint8)                              # :206
    areas = np.array([cv.contourArea(cnt) for cnt in cnts])                   # :208
    bounding_boxes = [cv.boundingRect(cnt) for cnt in cnts]                   # :212
    bounding_boxes = np.array([np.array((x + w / 2, y + h / 2, w, h)) for (x, y, w, h) in bounding_boxes])
    crop_data, crops = [], []
    c = 0
    for i, cnt in enumerate(cnts):                                            # :222
        if areas[i] < min_area_to_segment:                                    # :223
            continue
        name = f"{fn[:-4]}_{c}.png"                                           # :226-228
        x, y, w, h = bounding_boxes[i]                                        # :233
        if areas[i] > min_area_to_save:                                       # :235
            x = int(x - w / 2)                                                # :238-241
            y = int(y - h / 2)
            h = int(h)
            w = int(w)
            crop = cleaned[y:y + h, x:x + w]                                  # :243
            raw_crop = raw_bg_corr[y:y + h, x:x + w]                          # :244
            cv.drawContours(mask, [cnt], -1, (255), thickness=cv.FILLED)      # :245
            c_mask = mask[y:y + h, x:x + w]                                   # :246
            crop_masked = np.where(c_mask == 255, crop, 255)                  # :247
            crops.append((c, crop_masked.astype(np.uint8).copy(), raw_crop.copy(), c_mask.copy()))
            props = list(calculate_regionprops(c_mask, crop_masked).values())  # :258
            crop_data.append([c, os.path.basename(name), mean_raw[0], mean_raw[1], mean, std,
                              areas[i], x, y, w, h, 1, *props])                # :265
        else:
            crop_data.append([c, os.path.basename(name), mean_raw[0], mean_raw[1], mean, std,
                              areas[i], x, y, w, h, 0, *([np.nan] * 32)])      # :270-271
        c += 1                                                                # :273
    return dict(corrupt=False, rows=crop_data, crops=crops, labels=labels)


def batches_of(files, batch_size=200):
    """segmenter.py:209-213."""
    batches = []
    for i in range(len(files) // batch_size - 1):
        batches.append(files[i * batch_size:(i + 1) * batch_size])
    batches.append(files[batch_size * (len(files) // batch_size - 1):])
    return batches


def compute_radius(imgs):
    """segmenter.py:47-63 on the frames of :45 (``imgs``: the first 10 frames, already read and resized): the mask
    accumulates the growing filled circles exactly as the reference draws them into one array."""
    bg = np.max(imgs, axis=0)                                                 # :47
    bg = cv.threshold(cv.bitwise_not(bg), 180, 255, cv.THRESH_BINARY)[1].astype(np.uint8)   # :49
    bg = cv.bitwise_not(bg)                                                   # :50
    bg[np.where(bg == 0)] = 10                                                # :51
    mask = np.zeros(bg.shape, dtype=np.uint8)                                 # :53
    count_bg_px = 0
    radius = 1000
    while count_bg_px < 10000:                                                # :56
        radius += 50
        cv.circle(mask, (1280, 1280), radius, (255), -1)                      # :58
        masked = np.bitwise_and(bg, mask)                                     # :59
        count_bg_px = len(np.where(masked == 10)[0])                          # :60
        if radius > 4000:
            raise RuntimeError("compute_radius does not terminate on these frames (SURVEY.md F14)")
    radius -= 50                                                              # :62
    return radius
