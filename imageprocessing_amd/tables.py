"""Object tables as data frames: the crop table EcoTaxa_preparation.gen_crop_df
(/root/reference/EcoTaxa_preparation.py:20-129) builds by reading every per-frame CSV back, made directly from the
in-memory ObjectTable of a batch (or of merged batches) -- same columns, names, order, dtypes and row order."""
import os

import numpy as np

from .engine import PROP_KEYS

# EcoTaxa_preparation.py:65-72: names given to CSV columns 13..44 (regionprop2zooprocess order, detection.py:84-152)
OBJECT_COLUMNS = [
    "object_bound_box_w", "object_bound_box_h", "bound_box_x", "bound_box_y", "object_circularity", "object_area_exc",
    "object_area_rprops", "object_%area", "object_major_axis_len", "object_minor_axis_len", "object_centroid_y",
    "object_centroid_x", "object_convex_area", "object_min_intensity", "object_max_intensity", "object_mean_intensity",
    "object_int_density", "object_perimeter", "object_elongation", "object_range", "object_perim_area_excl",
    "object_perim_major", "object_circularity_area_excl", "object_angle", "object_boundbox_area", "object_eccentricity",
    "object_equivalent_diameter", "object_euler_nr", "object_extent", "object_local_centroid_col",
    "object_local_centroid_row", "object_solidity",
]
_INT_PROP_IDX = [i for i, k in enumerate(PROP_KEYS) if k in {
    "width", "height", "bx", "by", "area_exc", "area_rprops", "convex_area", "min_intensity", "max_intensity", "range",
    "bounding_box_area", "euler_number"}]
PIXEL_SIZE_UM = 13.5 * 2                                   # EcoTaxa_preparation.py:35, at 2560 x 2560


def area_to_esd(area):
    """EcoTaxa_preparation.py:34-36."""
    return 2 * np.sqrt(area * PIXEL_SIZE_UM ** 2 / np.pi)


def crop_dataframe(tab, filenames, mean_raw, data_dir, small=False, size_filter=0):
    """The data frame ``gen_crop_df(data_dir, small, size_filter)`` returns for the CSV files this table would be written to.

    tab: ObjectTable (engine) or any object / namespace with its arrays; filenames[f]: the frame's file name (the CSV is
    ``filenames[f][:-4] + '.csv'``, detection.py:293); mean_raw[f] = (mean, std) of the raw frame (CSV columns 3, 4);
    data_dir: the Data/ directory the CSVs live in (only used for the ``full_path`` column, :42-43,112).
    Frames are numbered (``img_id``) in the sorted order of their CSV names, skipping corrupt frames (their CSV has 14
    columns, :56-58) and frames without objects (empty CSV, :59-62), like the reference's loop (:45-62)."""
    import pandas as pd
    nf = int(tab.n_frames)
    off = np.asarray(tab.frame_row_offset, dtype=np.int64)
    stems = [str(fn)[:-4] for fn in filenames]
    if len(stems) != nf:
        raise ValueError("one file name per frame")
    order = sorted(range(nf), key=lambda f: stems[f] + ".csv")
    kept = [f for f in order if tab.frame_valid[f] and off[f + 1] > off[f]]
    if not kept:
        raise ValueError("No objects to concatenate")        # pd.concat([]) in the reference (:64)
    rows = np.concatenate([np.arange(off[f], off[f + 1]) for f in kept])
    frame_of = np.concatenate([np.full(off[f + 1] - off[f], f) for f in kept])
    img_id = np.concatenate([np.full(off[f + 1] - off[f], k + 1, dtype=np.int64) for k, f in enumerate(kept)])
    saved = np.asarray(tab.saved)[rows].astype(np.int64)
    x, y, w, h = (np.asarray(getattr(tab, k))[rows].astype(np.float64) for k in "xywh")
    uns = saved == 0
    x = np.where(uns, x + w / 2, x)                          # unsaved rows hold the box centre (detection.py:271)
    y = np.where(uns, y + h / 2, y)
    props = np.full((len(rows), len(PROP_KEYS)), np.nan)
    s_of_row = np.full(int(off[-1]) + 1, -1, dtype=np.int64)
    s_of_row[np.asarray(tab.saved_row, dtype=np.int64)] = np.arange(len(tab.saved_row))
    props[~uns] = np.asarray(tab.saved_props)[s_of_row[rows[~uns]]]
    # a CSV column parses as int64 only if every value of every file is an integer literal: any unsaved row (floats, nan)
    # turns the box and the integer properties into float64 for the whole table
    as_int = not uns.any()
    mr = np.asarray([(float(m[0]), float(m[1])) for m in mean_raw], dtype=np.float64)
    cvals = np.asarray(tab.c)[rows]
    cols = {
        "img_id": img_id,
        "filename": [f"{stems[f]}_{int(c)}.png" for f, c in zip(frame_of, cvals)],
        "mean_raw": mr[frame_of, 0], "std_raw": mr[frame_of, 1],
        "mean": np.asarray(tab.frame_mean, dtype=np.float64)[frame_of], "std": np.asarray(tab.frame_std, dtype=np.float64)[frame_of],
        "area": np.asarray(tab.area, dtype=np.float64)[rows],
        "x": x.astype(np.int64) if as_int else x, "y": y.astype(np.int64) if as_int else y,
        "w": w.astype(np.int64) if as_int else w, "h": h.astype(np.int64) if as_int else h,
        "saved": saved,
    }
    for j, name in enumerate(OBJECT_COLUMNS):
        cols[name] = props[:, j].astype(np.int64) if (as_int and j in _INT_PROP_IDX) else props[:, j]
    df = pd.DataFrame(cols)
    if not small:
        df = df[df["saved"] == 1]                            # :77-78
    # file-name fields (:83-106): <date>-<time>_<pressure>bar_<temperature>C_<index>.png
    parts = df["filename"].str.split("_", expand=True)
    if small:
        parts.columns = ["date", "time", "pressure", "temperature", "index", "mask_ext"]      # raises unless 6 fields (:87-90)
        parts = parts.drop(columns="mask_ext")
    elif parts.shape[1] == 5:
        parts.columns = ["date-time", "pressure", "temperature", "index", "drop"]
        parts = parts.drop(columns="drop")
    else:
        parts.columns = ["date-time", "pressure", "temperature", "index"]
    parts[["date", "time"]] = parts["date-time"].str.split("-", expand=True)
    parts = parts.drop(columns=["date-time"])
    parts["pressure"] = parts["pressure"].str.replace("bar", "", regex=False).astype(float)
    parts["temperature"] = parts["temperature"].str.replace("C", "", regex=False).astype(float)
    parts["index"] = parts["index"].str.replace(".png", "", regex=False).astype(int)
    df = pd.concat([parts, df], axis=1)
    df["full_path"] = [os.path.join(data_dir, fn) for fn in df["filename"]]
    df["esd"] = area_to_esd(df["area"]).round(2)             # :115
    df["pressure"] = (df["pressure"] - 1) * 10               # :116
    df = df.rename(columns={"pressure": "pressure [dbar]"})
    df = df.sort_values(by=["date", "time", "index"], ascending=True).reset_index(drop=True)
    return df[(df["w"] > size_filter) | (df["h"] > size_filter)]                               # :124
