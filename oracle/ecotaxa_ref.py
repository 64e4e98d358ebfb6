"""TEST INFRASTRUCTURE ONLY -- CPU restatement of gen_crop_df (/root/reference/EcoTaxa_preparation.py:20-129): the crop
table read back from the per-frame CSV files.  Parity unpinned against a run of the reference itself: the reference module
imports pyecotaxa at load time, which this image does not have; the function is restated statement by statement on
pandas, which both sides share."""
import os

import numpy as np
import pandas as pd
from pandas.errors import EmptyDataError

HEADERS = ["img_id", "index", "filename", "mean_raw", "std_raw", "mean", "std", "area", "x", "y", "w", "h", "saved",
           "object_bound_box_w", "object_bound_box_h", "bound_box_x", "bound_box_y", "object_circularity", "object_area_exc",
           "object_area_rprops", "object_%area", "object_major_axis_len", "object_minor_axis_len", "object_centroid_y",
           "object_centroid_x", "object_convex_area", "object_min_intensity", "object_max_intensity", "object_mean_intensity",
           "object_int_density", "object_perimeter", "object_elongation", "object_range", "object_perim_area_excl",
           "object_perim_major", "object_circularity_area_excl", "object_angle", "object_boundbox_area", "object_eccentricity",
           "object_equivalent_diameter", "object_euler_nr", "object_extent", "object_local_centroid_col",
           "object_local_centroid_row", "object_solidity"]


def gen_crop_df(path, small, size_filter=0):
    directory = os.path.join(os.path.dirname(path), "Data")                   # :42-43
    frames, img_id, n_empty = [], 1, 0
    for name in sorted(os.listdir(path)):                                     # :45-62
        if not name.endswith(".csv"):
            continue
        try:
            one = pd.read_csv(os.path.join(path, name), delimiter=",", header=None, index_col=None)
        except EmptyDataError:
            n_empty += 1
            continue
        if len(one.columns) != 44:
            continue
        one.insert(0, "", img_id)
        frames.append(one)
        img_id += 1
    df = pd.concat(frames, ignore_index=True)                                 # :64
    df.columns = HEADERS
    df.reset_index(drop=True, inplace=True)
    df.drop("index", axis=1, inplace=True)
    if not small:
        df = df[df["saved"] == 1]
    split_df = df["filename"].str.split("_", expand=True)                     # :84-106
    if small:
        split_df.columns = ["date", "time", "pressure", "temperature", "index", "mask_ext"]
        split_df.drop("mask_ext", axis=1, inplace=True)
    else:
        if split_df.shape[1] == 5:
            split_df.columns = ["date-time", "pressure", "temperature", "index", "drop"]
            split_df.drop("drop", axis=1, inplace=True)
        else:
            split_df.columns = ["date-time", "pressure", "temperature", "index"]
    split_df[["date", "time"]] = split_df["date-time"].str.split("-", expand=True)
    split_df.drop(columns=["date-time"], inplace=True)
    split_df["pressure"] = split_df["pressure"].str.replace("bar", "", regex=False).astype(float)
    split_df["temperature"] = split_df["temperature"].str.replace("C", "", regex=False).astype(float)
    split_df["index"] = split_df["index"].str.replace(".png", "", regex=False).astype(int)
    df = pd.concat([split_df, df], axis=1)                                    # :109
    df["full_path"] = df.apply(lambda r: os.path.join(directory, r["filename"]), axis=1)
    df["esd"] = df["area"].apply(lambda a: 2 * np.sqrt(a * (13.5 * 2) ** 2 / np.pi)).round(2)
    df["pressure"] = (df["pressure"] - 1) * 10
    df.rename(columns={"pressure": "pressure [dbar]"}, inplace=True)
    df = df.sort_values(by=["date", "time", "index"], ascending=True)
    df.reset_index(drop=True, inplace=True)
    df = df[(df["w"] > size_filter) | (df["h"] > size_filter)]
    return df, n_empty
