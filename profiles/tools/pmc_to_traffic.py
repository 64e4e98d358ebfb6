#!/usr/bin/env python3
"""Turn the counter_collection.csv files of two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into
profiles/traffic.json: HBM bytes per launch of each kernel group, corrected as MI355X_MICROARCH.md (HBM) says:
the CSV values are in kilobytes; gfx950 tallies 128-B read requests at 64 B, so reads are doubled; WRITE_SIZE is
taken as is.  Groups are the ones bench.py times with HIP events (k_reduce_partials belongs to the background group since round 2's closing code).
usage: pmc_to_traffic.py <fetch.csv> <write.csv> <out.json>"""
import csv
import json
import sys
from collections import defaultdict

# kernel name prefix -> group (template arguments vary: k_bg_stream<false>, k_runs_band<10>, ...); everything else that starts
# with k_ belongs to the sparse labelling chain
PREFIX = [("k_bg_stream", "bg_correct"), ("k_bg_generic", "bg_correct"), ("k_reduce_partials", "bg_correct"), ("k_label_fill", "label_map"),
          ("k_crops_sums", "crops_props"), ("k_convex", "crops_props"), ("k_band_map", "crops_props"), ("k_pack_out", "crops_props"),
          ("k_props", "crops_props"),
          # round 3: the table kernels (boxes, areas, ranks, rows, size scans) left the labelling chain (k_rank_ext feeds the label fill)
          ("k_rank_ext", "sparse_labelling"), ("k_bbox", "table_beside_label_map"), ("k_area", "table_beside_label_map"),
          ("k_rank", "table_beside_label_map"), ("k_scan_small", "table_beside_label_map"), ("k_emit", "table_beside_label_map"),
          ("k_scan_sizes", "table_beside_label_map")]
SKIP = ("k_synth", "k_fill_spans", "__amd", "k_mask_count")


def group_of(kernel):
    for pre, g in PREFIX:
        if kernel.startswith(pre):
            return g
    return "sparse_labelling" if kernel.startswith("k_") else None


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            tot[k] += float(r["Counter_Value"])
            cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    groups = defaultdict(float)
    for k in sorted(set(fetch) | set(write)):
        rd = fetch.get(k, 0.0) * 1024 * 2          # KB -> B, gfx950 read correction x2
        wr = write.get(k, 0.0) * 1024
        kernels[k] = {"read_bytes": rd, "write_bytes": wr, "launches_sampled": nf.get(k, 0)}
        g = group_of(k)
        if g is None or k.startswith(SKIP):
            continue
        groups[g] += rd + wr
    out = dict(groups)
    out["_per_kernel"] = kernels
    out["_note"] = ("bytes per launch (one 200-frame batch); FETCH_SIZE x 1024 x 2 (gfx950 counts 128-B reads as 64 B), "
                    "WRITE_SIZE x 1024; separate --pmc passes")
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k:24s} read {v['read_bytes']/1e6:10.1f} MB  write {v['write_bytes']/1e6:10.1f} MB")


if __name__ == "__main__":
    main()
