"""Transcribes the object-table rows that the reference's own notebook shows (pipeline.ipynb, cell 1: the DataFrame that
gen_crop_df made from a real cruise profile) into tests/golden/reference_notebook_rows.json -- VALUES ONLY, the visible
columns of the ten printed rows.  They are the only outputs of the hot path that the reference holds.

    python3 tests/golden/make_reference_notebook_rows.py            (needs /root/reference; run in the build container)
"""
import json
import os
import re

SRC = "/root/reference/pipeline.ipynb"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_notebook_rows.json")


def main():
    nb = json.load(open(SRC))
    cell = nb["cells"][1]
    plain = next("".join(o["data"]["text/plain"]) for o in cell["outputs"] if o.get("output_type") == "execute_result")
    # the frame is printed in column blocks; every block starts with a header line and has one line per printed row
    blocks, cur = [], []
    for ln in plain.splitlines():
        if not ln.strip():
            if cur:
                blocks.append(cur)
            cur = []
        elif not ln.startswith("["):
            cur.append(ln.rstrip(" \\"))
    if cur:
        blocks.append(cur)
    rows = {}
    for blk in blocks:
        head = re.split(r"\s{2,}", blk[0].strip())
        for ln in blk[1:]:
            parts = re.split(r"\s{2,}", ln.strip())
            if parts[0] == "...":
                continue
            idx, vals = int(parts[0]), parts[1:]
            assert len(vals) == len(head), (head, vals)
            for k, v in zip(head, vals):
                if k == "..." or v.endswith("..."):
                    continue
                try:
                    v = float(v) if re.fullmatch(r"-?\d+\.\d+", v) else int(v) if re.fullmatch(r"-?[1-9]\d*|0", v) else v
                except ValueError:
                    pass
                rows.setdefault(idx, {})[k] = v
    out = {"source": "pipeline.ipynb cell 1, execute_result text/plain (values only)",
           "n_rows_total": int(re.search(r"\[(\d+) rows x", plain).group(1)),
           "rows": [dict(row=i, **rows[i]) for i in sorted(rows)]}
    json.dump(out, open(OUT, "w"), indent=1)
    print(OUT, len(out["rows"]), "rows,", len(out["rows"][0]) - 1, "columns")


if __name__ == "__main__":
    main()
