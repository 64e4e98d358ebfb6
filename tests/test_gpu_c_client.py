"""The drop-in boundary seen from a C program: tests/c_client/abi_client.c includes include/piscoseg.h, links
libpiscoseg.so and the HIP runtime with gcc (built by __graft_entry__.build() in the build container) and runs the fused
path without Python or torch in its process.  Its printed results against the Python binding on the same input."""
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CLIENT = os.path.join(HERE, "c_client", "abi_client")
HEADER = os.path.join(HERE, "..", "include", "piscoseg.h")


def test_header_is_plain_c_and_cxx():
    for std, lang in (("-std=c99", "c"), ("-std=c11", "c"), ("-std=c++17", "c++")):
        subprocess.check_call(["gcc", std, "-x", lang, "-fsyntax-only", "-Wall", "-Werror", HEADER])


def test_c_client_is_built_with_the_library():
    import __graft_entry__ as g
    path = g.build_c_client()
    assert os.path.exists(path)
    out = subprocess.run(["ldd", path], capture_output=True, text=True).stdout
    assert "libpiscoseg.so" in out and "not found" not in out.split("libpiscoseg.so")[1].splitlines()[0]


@pytest.mark.gpu
def test_c_program_gets_the_python_bindings_results():
    import torch
    from imageprocessing_amd import synth
    from imageprocessing_amd.engine import Segmenter
    assert os.path.exists(CLIENT), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    H, W, n, seed = 256, 256, 14, 11
    objs, off = synth.make_objects(n, H, W, seed, per_frame=10, dmin=40, dmax=700)
    objs = np.asarray(objs, np.int32).reshape(-1, 8)
    text = f"{len(objs)}\n" + "\n".join(" ".join(str(int(v)) for v in o) for o in objs) + "\n" + " ".join(str(int(v)) for v in off) + "\n"
    env = dict(os.environ)
    env.pop("LD_LIBRARY_PATH", None)                       # the client's rpath decides (system HIP runtime, in-tree library)
    run = subprocess.run([CLIENT, str(H), str(W), str(n), str(seed)], input=text, capture_output=True, text=True, timeout=120, env=env)
    assert run.returncode == 0, run.stderr
    lines = run.stdout.splitlines()

    seg = Segmenter(H, W, max_frames=n, device=0)
    try:
        frames = seg.synth_frames(n, seed, objs, off)
        mask = seg.disk_mask(min(H, W) // 2 - 6)
        labels = torch.empty((n, H, W), dtype=torch.int32, device="cuda")
        cleaned, ms, valid, tab = seg.process(frames, mask=mask, labels=labels, min_area_to_save=150.0)
        head = lines[0].split()
        assert (int(head[1]), int(head[3]), int(head[5])) == (tab.n_rows, len(tab.saved_row), tab.crop_bytes)
        assert tab.n_rows > 20 and len(tab.saved_row) > 5
        fl = [ln.split() for ln in lines if ln.startswith("frame ")]
        assert len(fl) == n
        for f, t in enumerate(fl):
            assert float(t[3]) == ms[f, 0] and float(t[5]) == ms[f, 1] and int(t[7]) == valid[f]
            assert int(t[9]) == tab.frame_row_offset[f + 1] - tab.frame_row_offset[f]
        rl = [ln.split() for ln in lines if ln.startswith("row ")]
        assert len(rl) == tab.n_rows
        for i, t in enumerate(rl):
            assert [int(t[2]), int(t[3]), int(t[4])] == [tab.frame[i], tab.c[i], tab.label[i]]
            assert float(t[5]) == tab.area[i]
            assert [int(v) for v in t[6:11]] == [tab.x[i], tab.y[i], tab.w[i], tab.h[i], int(tab.saved[i])]
        pl = [ln.split() for ln in lines if ln.startswith("props ")]
        assert len(pl) == len(tab.saved_row)
        for s, t in enumerate(pl):
            got = np.array([float(v) for v in t[2:]])
            assert np.array_equal(got, tab.saved_props[s], equal_nan=True)
        sums = [ln.split() for ln in lines if ln.startswith("cleaned_sum")][0]
        ok = valid.astype(bool)
        assert int(sums[1]) == int(cleaned[ok].to(torch.int64).sum()) and int(sums[3]) == int(labels[ok].to(torch.int64).sum())
        err = [ln for ln in lines if ln.startswith("too_long")][0]
        assert "rc -1" in err and "out of range" in err
    finally:
        seg.close()
