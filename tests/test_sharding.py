"""Multi-GPU layout on CPU: the reference's batch partition, table packing / merging, and the
gather collective with world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest

from imageprocessing_amd import sharding
from oracle import pipeline_ref


@pytest.mark.parametrize("n", [1, 11, 150, 199, 200, 201, 399, 400, 401, 650, 1000, 10000])
def test_batches_match_reference_partition(n):
    files = list(range(n))
    want = pipeline_ref.batches_of(files)                      # restates segmenter.py:209-213
    got = sharding.batches_of(n)
    assert [(b[0], len(b)) for b in want] == got
    assert sum(m for _, m in got) == n


def test_assign_batches_contiguous_and_balanced():
    b = sharding.batches_of(10000)
    for world in (1, 2, 4, 8):
        parts = sharding.assign_batches(b, world)
        assert sum(parts, []) == b
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1


class _Tab:
    pass


def fake_table(n_frames, rng, crop_bytes_per_saved=50):
    t = _Tab()
    rows_per_frame = rng.integers(0, 6, n_frames)
    nr = int(rows_per_frame.sum())
    t.n_frames = n_frames
    t.frame_row_offset = np.concatenate([[0], np.cumsum(rows_per_frame)]).astype(np.int32)
    t.frame_n_contours = (rows_per_frame + 2).astype(np.int32)
    t.frame_valid = np.ones(n_frames, np.uint8)
    t.frame_mean = rng.random(n_frames)
    t.frame_std = rng.random(n_frames)
    t.frame = np.repeat(np.arange(n_frames), rows_per_frame).astype(np.int32)
    t.c = np.concatenate([np.arange(k) for k in rows_per_frame] + [np.zeros(0, int)]).astype(np.int32)
    t.label = t.c + 1
    t.area = rng.random(nr) * 1000
    t.x, t.y, t.w, t.h = (rng.integers(0, 100, nr).astype(np.int32) for _ in range(4))
    t.saved = (t.area > 400).astype(np.uint8)
    t.saved_row = np.nonzero(t.saved)[0].astype(np.int32)
    ns = len(t.saved_row)
    t.saved_props = rng.random((ns, 32))
    t.crop_offset = (np.arange(ns + 1) * crop_bytes_per_saved).astype(np.int64)
    t.sums = rng.integers(0, 1000, (ns, 24)).astype(np.int64)
    t.crop_bytes = ns * crop_bytes_per_saved
    return t


def test_pack_merge_is_independent_of_sharding():
    rng = np.random.default_rng(0)
    shards = [fake_table(7, rng), fake_table(5, rng), fake_table(9, rng)]
    bases = [0, 7, 12]
    blobs = [sharding.pack_table(t, b) for t, b in zip(shards, bases)]
    merged = sharding.merge_tables([sharding.unpack_table(b) for b in reversed(blobs)])
    assert list(merged["frame"]) == sum([list(t.frame + b) for t, b in zip(shards, bases)], [])
    assert merged["frame_row_offset"][-1] == sum(len(t.frame) for t in shards)
    assert len(merged["frame_row_offset"]) == 22
    assert np.array_equal(merged["area"], np.concatenate([t.area for t in shards]))
    for s, r in enumerate(merged["saved_row"]):
        assert merged["saved"][r] == 1
    assert merged["crop_offset"][-1] == sum(t.crop_bytes for t in shards)
    # several tables back to back in one message (bench.py gathers a few batches at a time)
    both = np.concatenate(blobs[:2])
    parts = sharding.split_tables(both)
    assert [p.size for p in parts] == [blobs[0].size, blobs[1].size]
    assert np.array_equal(sharding.unpack_table(parts[1])["area"], shards[1].area)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(rank)
    tab = fake_table(4 + rank, rng)
    blob = sharding.pack_table(tab, frame_base=100 * rank)
    got = sharding.gather_tables(blob)
    if rank == 0:
        merged = sharding.merge_tables([sharding.unpack_table(b) for b in got])
        q.put((len(got), int(merged["frame_row_offset"][-1]), [int(v) for v in merged["frame"][:3]]))
    else:
        assert got is None
        q.put(len(tab.frame))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_tables_gloo_world2():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    r0 = [r for r in res if isinstance(r, tuple)][0]
    r1 = [r for r in res if not isinstance(r, tuple)][0]
    rows0 = len(fake_table(4, np.random.default_rng(0)).frame)
    assert r0[0] == 2 and r0[1] == rows0 + r1


def _worker_plan(rank, world, port, q, n_files):
    """what bench.py --gpus N does with real tables: the rank's parts of the 10 000-frame plan, one fake table per part,
    packed with the part's global frame base, a few parts per gather message, merged on rank 0"""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    parts = sharding.plan_ranges(n_files, world)[rank]
    gat = sharding.TableGatherer()
    # every rank makes the same number of gather calls (a collective): the plan gives ranks equally many parts +- 1
    nparts = [len(p) for p in sharding.plan_ranges(n_files, world)]
    ncalls = max((k + 3) // 4 for k in nparts)
    got, rows = [], 0
    wire = np.empty(8 << 20, np.uint8)
    for g in range(ncalls):
        fill = 0
        for p in parts[4 * g:4 * g + 4]:
            tab = fake_table(p["n_emit"], np.random.default_rng(p["start"]))
            rows += len(tab.frame)
            fill += sharding.pack_table(tab, frame_base=p["start"], out=wire[fill:]).size
        res = gat.gather(wire[:fill])
        if rank == 0:
            for blob in res:
                got += [sharding.unpack_table(b) for b in sharding.split_tables(blob)]
    if rank == 0:
        merged = sharding.merge_tables(got)
        frames_with_rows = np.unique(merged["frame"])
        q.put(("merged", len(merged["frame_row_offset"]) - 1, int(merged["frame_row_offset"][-1]),
               bool((np.diff(merged["frame"]) >= 0).all()), int(frames_with_rows.max()), rows))
    else:
        q.put(("rows", rows))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_range_plan_gather_merge_gloo_world2():
    import torch.multiprocessing as mp
    n_files = 10000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_plan, args=(r, 2, port, q, n_files)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    m = [r for r in res if r[0] == "merged"][0]
    other = [r for r in res if r[0] == "rows"][0]
    assert m[1] == n_files                                  # every frame of the stack exactly once
    assert m[2] == m[5] + other[1]                          # every row of both ranks
    assert m[3] and m[4] <= n_files - 1                     # global (frame, c) order


@pytest.mark.parametrize("n_files,world", [(2000, 8), (10000, 8), (450, 3), (200, 2), (30, 4), (1999, 7), (13, 2), (5, 3)])
def test_plan_ranges_covers_every_frame_once_with_local_windows(n_files, world):
    """Every frame is emitted by exactly one part, and the background window the reference gives it
    (bg_correction.py:72, inside its own batch) lies inside the frames that part loads."""
    window = 12
    plan = sharding.plan_ranges(n_files, world, window=window)
    assert len(plan) == world
    batches = sharding.batches_of(n_files)
    owner = np.full(n_files, -1)
    for rank, parts in enumerate(plan):
        for p in parts:
            s, ne, na = p["start"], p["n_emit"], p["n_avail"]
            assert ne >= 1 and na >= ne
            bs, bl = next((bs, bl) for bs, bl in batches if bs <= s < bs + bl)
            assert s + na <= bs + bl, "a part never reads past its batch"
            assert (owner[s:s + ne] == -1).all()
            owner[s:s + ne] = rank
            for g in range(s, s + ne):
                i = g - bs                                   # index inside the reference batch
                if bl >= window:
                    start = i if i + window < bl else bl - window
                    assert s - bs <= start and start + window <= s - bs + na
                    il = g - s                               # what pseg_process_part computes locally
                    local = il if (na > ne or il + window < na) else na - window
                    assert local + (s - bs) == start
    assert (owner >= 0).all()
    assert (np.diff(owner) >= 0).all(), "ranks own contiguous, ordered ranges"


def _bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.parametrize("n", [2, 3])
def test_bench_gpus_n_starts_n_ranks_itself(n):
    """`python3 bench.py --gpus N` (no torchrun, no WORLD_SIZE) must run N ranks: the launcher starts them, rank 0 prints the
    one line, the parent checks n_gpus and ranks_seen against --gpus.  --plan-only: process group + plan_ranges + gathers of
    empty tables, no kernels (this container has no GPU)."""
    r, line = _bench("--gpus", str(n), "--backend", "gloo", "--plan-only", "--frames", "2000")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1, "exactly one line on stdout"
    assert line["n_gpus"] == n and line["ranks_seen"] == n and line["launch"] == "self-spawned"
    assert line["frames_merged"] == 2000 and sum(line["frames_per_rank"]) == 2000 and len(line["frames_per_rank"]) == n
    assert max(line["halo_frames_per_rank"]) <= 11 * max(line["parts_per_rank"])


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 4 inside a 1-rank world (the way round 2's bench silently ran one rank) is an error, not a 1-rank number"""
    r, line = _bench("--gpus", "4", "--backend", "gloo", "--plan-only", env={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and line is None
    assert "WORLD_SIZE" in r.stderr


def test_bench_launcher_fails_when_a_rank_fails():
    r, line = _bench("--gpus", "2", "--backend", "gloo", "--plan-only", "--frames", "-5")
    assert r.returncode != 0 and line is None
