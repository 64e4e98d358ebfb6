#!/usr/bin/env python3
"""Benchmark of the hot path (BASELINE.json): frames/s on 2560x2560 uint8 PISCO-like stacks.

One "step" = one pass of the fused hot path (background correction -> threshold -> external
objects -> area/bbox/label map -> crops + property sums) over one reference batch of 200 frames
(segmenter.py:209) already resident in HBM.  The N=1 workload is BASELINE.json configs[1]: a
2000-frame stack (10 batches), ~50 objects/frame.  With N>1 every rank runs its own 10 batches
(weak scaling; batches are independent in the reference) and the object tables are gathered to
rank 0 over RCCL inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel group, timed with HIP events
on the library's stream; `cpu_baseline` is the CPU oracle (a port of the reference's arithmetic) on
one batch of the same frames, run with the reference's process topology on the host cores of this box.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 2560
P = H * W
BATCH = 200
GATHER_EVERY = 4                           # batches per table gather (N > 1)
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec peak
PATH_BYTES_PER_FRAME = 6 * P               # raw read + cleaned write + int32 label write
# algorithmic HBM bytes per frame of each kernel group (DESIGN.md "Kernels and rooflines")
GROUP_BYTES = {
    "bg_correct": 2 * P + P // 8,          # read raw once, write cleaned once, write the 1-bit plane
    "label_map": 4 * P,                    # int32 label map written once
}
GROUP_NAMES = {0: "bg_correct", 2: "sparse_labelling", 3: "label_map", 4: "crops_props"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batches", type=int, default=10, help="distinct 200-frame batches resident per rank")
    ap.add_argument("--workload", default="pisco", choices=["pisco", "uvp5_dense"])
    ap.add_argument("--no-labels", action="store_true", help="streamed mode: label map not materialised (config #5)")
    ap.add_argument("--cpu-frames", type=int, default=200, help="frames of the CPU-baseline sample (0 = skip); 200 = one reference batch")
    ap.add_argument("--cpu-procs", default="3,8", help="background-correction,detection processes of the CPU baseline "
                    "(reference topology: bg_correction.py:120 / segmenter.py:252); '1,1' = single core, in-process")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --single-device rehearses the N > 1 path on one GPU")
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (rehearsal only; never a measurement)")
    ap.add_argument("--streams", type=int, default=3, help="contexts (own HIP stream + host thread each) that alternate batches")
    ap.add_argument("--batches-per-call", type=int, default=1, help="consecutive reference batches handed to the library in one call "
                    "(pseg_process_batches: every batch keeps its own background window; a step stays one batch of 200 frames)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from imageprocessing_amd import sharding, synth
    from imageprocessing_amd.engine import Segmenter

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_device:
        local_rank = 0
    # RCCL (and gloo) print banners to stdout when a communicator comes up; the contract is ONE JSON line on stdout, so
    # file descriptor 1 points at stderr until that line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    cdev = dev if args.backend == "nccl" else "cpu"        # where the collectives' tensors live

    import queue
    import threading
    ns = max(1, args.streams)
    bpc = max(1, args.batches_per_call)                      # batches per library call
    segs = [Segmenter(H, W, max_frames=BATCH * bpc, device=local_rank) for _ in range(ns)]
    seg = segs[0]
    mask = seg.disk_mask(1200)
    cfg = synth.CONFIGS[args.workload]
    nb = max(1, min(args.batches, args.steps + args.warmup))
    nb = ((nb + bpc - 1) // bpc) * bpc                       # whole groups of batches
    seed = 2 + 1000 * rank
    stacks = []                                              # nb / bpc resident groups of bpc consecutive batches each
    for g in range(nb // bpc):
        objs, off = synth.make_objects(BATCH * bpc, H, W, seed, frame0=g * bpc * BATCH, **cfg)
        stacks.append(seg.synth_frames(BATCH * bpc, seed, objs, off, frame0=g * bpc * BATCH))
    cleaned = [torch.empty((BATCH * bpc, H, W), dtype=torch.uint8, device=dev) for _ in range(ns)]
    labels = [None if args.no_labels else torch.empty((BATCH * bpc, H, W), dtype=torch.int32, device=dev) for _ in range(ns)]

    gatherer = sharding.TableGatherer(device=cdev) if world > 1 else None
    wire = np.empty(256 << 20, np.uint8) if world > 1 else None     # reused by pack_table
    group_ms = {k: 0.0 for k in GROUP_NAMES}
    stats = dict(rows=0, saved=0, crop_bytes=0, contours=0, gather_s=0.0)
    lock = threading.Lock()

    def call(c, nbat, k, timed, out_q=None):
        """library call number c: nbat consecutive batches (steps) through the fused hot path on context k"""
        grp = stacks[c % len(stacks)]
        m = nbat * BATCH
        _, _, _, tab = segs[k].process(grp[:m], mask=mask, cleaned=cleaned[k][:m], labels=None if labels[k] is None else labels[k][:m],
                                       batch_len=BATCH)
        if out_q is not None:
            out_q.put((c, tab))                              # packed and gathered by the main thread
        if timed:
            with lock:
                stats["rows"] += tab.n_rows
                stats["saved"] += int(tab.saved.sum())
                stats["crop_bytes"] += tab.crop_bytes
                stats["contours"] += int(tab.frame_n_contours.sum())

    def run(count, timed):
        """count steps (batches) as calls of bpc batches (the last call takes the remainder), dealt round-robin to the
        contexts (one host thread each); the object tables are gathered to rank 0 from the main thread, in call order,
        while later batches compute"""
        calls = [(c, min(bpc, count - c * bpc)) for c in range((count + bpc - 1) // bpc)]
        out_q = queue.Queue() if world > 1 else None
        ths = [threading.Thread(target=lambda k=k: [call(c, nbat, k, timed, out_q) for c, nbat in calls[k::ns]]) for k in range(ns)]
        for t in ths:
            t.start()
        if world > 1:
            # tables of about GATHER_EVERY consecutive batches travel in one message: fewer, larger collectives
            pending, fill, since = {}, 0, 0
            for want, nbat in calls:
                while want not in pending:
                    c, tab = out_q.get()
                    pending[c] = tab
                g0 = time.perf_counter()
                blob = sharding.pack_table(pending.pop(want), frame_base=(rank * nb + (want * bpc) % nb) * BATCH, out=wire[fill:])
                fill += blob.size
                since += nbat
                if since >= GATHER_EVERY or want == calls[-1][0]:
                    gatherer.gather(wire[:fill])
                    fill, since = 0, 0
                if timed:
                    stats["gather_s"] += time.perf_counter() - g0
        for t in ths:
            t.join()

    run(args.warmup, False)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # kernel-group durations for the roofline: HIP events on the library's stream, one context alone on the GPU
    # (with several contexts in flight the kernels of different batches overlap and share the bandwidth)
    nprobe = 3
    for i in range(nprobe):
        call(i, bpc, 0, False)
        for k in GROUP_NAMES:
            group_ms[k] += seg.last_ms(k)

    frames = args.steps * BATCH * world
    fps = frames / elapsed
    # dominant kernel group by device time (HIP events inside the library, on its own stream)
    dom = max(GROUP_NAMES, key=lambda k: group_ms[k])
    dom_name = GROUP_NAMES[dom]
    dom_ms = group_ms[dom] / nprobe
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom_name)    # PMC bytes of a 200-frame launch (profiles/README.md)
            traffic = traffic * bpc if traffic is not None else None      # a launch covers bpc batches; bytes are per frame
        except Exception:
            traffic = None
    if dom_name in GROUP_BYTES:
        launch_bytes = GROUP_BYTES[dom_name] * BATCH * bpc
    else:   # sparse groups: their share of the path's compulsory bytes is the object payload only
        launch_bytes = (stats["crop_bytes"] * 3 + stats["rows"] * 44 * 8) / max(args.steps, 1) * bpc
    achieved = launch_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    obj_bytes = (stats["crop_bytes"] * 3 + stats["rows"] * 44 * 8) / max(args.steps * BATCH, 1)     # rank 0's batches
    out = {
        "metric": "frames/sec (2560x2560 uint8), hot path bg_correction+detection",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"configs[1]: {nb * BATCH}-frame {args.workload} stack per GPU, batches of {BATCH}, bg window 12, "
                               f"~{cfg['per_frame']} objects/frame, label map {'off' if args.no_labels else 'on'}, {bpc} batches per library call",
                   "frames_per_step": BATCH, "objects_per_frame": round(stats["contours"] / (args.steps * BATCH), 1),
                   "rows_per_frame": round(stats["rows"] / (args.steps * BATCH), 1),
                   "saved_per_frame": round(stats["saved"] / (args.steps * BATCH), 1)},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "avg_launch_ms": round(dom_ms, 4), "bytes_per_launch": int(launch_bytes)},
        "path": {"bytes_per_frame": PATH_BYTES_PER_FRAME if not args.no_labels else 2 * P,
                 # compulsory bytes = the frame terms + the measured object payload (three crop planes, 44 x 8 B per row)
                 "object_bytes_per_frame": round(obj_bytes, 1),
                 "frac_of_hbm_peak_per_gpu_with_objects": round(fps / world * ((PATH_BYTES_PER_FRAME if not args.no_labels else 2 * P) + obj_bytes)
                                                                / (HBM_PEAK_GBS * 1e9), 4),
                 "frac_of_hbm_peak_per_gpu": round(fps / world * (PATH_BYTES_PER_FRAME if not args.no_labels else 2 * P) / (HBM_PEAK_GBS * 1e9), 4),
                 "contexts": ns, "table_gather_ms_per_step_rank0": round(stats["gather_s"] / args.steps * 1e3, 3) if world > 1 else None,
                 "batches_per_call": bpc,
                 "group_ms_per_step_single_context": {GROUP_NAMES[k]: round(group_ms[k] / nprobe / bpc, 3) for k in GROUP_NAMES}},
    }

    if rank == 0 and world == 1 and args.cpu_frames >= 12:
        # the oracle is the timed CPU port here, never the product path
        nf = min(args.cpu_frames, BATCH)
        sample = stacks[0][:nf].cpu().numpy()                # nf <= BATCH: inside the first batch
        hmask = mask.cpu().numpy()
        pb, pd = (int(v) for v in args.cpu_procs.split(","))
        if pb == 1 and pd == 1:
            from tests import helpers
            c0 = time.perf_counter()
            items = helpers.oracle_bgcorrect(sample)
            helpers.oracle_detect(items, hmask, want_labels=False)
            ct = time.perf_counter() - c0
            cores, how = 1, "single thread, in-process"
        else:
            from oracle import cpu_pool
            r = cpu_pool.run(sample, hmask, pb, pd)
            ct = r["seconds_bg"] + r["seconds_det"]
            cores = max(pb, pd)
            how = (f"reference topology: background correction in {pb} processes ({r['seconds_bg']:.2f} s), then detection in "
                   f"{pd} processes ({r['seconds_det']:.2f} s), frames through /dev/shm")
        out["cpu_baseline"] = {"value": round(nf / ct, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": f"first {nf} frames of batch 0 as one batch (window 12), NumPy + C oracle, {how}, "
                                         f"no PNG/CSV writes, host has {os.cpu_count()} logical CPUs"}
    if world > 1:
        dist.barrier()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    for sg in segs:
        sg.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
