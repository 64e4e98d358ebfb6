#!/usr/bin/env python3
"""Benchmark of the hot path (BASELINE.json): frames/s on 2560x2560 uint8 PISCO-like stacks.

One "step" = one pass of the fused hot path (background correction -> threshold -> external objects -> area / bbox /
label map -> crops + property sums) over the whole stack of the configuration, resident in HBM:

  N = 1   BASELINE.json configs[1]: a 2000-frame stack = 10 reference batches of 200 frames (segmenter.py:209), ~50
          objects per frame (--workload uvp5_dense / uvp5_dense_saved: configs[3], ~2000 particles per frame).
  N > 1   configs[2]: a 10 000-frame stack cut into contiguous frame ranges (sharding.plan_ranges): every rank holds its
          range plus the forward halos of background-window frames, runs pseg_process_part on its parts, and the object
          tables are gathered to rank 0 over RCCL inside the timed region.  The total work is fixed: "scaling": "strong".
          --scaling weak: every rank runs its own 2000-frame stack (whole batches, no halo).

`python3 bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes itself (children of
this process, started before anything here touches torch or the GPU; rank r uses GPU r) and relays rank 0's line; under
`python -m torch.distributed.run` (WORLD_SIZE set) every process is a rank.  Either way the run fails when the ranks that
took part are not --gpus.  --plan-only: process group, frame-range plan and the table gathers with empty tables, no kernels
(what the CPU test of the launcher runs).

A single caller thread keeps --streams contexts in flight with pseg_submit / pseg_wait (--driver threads: one host thread
per context, the round-1 form).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel group, timed with HIP events on the library's stream
during the timed region; `roofline_groups` has every group; `cpu_baseline` is the CPU oracle (a port of the reference's
arithmetic) on BASELINE.json configs[0] (100 frames, seed 1) with the reference's process topology on this box's cores.
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
GATHER_EVERY = 4                           # library calls per table gather (N > 1)
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: 8.0 TB/s spec peak
PATH_BYTES_PER_FRAME = 6 * P               # raw read + cleaned write + int32 label write
GROUP_NAMES = {0: "bg_correct", 2: "sparse_labelling", 3: "label_map", 4: "crops_props",   # (+ 1: "threshold" with --deconv-standin)
               6: "table_beside_label_map"}               # boxes, areas, ranks, rows, size scans on the side stream (hidden under group 3)


def group_bytes_per_frame(name, rows_pf, crop_bytes_pf, labels_on):
    """algorithmic HBM bytes per frame of a kernel group"""
    if name == "bg_correct":
        return 2 * P + P // 8              # raw read once, cleaned written once, the 1-bit threshold plane written
    if name == "label_map":
        return 4 * P if labels_on else 0   # int32 label map written once
    if name == "threshold":
        return P + P // 8                  # the (deconvolved) cleaned frame read once, the 1-bit threshold plane written
    if name == "sparse_labelling":
        return P // 8                      # the threshold plane read once
    if name == "table_beside_label_map":
        return 44 * 8 * rows_pf            # the rows of the table (what the group leaves; it reads the run tables)
    return 4 * crop_bytes_pf + 44 * 8 * rows_pf      # crops: one read + three planes written; the rows of the table


def launch_ranks(n):
    """--gpus n without a launcher: start the n ranks as child processes, one GPU each, and relay rank 0's JSON line.
    Nothing in this process has imported torch or touched the GPU (children are plain subprocesses, never an exec of a
    process that holds the device).  Exit status non-zero when a rank fails or the line does not report n ranks."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PSEG_BENCH_LAUNCH="self-spawned")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    failed = None
    live = set(range(n))
    while live and failed is None:
        time.sleep(0.05)                                     # (rank 0 writes one short line into its pipe: it cannot fill up)
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is not None:
                live.discard(r)
                if rc != 0:
                    failed = (r, rc)
                    break
    if failed is not None:
        for r in live:
            procs[r].terminate()                             # exactly the processes started above
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
        sys.stderr.write(f"bench: rank {failed[0]} exited with status {failed[1]}\n")
        return 1
    text = procs[0].stdout.read().decode()
    lines = [ln for ln in text.splitlines() if ln.strip().startswith("{")]
    if not lines:
        sys.stderr.write("bench: rank 0 printed no result line\n")
        return 1
    out = json.loads(lines[-1])
    seen = out.get("ranks_seen", out.get("path", {}).get("ranks_seen"))
    if out.get("n_gpus") != n or seen != n:
        sys.stderr.write(f"bench: --gpus {n} but the line reports n_gpus {out.get('n_gpus')}, ranks_seen {seen}\n")
        return 2
    print(lines[-1], flush=True)
    return 0


class _EmptyTable:
    """an object table of n frames without objects (--plan-only)"""
    def __init__(self, n):
        z32, z64, z8 = np.zeros(0, np.int32), np.zeros(0, np.float64), np.zeros(0, np.uint8)
        self.n_frames, self.crop_bytes = n, 0
        self.frame_row_offset = np.zeros(n + 1, np.int32)
        self.frame_n_contours = np.zeros(n, np.int32)
        self.frame_valid = np.ones(n, np.uint8)
        self.frame_mean, self.frame_std = np.zeros(n), np.zeros(n)
        self.frame = self.c = self.label = self.x = self.y = self.w = self.h = self.saved_row = z32
        self.area, self.saved = z64, z8
        self.saved_props = np.zeros((0, 32))
        self.crop_offset = np.zeros(1, np.int64)


def plan_only(args, world, rank, launch):
    """process group + frame-range plan + the gather messages of one pass with empty tables: no kernels, no GPU"""
    import torch.distributed as dist
    from imageprocessing_amd import sharding
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dist.init_process_group("gloo")
    total_frames = args.frames or 10000
    plan = sharding.plan_ranges(total_frames, world, window=2 * args.n_bg_imgs)
    parts = plan[rank]
    n_gather = max(max((len(p) + GATHER_EVERY - 1) // GATHER_EVERY for p in plan), 1)
    gat = sharding.TableGatherer()
    got = []
    for g in range(n_gather):
        mine = parts[(g * len(parts)) // n_gather:((g + 1) * len(parts)) // n_gather]
        blobs = [sharding.pack_table(_EmptyTable(p["n_emit"]), frame_base=p["start"]) for p in mine]
        res = gat.gather(np.concatenate(blobs) if blobs else np.zeros(8, np.uint8))
        if rank == 0:
            for blob in res:
                if blob.size > 8:
                    got += [sharding.unpack_table(b) for b in sharding.split_tables(blob)]
    line = None
    if rank == 0:
        merged = sharding.merge_tables(got)
        emitted = [sum(p["n_emit"] for p in pp) for pp in plan]
        loaded = [sum(p["n_avail"] for p in pp) for pp in plan]
        line = {"plan_only": True, "n_gpus": args.gpus, "ranks_seen": dist.get_world_size(), "launch": launch,
                "frames": total_frames, "frames_merged": int(len(merged["frame_row_offset"]) - 1),
                "frames_per_rank": emitted, "halo_frames_per_rank": [a - b for a, b in zip(loaded, emitted)],
                "parts_per_rank": [len(pp) for pp in plan], "table_gather_messages_per_step": n_gather}
    dist.barrier()
    dist.destroy_process_group()
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0, help="frames of the stack (default: 2000 at N = 1 / weak, 10000 strong)")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"], help="N > 1 only")
    ap.add_argument("--workload", default="pisco", choices=["pisco", "uvp5_dense", "uvp5_dense_saved"])
    ap.add_argument("--no-labels", action="store_true", help="streamed mode: label map not materialised (config #5)")
    ap.add_argument("--cpu", default="writes", choices=["on", "off", "writes"], help="CPU baseline: on = configs[0] without file "
                    "writes; writes = also with PNG / CSV writes; off = skip")
    ap.add_argument("--cpu-frames", type=int, default=100, help="frames of the CPU-baseline sample (configs[0]: 100)")
    ap.add_argument("--cpu-procs", default="3,8", help="background-correction,detection processes of the CPU baseline "
                    "(reference topology: bg_correction.py:120 / segmenter.py:252); '1,1' = single core, in-process")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo + --single-device rehearses the N > 1 path on one GPU")
    ap.add_argument("--single-device", action="store_true", help="every rank uses cuda:0 (rehearsal only; never a measurement)")
    ap.add_argument("--streams", type=int, default=3, help="contexts (own HIP stream each) that alternate library calls")
    ap.add_argument("--driver", default="submit", choices=["submit", "threads"])
    ap.add_argument("--force-dist", action="store_true", help="run the N > 1 code path (frame ranges, process group, table gathers on "
                    "their thread) with a world of one rank: all a one-GPU box allows through RCCL")
    ap.add_argument("--host-batches", type=int, default=6, help="batches of the host-input measurement (0 = skip)")
    ap.add_argument("--deconv-standin", action="store_true", help="the reference's default two-call path (run_pipeline.py:25, "
                    "segmenter.py:243-254): pseg_bgcorrect -> an elementwise stand-in for the deconvolution network on the device -> "
                    "pseg_detect(cleaned = its output, raw = the background-corrected frames); adds the `threshold` group")
    ap.add_argument("--plan-only", action="store_true", help="process group + frame-range plan + gathers of empty tables over gloo; "
                    "no kernels, no GPU (the launcher's CPU test)")
    ap.add_argument("--n-bg-imgs", type=int, default=6, help="half the background window (bg_correction.py:65; reference: 6)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))                    # before torch is imported: this process never touches a GPU
    launch = os.environ.get("PSEG_BENCH_LAUNCH") or ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "single process")
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench: --gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE')}: the ranks that run must be --gpus")
    if args.plan_only:
        sys.stdout.flush()
        real_stdout = os.dup(1)                              # (gloo prints a banner to stdout: ONE line is the contract)
        os.dup2(2, 1)
        line = plan_only(args, int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), launch)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        if line is not None:
            print(json.dumps(line), flush=True)
        return

    import torch
    import torch.distributed as dist
    from imageprocessing_amd import sharding, synth
    from imageprocessing_amd.engine import HostPipeline, Segmenter

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
    dist_on = world > 1 or args.force_dist
    if dist_on:
        # frame-sharded runs pack and gather a table between a wait and the next submit: the next call arrives too late for the
        # library's deferred label fills to pay (each adds a millisecond of latency to its call; 96 000 - 103 000 against
        # 103 000 - 106 000 frames/s with --force-dist on one GPU, profiles/r03/21_deferred_fill.txt), so they stay off here
        os.environ.setdefault("PSEG_DEFER_FILL", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            # (device_id: the communicator -- and RCCL's streams -- come up now, before the library's contexts create theirs.  A
            # process has four hardware queues and a stream goes to the one with the fewest users when it is first used; which
            # streams end up sharing one is a lottery the program cannot see -- a kernel trace with two contexts on one queue is in
            # profiles/r03/23_frame_sharded_queues.txt -- and six alternating pairs of --force-dist runs gave 102 800 - 105 300
            # frames/s this way round against 98 400 - 103 700 with the communicator created at the first collective
            # (BENCH_LAZY_COMM=1))
            if os.environ.get("BENCH_LAZY_COMM"):
                dist.init_process_group("nccl")
            else:
                dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group("gloo")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    cdev = dev if args.backend == "nccl" else "cpu"        # where the collectives' tensors live

    strong = dist_on and args.scaling == "strong"
    if args.deconv_standin:
        if dist_on:
            raise SystemExit("bench: --deconv-standin is a single-GPU variant")
        args.driver = "threads"                              # two library calls per batch: one host thread per context
        GROUP_NAMES[1] = "threshold"
    total_frames = args.frames or (10000 if strong else 2000)
    cfg = synth.CONFIGS[args.workload]
    ns = max(1, args.streams)
    labels_on = not args.no_labels

    # ---- the rank's frames and its library calls (start, n_emit, n_avail): whole batches, or parts with forward halos ----
    if strong:
        seed = 3
        parts = sharding.plan_ranges(total_frames, world, window=2 * args.n_bg_imgs)[rank]
        g0 = parts[0]["start"] if parts else 0
        g1 = max((p["start"] + p["n_avail"] for p in parts), default=0)
        calls = [(p["start"] - g0, p["n_emit"], p["n_avail"], p["start"]) for p in parts]
    else:
        seed = 2 + 1000 * rank
        g0, g1 = 0, total_frames
        calls = [(a, m, m, a + rank * total_frames) for a, m in sharding.batches_of(total_frames, BATCH)]
    max_call = max((c[2] for c in calls), default=1)
    segs = [Segmenter(H, W, max_frames=max(max_call, BATCH), device=local_rank) for _ in range(ns)]
    seg = segs[0]
    mask = seg.disk_mask(1200)
    stack = torch.empty((max(g1 - g0, 1), H, W), dtype=torch.uint8, device=dev)
    for a in range(0, g1 - g0, BATCH):                       # rendered on the device, 200 frames at a time
        m = min(BATCH, g1 - g0 - a)
        objs, off = synth.make_objects(m, H, W, seed, frame0=g0 + a, **cfg)
        seg.synth_frames(m, seed, objs, off, frame0=g0 + a, out=stack[a:a + m])
    cleaned = [torch.empty((max_call, H, W), dtype=torch.uint8, device=dev) for _ in range(ns)]
    labels = [torch.empty((max_call, H, W), dtype=torch.int32, device=dev) if labels_on else None for _ in range(ns)]
    deconv = [torch.empty((max_call, H, W), dtype=torch.uint8, device=dev) for _ in range(ns)] if args.deconv_standin else None
    emitted = sum(c[1] for c in calls)
    loaded = sum(c[2] for c in calls)

    gatherer = sharding.TableGatherer(device=cdev) if dist_on else None
    group_ms = {k: 0.0 for k in GROUP_NAMES}
    stats = dict(rows=0, saved=0, crop_bytes=0, contours=0, gather_s=0.0, calls=0, frames=0)

    def kwargs(c, k):
        a, ne, na, _ = calls[c]
        return dict(mask=mask, cleaned=cleaned[k][:ne], labels=None if labels[k] is None else labels[k][:ne],
                    n_emit=None if ne == na else ne, n_bg_imgs=args.n_bg_imgs)

    def account(c, k, tab, timed):
        if timed:
            stats["rows"] += tab.n_rows
            stats["saved"] += len(tab.saved_row)
            stats["crop_bytes"] += tab.crop_bytes
            stats["contours"] += int(tab.frame_n_contours.sum())
            stats["calls"] += 1
            stats["frames"] += calls[c][1]
            for g in GROUP_NAMES:
                group_ms[g] += segs[k].last_ms(g)

    # Table gathers: every rank makes the same number of gather calls per pass (a collective), whatever its number of parts;
    # the tables of about GATHER_EVERY consecutive calls travel in one message, and the collective itself runs on a thread
    # of its own so that the caller thread keeps submitting (it only packs the table: plain copies).
    import queue as _queue
    import threading as _threading
    if strong:
        n_gather = max(max((len(p) + GATHER_EVERY - 1) // GATHER_EVERY for p in sharding.plan_ranges(total_frames, world, window=2 * args.n_bg_imgs)), 1)
    else:
        n_gather = max((len(calls) + GATHER_EVERY - 1) // GATHER_EVERY, 1)
    nc_ = len(calls)
    group_end = [((g + 1) * nc_) // n_gather for g in range(n_gather)]      # calls [group_end[g-1], group_end[g]) form message g
    wires = [np.empty(64 << 20, np.uint8) for _ in range(3)] if dist_on else []
    gq = _queue.Queue()
    free_wires = _queue.Queue()
    for w_ in wires:
        free_wires.put(w_)

    def gather_main():
        torch.cuda.set_device(local_rank)                    # a new thread starts on device 0
        while True:
            it = gq.get()
            if it is None:
                return
            buf, fill, timed = it
            g0_ = time.perf_counter()
            gatherer.gather(buf[:max(fill, 8)])
            if timed:
                stats["gather_s"] += time.perf_counter() - g0_
            free_wires.put(buf)
    gth = None
    if dist_on:
        gth = _threading.Thread(target=gather_main, daemon=True)
        gth.start()

    class Gather:
        def __init__(self):
            self.fill, self.g, self.buf = 0, 0, None

        def add(self, c, tab, last, timed):
            if not dist_on:
                return
            if self.buf is None:
                self.buf = free_wires.get()
            need = sharding.packed_bytes(tab)
            if self.fill + need > self.buf.size:           # a dense table (256 B of properties per saved object): a larger wire
                grown = np.empty(max(2 * self.buf.size, self.fill + need), np.uint8)
                grown[:self.fill] = self.buf[:self.fill]
                self.buf = grown                           # (returns to free_wires in place of the small one)
            self.fill += sharding.pack_table(tab, frame_base=calls[c][3], out=self.buf[self.fill:]).size
            assert self.fill <= self.buf.size
            while self.g < n_gather and c + 1 >= group_end[self.g]:          # (also sends the empty messages of a rank with few parts)
                gq.put((self.buf, self.fill, timed))
                self.g += 1
                self.fill, self.buf = 0, None
                if self.g < n_gather and c + 1 >= group_end[self.g]:
                    self.buf = free_wires.get()

        def finish(self, timed):
            while dist_on and self.g < n_gather:                           # a rank without parts still takes part in every gather
                gq.put((self.buf if self.buf is not None else free_wires.get(), self.fill, timed))
                self.g += 1
                self.fill, self.buf = 0, None

    def gathers_done():
        if dist_on:
            while free_wires.qsize() < len(wires):                           # every message sent and its buffer returned
                time.sleep(0.0002)

    def run_passes(count, timed):
        """count passes over this rank's calls (count steps), as ONE pipelined sequence of library calls over the contexts:
        the pipeline is not drained between passes"""
        nc = len(calls)
        total = count * nc
        state = {"gat": None}

        def done(i, k, tab, ms=None):
            c = i % nc
            if c == 0:
                state["gat"] = Gather()
            if ms is None:
                account(c, k, tab, timed)
            elif timed:
                stats["rows"] += tab.n_rows; stats["saved"] += len(tab.saved_row); stats["crop_bytes"] += tab.crop_bytes
                stats["contours"] += int(tab.frame_n_contours.sum()); stats["calls"] += 1; stats["frames"] += calls[c][1]
                for g in GROUP_NAMES:
                    group_ms[g] += ms[g]
            state["gat"].add(c, tab, c == nc - 1, timed)
            if c == nc - 1:
                state["gat"].finish(timed)
        if args.driver == "submit":
            inflight = []
            hp = [] if os.environ.get("BENCH_HOST_PROF") else None   # (diagnostic: host time per loop turn in wait / submit / done, to stderr)
            for i in range(total):
                k = i % ns
                finished = None
                ta = time.perf_counter()
                if len(inflight) == ns:
                    i0, k0 = inflight.pop(0)
                    finished = (i0, k0, segs[k0].wait()[3], {g: segs[k0].last_ms(g) for g in GROUP_NAMES})
                tb = time.perf_counter()
                c = i % nc
                a, ne, na, _ = calls[c]
                segs[k].submit(stack[a:a + na], **kwargs(c, k))     # (the context goes back to work before its table is looked at:
                inflight.append((i, k))                              # the table is a copy, and the group times were read above)
                tc = time.perf_counter()
                if finished is not None:
                    done(*finished)
                if hp is not None:
                    hp.append((tb - ta, tc - tb, time.perf_counter() - tc))
            if hp and timed:
                h = np.array(hp) * 1e3
                print("host prof (ms) wait/submit/done mean", h.mean(0).round(3), "p99", np.percentile(h, 99, axis=0).round(3), "max", h.max(0).round(3), file=sys.stderr)
            for i0, k0 in inflight:
                done(i0, k0, segs[k0].wait()[3])
        else:
            import queue
            import threading
            out_q = queue.Queue()

            def work(k):
                for i in range(k, total, ns):
                    c = i % nc
                    a, ne, na, _ = calls[c]
                    if args.deconv_standin:
                        cb, _, v = segs[k].bgcorrect(stack[a:a + na], n_bg_imgs=args.n_bg_imgs, cleaned=cleaned[k][:na])
                        ms_bg = segs[k].last_ms(0)
                        torch.bitwise_or(cb, 0, out=deconv[k][:na])                # elementwise stand-in for the network (its output
                        tab = segs[k].detect(deconv[k][:na], raw=cb, mask=mask, valid=v,   # equals its input: results stay checkable)
                                             labels=None if labels[k] is None else labels[k][:na])
                        ms = {g: segs[k].last_ms(g) for g in GROUP_NAMES}
                        ms[0] = ms_bg
                        out_q.put((i, k, tab, ms))
                        continue
                    tab = segs[k].process(stack[a:a + na], **kwargs(c, k))[3]
                    out_q.put((i, k, tab, {g: segs[k].last_ms(g) for g in GROUP_NAMES}))
            ths = [threading.Thread(target=work, args=(k,)) for k in range(ns)]
            for t in ths:
                t.start()
            pending = {}
            for want in range(total):
                while want not in pending:
                    i, k, tab, ms = out_q.get()
                    pending[i] = (k, tab, ms)
                k, tab, ms = pending.pop(want)
                done(want, k, tab, ms)
            for t in ths:
                t.join()
        if nc == 0:
            for _ in range(count):
                Gather().finish(timed)

    run_passes(args.warmup, False)
    gathers_done()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_passes(args.steps, True)
    gathers_done()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- after the timed region: what was timed is checked, and the groups are timed once more with the GPU to themselves ----
    checked = None
    iso_ms = {k: 0.0 for k in GROUP_NAMES}
    if calls:
        # the first call of this rank on every context at once against a single context with the device to itself
        a, ne, na, _ = calls[0]
        want_c = torch.empty((ne, H, W), dtype=torch.uint8, device=dev)
        want_l = torch.empty((ne, H, W), dtype=torch.int32, device=dev) if labels_on else None
        _, ms0, v0, t0_ = segs[0].process(stack[a:a + na], mask=mask, cleaned=want_c, labels=want_l, n_emit=None if ne == na else ne,
                                          n_bg_imgs=args.n_bg_imgs)
        for k in GROUP_NAMES:
            iso_ms[k] = segs[0].last_ms(k)
        ref = {f: np.array(getattr(t0_, f)) for f in ("frame_row_offset", "frame_n_contours", "frame_mean", "frame_std", "c", "label",
                                                      "area", "x", "y", "w", "h", "saved", "saved_props", "crop_offset")}
        for k in range(ns):
            segs[k].submit(stack[a:a + na], **kwargs(0, k))
        checked = True
        okv = torch.from_numpy(v0.astype(bool)).to(dev)
        for k in range(ns):
            c_, ms_, v_, t_ = segs[k].wait()
            same = torch.equal(c_[okv], want_c[okv]) and np.array_equal(ms_, ms0) and np.array_equal(v_, v0)
            same = same and (not labels_on or torch.equal(labels[k][:ne], want_l))
            same = same and all(np.array_equal(getattr(t_, f), ref[f], equal_nan=True) for f in ref)
            checked = checked and bool(same)
        if args.deconv_standin:
            # the two-call path alone on one context: its groups with the GPU to themselves, its results against the fused call
            cb, ms_b, v_b = segs[0].bgcorrect(stack[a:a + na], n_bg_imgs=args.n_bg_imgs, cleaned=cleaned[0][:na])
            iso_ms[0] = segs[0].last_ms(0)
            torch.bitwise_or(cb, 0, out=deconv[0][:na])
            t_ = segs[0].detect(deconv[0][:na], raw=cb, mask=mask, valid=v_b, labels=None if labels[0] is None else labels[0][:na])
            for k in (1, 2, 3, 4):
                iso_ms[k] = segs[0].last_ms(k)
            same = torch.equal(cb[okv], want_c[okv]) and np.array_equal(ms_b, ms0) and np.array_equal(v_b, v0)
            same = same and (not labels_on or torch.equal(labels[0][:ne], want_l))
            same = same and all(np.array_equal(getattr(t_, f), ref[f], equal_nan=True) for f in ref)
            checked = checked and bool(same)
        del want_c, want_l
        if not checked:
            raise SystemExit("bench: the contexts' results differ from the single-context run")

    frames = args.steps * (total_frames if strong else total_frames * world)
    fps = frames / elapsed
    ncalls = max(stats["calls"], 1)
    fr = max(stats["frames"], 1)
    rows_pf, crop_pf = stats["rows"] / fr, stats["crop_bytes"] / fr
    obj_bytes = 3 * crop_pf + 44 * 8 * rows_pf               # three crop planes + 44 x 8 B per row, per frame
    frames_per_call = fr / ncalls
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath))                  # PMC bytes of a 200-frame launch (profiles/README.md)
        except Exception:
            traffic = {}
    groups = {}
    for k, name in GROUP_NAMES.items():
        ms = group_ms[k] / ncalls                            # average launch (group) duration over the timed region
        b = group_bytes_per_frame(name, rows_pf, crop_pf, labels_on) * frames_per_call
        ach = b / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        tr = traffic.get(name) if args.workload == "pisco" else None
        groups[name] = {"avg_ms": round(ms, 4), "avg_ms_alone": round(iso_ms[k], 4), "algorithmic_bytes": int(b),
                        "achieved_GBps": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4),
                        "counter_bytes": None if tr is None else int(tr * frames_per_call / BATCH),
                        "counter_over_algorithmic": None if (tr is None or b == 0) else round(tr * frames_per_call / BATCH / b, 3)}
    # the dominant kernel group: among the groups that own their time (with deferred fills the labelling chain runs beside the
    # previous call's label fill and the crop tail beside the next background kernel: their avg_ms is time spent beside a
    # streaming kernel, not device time of their own)
    # ... and with several contexts in flight a group's time in the region includes what it waited for in its hardware queue (the
    # crop tail of a frame-sharded run: 1.4 ms in the region, 0.28 ms alone).  The group is chosen by its time with the GPU to
    # itself, measured after the timed region; the figures reported for it are those of the timed region.
    n_deferred = sum(s_.tuning_state()["fills_deferred"] for s_ in segs)
    beside = {"table_beside_label_map"} | ({"sparse_labelling"} if n_deferred > 0 else set())
    own = [g for g in groups if g not in beside]
    dom_name = max(own, key=lambda g: groups[g]["avg_ms_alone"]) if any(groups[g]["avg_ms_alone"] > 0 for g in own) else \
        max(own, key=lambda g: groups[g]["avg_ms"])
    dom = groups[dom_name]
    path_bytes = PATH_BYTES_PER_FRAME if labels_on else 2 * P
    halo = (loaded - emitted) / max(emitted, 1)
    if strong:
        wl = (f"configs[2]: {total_frames}-frame {args.workload} stack frame-sharded over {world} GPUs (plan_ranges: "
              f"{emitted} frames + {loaded - emitted} halo frames on rank 0), RCCL gather of object tables")
    else:
        idx = 1 if args.workload == "pisco" else 3
        wl = (f"configs[{idx}]: {total_frames}-frame {args.workload} stack{' per GPU' if world > 1 else ''}, batches of {BATCH}")
        if args.deconv_standin:
            wl += ", two-call path (pseg_bgcorrect -> elementwise stand-in for the deconvolution network -> pseg_detect)"
    out = {
        "metric": "frames/sec (2560x2560 uint8), hot path bg_correction+detection",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 3), "higher_is_better": True,
        "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": wl + f", bg window {2 * args.n_bg_imgs}, ~{cfg['per_frame']} objects/frame, label map {'on' if labels_on else 'off'}",
                   "frames_per_step": total_frames if strong else total_frames * world,
                   "library_calls_per_step_rank0": len(calls), "objects_per_frame": round(stats["contours"] / fr, 1),
                   "rows_per_frame": round(rows_pf, 1), "saved_per_frame": round(stats["saved"] / fr, 1)},
        "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": dom["frac"], "traffic": dom["counter_bytes"],
                     "traffic_source": None if dom["counter_bytes"] is None else "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                                       "of this code kept from the build session (profiles/README.md), scaled to this launch's frames; "
                                       "not collected in this run", "avg_launch_ms": dom["avg_ms"],
                     # the same kernel group with the GPU to itself (one call after the timed region): in the pipeline the label fill
                     # shares the device with the next call's labelling chain, the background kernel with the previous call's crops
                     "avg_launch_ms_alone": dom["avg_ms_alone"],
                     "frac_alone": None if not dom["avg_ms_alone"] else round(dom["algorithmic_bytes"] / (dom["avg_ms_alone"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "bytes_per_launch": dom["algorithmic_bytes"], "frames_per_launch": round(frames_per_call, 1)},
        "roofline_groups": groups,
        "path": {"bytes_per_frame": path_bytes, "object_bytes_per_frame": round(obj_bytes, 1),
                 "frac_of_hbm_peak_per_gpu": round(fps / world * path_bytes / (HBM_PEAK_GBS * 1e9), 4),
                 "frac_of_hbm_peak_per_gpu_with_objects": round(fps / world * (path_bytes + obj_bytes) / (HBM_PEAK_GBS * 1e9), 4),
                 "contexts": ns, "driver": args.driver, "contexts_consistent_after_timed_region": checked,   # the contexts' results against a single-context run of the library
                                                                      # (consistency of what was timed; parity with the oracle is tests/)
                 # the band kernel's scan order of the contexts after the run (1: block index, 0: tickets) and how often one of them
                 # had to give the index order up (a timed-out scan: the batch is repeated inside the call)
                 "band_scan_order": [s_.tuning_state()["band_order"] for s_ in segs],
                 "band_scan_fallbacks": sum(s_.tuning_state()["band_order_fallbacks"] for s_ in segs),
                 # calls whose label fill waited for the next call's background kernel (the labelling chain then runs beside the
                 # previous call's fill: its avg_ms in roofline_groups is the time beside that fill, not device time of its own)
                 "fills_deferred": sum(s_.tuning_state()["fills_deferred"] for s_ in segs),
                 "calls_on_streaming_queue": sum(s_.tuning_state()["calls_on_streaming_queue"] for s_ in segs),
                 "deferred_fills_given_up": sum(s_.tuning_state()["deferred_fills_given_up"] for s_ in segs),
                 "ranks_seen": dist.get_world_size() if dist_on else 1, "launch": launch,
                 "halo_overhead": round(halo, 4) if strong else 0.0,
                 "table_gather_messages_per_step": n_gather if dist_on else None,
                 "table_gather_ms_per_step_rank0": round(stats["gather_s"] / max(args.steps, 1) * 1e3, 3) if dist_on else None},
    }

    if rank == 0 and not dist_on and args.host_batches > 0:
        # the same path with the frames starting in (page-locked) host memory and the table + crops ending there: PCIe-bound,
        # reported beside `value`, never as it
        nbh = args.host_batches
        pipe = HostPipeline(H, W, max_frames=BATCH, device=local_rank, depth=2, labels=False)
        try:
            for k in range(2):
                pipe.host[k].copy_(stack[k * BATCH:(k + 1) * BATCH])
            torch.cuda.synchronize()
            for _ in pipe.run(((BATCH, None) for _ in range(2)), mask=mask):
                pass
            h0 = time.perf_counter()
            got = 0
            for b, c_, ms_, v_, tab in pipe.run(((BATCH, None) for _ in range(nbh)), mask=mask):
                got += tab.n_frames
            ht = time.perf_counter() - h0
            out["path"]["host_input_fps"] = round(got / ht, 1)
            out["path"]["host_input_note"] = (f"{nbh} batches of {BATCH} frames from page-locked host memory: upload, fused path, table and three "
                                              "crop planes back into page-locked memory, two contexts, one caller thread (label map not "
                                              "materialised)")
        finally:
            pipe.close()

    if rank == 0 and not dist_on and args.cpu != "off" and args.cpu_frames >= 12:
        # the oracle is the timed CPU port here, never the product path.  BASELINE.json configs[0]: 100 frames, seed 1
        nf = args.cpu_frames
        objs, off = synth.make_objects(nf, H, W, 1, **synth.CONFIGS["pisco"])
        sample = seg.synth_frames(nf, 1, objs, off).cpu().numpy()
        hmask = mask.cpu().numpy()
        pb, pd = (int(v) for v in args.cpu_procs.split(","))
        extra = {}
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
            ct = r["seconds"]
            cores = pb + pd
            how = (f"reference topology: {pb} background-correction processes feeding {pd} detection processes through a queue of 10, "
                   f"both pools running at once ({ct:.2f} s; background pool done after {r['seconds_bg']:.2f} s), frames through /dev/shm")
            if args.cpu == "writes":
                import shutil
                import tempfile
                wd = tempfile.mkdtemp(prefix="pseg_cpu_out_")
                try:
                    rw = cpu_pool.run(sample, hmask, pb, pd, write_dir=wd)
                    extra = {"value_with_png_csv_writes": round(nf / rw["seconds"], 3)}
                finally:
                    shutil.rmtree(wd, ignore_errors=True)
        out["cpu_baseline"] = {"value": round(nf / ct, 3), "unit": "frames/s", "cores": cores, "kind": "port",
                               "sample": f"configs[0]: {nf} frames, seed 1, one batch (window 12), NumPy + C oracle, {how}, "
                                         f"no PNG/CSV writes, host has {os.cpu_count()} logical CPUs", **extra}
    if dist_on:
        gq.put(None)
        gth.join()
        dist.barrier()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    os.dup2(2, 1)
    for sg in segs:
        sg.close()
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
