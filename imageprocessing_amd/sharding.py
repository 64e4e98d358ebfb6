"""Multi-GPU layout of the hot path: one process per GPU, frames sharded, no pixel traffic
between GPUs, one small gather of the object tables.

The reference cuts a folder into independent batches of 200 frames (segmenter.py:209-213) and a
frame's background window never leaves its batch (bg_correction.py:72), so batches are the unit of
independence.  ``batches_of`` restates the reference's partition; ``assign_batches`` deals them to
ranks; ``gather_tables`` is the only collective (RCCL when the tensors live on the GPU, gloo on CPU).
"""
import io

import numpy as np


def batches_of(n_files, batch_size=200):
    """(start, length) of every batch for n_files frames -- segmenter.py:209-213: all batches have
    batch_size frames except the last one, which takes the remainder (batch_size..2*batch_size-1)."""
    nb = n_files // batch_size - 1
    out = [(i * batch_size, batch_size) for i in range(max(nb, 0))]
    start = batch_size * nb if nb > 0 else 0
    if nb < 0:                       # fewer than batch_size files: Python's negative slice keeps them all
        start = 0
    out.append((start, n_files - start))
    return out


def assign_batches(batches, world_size):
    """Contiguous blocks of batches per rank, sizes differing by at most one."""
    nb = len(batches)
    q, r = divmod(nb, world_size)
    out, a = [], 0
    for k in range(world_size):
        m = q + (1 if k < r else 0)
        out.append(batches[a:a + m])
        a += m
    return out


def plan_ranges(n_files, world_size, batch_size=200, window=12):
    """Contiguous frame ranges per rank: the stack is cut at k * n_files / world_size, every cut moved so
    that no part begins inside the last ``window`` frames of a reference batch (those frames share one background window,
    bg_correction.py:72, and stay with the part that ends the batch).  Returns, per rank, a list of parts
    ``dict(start, n_emit, n_avail)``: global index of the part's first frame, frames to process, frames to load
    (n_emit plus a forward halo of window - 1 frames unless the part ends its batch) -- the arguments of
    ``pseg_process_part`` / ``Segmenter.process(..., n_emit=)``.  No pixel crosses GPUs."""
    batches = batches_of(n_files, batch_size)

    def snap(cut):
        for bs, bl in batches:
            if bs < cut < bs + bl:
                r = cut - bs
                if r > bl - window:                      # inside the tail: move to whichever end of the tail is nearer
                    r = bl - window if (r - (bl - window)) <= (bl - r) and bl - window > 0 else bl
                return bs + r
        return cut

    cuts = [0] + [snap(k * n_files // world_size) for k in range(1, world_size)] + [n_files]
    cuts = [max(c, p) for c, p in zip(cuts, [0] + cuts[:-1])]          # keep them monotone
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    out = []
    for k in range(world_size):
        g0, g1 = cuts[k], cuts[k + 1]
        parts = []
        for bs, bl in batches:
            lo, hi = max(g0, bs), min(g1, bs + bl)
            if lo >= hi:
                continue
            ends_batch = hi == bs + bl
            parts.append(dict(start=lo, n_emit=hi - lo, n_avail=(hi - lo) if ends_batch else (hi - lo) + window - 1))
        out.append(parts)
    return out


TABLE_FIELDS = ["frame_row_offset", "frame_n_contours", "frame_valid", "frame_mean", "frame_std", "frame", "c", "label",
                "area", "x", "y", "w", "h", "saved", "saved_props", "saved_row", "crop_offset", "sums"]


def pack_table(tab, frame_base=0):
    """Object table -> one uint8 blob (npz container); frame indices are made global."""
    buf = io.BytesIO()
    d = {k: getattr(tab, k) for k in TABLE_FIELDS}
    d["frame"] = d["frame"] + np.int32(frame_base)
    d["meta"] = np.asarray([tab.n_frames, frame_base, tab.crop_bytes], np.int64)
    np.savez(buf, **d)
    return np.frombuffer(buf.getvalue(), np.uint8).copy()


def unpack_table(blob):
    d = np.load(io.BytesIO(blob.tobytes()), allow_pickle=False)
    return {k: d[k] for k in d.files}


def merge_tables(parts):
    """Concatenate per-shard tables (dicts from unpack_table) in global frame order.  The result is
    independent of how the frames were sharded (rows are ordered (frame, c))."""
    parts = sorted(parts, key=lambda p: int(p["meta"][1]))
    out = {}
    for k in ("frame_n_contours", "frame_valid", "frame_mean", "frame_std", "frame", "c", "label", "area", "x", "y", "w",
              "h", "saved", "saved_props", "sums"):
        out[k] = np.concatenate([p[k] for p in parts])
    rows = np.cumsum([0] + [len(p["frame"]) for p in parts])
    crops = np.cumsum([0] + [int(p["meta"][2]) for p in parts])
    out["frame_row_offset"] = np.concatenate([p["frame_row_offset"][:-1] + rows[i] for i, p in enumerate(parts)] + [rows[-1:]])
    out["saved_row"] = np.concatenate([p["saved_row"] + rows[i] for i, p in enumerate(parts)])
    out["crop_offset"] = np.concatenate([p["crop_offset"][:-1] + crops[i] for i, p in enumerate(parts)] + [crops[-1:]])
    out["shard_of_saved"] = np.concatenate([np.full(len(p["saved_row"]), i, np.int32) for i, p in enumerate(parts)])
    return out


def gather_tables(blob, device=None, dst=0):
    """Variable-length gather of the packed tables to rank ``dst``: an all-gather of the byte counts,
    then a gather of the padded blobs.  Returns the list of blobs on dst, None elsewhere.
    With the nccl backend (RCCL over xGMI) the tensors are staged on ``device``."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else "cpu"
    n = torch.tensor([blob.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    send = torch.zeros(cap, dtype=torch.uint8, device=dev)
    send[:blob.size] = torch.from_numpy(blob).to(dev)
    recv = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank != dst:
        return None
    return [r[:s].cpu().numpy() for r, s in zip(recv, sizes)]
