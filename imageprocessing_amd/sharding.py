"""Multi-GPU layout of the hot path: one process per GPU, frames sharded, no pixel traffic
between GPUs, one small gather of the object tables.

The reference cuts a folder into independent batches of 200 frames (segmenter.py:209-213) and a
frame's background window never leaves its batch (bg_correction.py:72), so batches are the unit of
independence.  ``batches_of`` restates the reference's partition; ``assign_batches`` deals them to
ranks; ``gather_tables`` is the only collective (RCCL when the tensors live on the GPU, gloo on CPU).
"""
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


# (name, dtype, shape as a function of n_frames F, rows R, saved S) in wire order
_WIRE = [
    ("frame_row_offset", np.int32, lambda F, R, S: (F + 1,)), ("frame_n_contours", np.int32, lambda F, R, S: (F,)),
    ("frame_valid", np.uint8, lambda F, R, S: (F,)), ("frame_mean", np.float64, lambda F, R, S: (F,)),
    ("frame_std", np.float64, lambda F, R, S: (F,)),
    ("frame", np.int32, lambda F, R, S: (R,)), ("c", np.int32, lambda F, R, S: (R,)), ("label", np.int32, lambda F, R, S: (R,)),
    ("area", np.float64, lambda F, R, S: (R,)), ("x", np.int32, lambda F, R, S: (R,)), ("y", np.int32, lambda F, R, S: (R,)),
    ("w", np.int32, lambda F, R, S: (R,)), ("h", np.int32, lambda F, R, S: (R,)), ("saved", np.uint8, lambda F, R, S: (R,)),
    ("saved_props", np.float64, lambda F, R, S: (S, 32)), ("saved_row", np.int32, lambda F, R, S: (S,)),
    ("crop_offset", np.int64, lambda F, R, S: (S + 1,)),
]                                                  # (the integer sums behind the properties stay on the producing rank)
TABLE_FIELDS = [w[0] for w in _WIRE]
_MAGIC = 0x50534547544142                          # "PSEGTAB"


def packed_bytes(tab):
    """bytes pack_table needs for this table"""
    F, R, S = int(tab.n_frames), len(tab.frame), len(tab.saved_row)
    return 48 + sum(int(np.prod(shape(F, R, S))) * np.dtype(dt).itemsize for _, dt, shape in _WIRE)


def pack_table(tab, frame_base=0, out=None):
    """Object table -> one uint8 blob: a 6-word int64 header, then the arrays back to back in _WIRE order (plain
    memcpy -- this runs once per batch next to the kernels).  Frame indices are made global.  ``out``: a uint8 buffer
    to write into (fresh pages cost more than the copy); the returned array is a view of it, and a buffer that is too
    small raises ValueError -- a caller that keeps its own fill count must never be handed a fresh array silently."""
    F, R, S = int(tab.n_frames), len(tab.frame), len(tab.saved_row)
    head = np.asarray([_MAGIC, F, frame_base, tab.crop_bytes, R, S], np.int64)
    sizes = [int(np.prod(shape(F, R, S))) * np.dtype(dt).itemsize for _, dt, shape in _WIRE]
    total = 48 + sum(sizes)
    if out is None:
        out = np.empty(total, np.uint8)
    elif out.size < total:
        raise ValueError(f"pack_table: the buffer holds {out.size} bytes, the table needs {total}")
    else:
        out = out[:total]
    out[:48] = head.view(np.uint8)
    o = 48
    for (name, dt, shape), nb in zip(_WIRE, sizes):
        a = getattr(tab, name)
        assert a.dtype == dt and a.shape == shape(F, R, S), (name, a.dtype, a.shape)
        dst = out[o:o + nb].view(dt).reshape(a.shape)
        if name == "frame":
            np.add(a, np.int32(frame_base), out=dst)
        else:
            dst[...] = a
        o += nb
    return out


def packed_size(blob):
    """bytes of the packed table that starts at blob[0] (several tables may be sent back to back in one message)"""
    head = np.ascontiguousarray(blob[:48]).view(np.int64)
    assert head[0] == _MAGIC, "not a packed object table"
    F, R, S = int(head[1]), int(head[4]), int(head[5])
    return 48 + sum(int(np.prod(shape(F, R, S))) * np.dtype(dt).itemsize for _, dt, shape in _WIRE)


def split_tables(buf):
    """the packed tables lying back to back in buf"""
    out, o = [], 0
    while o < buf.size:
        n = packed_size(buf[o:])
        out.append(buf[o:o + n])
        o += n
    return out


def unpack_table(blob):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    head = blob[:48].view(np.int64)
    assert head[0] == _MAGIC, "not a packed object table"
    F, R, S = int(head[1]), int(head[4]), int(head[5])
    out = {"meta": np.asarray([F, head[2], head[3]], np.int64)}
    o = 48
    for name, dt, shape in _WIRE:
        shp = shape(F, R, S)
        nb = int(np.prod(shp)) * np.dtype(dt).itemsize
        out[name] = blob[o:o + nb].view(dt).reshape(shp).copy()
        o += nb
    assert o == blob.size
    return out


def merge_tables(parts):
    """Concatenate per-shard tables (dicts from unpack_table) in global frame order.  The result is
    independent of how the frames were sharded (rows are ordered (frame, c))."""
    parts = sorted(parts, key=lambda p: int(p["meta"][1]))
    out = {}
    for k in ("frame_n_contours", "frame_valid", "frame_mean", "frame_std", "frame", "c", "label", "area", "x", "y", "w",
              "h", "saved", "saved_props"):
        out[k] = np.concatenate([p[k] for p in parts])
    rows = np.cumsum([0] + [len(p["frame"]) for p in parts])
    crops = np.cumsum([0] + [int(p["meta"][2]) for p in parts])
    out["frame_row_offset"] = np.concatenate([p["frame_row_offset"][:-1] + rows[i] for i, p in enumerate(parts)] + [rows[-1:]])
    out["saved_row"] = np.concatenate([p["saved_row"] + rows[i] for i, p in enumerate(parts)])
    out["crop_offset"] = np.concatenate([p["crop_offset"][:-1] + crops[i] for i, p in enumerate(parts)] + [crops[-1:]])
    out["shard_of_saved"] = np.concatenate([np.full(len(p["saved_row"]), i, np.int32) for i, p in enumerate(parts)])
    return out


class TableGatherer:
    """Variable-length gather of the packed tables to rank ``dst``: an all-gather of the byte counts, then a gather of
    the padded blobs -- the only collective of the path.  With the nccl backend (RCCL over xGMI) the blobs are staged on
    ``device``; the buffers (device send / receive, page-locked host mirrors) are kept between calls and grow together on
    all ranks (every rank sees the same byte counts).  ``gather`` returns the list of blobs on dst (views of a host
    buffer that the next call overwrites), None elsewhere."""

    def __init__(self, device=None, dst=0):
        import torch.distributed as dist
        self.dev = device if device is not None else "cpu"
        self.dst, self.world, self.rank = dst, dist.get_world_size(), dist.get_rank()
        self.cap = 0
        self.on_gpu = str(self.dev) != "cpu"

    def _grow(self, cap):
        import torch
        cap = (cap * 5 // 4 + 4095) & ~4095
        self.send = torch.zeros(cap, dtype=torch.uint8, device=self.dev)
        self.hsend = torch.zeros(cap, dtype=torch.uint8, pin_memory=self.on_gpu)
        if self.rank == self.dst:
            self.recv = torch.zeros((self.world, cap), dtype=torch.uint8, device=self.dev)
            self.hrecv = torch.zeros((self.world, cap), dtype=torch.uint8, pin_memory=self.on_gpu) if self.on_gpu else self.recv
        self.cap = cap

    def gather(self, blob):
        import torch
        import torch.distributed as dist
        n = torch.tensor([blob.size], dtype=torch.int64, device=self.dev)
        sizes = torch.zeros(self.world, dtype=torch.int64, device=self.dev)
        dist.all_gather(list(sizes.split(1)), n)
        sizes = sizes.cpu().tolist()
        need = max(sizes)
        if need > self.cap:
            self._grow(need)
        self.hsend[:blob.size].numpy()[...] = blob
        self.send[:need].copy_(self.hsend[:need], non_blocking=True)
        recv = [self.recv[r, :need] for r in range(self.world)] if self.rank == self.dst else None
        dist.gather(self.send[:need], recv, dst=self.dst)
        if self.rank != self.dst:
            return None
        if self.on_gpu:
            self.hrecv[:, :need].copy_(self.recv[:, :need], non_blocking=True)
            torch.cuda.current_stream(self.recv.device).synchronize()
        return [self.hrecv[r, :s].numpy() for r, s in enumerate(sizes)]


def gather_tables(blob, device=None, dst=0):
    """One-off form of TableGatherer.gather (the blobs are copied out of the gatherer's buffer)."""
    got = TableGatherer(device, dst).gather(blob)
    return None if got is None else [g.copy() for g in got]
