"""Does the physical layout of the label map decide the label fill's time?  One context, one 200-frame batch of the bench
workload, the fill's time (the library's events, GPU to itself) over a few calls, with the label map (and optionally the cleaned
frames) allocated by torch (hipMalloc) or with hipExtMallocWithFlags(hipDeviceMallocContiguous):
    contig_probe.py torch|contig [labels|all]
Prints one line: mode, median / min / max of the fill, of the background kernel."""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), "..", ".."))
from imageprocessing_amd import synth                      # noqa: E402
from imageprocessing_amd.engine import Segmenter           # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "torch"
what = sys.argv[2] if len(sys.argv) > 2 else "labels"
H = W = 2560
N = 200
hip = C.CDLL("libamdhip64.so.7")
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipFree.argtypes = [C.c_void_p]


class MemLocation(C.Structure):
    _fields_ = [("type", C.c_int), ("id", C.c_int)]


class AllocProp(C.Structure):
    _fields_ = [("type", C.c_int), ("handle_type", C.c_int), ("location", MemLocation), ("win32", C.c_void_p),
                ("compression", C.c_ubyte), ("rdma", C.c_ubyte), ("usage", C.c_ushort)]


class AccessDesc(C.Structure):
    _fields_ = [("location", MemLocation), ("flags", C.c_int)]


hip.hipMemAddressReserve.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
hip.hipMemCreate.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.POINTER(AllocProp), C.c_ulonglong]
hip.hipMemMap.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, C.c_ulonglong]
hip.hipMemSetAccess.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(AccessDesc), C.c_size_t]
hip.hipMemGetAllocationGranularity.argtypes = [C.POINTER(C.c_size_t), C.POINTER(AllocProp), C.c_int]


def chk(rc, what_):
    if rc != 0:
        raise SystemExit(f"{what_} -> {rc}")


class Raw:
    """modes: plain / contig / uncached / fine (hipExtMallocWithFlags), vmm<MiB> / vmmshuf<MiB> (one virtual range backed by physical chunks
    of that size, created in address order or in a shuffled order)"""
    def __init__(self, shape, typestr, itemsize):
        self.p = C.c_void_p()
        nbytes = int(np.prod(shape)) * itemsize
        if mode.startswith("vmm"):
            shuf = mode.startswith("vmmshuf")
            chunk = int(mode[7 if shuf else 3:]) << 20
            prop = AllocProp(1, 0, MemLocation(1, 0), None, 0, 0, 0)
            g = C.c_size_t()
            chk(hip.hipMemGetAllocationGranularity(C.byref(g), C.byref(prop), 1), "granularity")
            chunk = max(chunk, g.value) // g.value * g.value
            total = (nbytes + chunk - 1) // chunk * chunk
            chk(hip.hipMemAddressReserve(C.byref(self.p), total, 0, None, 0), "reserve")
            n = total // chunk
            handles = []
            for _ in range(n):
                h = C.c_void_p()
                chk(hip.hipMemCreate(C.byref(h), chunk, C.byref(prop), 0), "create")
                handles.append(h)
            order = list(range(n))
            if shuf:
                np.random.default_rng(7).shuffle(order)
            for i, j in enumerate(order):
                chk(hip.hipMemMap(C.c_void_p(self.p.value + i * chunk), chunk, 0, handles[j], 0), "map")
            desc = AccessDesc(MemLocation(1, 0), 3)
            chk(hip.hipMemSetAccess(self.p, total, C.byref(desc), 1), "access")
            self.note = f"{n} chunks of {chunk >> 20} MiB (granularity {g.value >> 10} KiB)"
        else:
            flag = {"contig": 0x4, "uncached": 0x3, "fine": 0x1, "plain": 0x0}[mode]
            chk(hip.hipExtMallocWithFlags(C.byref(self.p), nbytes, flag), f"hipExtMallocWithFlags({mode}, {nbytes})")
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (self.p.value, False), "version": 2}


def alloc(shape, dtype):
    if mode == "torch":
        return torch.empty(shape, dtype=dtype, device="cuda"), None
    raw = Raw(shape, "<i4" if dtype == torch.int32 else "|u1", 4 if dtype == torch.int32 else 1)
    return torch.as_tensor(raw, device="cuda"), raw


torch.cuda.set_device(0)
seg = Segmenter(H, W, max_frames=N, device=0)
mask = seg.disk_mask(1200)
cfg = synth.CONFIGS["pisco"]
objs, off = synth.make_objects(N, H, W, 2, frame0=0, **cfg)
if what == "all":
    frames, keep_f = alloc((N, H, W), torch.uint8)
    cleaned, keep_c = alloc((N, H, W), torch.uint8)
else:
    frames = torch.empty((N, H, W), dtype=torch.uint8, device="cuda")
    cleaned = torch.empty((N, H, W), dtype=torch.uint8, device="cuda")
seg.synth_frames(N, 2, objs, off, frame0=0, out=frames)
labels, keep_l = alloc((N, H, W), torch.int32)
fill, bg = [], []
for i in range(8):
    seg.process(frames, mask=mask, cleaned=cleaned, labels=labels)
    if i >= 2:
        fill.append(seg.last_ms(3))
        bg.append(seg.last_ms(0))
print(f"{mode:6s} {what:6s} fill median {np.median(fill):.3f} min {min(fill):.3f} max {max(fill):.3f} ms   bg median {np.median(bg):.3f} min {min(bg):.3f} ms"
      f"   labels at {labels.data_ptr():#x}", flush=True)
seg.close()
