"""ctypes binding of libpiscoseg.so (include/piscoseg.h).  Fails loudly when the library is
missing: there is no CPU fallback in the product path."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSEG_LIB") or os.path.join(HERE, "libpiscoseg.so")   # PSEG_LIB: another build of the same library
CSRC = os.path.join(HERE, "csrc")

NPROPS = 32
NSUMS = 24

SYMBOLS = [
    "pseg_create", "pseg_destroy", "pseg_last_error", "pseg_set_stream", "pseg_bgcorrect", "pseg_detect",
    "pseg_process", "pseg_process_part", "pseg_process_batches", "pseg_submit", "pseg_wait", "pseg_fetch_crops", "pseg_regionprops", "pseg_regionprops_noalloc", "pseg_vit_tensors",
    "pseg_disk_mask", "pseg_resize_frames", "pseg_synth_frames", "pseg_last_ms", "pseg_tuning_state",
]


class Table(C.Structure):
    _fields_ = [
        ("n_frames", C.c_int32), ("n_rows", C.c_int32), ("n_saved", C.c_int32), ("crop_bytes", C.c_int64),
        ("frame_row_offset", C.POINTER(C.c_int32)), ("frame_n_contours", C.POINTER(C.c_int32)),
        ("frame_valid", C.POINTER(C.c_uint8)), ("frame_mean", C.POINTER(C.c_double)), ("frame_std", C.POINTER(C.c_double)),
        ("row_frame", C.POINTER(C.c_int32)), ("row_c", C.POINTER(C.c_int32)), ("row_label", C.POINTER(C.c_int32)),
        ("row_area", C.POINTER(C.c_double)),
        ("row_x", C.POINTER(C.c_int32)), ("row_y", C.POINTER(C.c_int32)), ("row_w", C.POINTER(C.c_int32)), ("row_h", C.POINTER(C.c_int32)),
        ("row_saved", C.POINTER(C.c_uint8)),
        ("saved_row", C.POINTER(C.c_int32)), ("crop_offset", C.POINTER(C.c_int64)),
        ("saved_props", C.POINTER(C.c_double)), ("sums", C.POINTER(C.c_int64)),
        ("crops_raw_dev", C.c_void_p), ("crops_masked_dev", C.c_void_p), ("crops_mask_dev", C.c_void_p),
    ]


def build(force=False):
    """Compile the HIP library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    import subprocess
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))]
    hdr = os.path.join(HERE, "..", "include", "piscoseg.h")
    newest = max(os.path.getmtime(p) for p in srcs + [hdr])
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= newest:
        return LIB_PATH
    subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
                           "-o", LIB_PATH, os.path.join(CSRC, "piscoseg.hip")])
    return LIB_PATH


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    try:
        # torch ships its own libamdhip64.so.7 / libhsa-runtime64; a process holds one HIP runtime per SONAME, and torch finds
        # no GPU when the system copy was mapped first (this library links against it).  Tensors are this binding's device
        # memory, so torch goes first and the library binds to the runtime torch brought.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    L.pseg_create.argtypes = [i32, i32, i32, i32, C.POINTER(vp)]
    L.pseg_destroy.argtypes = [vp]
    L.pseg_last_error.argtypes = [vp]
    L.pseg_last_error.restype = C.c_char_p
    L.pseg_set_stream.argtypes = [vp, vp]
    L.pseg_bgcorrect.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    L.pseg_detect.argtypes = [vp, vp, vp, vp, vp, i32, i32, dbl, dbl, vp, C.POINTER(Table)]
    L.pseg_process.argtypes = [vp, vp, i32, i32, vp, i32, dbl, dbl, vp, vp, vp, vp, C.POINTER(Table)]
    L.pseg_process_part.argtypes = [vp, vp, i32, i32, i32, vp, i32, dbl, dbl, vp, vp, vp, vp, C.POINTER(Table)]
    L.pseg_process_batches.argtypes = [vp, vp, i32, i32, i32, vp, i32, dbl, dbl, vp, vp, vp, vp, C.POINTER(Table)]
    L.pseg_submit.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, dbl, dbl, vp, vp, vp, vp]
    L.pseg_wait.argtypes = [vp, C.POINTER(Table)]
    L.pseg_fetch_crops.argtypes = [vp, vp, vp, vp]
    L.pseg_regionprops.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.pseg_regionprops_noalloc.argtypes = [vp, vp, vp, i32, i32, i32, vp, vp]
    L.pseg_vit_tensors.argtypes = [vp, i32, i32, i32, vp, vp]
    L.pseg_disk_mask.argtypes = [vp, i32, vp]
    L.pseg_resize_frames.argtypes = [vp, vp, i32, i32, i32, vp]
    L.pseg_synth_frames.argtypes = [vp, i32, i32, C.c_uint64, vp, vp, vp]
    L.pseg_last_ms.argtypes = [vp, i32]
    L.pseg_last_ms.restype = C.c_float
    L.pseg_tuning_state.argtypes = [vp, i32]
    L.pseg_tuning_state.restype = C.c_longlong
    for f in ("pseg_create", "pseg_destroy", "pseg_set_stream", "pseg_bgcorrect", "pseg_detect", "pseg_process", "pseg_process_part", "pseg_process_batches", "pseg_submit", "pseg_wait",
              "pseg_fetch_crops", "pseg_regionprops", "pseg_regionprops_noalloc", "pseg_vit_tensors", "pseg_disk_mask", "pseg_resize_frames", "pseg_synth_frames"):
        getattr(L, f).restype = i32
    _lib = L
    return L
