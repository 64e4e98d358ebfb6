"""cv2-shaped front end of the oracle -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Each function restates one OpenCV call the reference makes on the hot path; the call sites
are cited per function.  PARITY UNPINNED for these (cv2 absent, no reference fixtures).
"""
import ctypes as C
import numpy as np

from .build import build

RETR_EXTERNAL = 0
CHAIN_APPROX_SIMPLE = 2
THRESH_BINARY = 0
FILLED = -1

_lib = None


class _Contours(C.Structure):
    _fields_ = [("n", C.c_int32), ("offsets", C.POINTER(C.c_int32)), ("pts", C.POINTER(C.c_int32))]


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_find_contours_external.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_Contours)]
        L.orc_find_contours_external.restype = C.c_int
        L.orc_free_contours.argtypes = [C.POINTER(_Contours)]
        L.orc_contour_area.argtypes = [C.c_void_p, C.c_int]
        L.orc_contour_area.restype = C.c_double
        L.orc_bounding_rect.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_draw_contour_filled.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_uint8]
        L.orc_circle_filled.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint8]
        L.orc_label_map.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(_Contours), C.c_void_p]
        L.orc_label_map.restype = C.c_int
        _lib = L
    return _lib


def absdiff(a, b):
    """cv.absdiff on uint8 (bg_correction.py:80)."""
    return np.where(a > b, a - b, b - a).astype(np.uint8)


def bitwise_not(a):
    """cv.bitwise_not on uint8 (bg_correction.py:81, detection.py:178)."""
    return (255 - a).astype(np.uint8)


def threshold(src, thresh, maxval, type_):
    """cv.threshold(src, thresh, maxval, THRESH_BINARY): strict '>' (detection.py:195-201)."""
    assert type_ == THRESH_BINARY
    return float(thresh), np.where(src > thresh, maxval, 0).astype(src.dtype)


def findContours(img, mode, method):
    """cv.findContours(thresh, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) (detection.py:204).
    Returns (list of int32 [n,1,2] arrays in OpenCV order, None)."""
    assert mode == RETR_EXTERNAL and method == CHAIN_APPROX_SIMPLE
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    c = _Contours()
    if lib().orc_find_contours_external(img.ctypes.data, H, W, C.byref(c)) != 0:
        raise MemoryError("oracle findContours")
    n = c.n
    offs = np.ctypeslib.as_array(c.offsets, shape=(n + 1,)).copy()
    total = int(offs[n])
    pts = np.ctypeslib.as_array(c.pts, shape=(max(total, 1) * 2,)).copy()[: total * 2].reshape(-1, 2)
    lib().orc_free_contours(C.byref(c))
    return [pts[offs[k]:offs[k + 1]].reshape(-1, 1, 2).copy() for k in range(n)], None


def contourArea(cnt):
    """cv.contourArea (detection.py:208)."""
    p = np.ascontiguousarray(cnt, dtype=np.int32).reshape(-1, 2)
    return float(lib().orc_contour_area(p.ctypes.data, p.shape[0]))


def boundingRect(cnt):
    """cv.boundingRect (detection.py:212) -> (x, y, w, h)."""
    p = np.ascontiguousarray(cnt, dtype=np.int32).reshape(-1, 2)
    out = np.zeros(4, np.int32)
    lib().orc_bounding_rect(p.ctypes.data, p.shape[0], out.ctypes.data)
    return tuple(int(v) for v in out)


def drawContours(mask, cnts, idx, color, thickness=FILLED):
    """cv.drawContours(mask, [cnt], -1, 255, thickness=cv.FILLED) (detection.py:245)."""
    assert thickness == FILLED and mask.dtype == np.uint8 and mask.flags.c_contiguous
    val = int(color[0] if isinstance(color, (tuple, list)) else color)
    H, W = mask.shape
    for k, cnt in enumerate(cnts):
        if idx >= 0 and k != idx:
            continue
        p = np.ascontiguousarray(cnt, dtype=np.int32).reshape(-1, 2)
        lib().orc_draw_contour_filled(mask.ctypes.data, H, W, p.ctypes.data, p.shape[0], val)
    return mask


def circle(img, center, radius, color, thickness):
    """cv.circle(mask, (cx, cy), r, 255, -1) (detection.py:320-326)."""
    assert thickness == -1 and img.dtype == np.uint8 and img.flags.c_contiguous
    val = int(color[0] if isinstance(color, (tuple, list)) else color)
    H, W = img.shape
    lib().orc_circle_filled(img.ctypes.data, H, W, int(center[0]), int(center[1]), int(radius), val)
    return img


def label_map(thresh, cnts):
    """Label map of SURVEY.md section 8(a): k on the component traced by cnts[k-1], 0 elsewhere."""
    thresh = np.ascontiguousarray(thresh, dtype=np.uint8)
    H, W = thresh.shape
    n = len(cnts)
    offs = np.zeros(n + 1, np.int32)
    for k, cnt in enumerate(cnts):
        offs[k + 1] = offs[k] + cnt.reshape(-1, 2).shape[0]
    pts = (np.concatenate([c.reshape(-1, 2) for c in cnts]).astype(np.int32) if n else np.zeros((1, 2), np.int32))
    pts = np.ascontiguousarray(pts)
    c = _Contours(n, offs.ctypes.data_as(C.POINTER(C.c_int32)), pts.ctypes.data_as(C.POINTER(C.c_int32)))
    labels = np.zeros((H, W), np.int32)
    if lib().orc_label_map(thresh.ctypes.data, H, W, C.byref(c), labels.ctypes.data) != 0:
        raise MemoryError("oracle label_map")
    return labels
