"""pseg_regionprops_noalloc (include/piscoseg.h): the region-property entry point for callers that cannot allocate.
Same kernels as pseg_regionprops with the scratch in a static device array; its output must equal that entry point's
bit for bit (integer sums and the float64 columns computed from them), also when a smaller window follows a larger one
in the same array."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _window(rng, h, w):
    """a blob with a hole plus a detached piece (several components, empty rows possible), spanning the window"""
    yy, xx = np.mgrid[0:h, 0:w]
    m = ((yy - h / 2) / (h / 2)) ** 2 + ((xx - w / 2) / (w / 2)) ** 2 <= 1.0
    m &= ~(((yy - h / 2) / (h / 5 + 1)) ** 2 + ((xx - w / 3) / (w / 6 + 1)) ** 2 <= 1.0)
    m[0, :] |= rng.random(w) < 0.3
    m[:, 0] |= rng.random(h) < 0.3
    m[h - 1, w - 1] = True
    m[0, 0] = True
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    return (m * 255).astype(np.uint8), img


def _both(seg, mask, img):
    import torch
    L = seg.L
    m = torch.from_numpy(mask).cuda()
    g = torch.from_numpy(img).cuda()
    torch.cuda.synchronize()
    h, w = mask.shape
    out = []
    for fn in (L.pseg_regionprops, L.pseg_regionprops_noalloc):
        props = np.full(32, -7.0)
        sums = np.full(24, -7, np.int64)
        rc = fn(seg.h, m.data_ptr(), g.data_ptr(), h, w, w, props.ctypes.data, sums.ctypes.data)
        assert rc == 0, L.pseg_last_error(seg.h)
        out.append((props, sums))
    return out


def test_noalloc_equals_regionprops_two_calls_in_a_row():
    from imageprocessing_amd.engine import Segmenter
    seg = Segmenter(256, 256, max_frames=2, device=0)
    rng = np.random.default_rng(5)
    try:
        for h, w in ((500, 520), (37, 90)):                 # the second, smaller window reuses the array the first one filled
            mask, img = _window(rng, h, w)
            (p0, s0), (p1, s1) = _both(seg, mask, img)
            assert np.array_equal(s0, s1), (h, w)
            assert np.array_equal(p0, p1, equal_nan=True), (h, w)
            assert s1[0] == int((mask == 255).sum()) and p1[0] == w and p1[1] == h
    finally:
        seg.close()


def test_noalloc_limits_and_second_context():
    import torch
    from imageprocessing_amd.engine import Segmenter
    a = Segmenter(256, 256, max_frames=2, device=0)
    b = Segmenter(64, 64, max_frames=2, device=0)            # another context on the same GPU shares the array
    try:
        rng = np.random.default_rng(6)
        big = torch.zeros((1025, 64), dtype=torch.uint8, device="cuda")
        props = np.zeros(32)
        assert a.L.pseg_regionprops_noalloc(a.h, big.data_ptr(), big.data_ptr(), 1025, 64, 64, props.ctypes.data, None) == -3
        assert b"static scratch" in a.L.pseg_last_error(a.h)
        wide = torch.zeros((600, 600), dtype=torch.uint8, device="cuda")
        assert a.L.pseg_regionprops_noalloc(a.h, wide.data_ptr(), wide.data_ptr(), 600, 600, 600, props.ctypes.data, None) == -3
        empty = torch.zeros((8, 8), dtype=torch.uint8, device="cuda")
        assert a.L.pseg_regionprops_noalloc(a.h, empty.data_ptr(), empty.data_ptr(), 8, 8, 8, props.ctypes.data, None) == -1
        for seg, (h, w) in ((a, (1024, 256)), (b, (3, 8192)), (a, (1, 1)), (b, (64, 40))):
            mask, img = _window(rng, h, w)
            (p0, s0), (p1, s1) = _both(seg, mask, img)
            assert np.array_equal(s0, s1) and np.array_equal(p0, p1, equal_nan=True), (h, w)
    finally:
        a.close()
        b.close()


def test_noalloc_moves_no_device_memory_over_100_calls():
    """free device memory (hipMemGetInfo through torch) is the same before and after 100 calls with changing window sizes;
    the growing-buffer entry point pseg_regionprops, by contrast, may allocate when a window outgrows its buffers"""
    import torch
    from imageprocessing_amd.engine import Segmenter
    seg = Segmenter(256, 256, max_frames=2, device=0)
    rng = np.random.default_rng(7)
    try:
        shapes = [(int(rng.integers(1, 1025)), 0) for _ in range(100)]
        shapes = [(h, int(rng.integers(1, min(8192, 262144 // h) + 1))) for h, _ in shapes]
        hmax = max(h for h, _ in shapes)
        wmax = max(w for _, w in shapes)
        mbuf = torch.zeros((1024, 8192), dtype=torch.uint8, device="cuda")
        gbuf = torch.randint(0, 256, (1024, 8192), dtype=torch.uint8, device="cuda")
        mbuf[:hmax, :wmax] = (torch.rand((hmax, wmax), device="cuda") < 0.6).to(torch.uint8) * 255
        mbuf[:, 0] = 255
        mbuf[0, :] = 255
        props = np.zeros(32)
        sums = np.zeros(24, np.int64)
        # warm-up: kernels loaded, the stream's first use
        assert seg.L.pseg_regionprops_noalloc(seg.h, mbuf.data_ptr(), gbuf.data_ptr(), 16, 16, 8192, props.ctypes.data, sums.ctypes.data) == 0
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for h, w in shapes:
            rc = seg.L.pseg_regionprops_noalloc(seg.h, mbuf.data_ptr(), gbuf.data_ptr(), h, w, 8192, props.ctypes.data, sums.ctypes.data)
            assert rc == 0, (h, w, seg.L.pseg_last_error(seg.h))
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, (free0, free1)
        h, w = shapes[-1]
        assert sums[0] == int((mbuf[:h, :w] == 255).sum().item())
    finally:
        seg.close()
