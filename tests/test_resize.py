"""the reader's ``cv.resize(img, (2560, 2560))`` (reader.py:79) on the device.

The oracle (oracle/cv_resize_ref.py) restates OpenCV's INTER_LINEAR for uint8; parity unpinned against OpenCV itself
(absent here, no fixture in the reference).  The CPU tests hold the restatement to values that follow from the
library's documented arithmetic; the GPU test compares ``pseg_resize_frames`` with it bit for bit."""
import numpy as np
import pytest

from oracle.cv_resize_ref import resize_linear_u8


def test_oracle_known_values():
    # pixel centres: dst 0..3 of a 2 -> 4 enlargement sample at -0.25 (clamped), 0.25, 0.75, 1.25 (clamped)
    assert resize_linear_u8(np.array([[0, 100]], np.uint8), 1, 4).tolist() == [[0, 25, 75, 100]]
    assert resize_linear_u8(np.array([[0], [100]], np.uint8), 4, 1).ravel().tolist() == [0, 25, 75, 100]
    # a constant image stays constant through the fixed-point passes, whatever the scale
    for v in (0, 1, 127, 200, 255):
        for shape in ((7, 9), (33, 20), (64, 64)):
            assert np.unique(resize_linear_u8(np.full(shape, v, np.uint8), 48, 56)).tolist() == [v]
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (32, 48), dtype=np.uint8)
    assert np.array_equal(resize_linear_u8(a, 32, 48), a)                          # equal sizes: a copy
    half = resize_linear_u8(a, 16, 24)                                             # 2 x 2 reduction: rounded area mean
    s = a.astype(np.int64)
    assert np.array_equal(half, ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8))
    up = resize_linear_u8(a, 64, 96)                                               # 2 x enlargement: weights 1/4, 3/4
    assert up[0, 0] == a[0, 0] and up[-1, -1] == a[-1, -1]
    assert abs(int(up[1, 1]) - (9 * int(a[0, 0]) + 3 * int(a[0, 1]) + 3 * int(a[1, 0]) + int(a[1, 1]) + 8) // 16) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("sh,sw", [(64, 128), (32, 64), (128, 256), (48, 100), (100, 90), (200, 131), (1, 1), (63, 127), (2, 300)])
def test_device_resize_equals_the_restatement(gpu_seg_factory, sh, sw):
    import torch
    H, W = 64, 128
    seg = gpu_seg_factory(H, W)
    rng = np.random.default_rng(sh * 1000 + sw)
    src = rng.integers(0, 256, (3, sh, sw), dtype=np.uint8)
    src[1] = np.arange(sh * sw, dtype=np.int64).reshape(sh, sw) % 251              # smooth ramps next to the noise
    out = seg.resize_frames(torch.from_numpy(src).cuda()).cpu().numpy()
    assert out.shape == (3, H, W)
    for k in range(3):
        assert np.array_equal(out[k], resize_linear_u8(src[k], H, W)), (sh, sw, k)


@pytest.mark.gpu
def test_device_resize_to_the_full_frame_size():
    """2048 x 2448 (a 5 MP camera frame) and 5120 x 5120 (2 x 2 reduction) to 2560 x 2560."""
    import torch
    from imageprocessing_amd.engine import Segmenter
    seg = Segmenter(2560, 2560, max_frames=2, device=0)
    rng = np.random.default_rng(5)
    for sh, sw in ((2048, 2448), (5120, 5120)):
        src = rng.integers(0, 256, (1, sh, sw), dtype=np.uint8)
        out = seg.resize_frames(torch.from_numpy(src).cuda()).cpu().numpy()
        assert np.array_equal(out[0], resize_linear_u8(src[0], 2560, 2560))
    seg.close()
