import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_seg_factory():
    """Factory of Segmenter contexts on cuda:0; contexts are cached per (H, W)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from imageprocessing_amd.engine import Segmenter
    cache = {}

    def get(H, W, max_frames=64):
        key = (H, W)
        if key not in cache or cache[key].max_frames < max_frames:
            cache[key] = Segmenter(H, W, max_frames=max_frames, device=0)
        return cache[key]

    yield get
    for s in cache.values():
        s.close()
