import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")
    # correctness tests run several frames in flight on whatever queue count the box exports
    config.addinivalue_line("filterwarnings", "ignore:frames_in_flight=.*GPU_MAX_HW_QUEUES:RuntimeWarning")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def small_scene():
    import wgpu_renderer_amd  # noqa: F401
    from wgpu_renderer_amd import scene
    return scene.sponza_standin(detail=0.2, tex_size=64)


@pytest.fixture(scope="session")
def tiny_scene():
    import wgpu_renderer_amd  # noqa: F401
    from wgpu_renderer_amd import scene
    return scene.sponza_standin(detail=0.08, tex_size=32)
