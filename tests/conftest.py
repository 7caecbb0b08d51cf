import os
import sys

import pytest

# Every GPU test runs on POISONED device memory (include/osmtile.h: osmt_debug_poison_enabled): buffers the library hands
# out are filled with 0xA5 first, so a kernel that reads something this render did not write cannot pass by reading the zero
# pages of a fresh process (round 4's list headers of empty tiles).  Set before libosmtile.so is loaded; an explicit
# OSMT_POISON_ALLOC=0 in the environment switches it off (timing runs).
os.environ.setdefault("OSMT_POISON_ALLOC", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py

    oracle_py.build()
    return oracle_py


@pytest.fixture(scope="session")
def gpu_ctx():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible (the raster path has no CPU fallback)")
    from osm_renderer_amd.renderer import Context

    ctx = Context(0)
    yield ctx
    ctx.close()
