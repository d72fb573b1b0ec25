import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a HIP device: without one they are skipped (not failed) unless `-m gpu` asked for them
    explicitly -- on the GPU box a missing device must fail loudly, not skip."""
    import torch

    if torch.cuda.is_available() or "gpu" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="no HIP device visible (the product path has no CPU fallback)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_lib():
    """The HIP library must be built (cross-compiles without a GPU)."""
    from open_universe_amd import _lib

    if not os.path.exists(_lib.lib_path()):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.load()


class _Steer:
    """Forces kernel choices through the C ABI (ou_set_option) on every live model object and on those created later -- what the
    OU_* environment variables did until ABI 4, without the library reading the environment.  Undone at the end of the test."""

    def __init__(self):
        self.touched = set()

    def set(self, **kw):
        from open_universe_amd.universe import Universe

        Universe.set_default_options(**kw)
        self.touched |= set(kw)

    def unset(self, *keys):
        from open_universe_amd.universe import Universe

        Universe.set_default_options(**{k: None for k in keys})


@pytest.fixture
def steer():
    s = _Steer()
    yield s
    if s.touched:
        s.unset(*s.touched)
