import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")     # the package's deployment default, set before anything initialises HIP
import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` via gpurun)")


def pytest_collection_modifyitems(config, items):
    """The product library carries no opt-in / experiment instantiation (EA_TOOLS builds only: the CPU emulation is one), so a
    (`gpu` backend x tools-only variant) parametrisation can never run: it is not collected at all, instead of showing up as
    119 permanent skips in every GPU run (round-5 verdict).  The emulation side of the same parametrisations stays."""
    tools_only = {2, 3, 4, 5, 6, 7, 8, 10, 11, 12, 13, 20, 21, 22, 23, 24, 31, 32}       # = test_kernels.TOOLS_ONLY_VARIANTS
    keep, drop = [], []
    for it in items:
        cs = getattr(it, "callspec", None)
        v = cs.params.get("variant") if cs is not None else None
        if cs is not None and cs.params.get("kb") == "gpu" and isinstance(v, (int, str)) and str(v).isdigit() and int(v) in tools_only:
            drop.append(it)
        else:
            keep.append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def emu_lib():
    """Host emulation build of the HIP kernels (tests/emu) bound through the same ctypes prototypes."""
    from editanything_amd.csrc import build
    from editanything_amd import _lib
    path = build.build_emu(verbose=False)
    return _lib.bind(path)


@pytest.fixture(params=["emu", pytest.param("gpu", marks=pytest.mark.gpu)])
def kb(request):
    """Kernel backend: the same op tests run on the CPU emulation and (with -m gpu) on the MI355X."""
    import emu_util
    if request.param == "emu":
        be = emu_util.HostBackend(request.getfixturevalue("emu_lib"))
    else:
        import torch
        assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
        from editanything_amd import _lib
        be = emu_util.GpuBackend(_lib.lib())
    emu_util.BACKEND = be
    yield be
    be.lib.ea_set_tuning(None)      # tests that pin an instantiation (test_kernels.tune) never leak it
    be.keep.clear()
    emu_util.BACKEND = None
