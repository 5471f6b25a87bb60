import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU should fail loudly rather than skip:
    # the product path has no CPU fallback.  Nothing to do here on purpose.
    # (the ABI coverage check looks back over the whole run: it goes last)
    last = [it for it in items if "test_zz_abi_coverage" in it.nodeid]
    if last:
        items[:] = [it for it in items if "test_zz_abi_coverage" not in it.nodeid] + last


# ---- which tm_* entry points did this run reach?  
# Every call into libtabmat_hip.so goes through an attribute of the ctypes handle (`_lib.call(name, ...)` and
# `_lib.lib().tm_x(...)` alike), so wrapping those attributes sees them all -- 180+ symbols behind ~40 dispatch
# predicates can otherwise go dark without a test failing.  tests/test_zz_abi_coverage.py reads ABI_CALLS at the end.
ABI_CALLS = {}


@pytest.fixture(scope="session", autouse=True)
def _abi_call_spy():
    if not _has_gpu():
        yield
        return
    from tabmat_amd import _lib

    handle = _lib.lib()
    for name in _lib.prototypes():
        fn = getattr(handle, name)
        if not hasattr(fn, "argtypes"):
            continue                    # (already wrapped)

        def spy(*args, _fn=fn, _name=name):
            ABI_CALLS[_name] = ABI_CALLS.get(_name, 0) + 1
            return _fn(*args)

        setattr(handle, name, spy)
    yield


@pytest.fixture(autouse=True)
def _column_major_kernels_for_f_order_cases(request):
    """Tests parametrised with order == "F" exist to exercise the column-major kernel variants:
    they run with the row-major twin of F-ordered dense blocks switched off
    (tabmat_amd/dense_matrix.py ROW_MAJOR_TWIN); everything else runs with the default."""
    params = getattr(getattr(request.node, "callspec", None), "params", {})
    if params.get("order") != "F":
        yield
        return
    import tabmat_amd.dense_matrix as dmod

    old = dmod.ROW_MAJOR_TWIN
    dmod.ROW_MAJOR_TWIN = False
    try:
        yield
    finally:
        dmod.ROW_MAJOR_TWIN = old
