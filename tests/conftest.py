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
    return


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
