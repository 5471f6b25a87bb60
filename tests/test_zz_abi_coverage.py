"""Every entry point declared in include/tabmat_hip.h must have been CALLED by the `-m gpu` run that ends here
(VERDICT r5 item 7a): the host side picks kernels through ~40 measured predicates (`*_pays`, `*_ok`, cost models,
thresholds), and a branch that no test reaches any more would go dark silently.  The spy lives in conftest.py; the
tests that reach the symbols compare their results with the oracle / dense algebra.  Needs the whole suite in ONE
process: skipped under xdist and for partial runs (-k, single files)."""
import os

import pytest

pytestmark = pytest.mark.gpu

# not products: version / error text, device and memory plumbing of the C-only host path (the Python
# host uses torch for these), events, profiling / tuning switches
RUNTIME = {
    "tm_version", "tm_last_error", "tm_device_count", "tm_set_device", "tm_device_info", "tm_malloc", "tm_free",
    "tm_memcpy_h2d", "tm_memcpy_d2h", "tm_memset", "tm_stream_synchronize", "tm_set_workspace",
    "tm_workspace_generation", "tm_event_create", "tm_event_destroy", "tm_event_record", "tm_event_elapsed_ms",
    "tm_tune_set", "tm_tune_get", "tm_profile_enable", "tm_profile_last_ms",
}


def test_zz_abi_coverage(request):
    from conftest import ABI_CALLS
    from tabmat_amd import _lib

    if os.environ.get("PYTEST_XDIST_WORKER"):
        pytest.skip("needs the whole suite in one process (run without -n)")
    n_gpu_tests = sum(1 for it in request.session.items if it.get_closest_marker("gpu"))
    if request.config.option.keyword or n_gpu_tests < 2000:
        pytest.skip("partial run")
    declared = set(_lib.prototypes())
    unknown = RUNTIME - declared
    assert not unknown, f"allow-list names symbols the header does not declare: {sorted(unknown)}"
    dark = sorted(declared - set(ABI_CALLS) - RUNTIME)
    out = os.environ.get("TABMAT_AMD_ABI_COVERAGE_OUT")         # e.g. gpurun_out/abi_coverage.json
    if out:
        import json

        os.makedirs(os.path.dirname(os.path.abspath(out)) or ".", exist_ok=True)
        with open(out, "w") as f:
            json.dump({"declared": len(declared), "called": {k: ABI_CALLS[k] for k in sorted(ABI_CALLS)},
                       "runtime_allow_list": sorted(RUNTIME), "never_called": dark}, f, indent=1)
    assert not dark, f"{len(dark)} of {len(declared)} declared entry points were never called by the GPU suite: {dark}"
