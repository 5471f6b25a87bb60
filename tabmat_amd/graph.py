"""HIP-graph replay of a whole device product.

One `SplitMatrix.sandwich` is ~40 small launches (block products, partial reductions, assembly
scatters); their launch cost and the Python between them is a few per cent of the step at 10M
rows and far more for small matrices.  `CapturedProduct` records the launch sequence of one
device-in/device-out call into a HIP graph (torch.cuda.CUDAGraph — on ROCm that is hipGraph) and
replays it: the per-call input (`d`) is copied into a static buffer, the result is a static
buffer.  The kernels, their order and their arguments are exactly those of the eager call.

The library's scratch pointer is baked into the captured launches, so the wrapper re-captures
when `tm_workspace_generation()` reports that the workspace moved (include/tabmat_hip.h)."""
import ctypes as C

import torch

from . import _lib


def _ws_generation() -> int:
    g = C.c_int64(0)
    _lib.call("tm_workspace_generation", C.byref(g))
    return int(g.value)


class CapturedProduct:
    """fn(static_input) -> device tensor, replayed from a HIP graph.

    fn must be a pure device-in/device-out product (no host synchronisation, no data-dependent
    Python control flow); every lazily built device twin it needs must exist already — the two
    eager warm-up calls before the capture take care of that."""

    def __init__(self, fn, example: torch.Tensor):
        self._fn = fn
        self._static_in = example.detach().clone()
        self._graph = None
        self._static_out = None
        self._gen = -1
        self._capture()

    def _capture(self):
        # Warm-up and capture run on ONE private stream: the library keeps a workspace per
        # (device, stream) and refuses to grow it during a capture, so the stream that is captured
        # must be the one that was warmed up.  The captured launches then use that stream's
        # workspace only -- eager products on other streams never share scratch with a replay.
        if getattr(self, "_side", None) is None:
            self._side = torch.cuda.Stream()
        side = self._side
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):          # builds lazy twins, grows workspace / caches
                self._fn(self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog) may touch the runtime meanwhile
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            out = self._fn(self._static_in)
        self._graph, self._static_out, self._gen = g, out, _ws_generation()

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self._gen != _ws_generation():
            self._capture()
        if x.data_ptr() != self._static_in.data_ptr():
            self._static_in.copy_(x)
        self._graph.replay()
        return self._static_out
