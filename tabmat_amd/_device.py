"""Device plumbing: torch is used ONLY for HBM allocation, host<->device copies, the
current HIP stream and (in distributed.py) the RCCL process group.  All arithmetic on
the hot path happens in libtabmat_hip.so."""
from __future__ import annotations

from typing import Optional

import numpy as np
import os

import torch

from . import _lib

_FSUF = {torch.float32: "f32", torch.float64: "f64"}
_NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64}


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.TabmatHipError(
            "tabmat_amd needs a ROCm GPU (MI355X): there is no CPU fallback for the "
            "sandwich / matvec path"
        )
    return torch.device("cuda", torch.cuda.current_device())


def is_dev(x) -> bool:
    return isinstance(x, torch.Tensor)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def fsuf(t: torch.Tensor) -> str:
    try:
        return _FSUF[t.dtype]
    except KeyError:
        raise TypeError(f"only float32/float64 are supported on the device, got {t.dtype}")


def same_float(fn: str, *tensors) -> None:
    """Every float tensor handed to one tm_*_{f32,f64} call must have the suffix dtype: the C ABI
    takes bare pointers, so a float32 operand behind a _f64 entry point would be read out of
    bounds.  The reference raises TypeError for mixed dtypes (util.py:62-67,
    sparse_matrix.py:218-223)."""
    for t in tensors:
        if t is not None and not t.is_floating_point():
            # an integer d / v behind a raw pointer would be reinterpreted as floats
            raise TypeError(f"{fn}: operands need to be np.float64 or np.float32 arrays, got {t.dtype}")
    dts = {t.dtype for t in tensors if t is not None}
    if len(dts) > 1:
        raise TypeError(f"{fn}: all floating-point operands need the same dtype, either "
                        f"np.float64 or np.float32; got {sorted(str(d) for d in dts)}")


def torch_dtype(np_dtype) -> torch.dtype:
    try:
        return _NP2T[np.dtype(np_dtype)]
    except KeyError:
        raise TypeError(f"only float32/float64 are supported on the device, got {np_dtype}")


def to_dev(x, dtype=None) -> torch.Tensor:
    """numpy / torch -> contiguous torch cuda tensor (no copy when already there)."""
    dev = require_gpu()
    if isinstance(x, torch.Tensor):
        t = x if x.is_cuda else x.to(dev)
        if dtype is not None and t.dtype != dtype:
            t = t.to(dtype)
        return t.contiguous()
    a = np.ascontiguousarray(x)
    if not a.flags.writeable:
        a = a.copy()  # torch.from_numpy refuses read-only buffers (reference accepts them)
    t = torch.from_numpy(a).to(dev)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def idx_dev(idx, dtype=torch.int32) -> Optional[torch.Tensor]:
    """Row / column list -> int32 device tensor (None stays None = "all")."""
    if idx is None:
        return None
    if isinstance(idx, torch.Tensor):
        return to_dev(idx, dtype)
    return to_dev(np.asarray(idx).astype(np.int64 if dtype == torch.int64 else np.int32), dtype)


def p(t: Optional[torch.Tensor]) -> Optional[int]:
    """device pointer for ctypes (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def nlen(t: Optional[torch.Tensor]) -> int:
    return 0 if t is None else int(t.numel())


def masked_d(d: torch.Tensor, rows: Optional[torch.Tensor], as_set: bool = False) -> torch.Tensor:
    """A row restriction as a masked weight vector for the kernels that make one full pass: rows outside
    `rows` get d = 0.  What a REPEATED row id means follows the reference product by product, and every
    masked-d path goes through here so that the masked pass and the row-list kernel of one product agree:
      as_set=False  a row that occurs k times counts k times (index_add_): the reference's `for k in rows` loops
                    (dense_helpers-tmpl.cpp:224, sparse_helpers-tmpl.cpp:67-131, ext/categorical.pyx,
                    ext/split.pyx) and its X[rows] indexing (categorical_matrix.py:825-838);
      as_set=True   it counts once (assignment): the sparse SELF sandwich, whose reference turns `rows` into a
                    uint8 mask (ext/sparse.pyx:46-48)."""
    if rows is None:
        return d
    dm = torch.zeros_like(d)
    r64 = rows.to(torch.int64)
    if as_set:
        dm[r64] = d[r64]
    else:
        dm.index_add_(0, r64, d[r64])
    return dm


def to_host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


def zeros(shape, dtype) -> torch.Tensor:
    return torch.zeros(shape, dtype=dtype, device=require_gpu())


def empty(shape, dtype) -> torch.Tensor:
    return torch.empty(shape, dtype=dtype, device=require_gpu())


# Result buffer of an entry point that OVERWRITES its output (the *_sandwich family): no fill
# launch.  TABMAT_AMD_POISON=1 fills it with NaN instead, so that the test-suite proves that every
# element is written by the kernels.
POISON = os.environ.get("TABMAT_AMD_POISON", "0") == "1"


def out_buf(shape, dtype) -> torch.Tensor:
    if POISON:
        return torch.full(shape, float("nan"), dtype=dtype, device=require_gpu())
    return torch.empty(shape, dtype=dtype, device=require_gpu())
