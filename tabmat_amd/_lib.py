"""ctypes binding of libtabmat_hip.so (the C ABI in include/tabmat_hip.h).

The prototypes are read from the header itself, so the Python side can never drift
from the declared ABI.  There is NO fallback: if the library is missing or a call
fails, an exception is raised (the product path must not silently run on the CPU).
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADER = os.path.join(ROOT, "include", "tabmat_hip.h")
# TABMAT_AMD_LIB: load another build of the same ABI (kernel experiments)
LIB_PATH = os.environ.get("TABMAT_AMD_LIB") or os.path.join(_HERE, "libtabmat_hip.so")
CSRC = os.path.join(_HERE, "csrc")


class TabmatHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-j4", "-s"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(args)
    return LIB_PATH


_CTYPE = {
    "int": C.c_int,
    "int64_t": C.c_int64,
    "size_t": C.c_size_t,
    "float": C.c_float,
}


def parse_header(path: str = HEADER) -> dict[str, list]:
    """Return {function name: [ctypes argtypes]} for every `int tm_*(...)` /
    `const char *tm_*(...)` declaration in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out: dict[str, list] = {}
    for m in re.finditer(r"(?:int|const char \*)\s*(tm_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    types.append(C.c_void_p)
                else:
                    base = a.replace("const ", "").rsplit(" ", 1)[0].strip()
                    types.append(_CTYPE[base])
        out[name] = types
    return out


_lib = None
_protos = None


def prototypes() -> dict[str, list]:
    global _protos
    if _protos is None:
        _protos = parse_header()
    return _protos


def lib() -> C.CDLL:
    """Load the library (must have been built: `python -c 'import __graft_entry__ as g; g.build()'`
    or `make -C tabmat_amd/csrc`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TabmatHipError(
                f"{LIB_PATH} not found: build it with `make -C {CSRC}` "
                "(tabmat_amd has no CPU fallback)"
            )
        handle = C.CDLL(LIB_PATH)
        lax = bool(os.environ.get("TABMAT_AMD_LIB")) and os.environ.get("TABMAT_AMD_LIB_LAX") == "1"
        for name, argtypes in prototypes().items():
            if lax and not hasattr(handle, name):
                continue                # (A/B against an OLDER build of the ABI: scripts/dev only)
            fn = getattr(handle, name)  # AttributeError if the .so lacks a declared symbol
            fn.argtypes = argtypes
            fn.restype = C.c_char_p if name == "tm_last_error" else C.c_int
        _lib = handle
    return _lib


def last_error() -> str:
    msg = lib().tm_last_error()
    return msg.decode() if msg else ""


def call(name: str, *args) -> None:
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise TabmatHipError(f"{name} failed with code {rc}: {last_error()}")
