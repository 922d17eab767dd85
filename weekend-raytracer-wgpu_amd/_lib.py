"""Loader of libmirt.so (HIP kernels + C ABI).  Fails loudly: there is no fallback path."""
from __future__ import annotations

import ctypes as C
import os
import sys
from pathlib import Path

from . import _abi

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = _PKG_DIR / "csrc" / "libmirt.so"

_lib: C.CDLL | None = None


class MirtError(RuntimeError):
    """A C-ABI call returned a negative MirtStatus."""

    def __init__(self, status: int, message: str):
        self.status = status
        self.status_name = _abi.STATUS.get(status, f"MIRT_ERR_UNKNOWN({status})")
        super().__init__(f"{self.status_name}: {message}")


def lib() -> C.CDLL:
    """The bound library.  Raises ImportError if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {LIB_PATH.parent}`.  There is no CPU fallback for the render path.")
    # If torch is (or will be) in the process, let it load its bundled libamdhip64.so.7 first:
    # both libraries then share ONE HIP runtime, so torch device pointers and streams are valid
    # arguments of mirt_ctx_render_device.
    if "torch" in sys.modules or os.environ.get("MIRT_WITH_TORCH", "1") == "1":
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a requirement of the library itself
            pass
    handle = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    _abi.bind(handle)
    _lib = handle
    return handle


def check(status: int) -> None:
    if status != _abi.MIRT_OK:
        msg = lib().mirt_last_error()
        raise MirtError(status, msg.decode("utf-8", "replace") if msg else "")
