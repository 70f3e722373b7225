"""Loader of the product shared library (raytracer_amd/libraytracer_amd.so, built in-tree by hipcc).

Fails loudly: there is no CPU fallback.  If the library is missing or a symbol of
include/raytracer_amd.h is not exported, importing callers get an exception.
"""
import ctypes as C
import os

from . import abi

# RAYTRACER_AMD_LIB names another build of the same library (an instrumented one, scripts/bvh_stats.sh)
LIB_PATH = os.environ.get("RAYTRACER_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libraytracer_amd.so")

_dll = None


class RtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"raytracer_amd status {status}: {message}")
        self.status = status


def dll():
    """The bound C-ABI.  Raises if the HIP library is absent (build with __graft_entry__.build())."""
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} not built: run `make -C raytracer_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
        # One HIP runtime per process: PyTorch's ROCm wheel bundles its own libamdhip64.so (soname
        # libamdhip64.so.7, the soname this library needs).  Importing torch first makes the dynamic
        # linker resolve our dependency to that already-loaded copy; loading /opt/rocm's copy beside
        # it would give two runtimes fighting over the device ("No HIP GPUs are available").
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _dll = abi.bind(C.CDLL(LIB_PATH))
    return _dll


def check(status: int):
    if status != abi.RT_OK:
        raise RtError(status, dll().rt_last_error().decode())
