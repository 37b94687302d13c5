"""ctypes loader for the C-ABI HIP library.  Fails loudly if the library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib_path() -> str:
    return os.path.join(_HERE, "lib", "libmedmoe_hip.so")


def load_library() -> ctypes.CDLL:
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(
                f"medmoe_amd: HIP library not built ({p}). Run `make` (or "
                "`python -c 'import __graft_entry__ as g; g.build()'`); there is no fallback path.")
        # torch first: it ships its own libamdhip64, and the stream handles this library receives come from that runtime.  Loaded before
        # torch, the library binds /opt/rocm's copy instead and its first launch on a torch stream fails (hipErrorInvalidResourceHandle;
        # seen as `python __graft_entry__.py smoke`, where build() loads the library before smoke() imports torch).
        import torch  # noqa: F401
        _LIB = ctypes.CDLL(p)
    return _LIB
