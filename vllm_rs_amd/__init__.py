"""vllm_rs_amd — MI355X (gfx950) native drop-in for the quantized-forward hot path of vllm.rs.

Only what the path needs lives here: `csrc/` (HIP kernels + C ABI), `host/` (C++ runtime behind the
same ABI) and this thin ctypes mirror of the reference's operator interface (`ops`, `engine`).
"""
from . import _lib  # noqa: F401

BF16, F16, F32 = 0, 1, 2
__all__ = ["_lib", "BF16", "F16", "F32"]
