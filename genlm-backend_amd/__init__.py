"""genlm-backend hot path, MI355X-native (gfx950).

Host side of the C-ABI HIP library `libglb_hip.so` (include/glb.h): the autobatched
`next_token_logprobs` path of genlm-backend (reference: genlm/backend/llm/hf.py, cache.py,
llm/base.py, README.md:46-115).  Import is cheap and works without a GPU; anything that computes
requires the built library and a HIP device and raises otherwise (no CPU fallback).
"""
from . import _lib  # noqa: F401
from ._lib import GlbError, LIB_PATH  # noqa: F401

__all__ = ["GlbError", "LIB_PATH", "load_model_by_name", "AsyncAmdLM", "HipEngine"]


def __getattr__(name):
    if name == "HipEngine":
        from .engine import HipEngine
        return HipEngine
    if name in ("load_model_by_name", "AsyncAmdLM", "AsyncLM"):
        from . import llm
        return getattr(llm, name)
    raise AttributeError(name)
