"""Host-side binding of libsnake_engine.so (the gfx950 self-play engine).

PyTorch is used for plumbing only: device memory (tensors), streams and torch.distributed.
Every compute call goes through the C ABI declared in include/snake_engine.h.  There is no
CPU fallback: importing works anywhere (so the symbol table can be checked on a CPU-only
box), but constructing an Engine without the library or without a GPU raises.
"""
from ._lib import lib, LIB_PATH, SnkGameState, check, EngineError  # noqa: F401
from .engine import Engine  # noqa: F401
