"""mi355q -- MI355X (gfx950) quantized-inference hot path behind TensorRT-LLM's plugin contracts.

Python here is host-side plumbing over the C ABI of libmi355q.so (include/mi355q.h); all arithmetic
lives in the HIP library.  Importing this package never falls back to a CPU implementation.
"""
from . import _lib
from ._lib import Mi355qError, LIB_PATH
from .plugin import Plugin, get_plugin_names
from . import functional

__all__ = ["_lib", "Mi355qError", "LIB_PATH", "Plugin", "get_plugin_names", "functional"]
