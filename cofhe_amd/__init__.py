"""cofhe_amd -- MI355X-native evaluation engine for CoFHE's local ciphertext-tensor path.

The product is the C-ABI shared library ``libcofhe_hip.so`` (include/cofhe_hip.h), built from
the hand-written HIP kernels in ``cofhe_amd/csrc``.  This package is the thin Python loader
used by the tests, the benchmark and ``__graft_entry__``; the C++20 host interface that mirrors
CoFHE's ``CryptoSystem`` API lives in ``cofhe_amd/host``.  There is no CPU fallback: every
call below fails loudly when the library or a GPU is missing.
"""
from .engine import Engine, CofheHipError, gather_plan, lib_path, load_library  # noqa: F401
