"""curve25519_amd -- MI355X-native batched X25519 / Ed25519 engine behind msotoodeh/curve25519's C API.

The product is curve25519_amd/libcurve25519_amd.so (hand-written gfx950 HIP kernels + a C-ABI shim,
declared in include/*.h).  This package is the thin Python host side: `api` mirrors the reference's
function names over batches, `synth` generates the benchmark inputs, `sharded` splits a batch across
one-process-per-GPU ranks.  There is no CPU implementation in this package.
"""
from . import synth  # noqa: F401

__all__ = ["api", "synth", "sharded", "build"]
