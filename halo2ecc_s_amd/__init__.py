"""MI355X-native witness-generation engine for the halo2ecc-s hot path.

Python is plumbing only (device memory via torch, streams, torch.distributed); the product is
`libh2e.so` — hand-written HIP for gfx950 behind the C ABI declared in include/h2e.h.  There is no CPU
fallback: importing the engine without the built library, or creating a context without a GPU, fails.
"""
from .engine import (Engine, Program, Records, Ring, H2EError, lib, lib_path, FIELD_BN256_FQ, FIELD_BLS12_381_FQ,  # noqa: F401
                     FIELD_BLS12_381_FR, ST_OK, ST_ASSERT_FAILED, ST_RETRY_ADD_SAME_OR_NEG_POINT,
                     ST_RETRY_ADD_IDENTITY, ST_ARITH, EXPORTED_SYMBOLS)

__all__ = ["Engine", "Program", "Records", "Ring", "H2EError", "lib", "lib_path"]
