"""plonky2-lib_amd: MI355X (gfx950) backend for the Goldilocks `CircuitData::prove()` path that
Orbiter-Finance/Plonky2-lib drives.  The product is `libglprover.so` (hand-written HIP kernels
behind the C ABI of include/glp.h); this package is the thin ctypes host layer used by the tests
and bench.py.  There is no CPU fallback anywhere in this package."""
from .binding import (GlpError, Context, Batch, Circuit, Session, StagedWitness, build_library, library_path, load_library,  # noqa: F401
                      exported_symbols, splitmix_field, P, CircuitFile, write_circuit_file)
