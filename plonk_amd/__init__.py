"""plonk_amd -- MI355X (gfx950) backend for the KZG / polynomial hot path of the
hjeljeli32/plonk prover.  The product is the C-ABI library built from plonk_amd/csrc
(include/plonk_hip.h); this package is the thin Python binding used by the tests, the
benchmark and the multi-GPU launcher."""
from .lib import load_library, LIB_PATH  # noqa: F401
from .backend import (Backend, MultiBackend, CommitError, ReferencePanic, PlonkHipError,  # noqa: F401
                      kzg_setup_tau_g2, kzg_verify, kzg_verify_batch)
