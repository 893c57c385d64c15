"""lurk_beta_amd - MI355X-native proving hot path for Lurk (Pedersen MSM over the Pasta curves,
neptune-compatible Poseidon batch/tree, radix-2 NTT) as hand-written gfx950 HIP kernels behind the
C ABI of include/lurk_hip.h.  This package is the thin host-side mirror of the reference's
interfaces for that path (PoseidonCache, Trie roots, CommitmentEngine::commit); all arithmetic runs
in liblurk_hip.so on the GPU - there is no CPU fallback."""
from . import _lib
from ._lib import LurkHipError

FIELD_PALLAS_FP, FIELD_PALLAS_FQ, FIELD_BN254_FR = 0, 1, 2
CURVE_PALLAS, CURVE_VESTA = 0, 1

from .poseidon import PoseidonCache, HashArity, poseidon_batch, poseidon_tree8, poseidon_constants  # noqa: E402

from .msm import CommitmentKey, MultiCommitmentKey, msm, point_sum, point_to_affine  # noqa: E402
from .ntt import ntt  # noqa: E402
from .fold import R1CSShape, fold_vec, fold_vecs  # noqa: E402
from .step import FoldingContext, NivcFoldingContext, nifs_challenge, nova_ro_squeeze, point_mul, public_io  # noqa: E402
from . import sumcheck, ipa, spartan  # noqa: E402,F401
from . import params  # noqa: E402,F401
from .witness import MultiFrameWitness, slot_witness, slot_witness_size  # noqa: E402

__all__ = [
    "CommitmentKey", "MultiCommitmentKey", "msm", "point_sum", "point_to_affine", "ntt", "R1CSShape", "fold_vec", "fold_vecs", "FoldingContext", "NivcFoldingContext", "nifs_challenge", "nova_ro_squeeze", "point_mul", "public_io", "MultiFrameWitness", "slot_witness", "slot_witness_size",
    "LurkHipError", "PoseidonCache", "HashArity", "poseidon_batch", "poseidon_tree8", "poseidon_constants",
    "FIELD_PALLAS_FP", "FIELD_PALLAS_FQ", "FIELD_BN254_FR", "CURVE_PALLAS", "CURVE_VESTA",
]
