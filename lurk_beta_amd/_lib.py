"""ctypes loader for liblurk_hip.so (the C-ABI drop-in boundary, include/lurk_hip.h).

There is no fallback: if the HIP library has not been built (``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C lurk_beta_amd/csrc``) loading raises, and every
compute entry point raises ``LurkHipError`` when no gfx950 device is usable."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# LURK_HIP_LIB: another build of the same library (A/B runs of two builds on one GPU box: bench_tools/ab.sh); never a fallback
LIB_PATH = os.environ.get("LURK_HIP_LIB") or os.path.join(_HERE, "liblurk_hip.so")
_lib = None

c_void_p, c_size_t, c_int, c_uint, c_u64 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_uint, ctypes.c_uint64


class LurkHipError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(message or f"lurk_hip error {code}")
        self.code = code


class RustError(ctypes.Structure):
    """sppark's RustError as pasta-msm's cuda_pippenger_* return it by value: {code, message (malloc'd, NULL on success)}"""
    _fields_ = [("code", c_int), ("message", ctypes.c_void_p)]


class RoParamsStruct(ctypes.Structure):
    """lurk_hip_ro_params"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("arity", ctypes.c_uint32), ("domain_separator", ctypes.c_uint32), ("absorb_tag_bit", ctypes.c_uint32),
                ("num_challenge_bits", ctypes.c_uint32), ("item_order", ctypes.c_uint32 * 4), ("relaxed_order", ctypes.c_uint32 * 4),
                ("fresh_order", ctypes.c_uint32 * 2), ("point_elements", ctypes.c_uint32), ("relaxed_x_limbs", ctypes.c_uint32),
                ("fresh_x_limbs", ctypes.c_uint32), ("limb_bits", ctypes.c_uint32), ("pattern_absorbs", ctypes.c_uint32), ("squeeze_element", ctypes.c_uint32)]


class CkParamsStruct(ctypes.Structure):
    """lurk_hip_ck_params"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("xof", ctypes.c_uint32), ("bytes_per_point", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("domain_prefix", ctypes.c_char * 32), ("curve_name_pallas", ctypes.c_char * 16), ("curve_name_vesta", ctypes.c_char * 16),
                ("suite", ctypes.c_char * 32)]


# every symbol include/lurk_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "lurk_hip_device_count": (c_int, []),
    "lurk_hip_last_error": (ctypes.c_char_p, []),
    "lurk_hip_version": (ctypes.c_char_p, []),
    "lurk_hip_abi_version": (c_int, []),
    "lurk_hip_scratch_trim": (c_int, [ctypes.POINTER(c_size_t)]),
    "lurk_hip_set_device": (c_int, [c_int]),
    "lurk_hip_profile_enable": (c_int, [c_int]),
    "lurk_hip_profile_reset": (c_int, []),
    "lurk_hip_profile_get": (c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_u64)]),
    "lurk_hip_msm_pallas": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    "lurk_hip_msm_vesta": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_int]),
    "lurk_hip_msm_oneshot_key_cache": (c_int, [c_int]),
    "mult_pippenger_pallas": (None, [c_void_p, c_void_p, c_size_t, c_void_p, ctypes.c_bool]),   # pasta-msm's own symbol names
    "mult_pippenger_vesta": (None, [c_void_p, c_void_p, c_size_t, c_void_p, ctypes.c_bool]),
    "cuda_pippenger_pallas": (RustError, [c_void_p, c_void_p, c_size_t, c_void_p, ctypes.c_bool]),   # pasta-msm's GPU symbol names
    "cuda_pippenger_vesta": (RustError, [c_void_p, c_void_p, c_size_t, c_void_p, ctypes.c_bool]),
    "lurk_hip_msm_ctx_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, c_int]),
    "lurk_hip_msm_ctx_create_dev": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "lurk_hip_msm_ctx_run": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "lurk_hip_msm_ctx_run_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "lurk_hip_msm_ctx_submit_dev": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "lurk_hip_msm_ctx_submit_dev_mode": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p, c_int]),
    "lurk_hip_msm_ctx_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "lurk_hip_msm_ctx_submit_pair_dev": (c_int, [c_void_p, c_int, c_void_p, c_size_t, c_int, c_void_p, c_int]),
    "lurk_hip_msm_ctx_wait_pair": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "lurk_hip_msm_ctx_destroy": (c_int, [c_void_p]),
    "lurk_hip_msm_ctx_rebind_dev": (c_int, [c_void_p, c_void_p, c_size_t]),
    "lurk_hip_msm_ctx_reserve": (c_int, [c_void_p, c_size_t, c_int]),
    "lurk_hip_shake256": (c_int, [ctypes.c_char_p, c_size_t, c_void_p, c_size_t]),
    "lurk_hip_ck_params_get": (c_int, [ctypes.POINTER(CkParamsStruct)]),
    "lurk_hip_ck_params_set": (c_int, [ctypes.POINTER(CkParamsStruct)]),
    "lurk_hip_ck_from_label_host": (c_int, [c_int, ctypes.c_char_p, c_size_t, c_size_t, c_void_p]),
    "lurk_hip_ck_hash_to_curve_dev": (c_int, [c_int, ctypes.c_char_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_ck_from_label_dev": (c_int, [c_int, ctypes.c_char_p, c_size_t, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_msm_ctx_from_label": (c_int, [ctypes.POINTER(c_void_p), c_int, ctypes.c_char_p, c_size_t, c_size_t, c_int]),
    "lurk_hip_msm_ctx_info": (c_int, [c_void_p, ctypes.POINTER(c_int), ctypes.POINTER(c_size_t), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "lurk_hip_msm_ctx_form": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "lurk_hip_msm_ctx_device": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "lurk_hip_msm_ctx_save": (c_int, [c_void_p, ctypes.c_char_p, c_int]),
    "lurk_hip_msm_ctx_load": (c_int, [ctypes.POINTER(c_void_p), ctypes.c_char_p, c_int]),
    "lurk_hip_msm_multi_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, ctypes.POINTER(c_int), c_int, c_int]),
    "lurk_hip_msm_multi_num_shards": (c_int, [c_void_p]),
    "lurk_hip_msm_multi_shard": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "lurk_hip_msm_multi_commit": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int]),
    "lurk_hip_msm_multi_commit_dev": (c_int, [c_void_p, c_void_p, ctypes.POINTER(c_void_p), c_size_t, c_size_t, c_int]),
    "lurk_hip_msm_multi_submit_dev": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), c_size_t, c_size_t, c_int, c_int]),
    "lurk_hip_msm_multi_wait": (c_int, [c_void_p, c_int, c_void_p]),
    "lurk_hip_msm_multi_destroy": (c_int, [c_void_p]),
    "lurk_hip_point_sum": (c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    "lurk_hip_point_sum_gathered": (c_int, [c_int, c_void_p, c_void_p, c_size_t]),
    "lurk_hip_point_mul": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "lurk_hip_point_to_affine_canonical": (c_int, [c_int, c_void_p, c_void_p]),
    "lurk_hip_poseidon_batch": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lurk_hip_poseidon_batch_dev": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_poseidon_tree8": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_poseidon_tree8_dev": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_poseidon_constants": (c_int, [c_int, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_void_p, c_void_p]),
    "lurk_hip_poseidon_hash_host": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "lurk_hip_store_hydrate": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, ctypes.POINTER(c_size_t)]),
    "lurk_hip_slot_witness_size": (c_int, [c_int, c_int, ctypes.POINTER(c_size_t)]),
    "lurk_hip_slot_witness_dev": (c_int, [c_int, c_int, c_void_p, c_size_t, c_int, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p]),
    "lurk_hip_slot_witness": (c_int, [c_int, c_int, c_void_p, c_size_t, c_int, c_void_p]),
    "lurk_hip_frames_witness_dev": (c_int, [c_int, c_size_t, ctypes.POINTER(c_size_t), ctypes.POINTER(c_void_p), c_int, c_void_p, c_size_t, c_size_t, c_void_p]),
    "lurk_hip_witness_blocks_dev": (c_int, [c_void_p, c_size_t, c_size_t, c_void_p, c_int, c_size_t, c_size_t, c_void_p]),
    "lurk_hip_ntt": (c_int, [c_int, c_void_p, c_uint, c_int]),
    "lurk_hip_ntt_dev": (c_int, [c_int, c_void_p, c_uint, c_int, c_void_p]),
    "lurk_hip_r1cs_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_size_t, c_size_t, c_size_t] + [c_void_p] * 9),
    "lurk_hip_r1cs_destroy": (c_int, [c_void_p]),
    "lurk_hip_r1cs_dims": (c_int, [c_void_p, ctypes.POINTER(c_int)] + [ctypes.POINTER(c_size_t)] * 3),
    "lurk_hip_r1cs_info": (c_int, [c_void_p] + [ctypes.POINTER(c_size_t)] * 4),
    "lurk_hip_r1cs_device": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "lurk_hip_r1cs_multiply_vec_dev": (c_int, [c_void_p] * 6),
    "lurk_hip_r1cs_cross_term_dev": (c_int, [c_void_p] * 5),
    "lurk_hip_r1cs_cross_term_cached_dev": (c_int, [c_void_p] * 15),
    "lurk_hip_fold_vecs_dev": (c_int, [c_int, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), ctypes.POINTER(c_void_p), c_void_p, c_void_p]),
    "lurk_hip_fold_vec_dev": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_r1cs_multiply_vec": (c_int, [c_void_p] * 5),
    "lurk_hip_r1cs_cross_term": (c_int, [c_void_p] * 4),
    "lurk_hip_fold_vec": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "lurk_hip_fold_ctx_create": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_create_multi": (c_int, [ctypes.POINTER(c_void_p), c_int, c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_add_helper": (c_int, [c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_destroy": (c_int, [c_void_p]),
    "lurk_hip_fold_ctx_set_running": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_step_begin": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_step_prefetch": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_int, c_void_p]),
    "lurk_hip_fold_step_begin_prefetched": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_step_finish": (c_int, [c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_set_pp_digest": (c_int, [c_void_p, c_void_p]),
    "lurk_hip_keccak_sumcheck_challenge": (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    "lurk_hip_keccak_ipa_challenge": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_set_submit_hook": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_step_challenge": (c_int, [c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_running_dev": (c_int, [c_void_p, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p)]),
    "lurk_hip_fold_ctx_read": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_set_instance": (c_int, [c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_ctx_instance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_fold_step": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_ro_params_get": (c_int, [ctypes.POINTER(RoParamsStruct)]),
    "lurk_hip_ro_params_set": (c_int, [ctypes.POINTER(RoParamsStruct)]),
    "lurk_hip_nifs_absorb_list": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t,
                                          ctypes.POINTER(c_size_t)]),
    "lurk_hip_nova_ro_squeeze": (c_int, [c_int, c_void_p, c_size_t, c_uint, c_void_p]),
    "lurk_hip_nova_ro_pattern_tag": (c_int, [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, c_void_p]),
    "lurk_hip_nifs_challenge": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_keccak256": (c_int, [ctypes.c_char_p, c_size_t, c_void_p]),
    "lurk_hip_keccak_transcript_new": (c_int, [ctypes.POINTER(c_void_p), ctypes.c_char_p, c_size_t]),
    "lurk_hip_keccak_transcript_destroy": (c_int, [c_void_p]),
    "lurk_hip_keccak_transcript_absorb": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, ctypes.c_char_p, c_size_t]),
    "lurk_hip_keccak_transcript_absorb_scalars": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, c_void_p, c_size_t]),
    "lurk_hip_keccak_transcript_absorb_point": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, c_int, c_void_p]),
    "lurk_hip_keccak_transcript_dom_sep": (c_int, [c_void_p, ctypes.c_char_p, c_size_t]),
    "lurk_hip_keccak_transcript_squeeze": (c_int, [c_void_p, ctypes.c_char_p, c_size_t, c_int, c_void_p]),
    "lurk_hip_sumcheck_round_dev": (c_int, [c_int, c_int, ctypes.POINTER(c_void_p), c_size_t, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_eq_evals_dev": (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "lurk_hip_inner_product_dev": (c_int, [c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_fold_halves_dev": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_ipa_round_scalars_dev": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_ipa_coef_fold_dev": (c_int, [c_int, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_points_fold_halves_dev": (c_int, [c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_sumcheck_prove_dev": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_sumcheck_prove_batch_dev": (c_int, [c_int, c_int, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_msm_ctx_fold_key_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_spartan_prove_dev": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_ipa_prove_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "lurk_hip_spartan_prove_batch_dev": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "lurk_hip_synth_scalars_dev": (c_int, [c_int, c_u64, c_int, c_size_t, c_size_t, c_void_p, c_int, c_void_p]),
    "lurk_hip_synth_bases_dev": (c_int, [c_int, c_size_t, c_size_t, c_void_p, c_void_p]),
}


def load():
    """Load liblurk_hip.so and bind every declared symbol.  Raises if the library is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is not built: the HIP extension is the product path and there is no CPU "
                "fallback.  Build it with `make -C lurk_beta_amd/csrc` (hipcc, gfx950)."
            )
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64 (same SONAME as
        # /opt/rocm's).  Importing torch first makes the dynamic loader bind liblurk_hip.so to that
        # copy; loading ours first would put two ROCr instances in the process and the second one
        # cannot open the device.  Without torch the library binds to /opt/rocm's runtime.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int):
    if rc != 0:
        raise LurkHipError(rc, load().lurk_hip_last_error().decode())


def ptr(x) -> c_void_p:
    """Pointer of a numpy array (host) or a torch tensor (host or device) or a raw int."""
    if x is None:
        return c_void_p(0)
    if isinstance(x, int):
        return c_void_p(x)
    if hasattr(x, "data_ptr"):
        return c_void_p(x.data_ptr())
    return c_void_p(x.ctypes.data)


class KeccakRoundBindingStruct(ctypes.Structure):
    """lurk_hip_keccak_round_binding"""
    _fields_ = [("transcript", ctypes.c_void_p), ("field_id", ctypes.c_int), ("n_scalars", ctypes.c_int), ("curve", ctypes.c_int),
                ("absorb_label", ctypes.c_char_p), ("absorb_label_len", ctypes.c_size_t),
                ("absorb_label2", ctypes.c_char_p), ("absorb_label2_len", ctypes.c_size_t),
                ("squeeze_label", ctypes.c_char_p), ("squeeze_label_len", ctypes.c_size_t),
                ("challenges_out", ctypes.c_void_p), ("challenges_cap", ctypes.c_size_t), ("n_rounds", ctypes.c_size_t)]


class SpartanProofStruct(ctypes.Structure):
    """lurk_hip_spartan_proof"""
    _fields_ = [(k, ctypes.c_void_p) for k in ("polys_outer", "claims_outer", "eval_e", "polys_inner", "eval_w", "polys_batch", "evals_batch", "ipa_l", "ipa_r", "ipa_a")]


class SpartanInstanceStruct(ctypes.Structure):
    """lurk_hip_spartan_instance"""
    _fields_ = [("shape", ctypes.c_void_p), ("shape_t", ctypes.c_void_p), ("num_cons", ctypes.c_size_t), ("num_vars", ctypes.c_size_t), ("num_io", ctypes.c_size_t),
                ("x32_canonical", ctypes.c_void_p), ("u32_canonical", ctypes.c_void_p), ("d_w32_mont", ctypes.c_void_p), ("d_e32_mont", ctypes.c_void_p),
                ("comm_w_jacobian96", ctypes.c_void_p), ("comm_e_jacobian96", ctypes.c_void_p)]


class SpartanBatchProofStruct(ctypes.Structure):
    """lurk_hip_spartan_batch_proof"""
    _fields_ = [(k, ctypes.c_void_p) for k in ("polys_outer", "claims_outer", "evals_e", "polys_inner", "evals_w", "polys_batch", "evals_batch", "ipa_l", "ipa_r", "ipa_a")]


class KeccakRounds:
    """A ``challenge`` argument that never leaves the library: the rounds of a sum-check / inner-product argument absorb into and squeeze
    from a Keccak transcript through lurk_hip_keccak_sumcheck_challenge / lurk_hip_keccak_ipa_challenge (no Python in the round loop).
    ``challenges()`` returns what was squeezed, in order."""

    def __init__(self, transcript_handle, field_id: int, absorb: bytes, squeeze: bytes, absorb2: bytes = b"", curve: int = 0, cap: int = 64):
        import numpy as np

        self._out = np.zeros((cap, 4), dtype=np.uint64)
        self._labels = (bytes(absorb), bytes(absorb2), bytes(squeeze))  # kept alive: the struct borrows them
        self.struct = KeccakRoundBindingStruct(transcript_handle, field_id, 0, curve, self._labels[0], len(absorb), self._labels[1], len(absorb2),
                                               self._labels[2], len(squeeze), self._out.ctypes.data, cap, 0)

    def callback(self, kind: str):
        """(function pointer, user pointer) for the library's round loops; kind: 'sumcheck' or 'ipa'."""
        fn = getattr(load(), "lurk_hip_keccak_sumcheck_challenge" if kind == "sumcheck" else "lurk_hip_keccak_ipa_challenge")
        return ctypes.cast(fn, ctypes.c_void_p), ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)

    def ipa_round(self, j: int, L, R) -> int:
        """One inner-product round from Python (the prover that folds its key drives its own loop): absorb L and R, squeeze."""
        import numpy as np

        out = np.zeros(4, dtype=np.uint64)
        l, r = np.ascontiguousarray(L, dtype=np.uint64), np.ascontiguousarray(R, dtype=np.uint64)
        check(load().lurk_hip_keccak_ipa_challenge(ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p), j, ptr(l), ptr(r), ptr(out)))
        return int(out[0]) | int(out[1]) << 64 | int(out[2]) << 128 | int(out[3]) << 192

    def challenges(self) -> list[int]:
        k = int(self.struct.n_rounds)
        return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in self._out[:k]]


# lurk_hip_ipa_challenge_fn: int (*)(void* user, int round, const void* L96, const void* R96, void* out_r32_canonical)
IPA_CHALLENGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
# lurk_hip_fold_submit_hook_fn: int (*)(void* user)
FOLD_SUBMIT_HOOK_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p)
# lurk_hip_sumcheck_challenge_fn: int (*)(void* user, int round, const void* coefficients, void* out_r32_canonical)
SUMCHECK_CHALLENGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p)
