"""ctypes binding of libvsp_hip.so -- the C ABI declared in include/vsp.h.

There is no CPU fallback: if the HIP library is missing this module raises on first use.
Build it with ``python __graft_entry__.py`` (or ``make -C vote_saver_protocol_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# VSP_LIB_PATH: a variant build of the same library (diagnostic / experiment builds made by `make OUT=... EXTRA=...`); never a fallback
SO_PATH = os.environ.get("VSP_LIB_PATH") or os.path.join(_HERE, "libvsp_hip.so")

_P = C.c_void_p
_SZ = C.c_size_t
_I = C.c_int
_U = C.c_uint

# name -> (restype, argtypes); must list every symbol include/vsp.h declares
PROTOTYPES = {
    "vsp_create": (_P, [_I]),
    "vsp_destroy": (None, [_P]),
    "vsp_last_error": (C.c_char_p, [_P]),
    "vsp_set_stream": (_I, [_P, _P]),
    "vsp_synchronize": (_I, [_P]),
    "vsp_get_stat": (C.c_double, [_P, C.c_char_p]),
    "vsp_stats_reset": (None, [_P]),
    "vsp_set_option": (_I, [_P, C.c_char_p, C.c_long]),
    "vsp_diag_clock": (_I, [_P, _I, _P, _P]),
    "vsp_diag_clock_ntt": (_I, [_P, _I, _P, _P]),
    "vsp_dmalloc": (_P, [_P, _SZ]),
    "vsp_dfree": (None, [_P, _P]),
    "vsp_h2d": (_I, [_P, _P, _P, _SZ]),
    "vsp_d2h": (_I, [_P, _P, _P, _SZ]),
    "vsp_host_register": (_I, [_P, _P, _SZ]),
    "vsp_host_unregister": (_I, [_P, _P]),
    "vsp_msm_g1": (_I, [_P, _P, _P, _SZ, _P, _P]),
    "vsp_msm_g2": (_I, [_P, _P, _P, _SZ, _P, _P]),
    "vsp_bases_upload_g1": (_P, [_P, _P, _SZ]),
    "vsp_bases_upload_g2": (_P, [_P, _P, _SZ]),
    "vsp_bases_from_device_g1": (_P, [_P, _P, _SZ]),
    "vsp_bases_from_device_g2": (_P, [_P, _P, _SZ]),
    "vsp_bases_precompute": (_I, [_P, _P, _U]),
    "vsp_bases_precompute_split": (_I, [_P, _P, _U]),
    "vsp_bases_count": (_SZ, [_P]),
    "vsp_bases_device_bytes": (_SZ, [_P]),
    "vsp_keypair_device_bytes": (_SZ, [_P]),
    "vsp_bases_free": (None, [_P, _P]),
    "vsp_msm_resident": (_I, [_P, _P, _SZ, _SZ, _P, _P, _P]),
    "vsp_msm_resident_batch": (_I, [_P, _P, _SZ, _SZ, _P, _SZ, _SZ, _P, _P]),
    "vsp_msm_resident_jacobian": (_I, [_P, _P, _SZ, _SZ, _P, _P]),
    "vsp_msm_launch": (_I, [_P, _U, _P, _SZ, _SZ, _P]),
    "vsp_msm_finish_jacobian": (_I, [_P, _U, _P]),
    "vsp_fold_jacobian": (_I, [_P, _I, _P, _SZ, _P, _P]),
    "vsp_msm_finish_jacobian_device": (_I, [_P, _U, _P, _P]),
    "vsp_fold_jacobian_device": (_I, [_P, _I, _P, _SZ, _P, _P, _P]),
    "vsp_ntt_fr": (_I, [_P, _P, _U, _I, _P]),
    "vsp_ntt_fr_device": (_I, [_P, _P, _U, _I, _P]),
    "vsp_witness_map_h": (_I, [_P, _P, _P, _P, _U, _P]),
    "vsp_witness_map_h_device": (_I, [_P, _P, _P, _P, _U, _P]),
    "vsp_r1cs_upload": (_P, [_P, _SZ, _SZ, _SZ] + [_P] * 9),
    "vsp_r1cs_free": (None, [_P, _P]),
    "vsp_r1cs_domain_size": (_SZ, [_P]),
    "vsp_r1cs_domain_kind": (_I, [_P]),
    "vsp_domain_create": (_P, [_P, _SZ]),
    "vsp_domain_free": (None, [_P, _P]),
    "vsp_domain_size": (_SZ, [_P]),
    "vsp_domain_kind": (_I, [_P]),
    "vsp_domain_fft": (_I, [_P, _P, _P, _I, _P]),
    "vsp_domain_fft_device": (_I, [_P, _P, _P, _I, _P]),
    "vsp_domain_lagrange": (_I, [_P, _P, _P, _P]),
    "vsp_domain_element": (_I, [_P, _P, _SZ, _P]),
    "vsp_domain_vanishing": (_I, [_P, _P, _P, _P]),
    "vsp_domain_add_poly_z": (_I, [_P, _P, _P, _P]),
    "vsp_domain_divide_by_z_on_coset": (_I, [_P, _P, _P]),
    "vsp_domain_witness_map_h": (_I, [_P, _P, _P, _P, _P, _P]),
    "vsp_pk_create": (_P, [_P] * 11),
    "vsp_pk_free": (None, [_P, _P]),
    "vsp_groth16_prove": (_I, [_P] * 12),
    "vsp_groth16_prove_batch": (_I, [_P, _P, _P, _P, _SZ, _P, _P, _P, _P, _P, _P]),
    "vsp_groth16_prove_batch_launch": (_I, [_P, _P, _P, _P, _SZ, _P, _P]),
    "vsp_groth16_prove_batch_finish": (_I, [_P, _P, _P, _P, _P]),
    "vsp_groth16_prove_launch": (_I, [_P] * 8),
    "vsp_groth16_prove_finish": (_I, [_P] * 5),
    "vsp_witness_pack_words": (_SZ, [_SZ]),
    "vsp_witness_pack": (_I, [_P, _SZ, _P, _P, _P, _SZ, _P]),
    "vsp_groth16_prove_launch_packed": (_I, [_P, _P, _P, _P, _P, _P, _SZ, _P, _P, _P, _P]),
    "vsp_groth16_generate": (_P, [_P, _P, _P, _I]),
    "vsp_keypair_pk": (_P, [_P]),
    "vsp_keypair_count": (_SZ, [_P, _I]),
    "vsp_keypair_export": (_I, [_P, _P, _I, _P]),
    "vsp_keypair_free": (None, [_P, _P]),
    "vsp_saver_pk_words": (_SZ, [_SZ]),
    "vsp_saver_vk_words": (_SZ, [_SZ]),
    "vsp_saver_keygen": (_I, [_P, _SZ, _P, _P, _P, _P, _P, _P, _P]),
    "vsp_saver_pk_load": (_P, [_P, _SZ, _P, _P]),
    "vsp_saver_pk_free": (None, [_P, _P]),
    "vsp_saver_pk_msg_size": (_SZ, [_P]),
    "vsp_saver_encrypt": (_I, [_P] * 14),
    "vsp_saver_rerandomize": (_I, [_P] * 9),
    "vsp_fr_vector_blob_size": (_SZ, [_SZ]),
    "vsp_fr_vector_to_blob": (_I, [_P, _SZ, _P]),
    "vsp_fr_vector_from_blob": (_I, [_P, _SZ, _P, _SZ, _P]),
    "vsp_g1_vector_blob_size": (_SZ, [_SZ]),
    "vsp_g1_vector_to_blob": (_I, [_P, _SZ, _P]),
    "vsp_g1_vector_from_blob": (_I, [_P, _SZ, _I, _P, _SZ, _P]),
    "vsp_proof_to_blob": (_I, [_P, _P, _P, _P]),
    "vsp_proof_from_blob": (_I, [_P, _I, _P, _P, _P]),
    "vsp_vk_blob_size": (_SZ, [_SZ]),
    "vsp_vk_to_blob": (_I, [C.c_uint32, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "vsp_vk_from_blob": (_I, [_P, _SZ, _I, _P, _P, _P, _P, _P, _P, _SZ, _P, _P]),
    "vsp_pk_blob_size": (_SZ, [_P]),
    "vsp_pk_to_blob": (_I, [_P, _P, _P]),
    "vsp_pk_from_blob": (_P, [_P, _P, _SZ, _I]),
    "vsp_fixed_base_mul_g1": (_I, [_P, _P, _SZ, _P]),
    "vsp_fixed_base_mul_g2": (_I, [_P, _P, _SZ, _P]),
    "vsp_selftest_field": (_I, [_P, _I, _I, _P, _P, _P, _SZ]),
    "vsp_selftest_xyzz_add": (_I, [_P, _I, _I, _P, _P, _P, _SZ]),
    "vsp_g1_compress": (_I, [_P, _P]),
    "vsp_g2_compress": (_I, [_P, _P]),
    "vsp_g1_decompress": (_I, [_P, _I, _P, _P]),
    "vsp_g2_decompress": (_I, [_P, _I, _P, _P]),
}

_lib = None


class VspLibraryMissing(RuntimeError):
    pass


def load():
    """Load libvsp_hip.so and attach prototypes.  Raises VspLibraryMissing if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise VspLibraryMissing(
            f"{SO_PATH} not found: the HIP extension has not been built (run `python __graft_entry__.py`); "
            "there is no CPU fallback for this path")
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
