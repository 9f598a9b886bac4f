"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the C oracle (oracle/vsp_ref.c).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product.
All arrays are numpy uint64, canonical little-endian limbs (Fr 4, G1 12, G2 24 per element).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("VSP_REF_SO") or os.path.join(_HERE, "build", "libvsp_ref.so")      # VSP_REF_SO: another build of vsp_ref.c (the sanitizer run of tests/test_oracle.py)


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.ref_init()
    lib.ref_r1cs_synth.restype = C.c_void_p
    lib.ref_r1cs_from_csr.restype = C.c_void_p
    lib.ref_groth16_generate.restype = C.c_void_p
    lib.ref_r1cs_num_vars.restype = C.c_size_t
    lib.ref_r1cs_nnz.restype = C.c_size_t
    lib.ref_keypair_count.restype = C.c_size_t
    lib.ref_r1cs_domain_size.restype = C.c_size_t
    lib.ref_domain_size.restype = C.c_size_t
    lib.ref_domain_op.restype = C.c_size_t
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def fp_mul(a, b):
    out = np.zeros(6, np.uint64); lib().ref_fp_mul(_p(_u64(a)), _p(_u64(b)), _p(out)); return out


def fp_inv(a):
    out = np.zeros(6, np.uint64); lib().ref_fp_inv(_p(_u64(a)), _p(out)); return out


def fr_mul(a, b):
    out = np.zeros(4, np.uint64); lib().ref_fr_mul(_p(_u64(a)), _p(_u64(b)), _p(out)); return out


def fr_inv(a):
    out = np.zeros(4, np.uint64); lib().ref_fr_inv(_p(_u64(a)), _p(out)); return out


def g1_add(p, q):
    out = np.zeros(12, np.uint64); lib().ref_g1_add(_p(_u64(p)), _p(_u64(q)), _p(out)); return out


def g1_mul(p, k):
    out = np.zeros(12, np.uint64); lib().ref_g1_mul(_p(_u64(p)), _p(_u64(k)), _p(out)); return out


def g2_add(p, q):
    out = np.zeros(24, np.uint64); lib().ref_g2_add(_p(_u64(p)), _p(_u64(q)), _p(out)); return out


def g2_mul(p, k):
    out = np.zeros(24, np.uint64); lib().ref_g2_mul(_p(_u64(p)), _p(_u64(k)), _p(out)); return out


def msm_g1(bases, scalars, mixed=False):
    bases, scalars = _u64(bases), _u64(scalars)
    n = scalars.size // 4
    out = np.zeros(12, np.uint64)
    lib().ref_msm_g1(_p(bases), _p(scalars), C.c_size_t(n), _p(out), C.c_int(int(mixed)))
    return out


def msm_g2(bases, scalars, mixed=False):
    bases, scalars = _u64(bases), _u64(scalars)
    n = scalars.size // 4
    out = np.zeros(24, np.uint64)
    lib().ref_msm_g2(_p(bases), _p(scalars), C.c_size_t(n), _p(out), C.c_int(int(mixed)))
    return out


def g1_batch_mul_gen(scalars):
    scalars = _u64(scalars); n = scalars.size // 4
    out = np.zeros((n, 12), np.uint64); lib().ref_g1_batch_mul_gen(_p(scalars), C.c_size_t(n), _p(out)); return out


def g2_batch_mul_gen(scalars):
    scalars = _u64(scalars); n = scalars.size // 4
    out = np.zeros((n, 24), np.uint64); lib().ref_g2_batch_mul_gen(_p(scalars), C.c_size_t(n), _p(out)); return out


def ntt_fr(a, inverse=False, coset=None):
    a = _u64(a).copy().reshape(-1, 4)
    n = a.shape[0]; log_m = n.bit_length() - 1
    assert 1 << log_m == n
    g = None if coset is None else _u64(coset)
    lib().ref_ntt_fr(_p(a), C.c_uint(log_m), C.c_int(int(inverse)), _p(g))
    return a


class Domain:
    """make_evaluation_domain(min_size) of the C oracle: basic radix-2 or step radix-2 (libfqfft order)."""
    OPS = {"fft": 0, "inverse_fft": 1, "coset_fft": 2, "inverse_coset_fft": 3, "lagrange": 4, "divide_by_z_on_coset": 5}

    def __init__(self, min_size):
        self.min_size = min_size
        self.m = lib().ref_domain_size(C.c_size_t(min_size))
        if self.m == 0:
            raise ValueError("no radix-2 family domain for this size")
        self.is_step = bool(lib().ref_domain_is_step(C.c_size_t(min_size)))

    def _op(self, op, a, aux):
        a = _u64(a).copy().reshape(-1, 4)
        assert a.shape[0] == self.m
        lib().ref_domain_op(C.c_size_t(self.min_size), C.c_int(self.OPS[op]), _p(a), _p(None if aux is None else _u64(aux)))
        return a

    def fft(self, a): return self._op("fft", a, None)
    def inverse_fft(self, a): return self._op("inverse_fft", a, None)
    def coset_fft(self, a, g): return self._op("coset_fft", a, g)
    def inverse_coset_fft(self, a, g): return self._op("inverse_coset_fft", a, g)
    def divide_by_z_on_coset(self, a, g): return self._op("divide_by_z_on_coset", a, g)

    def evaluate_all_lagrange_polynomials(self, t):
        return self._op("lagrange", np.zeros((self.m, 4), np.uint64), t)

    def get_domain_element(self, idx):
        out = np.zeros(4, np.uint64); lib().ref_domain_element(C.c_size_t(self.min_size), C.c_size_t(idx), _p(out)); return out

    def compute_vanishing_polynomial(self, t):
        out = np.zeros(4, np.uint64); lib().ref_domain_vanishing(C.c_size_t(self.min_size), _p(_u64(t)), _p(out)); return out


class R1CS:
    """Handle on a C-side constraint system (CSR triples A, B, C)."""

    def __init__(self, handle, num_constraints, num_inputs):
        self.h = C.c_void_p(handle)
        self.num_constraints, self.num_inputs = num_constraints, num_inputs
        self.num_vars = lib().ref_r1cs_num_vars(self.h)
        self.m = lib().ref_r1cs_domain_size(self.h)          # make_evaluation_domain(num_constraints + num_inputs + 1)
        self.is_step = bool(lib().ref_r1cs_domain_is_step(self.h))

    @classmethod
    def synth(cls, num_constraints, num_inputs, seed, ballot=None):
        """ballot = (msg_size, vote): the first msg_size public inputs are the one-hot ballot m[vote] = 1 (common.hpp:1029-1040)"""
        nv = num_constraints + num_inputs
        wit = np.zeros((nv, 4), np.uint64)
        lib().ref_r1cs_synth_ballot.restype = C.c_void_p
        msg_size, vote = ballot if ballot else (0, 0)
        h = lib().ref_r1cs_synth_ballot(C.c_size_t(num_constraints), C.c_size_t(num_inputs), C.c_uint64(seed), C.c_size_t(msg_size), C.c_size_t(vote), _p(wit))
        cs = cls(h, num_constraints, num_inputs)
        return cs, wit

    def export(self):
        """-> [(row_ptr u32[nc+1], col_idx u32[nnz], coeff u64[nnz,4])] for A, B, C."""
        out = []
        for m in range(3):
            nnz = lib().ref_r1cs_nnz(self.h, C.c_int(m))
            rp = np.zeros(self.num_constraints + 1, np.uint32)
            ci = np.zeros(max(nnz, 1), np.uint32)
            co = np.zeros((max(nnz, 1), 4), np.uint64)
            lib().ref_r1cs_export(self.h, C.c_int(m), _p(rp), _p(ci), _p(co))
            out.append((rp, ci[:nnz], co[:nnz]))
        return out

    def is_satisfied(self, witness):
        return bool(lib().ref_r1cs_is_satisfied(self.h, _p(_u64(witness))))

    def witness_map(self, witness, want_abc=False):
        m = self.m
        H = np.zeros((m, 4), np.uint64)
        if want_abc:
            Az, Bz, Cz = (np.zeros((m, 4), np.uint64) for _ in range(3))
            lib().ref_witness_map(self.h, _p(_u64(witness)), _p(H), _p(Az), _p(Bz), _p(Cz))
            return H, Az, Bz, Cz
        lib().ref_witness_map(self.h, _p(_u64(witness)), _p(H), None, None, None)
        return H

    def key_scalars(self, toxic):
        """The generator's exponents (canonical Fr) for (A_query, B_query, H_query, L_query, gamma_ABC)."""
        m = self.m
        nv, ni = self.num_vars, self.num_inputs
        A = np.zeros((nv + 1, 4), np.uint64); B = np.zeros((nv + 1, 4), np.uint64)
        H = np.zeros((m - 1, 4), np.uint64); Lq = np.zeros((nv - ni, 4), np.uint64); ABC = np.zeros((ni + 1, 4), np.uint64)
        lib().ref_groth16_key_scalars(self.h, _p(_u64(toxic)), _p(A), _p(B), _p(H), _p(Lq), _p(ABC))
        return dict(A=A, B=B, H=H, L=Lq, ABC=ABC)

    def free(self):
        if self.h:
            lib().ref_r1cs_free(self.h); self.h = None


KEY_PARTS = {"A_query": (0, 12), "B_query_g1": (1, 12), "B_query_g2": (2, 24), "H_query": (3, 12),
             "L_query": (4, 12), "gamma_ABC_g1": (5, 12), "alpha_g1": (6, 12), "beta_g1": (7, 12),
             "delta_g1": (8, 12), "beta_g2": (9, 24), "delta_g2": (10, 24), "gamma_g2": (11, 24), "gamma_g1": (12, 12)}


class Keypair:
    def __init__(self, cs, toxic):
        """toxic: uint64[5,4] = (t, alpha, beta, gamma, delta), canonical Fr."""
        self.cs = cs
        self.h = C.c_void_p(lib().ref_groth16_generate(cs.h, _p(_u64(toxic))))

    def part(self, name):
        which, width = KEY_PARTS[name]
        n = lib().ref_keypair_count(self.h, C.c_int(which))
        out = np.zeros((n, width), np.uint64)
        lib().ref_keypair_export(self.h, C.c_int(which), _p(out))
        return out

    def prove(self, witness, r, s, P1=None, r_enc=None):
        A = np.zeros(12, np.uint64); B = np.zeros(24, np.uint64); Cc = np.zeros(12, np.uint64)
        lib().ref_groth16_prove(self.cs.h, self.h, _p(_u64(witness)), _p(_u64(r)), _p(_u64(s)),
                                _p(None if P1 is None else _u64(P1)), _p(None if r_enc is None else _u64(r_enc)),
                                _p(A), _p(B), _p(Cc))
        return A, B, Cc

    def free(self):
        if self.h:
            lib().ref_keypair_free(self.h); self.h = None


# ---- SAVER (elgamal_verifiable around the prover): the C restatement; oracle/saver.py is the big-int one with the pairing checks
def saver_pk_words(n):
    lib().ref_saver_pk_words.restype = C.c_size_t
    return lib().ref_saver_pk_words(C.c_size_t(n))


def saver_vk_words(n):
    lib().ref_saver_vk_words.restype = C.c_size_t
    return lib().ref_saver_vk_words(C.c_size_t(n))


def saver_keygen(n, delta_g1, gamma_g1, gamma_abc, rnd):
    pk = np.zeros(saver_pk_words(n), np.uint64); sk = np.zeros(4, np.uint64); vk = np.zeros(saver_vk_words(n), np.uint64)
    lib().ref_saver_keygen(C.c_size_t(n), _p(_u64(delta_g1)), _p(_u64(gamma_g1)), _p(_u64(gamma_abc)), _p(_u64(rnd)), _p(pk), _p(sk), _p(vk))
    return pk, sk, vk


def saver_encrypt_ct(n, pk, gamma_abc, msg, r):
    ct = np.zeros((n + 2, 12), np.uint64)
    lib().ref_saver_encrypt_ct(C.c_size_t(n), _p(_u64(pk)), _p(_u64(gamma_abc)), _p(_u64(msg)), _p(_u64(r)), _p(ct))
    return ct


def saver_rerandomize(n, pk, delta_g2, rnd3, ct, A, B, Cc):
    ct, A, B, Cc = (_u64(x).copy() for x in (ct, A, B, Cc))
    lib().ref_saver_rerandomize(C.c_size_t(n), _p(_u64(pk)), _p(_u64(delta_g2)), _p(_u64(rnd3)), _p(ct), _p(A), _p(B), _p(Cc))
    return ct.reshape(n + 2, 12), A, B, Cc
