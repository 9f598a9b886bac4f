"""Host-side mirror of the reference's interface for the prover hot path, on top of the C ABI.

The reference exposes this path as C++ templates (absent crypto3 submodules):
    algebra::multiexp<multiexp_method_BDLO12>(bases_begin, bases_end, scalars_begin, scalars_end, chunks)
    algebra::multiexp_with_mixed_addition<...>(...)
    math::make_evaluation_domain<Fr>(m) -> evaluation_domain { m, fft, inverse_fft, ... }
    zk::snark::r1cs_gg_ppzksnark prover (reached from bin/cli/include/nil/vote_saver/common.hpp:1132-1135)
This module keeps the same names and argument meaning with numpy arrays of canonical little-endian
uint64 limbs (Fr: [n,4], G1 affine: [n,12], G2 affine: [n,24]; infinity = all zero).  Every call goes
through libvsp_hip.so; nothing here computes field or curve arithmetic on the CPU.
"""
import ctypes as C

import numpy as np

from . import _lib


class VspError(RuntimeError):
    pass


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, "data_ptr"):          # torch tensor (device or host memory)
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


def _u64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


class Context:
    """One GPU + one HIP stream + grow-only device workspaces (vsp_ctx)."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        self.h = self.lib.vsp_create(int(device))
        if not self.h:
            raise VspError(f"vsp_create({device}) failed: no such HIP device (this path has no CPU fallback)")
        self.device = device

    def close(self):
        if self.h:
            self.lib.vsp_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc):
        if rc != 0:
            raise VspError(f"vsp error {rc}: {self.lib.vsp_last_error(self.h).decode()}")

    def last_error(self):
        return self.lib.vsp_last_error(self.h).decode()

    def set_stream(self, stream_handle):
        self.check(self.lib.vsp_set_stream(self.h, C.c_void_p(stream_handle) if stream_handle else None))

    def synchronize(self):
        self.check(self.lib.vsp_synchronize(self.h))

    def stat(self, name):
        return self.lib.vsp_get_stat(self.h, name.encode())

    def stats_reset(self):
        self.lib.vsp_stats_reset(self.h)

    def set_option(self, name, value):
        self.check(self.lib.vsp_set_option(self.h, name.encode(), int(value)))

    # ---- raw device memory
    def dmalloc(self, nbytes):
        p = self.lib.vsp_dmalloc(self.h, nbytes)
        if not p:
            raise VspError("vsp_dmalloc failed: " + self.last_error())
        return p

    def dfree(self, p):
        self.lib.vsp_dfree(self.h, C.c_void_p(p))

    def h2d(self, dptr, host):
        host = np.ascontiguousarray(host)
        self.check(self.lib.vsp_h2d(self.h, C.c_void_p(dptr), _ptr(host), host.nbytes))

    def d2h(self, host, dptr):
        self.check(self.lib.vsp_d2h(self.h, _ptr(host), C.c_void_p(dptr), host.nbytes))

    def host_register(self, arr):
        """Page-lock a host array the library copies from on every call (the witness): its transfers become asynchronous DMA."""
        self.check(self.lib.vsp_host_register(self.h, _ptr(arr), arr.nbytes))
        return arr

    def host_unregister(self, arr):
        self.check(self.lib.vsp_host_unregister(self.h, _ptr(arr)))

    def to_device(self, host):
        host = np.ascontiguousarray(host)
        p = self.dmalloc(max(host.nbytes, 16))
        self.h2d(p, host)
        return p

    # ---- bases
    def upload_bases(self, bases, group=1):
        cols = 12 if group == 1 else 24
        bases = _u64(bases, cols)
        fn = self.lib.vsp_bases_upload_g1 if group == 1 else self.lib.vsp_bases_upload_g2
        h = fn(self.h, _ptr(bases), bases.shape[0])
        if not h:
            raise VspError("bases upload failed: " + self.last_error())
        return Bases(self, h, group)

    def bases_from_device(self, dptr, n, group=1):
        fn = self.lib.vsp_bases_from_device_g1 if group == 1 else self.lib.vsp_bases_from_device_g2
        h = fn(self.h, _ptr(dptr), n)
        if not h:
            raise VspError("bases_from_device failed: " + self.last_error())
        return Bases(self, h, group)


class Bases:
    """Device-resident MSM bases (a proving-key query), vsp_bases."""

    def __init__(self, ctx, handle, group):
        self.ctx, self.h, self.group = ctx, handle, group
        self.n = ctx.lib.vsp_bases_count(handle)

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.vsp_bases_free(self.ctx.h, self.h)
        self.h = None

    def precompute(self, window_bits=0, split=False):
        """Store the window multiples 2^(c*w) * P once (16x memory at c = 16); later multi-exponentiations share one bucket set.
        split=True (bases whose scalars are dense, e.g. the H query): the table carries the endomorphism rows for split scalars."""
        fn = self.ctx.lib.vsp_bases_precompute_split if split else self.ctx.lib.vsp_bases_precompute
        self.ctx.check(fn(self.ctx.h, self.h, window_bits))
        return self

    def msm(self, d_scalars, n=None, first=0):
        """sum_i scalars[i] * bases[first+i]; d_scalars is a device pointer / torch tensor.  -> (affine, is_inf)."""
        n = self.n - first if n is None else n
        out = np.zeros(12 if self.group == 1 else 24, np.uint64)
        inf = C.c_int(0)
        self.ctx.check(self.ctx.lib.vsp_msm_resident(self.ctx.h, self.h, first, n, _ptr(d_scalars), _ptr(out), C.byref(inf)))
        return out, bool(inf.value)

    def msm_batch(self, d_scalars, batch, n=None, first=0, stride=None):
        """`batch` multi-exponentiations over the same bases in one pass (vsp_msm_resident_batch): vector k at d_scalars + 32 * k * stride
        bytes.  -> (affine [batch, 12 | 24], is_inf [batch])."""
        n = self.n - first if n is None else n
        stride = n if stride is None else stride
        out = np.zeros((batch, 12 if self.group == 1 else 24), np.uint64)
        inf = np.zeros(batch, np.int32)
        self.ctx.check(self.ctx.lib.vsp_msm_resident_batch(self.ctx.h, self.h, first, n, _ptr(d_scalars), batch, stride, _ptr(out), _ptr(inf)))
        return out, inf.astype(bool)

    def msm_launch(self, slot, d_scalars, n=None, first=0):
        """Enqueue the multi-exponentiation on work slot `slot` (own stream); pair with msm_finish_jacobian(slot)."""
        n = self.n - first if n is None else n
        self.ctx.check(self.ctx.lib.vsp_msm_launch(self.ctx.h, slot, self.h, first, n, _ptr(d_scalars)))

    def msm_finish_jacobian(self, slot):
        out = np.zeros(18 if self.group == 1 else 36, np.uint64)
        self.ctx.check(self.ctx.lib.vsp_msm_finish_jacobian(self.ctx.h, slot, _ptr(out)))
        return out

    def msm_finish_jacobian_device(self, slot, d_out, stream=None):
        """Same, the record left in DEVICE memory at d_out (pointer / torch tensor) by an asynchronous copy queued on `stream`
        (a hipStream_t handle; None = the context's stream) -- where the RCCL all-gather of the sharded MSM reads it."""
        self.ctx.check(self.ctx.lib.vsp_msm_finish_jacobian_device(self.ctx.h, slot, _ptr(d_out), C.c_void_p(stream) if stream else None))

    def msm_jacobian(self, d_scalars, n=None, first=0):
        """Same, result as the Jacobian partial-sum record (18 / 36 uint64) ranks exchange."""
        n = self.n - first if n is None else n
        out = np.zeros(18 if self.group == 1 else 36, np.uint64)
        self.ctx.check(self.ctx.lib.vsp_msm_resident_jacobian(self.ctx.h, self.h, first, n, _ptr(d_scalars), _ptr(out)))
        return out


# ---- multiexp (a1-a3) ------------------------------------------------------------------------
def multiexp(ctx, bases, scalars, group=1):
    """algebra::multiexp<multiexp_method_BDLO12>: sum_i scalars[i] * bases[i] as an affine point
    (zero array = infinity).  bases [n,12|24], scalars [n,4], host arrays."""
    cols = 12 if group == 1 else 24
    bases, scalars = _u64(bases, cols), _u64(scalars, 4)
    if bases.shape[0] != scalars.shape[0]:
        raise ValueError("multiexp: bases and scalars differ in length")   # the reference asserts equal ranges
    out = np.zeros(cols, np.uint64)
    inf = C.c_int(0)
    fn = ctx.lib.vsp_msm_g1 if group == 1 else ctx.lib.vsp_msm_g2
    ctx.check(fn(ctx.h, _ptr(bases), _ptr(scalars), bases.shape[0], _ptr(out), C.byref(inf)))
    return out


def multiexp_with_mixed_addition(ctx, bases, scalars, group=1):
    """algebra::multiexp_with_mixed_addition: same value; the 0 / 1 scalar special cases are handled
    inside the bucket pipeline (zero digits are skipped, the ones share one bucket)."""
    return multiexp(ctx, bases, scalars, group)


def fold_jacobian(ctx, records, group=1):
    """Fold Jacobian partial sums (the multi-GPU exchange records) into one affine point."""
    cols = 18 if group == 1 else 36
    records = _u64(records, cols)
    out = np.zeros(12 if group == 1 else 24, np.uint64)
    inf = C.c_int(0)
    ctx.check(ctx.lib.vsp_fold_jacobian(ctx.h, group, _ptr(records), records.shape[0], _ptr(out), C.byref(inf)))
    return out


def fold_jacobian_device(ctx, d_records, count, group=1, stream=None):
    """Fold `count` records that sit in device memory (the all-gather's output) behind the work already queued on `stream`."""
    out = np.zeros(12 if group == 1 else 24, np.uint64)
    inf = C.c_int(0)
    ctx.check(ctx.lib.vsp_fold_jacobian_device(ctx.h, group, _ptr(d_records), count, C.c_void_p(stream) if stream else None, _ptr(out), C.byref(inf)))
    return out


# ---- evaluation_domain (a6) -------------------------------------------------------------------
class EvaluationDomain:
    """math::evaluation_domain<Fr>: the basic radix-2 domain (m a power of two) or the step radix-2 domain (m = big + small,
    both powers of two).  Methods take and return host arrays [m,4] of canonical Fr; *_device variants work in place on device
    memory.  Method names are upstream's (crypto3-math domains/evaluation_domain.hpp); the libfqfft spellings are aliases."""

    def __init__(self, ctx, m, _min_size=None):
        self.ctx, self.h = ctx, None
        if m == 1 and _min_size is None:          # degenerate size kept for completeness: every transform is the identity
            self.m, self.kind = 1, "basic_radix2"
            return
        self.h = ctx.lib.vsp_domain_create(ctx.h, m if _min_size is None else _min_size)
        if not self.h:
            raise ValueError("evaluation_domain: " + ctx.last_error())          # upstream: std::invalid_argument
        self.m = ctx.lib.vsp_domain_size(self.h)
        self.kind = ("basic_radix2", "step_radix2")[ctx.lib.vsp_domain_kind(self.h)]
        if _min_size is None and self.m != m:
            self.free()
            raise ValueError("evaluation_domain: %d is neither a power of two nor big + small with both powers of two" % m)

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.vsp_domain_free(self.ctx.h, self.h)
        self.h = None

    def _vec(self, a, rows=None):
        a = _u64(a, 4).copy()
        if a.shape[0] != (self.m if rows is None else rows):
            raise ValueError("evaluation_domain: expected vector of size m")     # upstream: std::invalid_argument
        return a

    def _run(self, a, inverse, coset):
        a = self._vec(a)
        g = None if coset is None else _u64(coset)
        if self.h is None:
            self.ctx.check(self.ctx.lib.vsp_ntt_fr(self.ctx.h, _ptr(a), 0, int(inverse), _ptr(g)))
        else:
            self.ctx.check(self.ctx.lib.vsp_domain_fft(self.ctx.h, self.h, _ptr(a), int(inverse), _ptr(g)))
        return a

    def fft(self, a): return self._run(a, False, None)
    def inverse_fft(self, a): return self._run(a, True, None)
    def coset_fft(self, a, g): return self._run(a, False, g)
    def inverse_coset_fft(self, a, g): return self._run(a, True, g)
    cosetFFT = coset_fft
    icosetFFT = inverse_coset_fft
    iFFT = inverse_fft
    FFT = fft

    def fft_device(self, d_a, inverse=False, coset=None):
        g = None if coset is None else _u64(coset)
        if self.h is None:
            self.ctx.check(self.ctx.lib.vsp_ntt_fr_device(self.ctx.h, _ptr(d_a), 0, int(inverse), _ptr(g)))
        else:
            self.ctx.check(self.ctx.lib.vsp_domain_fft_device(self.ctx.h, self.h, _ptr(d_a), int(inverse), _ptr(g)))

    def evaluate_all_lagrange_polynomials(self, t):
        out = np.zeros((self.m, 4), np.uint64)
        self.ctx.check(self.ctx.lib.vsp_domain_lagrange(self.ctx.h, self.h, _ptr(_u64(t)), _ptr(out)))
        return out

    def get_domain_element(self, idx):
        out = np.zeros(4, np.uint64)
        self.ctx.check(self.ctx.lib.vsp_domain_element(self.ctx.h, self.h, idx, _ptr(out)))
        return out

    def compute_vanishing_polynomial(self, t):
        out = np.zeros(4, np.uint64)
        self.ctx.check(self.ctx.lib.vsp_domain_vanishing(self.ctx.h, self.h, _ptr(_u64(t)), _ptr(out)))
        return out

    def add_poly_z(self, coeff, H):
        H = self._vec(H, self.m + 1)
        self.ctx.check(self.ctx.lib.vsp_domain_add_poly_z(self.ctx.h, self.h, _ptr(_u64(coeff)), _ptr(H)))
        return H

    def divide_by_z_on_coset(self, P):
        P = self._vec(P)
        self.ctx.check(self.ctx.lib.vsp_domain_divide_by_z_on_coset(self.ctx.h, self.h, _ptr(P)))
        return P

    def witness_map_h(self, Az, Bz, Cz):
        Az, Bz, Cz = (self._vec(x) for x in (Az, Bz, Cz))
        H = np.zeros((self.m, 4), np.uint64)
        self.ctx.check(self.ctx.lib.vsp_domain_witness_map_h(self.ctx.h, self.h, _ptr(Az), _ptr(Bz), _ptr(Cz), _ptr(H)))
        return H


def BasicRadix2Domain(ctx, m):
    if m < 1 or m & (m - 1):
        raise ValueError("basic_radix2_domain: m must be a power of two")       # upstream throws for other sizes
    return EvaluationDomain(ctx, m)


def StepRadix2Domain(ctx, m):
    d = EvaluationDomain(ctx, m)
    if d.kind != "step_radix2":
        d.free()
        raise ValueError("step_radix2_domain: m must be big + small with small < big, both powers of two")
    return d


def make_evaluation_domain(ctx, min_size):
    """math::make_evaluation_domain<Fr>(k): the first of basic_radix2(k), step_radix2(k), basic_radix2(big + rounded_small),
    step_radix2(big + rounded_small) that exists -- the domain r1cs_to_qap works over for k = constraints + inputs + 1."""
    return EvaluationDomain(ctx, None, _min_size=min_size)


def witness_map_h(ctx, Az, Bz, Cz):
    """r1cs_to_qap::witness_map tail (d1=d2=d3=0): coefficients of H from the evaluation vectors over the domain whose size the
    vectors have."""
    m = _u64(Az, 4).shape[0]
    if m == 1:
        Az, Bz, Cz = (_u64(x, 4).copy() for x in (Az, Bz, Cz))
        H = np.zeros((1, 4), np.uint64)
        ctx.check(ctx.lib.vsp_witness_map_h(ctx.h, _ptr(Az), _ptr(Bz), _ptr(Cz), 0, _ptr(H)))
        return H
    d = EvaluationDomain(ctx, m)
    try:
        return d.witness_map_h(Az, Bz, Cz)
    finally:
        d.free()


# ---- Groth16 prover (a8, a9) ---------------------------------------------------------------------
class R1CS:
    """Device-resident constraint system: three CSR matrices (row_ptr u32, col u32, coeff [nnz,4] u64)."""

    def __init__(self, ctx, num_constraints, num_inputs, num_vars, A, B, Cm):
        self.ctx = ctx
        self.num_constraints, self.num_inputs, self.num_vars = num_constraints, num_inputs, num_vars
        args = []
        self._keep = []
        for rp, ci, co in (A, B, Cm):
            rp = np.ascontiguousarray(rp, dtype=np.uint32); ci = np.ascontiguousarray(ci, dtype=np.uint32); co = _u64(co, 4)
            if rp.shape[0] != num_constraints + 1 or ci.shape[0] != rp[-1] or co.shape[0] != rp[-1]:
                raise ValueError("R1CS: malformed CSR")
            self._keep += [rp, ci, co]
            args += [_ptr(rp), _ptr(ci), _ptr(co)]
        self.h = ctx.lib.vsp_r1cs_upload(ctx.h, num_constraints, num_inputs, num_vars, *args)
        if not self.h:
            raise VspError("r1cs upload failed: " + ctx.last_error())
        self._keep = None
        self.m = ctx.lib.vsp_r1cs_domain_size(self.h)              # make_evaluation_domain(num_constraints + num_inputs + 1)
        self.domain_kind = ("basic_radix2", "step_radix2")[ctx.lib.vsp_r1cs_domain_kind(self.h)]

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.vsp_r1cs_free(self.ctx.h, self.h)
        self.h = None


class ProvingKey:
    """r1cs_gg_ppzksnark proving key with device-resident queries."""

    def __init__(self, ctx, alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2, A_query, B_query_g1, B_query_g2, H_query, L_query):
        self.ctx = ctx
        self.queries = (A_query, B_query_g1, B_query_g2, H_query, L_query)
        consts = [_u64(x) for x in (alpha_g1, beta_g1, beta_g2, delta_g1, delta_g2)]
        self.h = ctx.lib.vsp_pk_create(ctx.h, *[_ptr(c) for c in consts], *[q.h for q in self.queries])
        if not self.h:
            raise VspError("pk_create failed: " + ctx.last_error())

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.vsp_pk_free(self.ctx.h, self.h)
        self.h = None


KEY_PARTS = {"A_query": (0, 12), "B_query_g1": (1, 12), "B_query_g2": (2, 24), "H_query": (3, 12), "L_query": (4, 12),
             "gamma_ABC_g1": (5, 12), "alpha_g1": (6, 12), "beta_g1": (7, 12), "delta_g1": (8, 12), "beta_g2": (9, 24),
             "delta_g2": (10, 24), "gamma_g2": (11, 24), "gamma_g1": (12, 12)}


class Keypair:
    """zk::generate<proof_system>(constraint_system) on the GPU with explicit toxic waste [5,4] = (t, alpha, beta, gamma, delta)."""

    def __init__(self, ctx, cs, toxic, precompute=False, precompute_window=0, _handle=None):
        """precompute: False / 0 = a plain key; True / 1 = tables of window multiples on A, B(G1), B(G2), L; else a bit mask (vsp.h; 17 = all
        five queries).  precompute_window: the tables' window in bits (0 = by the query's size).  For batched proving at the real circuit's
        size (2^15..2^16 constraints): precompute=17, precompute_window=14."""
        self.ctx = ctx
        if _handle is None:
            toxic = _u64(toxic).reshape(20)
            if precompute_window:
                ctx.set_option("generate_precompute_window", int(precompute_window))
            try:
                _handle = ctx.lib.vsp_groth16_generate(ctx.h, cs.h, _ptr(toxic), int(precompute))
            finally:
                if precompute_window:
                    ctx.set_option("generate_precompute_window", 0)
            if not _handle:
                raise VspError("groth16_generate failed: " + ctx.last_error())
        self.h = _handle
        self.pk = _BorrowedPk(ctx.lib.vsp_keypair_pk(self.h))

    @classmethod
    def from_blob(cls, ctx, blob, precompute=False):
        """deserialize_pk_crs (common.hpp:749-754): the big-endian "fast" proving-key blob -> a resident key, converted on the GPU."""
        buf = np.frombuffer(bytes(blob), dtype=np.uint8)
        h = ctx.lib.vsp_pk_from_blob(ctx.h, _ptr(buf), buf.shape[0], int(bool(precompute)))
        if not h:
            raise VspError("pk_from_blob failed: " + ctx.last_error())
        return cls(ctx, None, None, _handle=h)

    def to_blob(self):
        out = np.zeros(self.ctx.lib.vsp_pk_blob_size(self.h), np.uint8)
        self.ctx.check(self.ctx.lib.vsp_pk_to_blob(self.ctx.h, self.h, _ptr(out)))
        return out.tobytes()

    def device_bytes(self):
        """device memory the key's six queries hold (points, tables of window multiples, 28-bit-limb copies)"""
        return self.ctx.lib.vsp_keypair_device_bytes(self.h)

    def part(self, name):
        which, width = KEY_PARTS[name]
        n = self.ctx.lib.vsp_keypair_count(self.h, which)
        out = np.zeros((n, width), np.uint64)
        self.ctx.check(self.ctx.lib.vsp_keypair_export(self.ctx.h, self.h, which, _ptr(out)))
        return out

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.vsp_keypair_free(self.ctx.h, self.h)
        self.h = None


class _BorrowedPk:
    def __init__(self, handle):
        self.h = handle


def groth16_prove(ctx, cs, pk, witness, r, s, saver_P1=None, saver_r_enc=None):
    """r1cs_gg_ppzksnark_prover::process with explicit r, s.  -> (A[12], B[24], C[12], proof_bytes[192])."""
    witness = _u64(witness, 4)
    if witness.shape[0] != cs.num_vars:
        raise ValueError("prove: witness must have num_vars entries (primary || auxiliary)")
    A = np.zeros(12, np.uint64); B = np.zeros(24, np.uint64); Cc = np.zeros(12, np.uint64)
    proof = np.zeros(192, np.uint8)
    p1 = None if saver_P1 is None else _u64(saver_P1)
    re = None if saver_r_enc is None else _u64(saver_r_enc)
    ctx.check(ctx.lib.vsp_groth16_prove(ctx.h, cs.h, pk.h, _ptr(witness), _ptr(_u64(r)), _ptr(_u64(s)), _ptr(p1), _ptr(re),
                                        _ptr(A), _ptr(B), _ptr(Cc), _ptr(proof)))
    return A, B, Cc, proof.tobytes()


def groth16_prove_batch(ctx, cs, pk, witnesses, r, s):
    """K proofs over one PLAIN key in one pass (vsp_groth16_prove_batch): witnesses [K, num_vars, 4], r / s [K, 4].
    -> (A [K, 12], B [K, 24], C [K, 12], [proof bytes] * K); proof k is byte-identical to groth16_prove(witness_k, r_k, s_k)."""
    witnesses = np.ascontiguousarray(witnesses, dtype=np.uint64)
    if witnesses.ndim != 3 or witnesses.shape[1:] != (cs.num_vars, 4):
        raise ValueError("prove_batch: witnesses must be [K, num_vars, 4]")
    K = witnesses.shape[0]
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(K, 4); s = np.ascontiguousarray(s, dtype=np.uint64).reshape(K, 4)
    A = np.zeros((K, 12), np.uint64); B = np.zeros((K, 24), np.uint64); Cc = np.zeros((K, 12), np.uint64)
    proofs = np.zeros((K, 192), np.uint8)
    ctx.check(ctx.lib.vsp_groth16_prove_batch(ctx.h, cs.h, pk.h, _ptr(witnesses), K, _ptr(r), _ptr(s), _ptr(A), _ptr(B), _ptr(Cc), _ptr(proofs)))
    return A, B, Cc, [proofs[k].tobytes() for k in range(K)]


def groth16_prove_batch_launch(ctx, cs, pk, witnesses, r, s):
    """first half of groth16_prove_batch: copy the witnesses, queue every kernel of the K proofs, return K (one batch in flight per context)"""
    witnesses = np.ascontiguousarray(witnesses, dtype=np.uint64)
    if witnesses.ndim != 3 or witnesses.shape[1:] != (cs.num_vars, 4):
        raise ValueError("prove_batch: witnesses must be [K, num_vars, 4]")
    K = witnesses.shape[0]
    r = np.ascontiguousarray(r, dtype=np.uint64).reshape(K, 4); s = np.ascontiguousarray(s, dtype=np.uint64).reshape(K, 4)
    ctx.check(ctx.lib.vsp_groth16_prove_batch_launch(ctx.h, cs.h, pk.h, _ptr(witnesses), K, _ptr(r), _ptr(s)))
    ctx._prove_batch_count = K
    return K


def groth16_prove_batch_finish(ctx):
    """second half: -> (A [K, 12], B [K, 24], C [K, 12], [proof bytes] * K) of the batch of K launched on this context"""
    K = getattr(ctx, "_prove_batch_count", 0)
    if not K:
        raise RuntimeError("prove_batch_finish: no batch in flight on this context")
    ctx._prove_batch_count = 0
    A = np.zeros((K, 12), np.uint64); B = np.zeros((K, 24), np.uint64); Cc = np.zeros((K, 12), np.uint64)
    proofs = np.zeros((K, 192), np.uint8)
    ctx.check(ctx.lib.vsp_groth16_prove_batch_finish(ctx.h, _ptr(A), _ptr(B), _ptr(Cc), _ptr(proofs)))
    return A, B, Cc, [proofs[k].tobytes() for k in range(K)]


class PackedWitness:
    """the witness in the packed form of vsp_witness_pack: two class bits per wire, per-word offsets, the dense values"""

    def __init__(self, witness):
        witness = _u64(witness, 4)
        lib = _lib.load()
        self.num_vars = witness.shape[0]
        words = lib.vsp_witness_pack_words(self.num_vars)
        self.class_words = np.zeros(words, np.uint64); self.word_offsets = np.zeros(words, np.uint32)
        n = C.c_size_t(0)
        if lib.vsp_witness_pack(_ptr(witness), self.num_vars, _ptr(self.class_words), _ptr(self.word_offsets), None, 0, C.byref(n)) != 0:
            raise ValueError("witness_pack failed")
        self.dense = np.zeros((max(n.value, 1), 4), np.uint64)
        if lib.vsp_witness_pack(_ptr(witness), self.num_vars, _ptr(self.class_words), _ptr(self.word_offsets), _ptr(self.dense), n.value, C.byref(n)) != 0:
            raise ValueError("witness_pack failed")
        self.n_dense = n.value

    @property
    def nbytes(self):
        return self.class_words.nbytes + self.word_offsets.nbytes + self.n_dense * 32


def groth16_prove_launch(ctx, cs, pk, witness, r, s, saver_P1=None, saver_r_enc=None):
    """first half of groth16_prove: queue the whole proof on the context's streams and return (one proof in flight per context).
    witness: an array [num_vars, 4] (keep it alive and unchanged until the finish when it is page-locked) or a PackedWitness."""
    p1 = None if saver_P1 is None else _u64(saver_P1)
    re = None if saver_r_enc is None else _u64(saver_r_enc)
    r, s = _u64(r), _u64(s)
    if isinstance(witness, PackedWitness):
        if witness.num_vars != cs.num_vars:
            raise ValueError("prove: witness must have num_vars entries (primary || auxiliary)")
        ctx.check(ctx.lib.vsp_groth16_prove_launch_packed(ctx.h, cs.h, pk.h, _ptr(witness.class_words), _ptr(witness.word_offsets), _ptr(witness.dense),
                                                          witness.n_dense, _ptr(r), _ptr(s), _ptr(p1), _ptr(re)))
        return
    witness = _u64(witness, 4)
    if witness.shape[0] != cs.num_vars:
        raise ValueError("prove: witness must have num_vars entries (primary || auxiliary)")
    ctx.check(ctx.lib.vsp_groth16_prove_launch(ctx.h, cs.h, pk.h, _ptr(witness), _ptr(r), _ptr(s), _ptr(p1), _ptr(re)))


def groth16_prove_finish(ctx):
    """second half: host-side assembly work, the wait, the proof.  -> (A[12], B[24], C[12], proof_bytes[192])"""
    A = np.zeros(12, np.uint64); B = np.zeros(24, np.uint64); Cc = np.zeros(12, np.uint64)
    proof = np.zeros(192, np.uint8)
    ctx.check(ctx.lib.vsp_groth16_prove_finish(ctx.h, _ptr(A), _ptr(B), _ptr(Cc), _ptr(proof)))
    return A, B, Cc, proof.tobytes()


# ---- SAVER wrapper (f.3): elgamal_verifiable over BLS12-381 around the prover ---------------------------------------------
def saver_generate_keypair(ctx, rnd, gamma_abc_g1, delta_g1, gamma_g1, msg_size):
    """generate_keypair<elgamal_verifiable>(rnd, {gg_keypair, msg_size}) (common.hpp:921-931) with the 3 * msg_size + 2 random scalars
    explicit.  -> (public key words, secret key rho [4], verification key words)"""
    n = int(msg_size)
    rnd = _u64(rnd, 4)
    if rnd.shape[0] != 3 * n + 2:
        raise ValueError("generate_keypair: 3 * msg_size + 2 random values are needed")
    gabc = _u64(gamma_abc_g1, 12)
    if gabc.shape[0] < n + 1:
        raise ValueError("generate_keypair: the verification key has fewer than msg_size + 1 accumulation elements")
    pk = np.zeros(ctx.lib.vsp_saver_pk_words(n), np.uint64); sk = np.zeros(4, np.uint64); vk = np.zeros(ctx.lib.vsp_saver_vk_words(n), np.uint64)
    ctx.check(ctx.lib.vsp_saver_keygen(ctx.h, n, _ptr(_u64(delta_g1)), _ptr(_u64(gamma_g1)), _ptr(gabc), _ptr(rnd), _ptr(pk), _ptr(sk), _ptr(vk)))
    return pk, sk, vk


class SaverPublicKey:
    """elgamal_verifiable public key resident for many votes (fixed-base tables of its bases)."""

    def __init__(self, ctx, pk_words, gamma_abc_g1, msg_size):
        self.ctx, self.n = ctx, int(msg_size)
        self.words = _u64(pk_words).reshape(-1)
        gabc = _u64(gamma_abc_g1, 12)
        # the C side reads vsp_saver_pk_words(n) words and n + 1 accumulation elements unconditionally (the C ABI carries no lengths)
        if self.n < 1 or self.words.shape[0] != ctx.lib.vsp_saver_pk_words(self.n):
            raise ValueError("SaverPublicKey: the flat public key must have vsp_saver_pk_words(msg_size) words")
        if gabc.shape[0] < self.n + 1:
            raise ValueError("SaverPublicKey: the verification key has fewer than msg_size + 1 accumulation elements")
        self.h = ctx.lib.vsp_saver_pk_load(ctx.h, self.n, _ptr(self.words), _ptr(gabc))
        if not self.h:
            raise VspError("saver_pk_load failed: " + ctx.last_error())

    def free(self):
        if self.h and self.ctx.h:
            self.ctx.lib.vsp_saver_pk_free(self.ctx.h, self.h)
        self.h = None


def saver_encrypt(ctx, spk, cs, pk, msg, witness, r_enc, r, s):
    """encrypt<elgamal_verifiable, verifiable_encryption>(m, {r_enc, pk_eid, gg_keypair, primary, auxiliary}) (common.hpp:1131-1135):
    -> (ciphertext [msg_size + 2, 12], (A, B, C), proof bytes)"""
    witness = _u64(witness, 4); msg = _u64(msg, 4)
    if witness.shape[0] != cs.num_vars or msg.shape[0] != spk.n:
        raise ValueError("encrypt: witness must have num_vars entries and the message msg_size blocks")
    ct = np.zeros((spk.n + 2, 12), np.uint64)
    A = np.zeros(12, np.uint64); B = np.zeros(24, np.uint64); Cc = np.zeros(12, np.uint64); proof = np.zeros(192, np.uint8)
    ctx.check(ctx.lib.vsp_saver_encrypt(ctx.h, spk.h, cs.h, pk.h, _ptr(msg), _ptr(witness), _ptr(_u64(r_enc)), _ptr(_u64(r)), _ptr(_u64(s)),
                                        _ptr(ct), _ptr(A), _ptr(B), _ptr(Cc), _ptr(proof)))
    return ct, (A, B, Cc), proof.tobytes()


def saver_rerandomize(ctx, spk, delta_g2, rnd3, ct, proof_abc):
    """rerandomize<elgamal_verifiable>(rnd[3], ct, {pk_eid, gg_keypair, proof}) (common.hpp:1138-1145), rnd3 = (r', z1, z2)."""
    ct = _u64(ct, 12).copy()
    A, B, Cc = (_u64(x).copy() for x in proof_abc)
    if ct.shape[0] != spk.n + 2 or A.size != 12 or B.size != 24 or Cc.size != 12 or _u64(delta_g2).size != 24 or _u64(rnd3).size != 12:
        raise ValueError("rerandomize: ciphertext of msg_size + 2 points, proof (A[12], B[24], C[12]), delta_g2[24] and three random scalars expected")
    proof = np.zeros(192, np.uint8)
    ctx.check(ctx.lib.vsp_saver_rerandomize(ctx.h, spk.h, _ptr(_u64(delta_g2)), _ptr(_u64(rnd3).reshape(12)), _ptr(ct), _ptr(A), _ptr(B), _ptr(Cc), _ptr(proof)))
    return ct, (A, B, Cc), proof.tobytes()


def fixed_base_mul(ctx, d_scalars, n, group=1):
    """Generator-side batch_exp: device array out[i] = scalars[i] * generator (canonical affine).  Returns a device pointer."""
    width = 96 if group == 1 else 192
    d_out = ctx.dmalloc(max(n, 1) * width)
    fn = ctx.lib.vsp_fixed_base_mul_g1 if group == 1 else ctx.lib.vsp_fixed_base_mul_g2
    ctx.check(fn(ctx.h, _ptr(d_scalars), n, C.c_void_p(d_out)))
    return d_out


def g1_compress(affine):
    out = np.zeros(48, np.uint8)
    if _lib.load().vsp_g1_compress(_ptr(_u64(affine).reshape(12)), _ptr(out)) != 0:
        raise ValueError("g1_compress: a coordinate is not canonical (>= p)")
    return out.tobytes()


def g2_compress(affine):
    out = np.zeros(96, np.uint8)
    if _lib.load().vsp_g2_compress(_ptr(_u64(affine).reshape(24)), _ptr(out)) != 0:
        raise ValueError("g2_compress: a coordinate is not canonical (>= p)")
    return out.tobytes()


def g1_decompress(data, check_subgroup=True):
    """48 ZCash-compressed bytes -> affine limbs [12] (all zero for infinity); ValueError for an invalid encoding."""
    buf = (C.c_uint8 * 48).from_buffer_copy(bytes(data))
    out = np.zeros(12, np.uint64); inf = C.c_int(0)
    if _lib.load().vsp_g1_decompress(buf, int(check_subgroup), _ptr(out), C.byref(inf)) != 0:
        raise ValueError("g1_decompress: invalid encoding")
    return out


def g2_decompress(data, check_subgroup=True):
    """96 ZCash-compressed bytes -> affine limbs [24] (all zero for infinity); ValueError for an invalid encoding."""
    buf = (C.c_uint8 * 96).from_buffer_copy(bytes(data))
    out = np.zeros(24, np.uint64); inf = C.c_int(0)
    if _lib.load().vsp_g2_decompress(buf, int(check_subgroup), _ptr(out), C.byref(inf)) != 0:
        raise ValueError("g2_decompress: invalid encoding")
    return out


# ---- wire formats (f.2): the big-endian blobs of the reference's marshaling_policy (common.hpp:168-203, 462-485, 749-799) ---------
def fr_vector_to_blob(vals):
    """serialize a scalar vector (primary input, eid, sn, rt, voting result): 8-byte count + 32-byte big-endian elements"""
    vals = _u64(vals, 4)
    out = np.zeros(_lib.load().vsp_fr_vector_blob_size(vals.shape[0]), np.uint8)
    if _lib.load().vsp_fr_vector_to_blob(_ptr(vals), vals.shape[0], _ptr(out)) != 0:
        raise ValueError("fr_vector_to_blob: value not canonical")
    return out.tobytes()


def fr_vector_from_blob(blob):
    """deserialize_scalar_vector (common.hpp:529-535)"""
    buf = np.frombuffer(bytes(blob), dtype=np.uint8); n = C.c_size_t(0)
    if _lib.load().vsp_fr_vector_from_blob(_ptr(buf), buf.shape[0], None, 0, C.byref(n)) != 0:
        raise ValueError("fr_vector_from_blob: malformed blob")
    out = np.zeros((n.value, 4), np.uint64)
    if _lib.load().vsp_fr_vector_from_blob(_ptr(buf), buf.shape[0], _ptr(out), n.value, C.byref(n)) != 0:
        raise ValueError("fr_vector_from_blob: element not canonical")
    return out


def g1_vector_to_blob(pts):
    """serialize the ciphertext (common.hpp:471-474): 8-byte count + compressed G1 points"""
    pts = _u64(pts, 12)
    out = np.zeros(_lib.load().vsp_g1_vector_blob_size(pts.shape[0]), np.uint8)
    if _lib.load().vsp_g1_vector_to_blob(_ptr(pts), pts.shape[0], _ptr(out)) != 0:
        raise ValueError("g1_vector_to_blob failed")
    return out.tobytes()


def g1_vector_from_blob(blob, check_subgroup=True):
    """deserialize_ct (common.hpp:773-781)"""
    buf = np.frombuffer(bytes(blob), dtype=np.uint8); n = C.c_size_t(0)
    if _lib.load().vsp_g1_vector_from_blob(_ptr(buf), buf.shape[0], int(check_subgroup), None, 0, C.byref(n)) != 0:
        raise ValueError("g1_vector_from_blob: malformed blob")
    out = np.zeros((n.value, 12), np.uint64)
    if _lib.load().vsp_g1_vector_from_blob(_ptr(buf), buf.shape[0], int(check_subgroup), _ptr(out), n.value, C.byref(n)) != 0:
        raise ValueError("g1_vector_from_blob: invalid point")
    return out


def proof_from_blob(blob, check_subgroup=True):
    buf = np.frombuffer(bytes(blob), dtype=np.uint8)
    if buf.shape[0] != 192:
        raise ValueError("proof_from_blob: a proof is 192 bytes")
    A = np.zeros(12, np.uint64); B = np.zeros(24, np.uint64); Cc = np.zeros(12, np.uint64)
    if _lib.load().vsp_proof_from_blob(_ptr(buf), int(check_subgroup), _ptr(A), _ptr(B), _ptr(Cc)) != 0:
        raise ValueError("proof_from_blob: invalid encoding")
    return A, B, Cc


def vk_to_blob(gt_bytes, gamma_g2, delta_g2, delta_g1, gamma_abc_g1, gamma_g1, head=0):
    """extended verification key; gt_bytes = alpha_g1_beta_g2 as 576 opaque bytes (12 little-endian Fp, computed by whoever has the pairing)"""
    gabc = _u64(gamma_abc_g1, 12); gt = np.frombuffer(bytes(gt_bytes), dtype=np.uint8)
    if gt.shape[0] != 576:
        raise ValueError("vk_to_blob: alpha_g1_beta_g2 is 576 bytes")
    out = np.zeros(_lib.load().vsp_vk_blob_size(gabc.shape[0]), np.uint8)
    if _lib.load().vsp_vk_to_blob(int(head), _ptr(gt), _ptr(_u64(gamma_g2)), _ptr(_u64(delta_g2)), _ptr(_u64(delta_g1)), _ptr(gabc), gabc.shape[0],
                                  _ptr(_u64(gamma_g1)), _ptr(out)) != 0:
        raise ValueError("vk_to_blob failed")
    return out.tobytes()


def vk_from_blob(blob, check_subgroup=True):
    buf = np.frombuffer(bytes(blob), dtype=np.uint8); n = C.c_size_t(0)
    lib = _lib.load()
    if lib.vsp_vk_from_blob(_ptr(buf), buf.shape[0], 0, None, None, None, None, None, None, 0, C.byref(n), None) != 0:
        raise ValueError("vk_from_blob: malformed blob")
    head = C.c_uint32(0); gt = np.zeros(576, np.uint8)
    g2 = np.zeros(24, np.uint64); d2 = np.zeros(24, np.uint64); d1 = np.zeros(12, np.uint64); g1 = np.zeros(12, np.uint64)
    gabc = np.zeros((n.value, 12), np.uint64)
    if lib.vsp_vk_from_blob(_ptr(buf), buf.shape[0], int(check_subgroup), C.byref(head), _ptr(gt), _ptr(g2), _ptr(d2), _ptr(d1), _ptr(gabc), n.value,
                            C.byref(n), _ptr(g1)) != 0:
        raise ValueError("vk_from_blob: invalid point")
    return dict(head=head.value, alpha_g1_beta_g2=gt.tobytes(), gamma_g2=g2, delta_g2=d2, delta_g1=d1, gamma_ABC_g1=gabc, gamma_g1=g1)
