"""ctypes binding of oracle/liboracle.so (the CPU oracle) -- the CHECKER used by tests, smoke() and
bench.py's cpu_baseline leg.  Never imported by the product package."""
import ctypes
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIR = os.path.join(_ROOT, "oracle")
_SO = os.environ.get("BP_ORACLE_SO") or os.path.join(_DIR, "liboracle.so")   # override: sanitizer builds (scripts/sanitize_cpu.sh)

BLS12_381, BN254 = 0, 1
CURVE_IDS = {"bls12_381": 0, "bn254": 1}


def build():
    subprocess.check_call(["make", "-s", "-C", _DIR])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.orc_msm_timed.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                    ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        L.orc_transcript_size.restype = ctypes.c_size_t
        for name in ("orc_msm", "orc_g1_fixed_base_batch", "orc_fr_inner", "orc_random_scalars", "orc_ipp_create",
                     "orc_ipp_verify", "orc_ipp_verification_scalars", "orc_transcript_new",
                     "orc_transcript_append_message", "orc_transcript_challenge_bytes", "orc_r1cs_prove", "orc_r1cs_verify",
                     "orc_r1cs_flattened_constraints", "orc_set_threads"):
            getattr(L, name).argtypes = None
        _lib = L
    return _lib


def fp_bytes(curve):
    return 48 if curve == 0 else 32


def pt_bytes(curve):
    return 2 * fp_bytes(curve)


FR_BYTES = 32


def _buf(n):
    return ctypes.create_string_buffer(n)


def field_op(curve, which, op, a, b):
    n = fp_bytes(curve) if which == 0 else FR_BYTES
    out = _buf(n)
    rc = lib().orc_field_op(curve, which, op, bytes(a), bytes(b), out)
    assert rc == 0
    return out.raw


def on_curve(curve, p):
    return bool(lib().orc_g1_on_curve(curve, bytes(p)))


def generator(curve):
    out = _buf(pt_bytes(curve))
    assert lib().orc_g1_generator(curve, out) == 0
    return out.raw


def g1_add(curve, p, q):
    out = _buf(pt_bytes(curve))
    assert lib().orc_g1_add(curve, bytes(p), bytes(q), out) == 0
    return out.raw


def g1_mul(curve, k, p):
    out = _buf(pt_bytes(curve))
    assert lib().orc_g1_mul(curve, bytes(k), bytes(p), out) == 0
    return out.raw


def binary_scalar_mul(curve, p, h, r1, r2):
    out = _buf(pt_bytes(curve))
    assert lib().orc_g1_binary_scalar_mul(curve, bytes(p), bytes(h), bytes(r1), bytes(r2), out) == 0
    return out.raw


def fixed_base_batch(curve, ks, n, nthreads=1):
    out = _buf(max(1, n) * pt_bytes(curve))
    assert lib().orc_g1_fixed_base_batch(ctypes.c_int(curve), bytes(ks), ctypes.c_size_t(n), ctypes.c_int(nthreads), out) == 0
    return out.raw[: n * pt_bytes(curve)]


def g1_to_amcl(curve, p):
    out = _buf(2 * fp_bytes(curve) + 1)
    assert lib().orc_g1_to_amcl(curve, bytes(p), out) == 0
    return out.raw


NAIVE, STRAUSS, PIPPENGER = 0, 1, 2


def msm(curve, points, scalars, n, algo=PIPPENGER, nthreads=1):
    out = _buf(pt_bytes(curve))
    rc = lib().orc_msm(ctypes.c_int(curve), ctypes.c_int(algo), bytes(points), bytes(scalars), ctypes.c_size_t(n),
                       ctypes.c_int(nthreads), out)
    assert rc == 0
    return out.raw


def msm_timed(curve, points, scalars, n, algo, nthreads=1):
    out = _buf(pt_bytes(curve))
    sec = ctypes.c_double(0)
    rc = lib().orc_msm_timed(curve, algo, bytes(points), bytes(scalars), n, nthreads, out, ctypes.byref(sec))
    assert rc == 0
    return out.raw, sec.value


def fr_inner(curve, a, b, n):
    out = _buf(FR_BYTES)
    assert lib().orc_fr_inner(ctypes.c_int(curve), bytes(a), bytes(b), ctypes.c_size_t(n), out) == 0
    return out.raw


def random_scalars(curve, seed, n):
    out = _buf(max(1, n) * FR_BYTES)
    assert lib().orc_random_scalars(ctypes.c_int(curve), ctypes.c_uint64(seed), ctypes.c_size_t(n), out) == 0
    return out.raw[: n * FR_BYTES]


class Transcript:
    def __init__(self, label: bytes):
        self.buf = _buf(lib().orc_transcript_size())
        lib().orc_transcript_new(self.buf, label, ctypes.c_size_t(len(label)))

    def append_message(self, label, msg):
        lib().orc_transcript_append_message(self.buf, label, ctypes.c_size_t(len(label)), bytes(msg), ctypes.c_size_t(len(msg)))

    def challenge_bytes(self, label, n):
        out = _buf(max(1, n))
        lib().orc_transcript_challenge_bytes(self.buf, label, ctypes.c_size_t(len(label)), out, ctypes.c_size_t(n))
        return out.raw[:n]

    def commit_point(self, curve, label, p):
        assert lib().orc_transcript_commit_point(curve, self.buf, label, bytes(p)) == 0

    def challenge_scalar(self, curve, label):
        out = _buf(FR_BYTES)
        assert lib().orc_transcript_challenge_scalar(curve, self.buf, label, out) == 0
        return out.raw

    def commit_scalar(self, curve, label, x):
        assert lib().orc_transcript_commit_scalar(curve, self.buf, label, bytes(x)) == 0


def ipp_create(curve, tr, Q, Gf, Hf, G, H, a, b, n):
    lg = max(0, n.bit_length() - 1)
    L, R = _buf(max(1, lg) * pt_bytes(curve)), _buf(max(1, lg) * pt_bytes(curve))
    ao, bo = _buf(FR_BYTES), _buf(FR_BYTES)
    rc = lib().orc_ipp_create(ctypes.c_int(curve), tr.buf, bytes(Q), bytes(Gf), bytes(Hf), bytes(G), bytes(H), bytes(a), bytes(b),
                              ctypes.c_size_t(n), L, R, ao, bo)
    if rc:
        return rc, None
    pb = pt_bytes(curve)
    return 0, (L.raw[: lg * pb], R.raw[: lg * pb], ao.raw, bo.raw)


def ipp_verify(curve, tr, n, Gf, Hf, P, Q, G, H, a, b, L, R, lg_n):
    return lib().orc_ipp_verify(ctypes.c_int(curve), tr.buf, ctypes.c_size_t(n), bytes(Gf), bytes(Hf), bytes(P), bytes(Q),
                                bytes(G), bytes(H), bytes(a), bytes(b), bytes(L), bytes(R), ctypes.c_size_t(lg_n))


def ipp_verification_scalars(curve, tr, L, R, lg_n, n):
    us, uis, s = _buf(max(1, lg_n) * 32), _buf(max(1, lg_n) * 32), _buf(max(1, n) * 32)
    rc = lib().orc_ipp_verification_scalars(ctypes.c_int(curve), tr.buf, bytes(L), bytes(R), ctypes.c_size_t(lg_n),
                                            ctypes.c_size_t(n), us, uis, s)
    if rc:
        return rc, None
    return 0, (us.raw[: lg_n * 32], uis.raw[: lg_n * 32], s.raw[: n * 32])


def g1_from_msg_hash(curve, msg):
    out = _buf(pt_bytes(curve))
    assert lib().orc_g1_from_msg_hash(curve, bytes(msg), ctypes.c_size_t(len(msg)), out) == 0
    return out.raw


def get_generators(curve, prefix, n, first=1, nthreads=1):
    """src/utils/mod.rs:16-23: from_msg_hash(prefix || decimal(i)), i = first .. first + n - 1."""
    prefix = prefix.encode() if isinstance(prefix, str) else bytes(prefix)
    out = _buf(pt_bytes(curve) * max(n, 1))
    assert lib().orc_get_generators(curve, prefix, ctypes.c_size_t(len(prefix)), ctypes.c_uint64(first), ctypes.c_size_t(n),
                                    nthreads, out) == 0
    return out.raw[: pt_bytes(curve) * n]


def group_order(curve):
    """r of the curve (public constant)."""
    return (0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001 if curve == 0
            else 0x2523648240000001BA344D8000000007FF9F800000000010A10000000000000D)


def set_threads(k):
    """Worker threads inside orc_ipp_* / orc_r1cs_* (MSMs and the fold loop); results do not depend on k."""
    lib().orc_set_threads(ctypes.c_int(k))


class R1CSTerms:
    """A constraint system as the flat term arrays orc_r1cs_* take: terms = [(constraint, kind, index, coeff int or 32-byte LE)]."""

    def __init__(self, terms, n_constraints, n, m):
        import numpy as np
        k = len(terms)
        self.k, self.nq, self.n, self.m = k, n_constraints, n, m
        self.con = np.ascontiguousarray([t[0] for t in terms], dtype=np.uint32)
        self.kind = np.ascontiguousarray([t[1] for t in terms], dtype=np.uint8)
        self.idx = np.ascontiguousarray([t[2] for t in terms], dtype=np.uint32)
        self.coeff = b"".join(t[3] if isinstance(t[3], (bytes, bytearray)) else int(t[3]).to_bytes(32, "little") for t in terms)

    def args(self):
        return (ctypes.c_size_t(self.k), self.con.ctypes.data_as(ctypes.c_void_p), self.kind.ctypes.data_as(ctypes.c_void_p),
                self.idx.ctypes.data_as(ctypes.c_void_p), self.coeff, ctypes.c_size_t(self.nq), ctypes.c_size_t(self.n), ctypes.c_size_t(self.m))


def r1cs_start_transcript(curve, label, V_list):
    """What Prover::new / Verifier::new and commit put on the transcript (src/r1cs/prover.rs:84-127)."""
    t = Transcript(label)
    t.append_message(b"dom-sep", b"r1cs v1")
    for V in V_list:
        t.commit_point(curve, b"V", V)
    return t


def r1cs_proof_bytes(curve, n):
    lg = max(0, (n - 1).bit_length())
    return 11 * pt_bytes(curve) + 3 * 32 + 2 * lg * pt_bytes(curve) + 2 * 32


def r1cs_prove(curve, tr, cs, g, h, G, H, ngens, aL, aR, aO, v_blinding, sL, sR, blindings):
    """Prover::prove (src/r1cs/prover.rs:322-593) in the C oracle -> (rc, proof bytes)."""
    out = _buf(r1cs_proof_bytes(curve, cs.n))
    rc = lib().orc_r1cs_prove(ctypes.c_int(curve), tr.buf, *cs.args(), bytes(g), bytes(h), bytes(G), bytes(H), ctypes.c_size_t(ngens),
                              bytes(aL), bytes(aR), bytes(aO), bytes(v_blinding), bytes(sL), bytes(sR), bytes(blindings), out)
    return rc, out.raw


def r1cs_verify(curve, tr, cs, V, proof, g, h, G, H, ngens, rnd):
    """Verifier::verify (src/r1cs/verifier.rs:267-457) in the C oracle -> 0 accepted / 3 rejected."""
    return lib().orc_r1cs_verify(ctypes.c_int(curve), tr.buf, *cs.args(), bytes(V), bytes(proof), ctypes.c_size_t(len(proof)), bytes(g), bytes(h),
                                 bytes(G), bytes(H), ctypes.c_size_t(ngens), bytes(rnd))


def r1cs_flattened_constraints(curve, cs, z):
    n, m = cs.n, cs.m
    wL, wR, wO, wV, wc = _buf(max(1, n) * 32), _buf(max(1, n) * 32), _buf(max(1, n) * 32), _buf(max(1, m) * 32), _buf(32)
    assert lib().orc_r1cs_flattened_constraints(ctypes.c_int(curve), *cs.args(), bytes(z), wL, wR, wO, wV, wc) == 0
    return wL.raw[: n * 32], wR.raw[: n * 32], wO.raw[: n * 32], wV.raw[: m * 32], wc.raw
