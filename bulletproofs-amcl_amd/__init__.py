"""bulletproofs-amcl_amd -- MI355X-native MSM / inner-product-argument engine (host-side Python mirror).

The directory name carries a hyphen, so the package is loaded under the import name
``bulletproofs_amcl_amd`` by ``__graft_entry__.load_package()``.

This module is plumbing above the C ABI (include/bpmsm.h, libbpmsm.so): thin ctypes wrappers whose names
follow the reference's types -- ``G1Vector`` / ``FieldElementVector`` (amcl_wrapper types used throughout
/root/reference src/ipp.rs) -- so that the parity tests read like the reference's own tests.  It contains no
arithmetic and no CPU fallback: every compute call goes to the HIP library and raises ``DeviceError`` when there
is no GPU.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("BPMSM_SO") or os.path.join(_HERE, "libbpmsm.so")   # override: another build of the SAME library (sanitizer runs)

BP_OK, BP_ERR_LENGTH, BP_ERR_ARG, BP_ERR_VERIFY, BP_ERR_DEVICE = 0, 1, 2, 3, 4
BLS12_381, BN254 = 0, 1
FMT_LE, FMT_AMCL = 0, 1
TUNE_TILE, TUNE_REDUCE_M, TUNE_TASK_TARGET, TUNE_SMALL_MSM, TUNE_TAIL_CHAINS, TUNE_COMPACT_AT, TUNE_GLV, TUNE_VERIFY_TABLES = 1, 2, 3, 4, 5, 6, 7, 8   # bp_ctx_set_tuning knobs (include/bpmsm.h)
CURVE_IDS = {"bls12_381": BLS12_381, "bn254": BN254}


class BpError(RuntimeError):
    code = -1


class ValueError_(BpError):
    """amcl_wrapper::errors::ValueError (length mismatch)."""
    code = BP_ERR_LENGTH


class ArgError(BpError):
    """assert!/assert_eq! panics of the reference (src/ipp.rs:48-55)."""
    code = BP_ERR_ARG


class VerificationError(BpError):
    """R1CSError::VerificationError (src/errors.rs:7-28)."""
    code = BP_ERR_VERIFY


class DeviceError(BpError):
    code = BP_ERR_DEVICE


_ERRS = {BP_ERR_LENGTH: ValueError_, BP_ERR_ARG: ArgError, BP_ERR_VERIFY: VerificationError, BP_ERR_DEVICE: DeviceError}


def _check(rc, what=""):
    if rc != BP_OK:
        raise _ERRS.get(rc, BpError)("%s failed with status %d" % (what, rc))


class CurveInfo(ctypes.Structure):
    _fields_ = [("curve_id", ctypes.c_int), ("fp_bytes", ctypes.c_int), ("fr_bytes", ctypes.c_int), ("modbytes", ctypes.c_int),
                ("fr_bits", ctypes.c_int), ("p_le", ctypes.c_uint8 * 48), ("r_le", ctypes.c_uint8 * 32), ("gen_le", ctypes.c_uint8 * 96)]


# every symbol include/bpmsm.h declares: name -> (restype, argtypes)
_P, _SZ, _I, _U8P = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_char_p
_PP = ctypes.POINTER(ctypes.c_void_p)
SYMBOLS = {
    "bp_version": (ctypes.c_char_p, []),
    "bp_curve_params": (_I, [_I, ctypes.POINTER(CurveInfo)]),
    "bp_device_count": (_I, []),
    "bp_ctx_create": (_I, [_I, _I, _PP]),
    "bp_ctx_destroy": (_I, [_P]),
    "bp_ctx_set_stream": (_I, [_P, _P]),
    "bp_ctx_synchronize": (_I, [_P]),
    "bp_ctx_set_window_bits": (_I, [_P, _I]),
    "bp_ctx_set_tuning": (_I, [_P, _I, ctypes.c_long]),
    "bp_ctx_set_device_tail": (_I, [_P, _I]),
    "bp_ctx_enable_timing": (_I, [_P, _I]),
    "bp_g1vec_upload": (_I, [_P, _U8P, _SZ, _I, _PP]),
    "bp_g1vec_alloc": (_I, [_P, _SZ, _PP]),
    "bp_g1vec_download": (_I, [_P, _P, _SZ, _SZ, _I, _U8P]),
    "bp_g1vec_free": (_I, [_P]),
    "bp_g1vec_precompute": (_I, [_P, _P, _I]),
    "bp_g1vec_drop_table": (_I, [_P]),
    "bp_ctx_verify_table_info": (_I, [_P, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)]),
    "bp_ctx_drop_verify_table": (_I, [_P]),
    "bp_g1vec_table_info": (_I, [_P, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_size_t)]),
    "bp_g1vec_len": (_SZ, [_P]),
    "bp_g1vec_device_ptr": (_P, [_P]),
    "bp_g1vec_wrap_device": (_I, [_P, _P, _SZ, _PP]),
    "bp_g1vec_fixed_base_mul": (_I, [_P, _P, _PP]),
    "bp_g1vec_scalar_mul": (_I, [_P, _P, _P, _PP]),
    "bp_g1vec_commit_pairs": (_I, [_P, _U8P, _U8P, _P, _P, _PP]),
    "bp_g1vec_from_msg_hash": (_I, [_P, _U8P, _P, _SZ, _PP]),
    "bp_get_generators": (_I, [_P, _U8P, _SZ, ctypes.c_uint64, _SZ, _PP]),
    "bp_frvec_upload": (_I, [_P, _U8P, _SZ, _PP]),
    "bp_frvec_alloc": (_I, [_P, _SZ, _PP]),
    "bp_frvec_download": (_I, [_P, _P, _SZ, _SZ, _U8P]),
    "bp_frvec_copy": (_I, [_P, _P, _SZ, _P, _SZ, _SZ]),
    "bp_frvec_free": (_I, [_P]),
    "bp_frvec_len": (_SZ, [_P]),
    "bp_frvec_device_ptr": (_P, [_P]),
    "bp_frvec_wrap_device": (_I, [_P, _P, _SZ, _PP]),
    "bp_msm_g1": (_I, [_P, _P, _P, _U8P]),
    "bp_msm_g1_range": (_I, [_P, _P, _SZ, _P, _SZ, _SZ, _U8P]),
    "bp_msm_g1_pair": (_I, [_P, _P, _P, _P, _U8P, _U8P]),
    "bp_msm_g1_begin": (_I, [_P, _P, _P]),
    "bp_msm_g1_end": (_I, [_P, _U8P]),
    "bp_msm_window_records": (_SZ, [_P, _SZ]),
    "bp_msm_record_bytes": (_SZ, [_I]),
    "bp_msm_g1_windows": (_I, [_P, _P, _SZ, _P, _SZ, _SZ, _P]),
    "bp_msm_g1_finish": (_I, [_P, _P, _SZ, _SZ, _U8P]),
    "bp_msm_window_records_subset": (_SZ, [_P, _SZ, _I, _I]),
    "bp_msm_g1_windows_subset": (_I, [_P, _P, _SZ, _P, _SZ, _SZ, _I, _I, _SZ, _P]),
    "bp_msm_g1_finish_blocks": (_I, [_P, _P, _SZ, _SZ, _SZ, _U8P]),
    "bp_msm_g1_finish_blocks_host": (_I, [_I, _U8P, _SZ, _SZ, _SZ, _I, _U8P]),
    "bp_msm_record_positions_subset": (_I, [_I, _SZ, _I, _I, _I, ctypes.POINTER(ctypes.c_int), _P]),
    "bp_msm_record_header_subset": (_I, [_I, _SZ, _I, _I, _I, _P]),
    "bp_msm_g1_finish_host": (_I, [_I, _U8P, _SZ, _SZ, _I, _U8P]),
    "bp_msm_geometry": (_I, [_I, _SZ, _I, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), _P, _P, _U8P]),
    "bp_msm_record_from_affine": (_I, [_I, _U8P, _P]),
    "bp_msm_record_positions": (_I, [_I, _SZ, _I, ctypes.POINTER(ctypes.c_int), _P]),
    "bp_msm_record_header": (_I, [_I, _SZ, _I, _P]),
    "bp_msm_g1_multi": (_I, [_PP, _PP, _PP, _SZ, _U8P]),
    "bp_ctx_trim": (_I, [_P]),
    "bp_g1_compressed_bytes": (_SZ, [_I]),
    "bp_g1vec_compress": (_I, [_P, _P, _SZ, _SZ, _U8P]),
    "bp_g1vec_decompress": (_I, [_P, _U8P, _SZ, _PP]),
    "bp_r1cs_proof_compressed_bytes": (_SZ, [_I, _SZ]),
    "bp_r1cs_proof_compress": (_I, [_P, _SZ, _U8P, _SZ, _U8P, _SZ]),
    "bp_r1cs_proof_decompress": (_I, [_P, _SZ, _U8P, _SZ, _U8P, _SZ]),
    "bp_fr_random": (_I, [_I, _U8P, _SZ]),
    "bp_fr_is_canonical_nonzero": (_I, [_I, _U8P]),
    "bp_msm_last_timing": (_I, [_P, ctypes.POINTER(ctypes.c_float), _I]),
    "bp_fr_inner_product": (_I, [_P, _P, _SZ, _P, _SZ, _SZ, _U8P]),
    "bp_fr_hadamard": (_I, [_P, _P, _P, _PP]),
    "bp_fr_scaled_by": (_I, [_P, _P, _U8P, _PP]),
    "bp_fr_glv_split": (_I, [_P, _P, _PP]),
    "bp_fr_vandermonde": (_I, [_P, _U8P, _SZ, _PP]),
    "bp_fr_inverse": (_I, [_I, _U8P, _U8P]),
    "bp_vecpoly3_special_inner_product": (_I, [_P, _PP, _PP, _U8P]),
    "bp_vecpoly1_inner_product": (_I, [_P, _PP, _PP, _U8P]),
    "bp_vecpoly_eval": (_I, [_P, _PP, _I, _U8P, _PP]),
    "bp_r1cs_plan_create": (_I, [_P, _SZ, _P, _U8P, _P, _U8P, _SZ, _SZ, _SZ, _PP]),
    "bp_r1cs_plan_free": (_I, [_P]),
    "bp_r1cs_flattened_constraints": (_I, [_P, _P, _U8P, _PP, _U8P]),
    "bp_r1cs_proof_bytes": (_SZ, [_I, _SZ]),
    "bp_r1cs_prove": (_I, [_P, _P, _P, _P, _P, _U8P, _U8P, _P, _P, _P, _P, _P, _P, _U8P, _U8P, _SZ]),
    "bp_r1cs_verify": (_I, [_P, _P, _P, _P, _P, _U8P, _U8P, _U8P, _SZ, _SZ, _U8P, _SZ, _U8P]),
    "bp_r1cs_phase1_bytes": (_SZ, []),
    "bp_r1cs_prove_begin": (_I, [_P, _P, _P, _P, _U8P, _SZ, _P, _P, _P, _P, _P, _U8P, _U8P, _SZ]),
    "bp_r1cs_prove_finish": (_I, [_P, _P, _P, _P, _P, _U8P, _U8P, _U8P, _P, _P, _P, _P, _P, _P, _U8P, _U8P, _SZ]),
    "bp_r1cs_verify_begin": (_I, [_P, _I, _SZ, _U8P, _SZ]),
    "bp_r1cs_verify_finish": (_I, [_P, _P, _P, _P, _P, _U8P, _U8P, _U8P, _SZ, _SZ, _SZ, _U8P, _SZ, _U8P]),
    "bp_r1cs_prover_polys": (_I, [_P, _PP, _U8P, _PP]),
    "bp_r1cs_ipp_inputs": (_I, [_P, _P, _P, _U8P, _U8P, _SZ, _SZ, _PP]),
    "bp_r1cs_verifier_scalars": (_I, [_P, _P, _U8P, _U8P, _SZ, _SZ, _SZ, _P, _P, _P, _U8P, _U8P, _U8P, _U8P, _U8P, _U8P, _U8P, _PP, _PP]),
    "bp_transcript_new": (_I, [_U8P, _SZ, _PP]),
    "bp_transcript_free": (_I, [_P]),
    "bp_transcript_append_message": (_I, [_P, _U8P, _SZ, _U8P, _SZ]),
    "bp_transcript_append_u64": (_I, [_P, _U8P, _SZ, ctypes.c_uint64]),
    "bp_transcript_challenge_bytes": (_I, [_P, _U8P, _SZ, _U8P, _SZ]),
    "bp_transcript_commit_point": (_I, [_P, _I, _U8P, _U8P]),
    "bp_transcript_commit_points": (_I, [_P, _I, _U8P, _U8P, _SZ]),
    "bp_transcript_commit_scalar": (_I, [_P, _I, _U8P, _U8P]),
    "bp_transcript_challenge_scalar": (_I, [_P, _I, _U8P, _U8P]),
    "bp_ipp_state_create": (_I, [_P, _P, _P, _U8P, _P, _P, _P, _P, _PP]),
    "bp_ipp_state_len": (_SZ, [_P]),
    "bp_ctx_set_ipp_fold_generators": (_I, [_P, _I]),
    "bp_ipp_round": (_I, [_P, _U8P, _U8P]),
    "bp_ipp_fold": (_I, [_P, _U8P, _U8P]),
    "bp_ipp_state_finish": (_I, [_P, _U8P, _U8P]),
    "bp_ipp_state_free": (_I, [_P]),
    "bp_ipp_create": (_I, [_P, _P, _U8P, _P, _P, _P, _P, _P, _P, _U8P, _U8P, ctypes.POINTER(ctypes.c_size_t), _U8P, _U8P]),
    "bp_ipp_create_multi": (_I, [_P, _SZ, _P, _U8P, _P, _P, _P, _P, _U8P, _U8P, _SZ, _U8P, _U8P, ctypes.POINTER(ctypes.c_size_t), _U8P, _U8P]),
    "bp_ipp_verify": (_I, [_P, _P, _SZ, _P, _P, _U8P, _U8P, _P, _P, _U8P, _U8P, _U8P, _U8P, _SZ]),
    "bp_ipp_verify_batch": (_I, [_P, _SZ, _SZ, _P, _P, _P, _P, _P, _SZ, _U8P]),
    "bp_ipp_verification_scalars": (_I, [_I, _P, _U8P, _U8P, _SZ, _SZ, _U8P, _U8P, _U8P]),
}

_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64.so.7.  Two HIP runtimes in one process cannot both own the GPU, so
    when torch is installed its copy is loaded first and libbpmsm.so (NEEDED libamdhip64.so.7) binds to it; the
    bench and the sharded path then share streams and device memory with torch / RCCL."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Load libbpmsm.so.  Fails loudly if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise ImportError("libbpmsm.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        _preload_torch_hip_runtime()
        L = ctypes.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("BPMSM_SO") and not hasattr(L, name):
                continue                                  # an OLDER build of the library named explicitly (same-box A/B runs): calls it lacks fail when made
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def curve_info(curve):
    info = CurveInfo()
    _check(lib().bp_curve_params(curve, ctypes.byref(info)), "bp_curve_params")
    return info


def device_count():
    return lib().bp_device_count()


class Context:
    """One per host thread; owns a HIP stream and the MSM workspace (bp_ctx)."""

    def __init__(self, curve=BLS12_381, device=0):
        self.curve = curve
        self.h = ctypes.c_void_p()
        _check(lib().bp_ctx_create(curve, device, ctypes.byref(self.h)), "bp_ctx_create")
        info = curve_info(curve)
        self.fp_bytes, self.fr_bytes, self.modbytes = info.fp_bytes, info.fr_bytes, info.modbytes
        self.point_bytes = 2 * info.fp_bytes
        self.r = int.from_bytes(bytes(info.r_le), "little")
        self.fr_bits = info.fr_bits

    def close(self):
        if self.h:
            lib().bp_ctx_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        _check(lib().bp_ctx_set_stream(self.h, ctypes.c_void_p(hip_stream)), "bp_ctx_set_stream")

    def synchronize(self):
        _check(lib().bp_ctx_synchronize(self.h), "bp_ctx_synchronize")

    def trim(self):
        """give the cached blocks of the context's memory pool back to the driver"""
        _check(lib().bp_ctx_trim(self.h), "bp_ctx_trim")

    def set_device_tail(self, on):
        _check(lib().bp_ctx_set_device_tail(self.h, 1 if on else 0), "bp_ctx_set_device_tail")

    def set_window_bits(self, c):
        _check(lib().bp_ctx_set_window_bits(self.h, c), "bp_ctx_set_window_bits")

    def set_tuning(self, knob, value):
        """validated engineering knobs of the MSM pipeline (TUNE_TILE, TUNE_REDUCE_M, TUNE_TASK_TARGET, TUNE_SMALL_MSM, TUNE_TAIL_CHAINS, TUNE_COMPACT_AT, TUNE_GLV, TUNE_VERIFY_TABLES); 0 = automatic"""
        _check(lib().bp_ctx_set_tuning(self.h, knob, value), "bp_ctx_set_tuning")

    def verify_table_info(self):
        """-> (generators per vector, bytes) of the [G | H] table this context keeps for its verifiers (TUNE_VERIFY_TABLES); zeros without one"""
        n, b = ctypes.c_size_t(), ctypes.c_size_t()
        _check(lib().bp_ctx_verify_table_info(self.h, ctypes.byref(n), ctypes.byref(b)), "bp_ctx_verify_table_info")
        return n.value, b.value

    def drop_verify_table(self):
        _check(lib().bp_ctx_drop_verify_table(self.h), "bp_ctx_drop_verify_table")

    def set_ipp_fold_generators(self, on):
        _check(lib().bp_ctx_set_ipp_fold_generators(self.h, 1 if on else 0), "bp_ctx_set_ipp_fold_generators")

    def enable_timing(self, on=True):
        _check(lib().bp_ctx_enable_timing(self.h, 1 if on else 0), "bp_ctx_enable_timing")

    def last_timing(self):
        buf = (ctypes.c_float * 8)()
        k = lib().bp_msm_last_timing(self.h, buf, 8)
        return [buf[i] for i in range(k)]


class HostPtr:
    """A raw host address + length handed to the upload calls instead of a bytes object (no copy on the Python side)."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = int(ptr), int(nbytes)


class G1Vector:
    """amcl_wrapper::group_elem_g1::G1Vector, resident in HBM."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def from_bytes(cls, ctx, data, n, fmt=FMT_LE):
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_upload(ctx.h, bytes(data), n, fmt, ctypes.byref(h)), "bp_g1vec_upload")
        return cls(ctx, h)

    @classmethod
    def new(cls, ctx, n):
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_alloc(ctx.h, n, ctypes.byref(h)), "bp_g1vec_alloc")
        return cls(ctx, h)

    @classmethod
    def wrap_device(cls, ctx, device_ptr, n):
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_wrap_device(ctx.h, ctypes.c_void_p(device_ptr), n, ctypes.byref(h)), "bp_g1vec_wrap_device")
        return cls(ctx, h)

    @classmethod
    def fixed_base(cls, ctx, scalars):
        """[k_i * G]"""
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_fixed_base_mul(ctx.h, scalars.h, ctypes.byref(h)), "bp_g1vec_fixed_base_mul")
        return cls(ctx, h)

    @classmethod
    def from_msg_hash(cls, ctx, messages):
        """[G1::from_msg_hash(m) for m in messages] (amcl_wrapper), hashed and mapped on the device."""
        messages = [bytes(m) for m in messages]
        offs = (ctypes.c_uint64 * (len(messages) + 1))()
        for i, m in enumerate(messages):
            offs[i + 1] = offs[i] + len(m)
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_from_msg_hash(ctx.h, b"".join(messages), ctypes.cast(offs, ctypes.c_void_p), len(messages), ctypes.byref(h)),
               "bp_g1vec_from_msg_hash")
        return cls(ctx, h)

    @classmethod
    def commit_pairs(cls, ctx, g_le, h_le, k1, k2):
        """[k1_i * g + k2_i * h]: batched commit_to_field_element / binary_scalar_mul with fixed g, h (src/r1cs/prover.rs:123)"""
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_commit_pairs(ctx.h, bytes(g_le), bytes(h_le), k1.h, k2.h, ctypes.byref(h)), "bp_g1vec_commit_pairs")
        return cls(ctx, h)

    @classmethod
    def from_compressed(cls, ctx, data, n):
        """tag || X per point (bp_g1vec_decompress); ArgError if a point does not decode"""
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_decompress(ctx.h, bytes(data), n, ctypes.byref(h)), "bp_g1vec_decompress")
        return cls(ctx, h)

    def to_compressed(self, offset=0, n=None):
        n = len(self) - offset if n is None else n
        buf = ctypes.create_string_buffer(max(1, n * lib().bp_g1_compressed_bytes(self.ctx.curve)))
        _check(lib().bp_g1vec_compress(self.ctx.h, self.h, offset, n, buf), "bp_g1vec_compress")
        return buf.raw[:n * lib().bp_g1_compressed_bytes(self.ctx.curve)]

    def scaled_by(self, scalars):
        """[k_i * P_i]"""
        h = ctypes.c_void_p()
        _check(lib().bp_g1vec_scalar_mul(self.ctx.h, self.h, scalars.h, ctypes.byref(h)), "bp_g1vec_scalar_mul")
        return G1Vector(self.ctx, h)

    def __len__(self):
        return lib().bp_g1vec_len(self.h)

    def device_ptr(self):
        return lib().bp_g1vec_device_ptr(self.h)

    def to_bytes(self, offset=0, n=None, fmt=FMT_LE):
        n = len(self) - offset if n is None else n
        per = self.ctx.point_bytes if fmt == FMT_LE else self.ctx.point_bytes + 1
        buf = ctypes.create_string_buffer(max(1, n * per))
        _check(lib().bp_g1vec_download(self.ctx.h, self.h, offset, n, fmt, buf), "bp_g1vec_download")
        return buf.raw[: n * per]

    # G1Vector::multi_scalar_mul_var_time / inner_product_var_time_with_ref_vecs / inner_product_const_time
    def multi_scalar_mul_var_time(self, scalars):
        out = ctypes.create_string_buffer(self.ctx.point_bytes)
        _check(lib().bp_msm_g1(self.ctx.h, self.h, scalars.h, out), "bp_msm_g1")
        return out.raw

    inner_product_var_time = multi_scalar_mul_var_time
    inner_product_const_time = multi_scalar_mul_var_time

    def msm_begin(self, scalars):
        """queue the MSM on this vector's context; finish with msm_end() (one in flight per context)"""
        _check(lib().bp_msm_g1_begin(self.ctx.h, self.h, scalars.h), "bp_msm_g1_begin")

    def msm_end(self):
        out = ctypes.create_string_buffer(self.ctx.point_bytes)
        _check(lib().bp_msm_g1_end(self.ctx.h, out), "bp_msm_g1_end")
        return out.raw

    def multi_scalar_mul_pair(self, scalars1, scalars2):
        """(<s1, P>, <s2, P>) in one pipeline pass"""
        o1, o2 = ctypes.create_string_buffer(self.ctx.point_bytes), ctypes.create_string_buffer(self.ctx.point_bytes)
        _check(lib().bp_msm_g1_pair(self.ctx.h, self.h, scalars1.h, scalars2.h, o1, o2), "bp_msm_g1_pair")
        return o1.raw, o2.raw

    def precompute(self, window_bits=0):
        """build the window-multiples table of this (fixed) vector: every later MSM over the whole vector runs the merged-window
        pipeline (bp_g1vec_precompute); same bytes with and without"""
        _check(lib().bp_g1vec_precompute(self.ctx.h, self.h, window_bits), "bp_g1vec_precompute")
        return self

    def drop_table(self):
        _check(lib().bp_g1vec_drop_table(self.h), "bp_g1vec_drop_table")

    def table_info(self):
        """-> (window_bits, windows, bytes); zeros without a table"""
        c, w, b = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        _check(lib().bp_g1vec_table_info(self.h, ctypes.byref(c), ctypes.byref(w), ctypes.byref(b)), "bp_g1vec_table_info")
        return c.value, w.value, b.value

    def msm_range(self, poff, scalars, soff, n):
        out = ctypes.create_string_buffer(self.ctx.point_bytes)
        _check(lib().bp_msm_g1_range(self.ctx.h, self.h, poff, scalars.h, soff, n, out), "bp_msm_g1_range")
        return out.raw

    def free(self):
        if self.h:
            lib().bp_g1vec_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class FieldElementVector:
    """amcl_wrapper::field_elem::FieldElementVector, resident in HBM (32-byte canonical LE scalars)."""

    def __init__(self, ctx, handle):
        self.ctx, self.h = ctx, handle

    @classmethod
    def from_bytes(cls, ctx, data, n):
        """data: bytes-like, or a HostPtr (a caller-owned host buffer -- e.g. page-locked -- passed to bp_frvec_upload as it is)"""
        h = ctypes.c_void_p()
        src = ctypes.cast(ctypes.c_void_p(data.ptr), _U8P) if isinstance(data, HostPtr) else bytes(data)
        _check(lib().bp_frvec_upload(ctx.h, src, n, ctypes.byref(h)), "bp_frvec_upload")
        return cls(ctx, h)

    @classmethod
    def from_ints(cls, ctx, values):
        return cls.from_bytes(ctx, b"".join((v % ctx.r).to_bytes(32, "little") for v in values), len(values))

    @classmethod
    def new(cls, ctx, n):
        h = ctypes.c_void_p()
        _check(lib().bp_frvec_alloc(ctx.h, n, ctypes.byref(h)), "bp_frvec_alloc")
        return cls(ctx, h)

    @classmethod
    def wrap_device(cls, ctx, device_ptr, n):
        h = ctypes.c_void_p()
        _check(lib().bp_frvec_wrap_device(ctx.h, ctypes.c_void_p(device_ptr), n, ctypes.byref(h)), "bp_frvec_wrap_device")
        return cls(ctx, h)

    def __len__(self):
        return lib().bp_frvec_len(self.h)

    def device_ptr(self):
        return lib().bp_frvec_device_ptr(self.h)

    def to_bytes(self, offset=0, n=None):
        n = len(self) - offset if n is None else n
        buf = ctypes.create_string_buffer(max(1, n * 32))
        _check(lib().bp_frvec_download(self.ctx.h, self.h, offset, n, buf), "bp_frvec_download")
        return buf.raw[: n * 32]

    def free(self):
        if self.h:
            lib().bp_frvec_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def get_generators(ctx, prefix, n, first=1):
    """utils::get_generators(prefix, n) (reference src/utils/mod.rs:16-23): [from_msg_hash(prefix || str(i)) for i in 1..=n],
    returned resident in HBM.  `first` shifts the counter (rank r of a sharded setup generates its own index range)."""
    prefix = prefix.encode() if isinstance(prefix, str) else bytes(prefix)
    h = ctypes.c_void_p()
    _check(lib().bp_get_generators(ctx.h, prefix, len(prefix), first, n, ctypes.byref(h)), "bp_get_generators")
    return G1Vector(ctx, h)


def msm_windows(ctx, points, poff, scalars, soff, n, device_out_ptr):
    """Stage 1 of the sharded MSM: W window records into caller-owned HBM (see include/bpmsm.h)."""
    _check(lib().bp_msm_g1_windows(ctx.h, points.h, poff, scalars.h, soff, n, ctypes.c_void_p(device_out_ptr)), "bp_msm_g1_windows")


def msm_finish(ctx, device_records_ptr, sets, n_per_set):
    out = ctypes.create_string_buffer(ctx.point_bytes)
    _check(lib().bp_msm_g1_finish(ctx.h, ctypes.c_void_p(device_records_ptr), sets, n_per_set, out), "bp_msm_g1_finish")
    return out.raw


def msm_window_records_subset(ctx, n, w_first, w_count):
    """records of the block of one WINDOW GROUP (tail records + header) -- the 2-D sharding of include/bpmsm.h"""
    return lib().bp_msm_window_records_subset(ctx.h, n, w_first, w_count)


def msm_windows_subset(ctx, points, poff, scalars, soff, n, w_first, w_count, block_records, device_out_ptr):
    _check(lib().bp_msm_g1_windows_subset(ctx.h, points.h, poff, scalars.h, soff, n, w_first, w_count, block_records, ctypes.c_void_p(device_out_ptr)),
           "bp_msm_g1_windows_subset")


def msm_finish_blocks(ctx, device_records_ptr, n_blocks, block_records, n_per_set):
    out = ctypes.create_string_buffer(ctx.point_bytes)
    _check(lib().bp_msm_g1_finish_blocks(ctx.h, ctypes.c_void_p(device_records_ptr), n_blocks, block_records, n_per_set, out), "bp_msm_g1_finish_blocks")
    return out.raw


def msm_finish_blocks_host(curve, host_records, n_blocks, block_records, n_per_set, window_bits=0):
    out = ctypes.create_string_buffer(2 * (48 if curve == BLS12_381 else 32))
    _check(lib().bp_msm_g1_finish_blocks_host(curve, bytes(host_records), n_blocks, block_records, n_per_set, window_bits, out), "bp_msm_g1_finish_blocks_host")
    return out.raw


def msm_record_positions_subset(curve, n, window_bits, w_first, w_count):
    nrec = ctypes.c_int()
    pos = (ctypes.c_uint16 * 4096)()
    _check(lib().bp_msm_record_positions_subset(curve, n, window_bits, w_first, w_count, ctypes.byref(nrec), ctypes.cast(pos, ctypes.c_void_p)), "bp_msm_record_positions_subset")
    return [pos[i] for i in range(nrec.value)]


def msm_record_header_subset(curve, n, window_bits, w_first, w_count):
    out = ctypes.create_string_buffer(lib().bp_msm_record_bytes(curve))
    _check(lib().bp_msm_record_header_subset(curve, n, window_bits, w_first, w_count, out), "bp_msm_record_header_subset")
    return out.raw


def msm_window_records(ctx, n):
    """records per block: W window records + the geometry header"""
    return lib().bp_msm_window_records(ctx.h, n)


def msm_finish_host(curve, host_records, sets, n_per_set, window_bits=0):
    """Stage 2 of the sharded MSM on host memory (no GPU): validates the headers, folds `sets` record blocks."""
    out = ctypes.create_string_buffer(2 * (48 if curve == BLS12_381 else 32))
    _check(lib().bp_msm_g1_finish_host(curve, bytes(host_records), sets, n_per_set, window_bits, out), "bp_msm_g1_finish_host")
    return out.raw


def msm_geometry(curve, n, window_bits=0):
    """-> (c, [cw_w], [off_w], bias int): the window table an MSM of n terms uses (bp_msm_geometry)."""
    c, W = ctypes.c_int(), ctypes.c_int()
    cw, off, bias = (ctypes.c_uint8 * 256)(), (ctypes.c_uint16 * 256)(), ctypes.create_string_buffer(32)
    _check(lib().bp_msm_geometry(curve, n, window_bits, ctypes.byref(c), ctypes.byref(W), ctypes.cast(cw, ctypes.c_void_p),
                                 ctypes.cast(off, ctypes.c_void_p), bias), "bp_msm_geometry")
    return c.value, list(cw[:W.value]), list(off[:W.value]), int.from_bytes(bias.raw, "little")


def msm_record_positions(curve, n, window_bits=0):
    """-> [pos_r]: record r of a block carries weight 2^pos_r (bp_msm_record_positions)"""
    k = ctypes.c_int()
    pos = (ctypes.c_uint16 * 4096)()
    _check(lib().bp_msm_record_positions(curve, n, window_bits, ctypes.byref(k), ctypes.cast(pos, ctypes.c_void_p)), "bp_msm_record_positions")
    return list(pos[:k.value])


def msm_record_from_affine(curve, point_le):
    out = ctypes.create_string_buffer(lib().bp_msm_record_bytes(curve))
    _check(lib().bp_msm_record_from_affine(curve, bytes(point_le), ctypes.cast(out, ctypes.c_void_p)), "bp_msm_record_from_affine")
    return out.raw


def msm_record_header(curve, n, window_bits=0):
    out = ctypes.create_string_buffer(lib().bp_msm_record_bytes(curve))
    _check(lib().bp_msm_record_header(curve, n, window_bits, ctypes.cast(out, ctypes.c_void_p)), "bp_msm_record_header")
    return out.raw


def msm_multi(ctxs, points, scalars):
    """bp_msm_g1_multi: shard i = (points[i], scalars[i]) resident with ctxs[i] (one context per device, or several on one
    device); one host thread, no torch / RCCL.  -> the affine sum over all shards, BP_FMT_LE."""
    k = len(ctxs)
    A = lambda hs: (ctypes.c_void_p * k)(*[h.h for h in hs])
    out = ctypes.create_string_buffer(ctxs[0].point_bytes)
    _check(lib().bp_msm_g1_multi(A(ctxs), A(points), A(scalars), k, out), "bp_msm_g1_multi")
    return out.raw


def fr_random(curve, n=1):
    """FieldElement::random() x n (getrandom)"""
    out = ctypes.create_string_buffer(32 * max(1, n))
    _check(lib().bp_fr_random(curve, out, n), "bp_fr_random")
    return out.raw[:32 * n]


def msm_record_bytes(curve):
    return lib().bp_msm_record_bytes(curve)


# ---- FieldElementVector helpers -------------------------------------------------------------------------------------

def _fr_copy_from(self, dst_off, src, src_off=0, n=None):
    """self[dst_off : dst_off + n] = src[src_off : src_off + n] on the device (bp_frvec_copy); returns self"""
    n = len(src) - src_off if n is None else n
    _check(lib().bp_frvec_copy(self.ctx.h, self.h, dst_off, src.h, src_off, n), "bp_frvec_copy")
    return self


def _fr_inner_product(self, other, aoff=0, boff=0, n=None):
    """FieldElementVector::inner_product"""
    if n is None:
        if len(self) != len(other):
            raise ValueError_("inner_product: unequal lengths")
        n = len(self)
    out = ctypes.create_string_buffer(32)
    _check(lib().bp_fr_inner_product(self.ctx.h, self.h, aoff, other.h, boff, n, out), "bp_fr_inner_product")
    return out.raw


def _fr_hadamard(self, other):
    """FieldElementVector::hadamard_product"""
    h = ctypes.c_void_p()
    _check(lib().bp_fr_hadamard(self.ctx.h, self.h, other.h, ctypes.byref(h)), "bp_fr_hadamard")
    return FieldElementVector(self.ctx, h)


def _fr_glv_split(self):
    """the two 128-bit halves of every scalar under the curve's GLV endomorphism (bp_fr_glv_split): 16 bytes s1 | 16 bytes s2 per element"""
    h = ctypes.c_void_p()
    _check(lib().bp_fr_glv_split(self.ctx.h, self.h, ctypes.byref(h)), "bp_fr_glv_split")
    return FieldElementVector(self.ctx, h)


def _fr_scaled_by(self, s_le32):
    """FieldElementVector::scaled_by"""
    h = ctypes.c_void_p()
    _check(lib().bp_fr_scaled_by(self.ctx.h, self.h, bytes(s_le32), ctypes.byref(h)), "bp_fr_scaled_by")
    return FieldElementVector(self.ctx, h)


def _fr_vandermonde(cls, ctx, e_le32, n):
    """FieldElementVector::new_vandermonde_vector"""
    h = ctypes.c_void_p()
    _check(lib().bp_fr_vandermonde(ctx.h, bytes(e_le32), n, ctypes.byref(h)), "bp_fr_vandermonde")
    return cls(ctx, h)


FieldElementVector.inner_product = _fr_inner_product
FieldElementVector.copy_from = _fr_copy_from
FieldElementVector.hadamard_product = _fr_hadamard
FieldElementVector.scaled_by = _fr_scaled_by
FieldElementVector.glv_split = _fr_glv_split
FieldElementVector.new_vandermonde_vector = classmethod(_fr_vandermonde)


def fr_inverse(curve, x_le32):
    """FieldElement::inverse (host)"""
    out = ctypes.create_string_buffer(32)
    _check(lib().bp_fr_inverse(curve, bytes(x_le32), out), "bp_fr_inverse")
    return out.raw


# ---- transcript (merlin::Transcript + TranscriptProtocol, src/transcript.rs) ---------------------------------------

class Transcript:
    def __init__(self, label: bytes):
        self.h = ctypes.c_void_p()
        _check(lib().bp_transcript_new(label, len(label), ctypes.byref(self.h)), "bp_transcript_new")

    def append_message(self, label, msg):
        _check(lib().bp_transcript_append_message(self.h, label, len(label), bytes(msg), len(msg)), "append_message")

    def append_u64(self, label, x):
        _check(lib().bp_transcript_append_u64(self.h, label, len(label), x), "append_u64")

    def challenge_bytes(self, label, n):
        out = ctypes.create_string_buffer(max(1, n))
        _check(lib().bp_transcript_challenge_bytes(self.h, label, len(label), out, n), "challenge_bytes")
        return out.raw[:n]

    def commit_point(self, curve, label, point_le):
        _check(lib().bp_transcript_commit_point(self.h, curve, label, bytes(point_le)), "commit_point")

    def commit_points(self, curve, label, points_le, n):
        """n commit_point calls with the same label in one library call (the V commitments of a statement)."""
        _check(lib().bp_transcript_commit_points(self.h, curve, label, bytes(points_le), n), "commit_points")

    def commit_scalar(self, curve, label, scalar_le32):
        _check(lib().bp_transcript_commit_scalar(self.h, curve, label, bytes(scalar_le32)), "commit_scalar")

    def challenge_scalar(self, curve, label):
        out = ctypes.create_string_buffer(32)
        _check(lib().bp_transcript_challenge_scalar(self.h, curve, label, out), "challenge_scalar")
        return out.raw

    def free(self):
        if self.h:
            lib().bp_transcript_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ---- inner-product argument (src/ipp.rs) -----------------------------------------------------------------------------

class InnerProductArgumentProof:
    """src/ipp.rs:13-20: {L: G1Vector, R: G1Vector, a, b}; points as BP_FMT_LE bytes here."""

    def __init__(self, L, R, a, b, lg_n):
        self.L, self.R, self.a, self.b, self.lg_n = L, R, a, b, lg_n


class IPPState:
    """Device-resident prover state for a host that owns its own transcript (bp_ipp_state_*)."""

    def __init__(self, ctx, G, H, Q_le, G_factors, H_factors, a, b):
        self.ctx = ctx
        self.h = ctypes.c_void_p()
        _check(lib().bp_ipp_state_create(ctx.h, G.h, H.h, bytes(Q_le), G_factors.h, H_factors.h, a.h, b.h, ctypes.byref(self.h)), "bp_ipp_state_create")

    def __len__(self):
        return lib().bp_ipp_state_len(self.h)

    def round(self):
        L, R = ctypes.create_string_buffer(self.ctx.point_bytes), ctypes.create_string_buffer(self.ctx.point_bytes)
        _check(lib().bp_ipp_round(self.h, L, R), "bp_ipp_round")
        return L.raw, R.raw

    def fold(self, u_le32, u_inv_le32):
        _check(lib().bp_ipp_fold(self.h, bytes(u_le32), bytes(u_inv_le32)), "bp_ipp_fold")

    def finish(self):
        a, b = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        _check(lib().bp_ipp_state_finish(self.h, a, b), "bp_ipp_state_finish")
        return a.raw, b.raw

    def free(self):
        if self.h:
            lib().bp_ipp_state_free(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class _ProofRef(ctypes.Structure):
    """struct bp_ipp_proof_ref (include/bpmsm.h)"""
    _fields_ = [("transcript", ctypes.c_void_p), ("P_le", ctypes.c_void_p), ("Q_le", ctypes.c_void_p), ("a_le32", ctypes.c_void_p),
                ("b_le32", ctypes.c_void_p), ("L_le", ctypes.c_void_p), ("R_le", ctypes.c_void_p)]


class IPP:
    """IPP::create_ipp / verify_ipp / verification_scalars (src/ipp.rs:22-316)."""

    @staticmethod
    def create_ipp(ctx, transcript, Q_le, G_factors, H_factors, G_vec, H_vec, a_vec, b_vec):
        n = len(G_vec)
        lg = max(1, n.bit_length())
        L = ctypes.create_string_buffer(lg * ctx.point_bytes)
        R = ctypes.create_string_buffer(lg * ctx.point_bytes)
        a, b = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        lg_n = ctypes.c_size_t(0)
        _check(lib().bp_ipp_create(ctx.h, transcript.h, bytes(Q_le), G_factors.h, H_factors.h, G_vec.h, H_vec.h, a_vec.h, b_vec.h, L, R,
                                   ctypes.byref(lg_n), a, b), "bp_ipp_create")
        k = lg_n.value
        return InnerProductArgumentProof(L.raw[: k * ctx.point_bytes], R.raw[: k * ctx.point_bytes], a.raw, b.raw, k)

    @staticmethod
    def create_ipp_multi(ctxs, transcript, Q_le, G_factors, H_factors, G_vecs, H_vecs, a_le, b_le):
        """create_ipp with the generators sharded by index range over several contexts (bp_ipp_create_multi): shard i of every list
        lives with ctxs[i]; a_le / b_le are the n x 32-byte host scalars.  Same proof bytes as create_ipp."""
        k = len(ctxs)
        n = sum(len(g) for g in G_vecs)
        A = lambda xs: (ctypes.c_void_p * k)(*[x.h for x in xs])
        pb = ctxs[0].point_bytes
        lg = max(1, n.bit_length())
        L, R = ctypes.create_string_buffer(lg * pb), ctypes.create_string_buffer(lg * pb)
        a, b = ctypes.create_string_buffer(32), ctypes.create_string_buffer(32)
        lg_n = ctypes.c_size_t(0)
        _check(lib().bp_ipp_create_multi(A(ctxs), k, transcript.h, bytes(Q_le), A(G_factors), A(H_factors), A(G_vecs), A(H_vecs), bytes(a_le), bytes(b_le), n,
                                         L, R, ctypes.byref(lg_n), a, b), "bp_ipp_create_multi")
        kk = lg_n.value
        return InnerProductArgumentProof(L.raw[: kk * pb], R.raw[: kk * pb], a.raw, b.raw, kk)

    @staticmethod
    def verify_ipp(ctx, n, transcript, G_factors, H_factors, P_le, Q_le, G, H, a_le32, b_le32, L_le, R_le):
        """Returns None on success, raises VerificationError otherwise (Result<(), R1CSError>)."""
        lg_n = len(L_le) // ctx.point_bytes
        _check(lib().bp_ipp_verify(ctx.h, transcript.h, n, G_factors.h, H_factors.h, bytes(P_le), bytes(Q_le), G.h, H.h, bytes(a_le32),
                                   bytes(b_le32), bytes(L_le), bytes(R_le), lg_n), "bp_ipp_verify")

    @staticmethod
    def verify_batch(ctx, n, G_factors, H_factors, G, H, proofs, weights=None):
        """Accept m proofs over the same generators with ONE MSM (random linear combination, bp_ipp_verify_batch).
        proofs: iterable of (transcript, P_le, Q_le, a_le32, b_le32, L_le, R_le).  weights: m 32-byte LE scalars; by default
        128 fresh random bits each from os.urandom.  Raises VerificationError if the combination is not the identity."""
        proofs = list(proofs)
        m = len(proofs)
        lg_n = max(0, n.bit_length() - 1)
        arr = (_ProofRef * max(1, m))()
        keep = []
        for i, (tr, P, Q, a, b, L, R) in enumerate(proofs):
            bufs = [ctypes.create_string_buffer(bytes(x), max(1, len(x))) for x in (P, Q, a, b, L, R)]
            if len(L) != lg_n * ctx.point_bytes or len(R) != lg_n * ctx.point_bytes:
                raise VerificationError("bp_ipp_verify_batch: proof %d has the wrong number of L/R points" % i)
            keep.append(bufs)
            arr[i].transcript = tr.h
            for name, bf in zip(("P_le", "Q_le", "a_le32", "b_le32", "L_le", "R_le"), bufs):
                setattr(arr[i], name, ctypes.cast(bf, ctypes.c_void_p))
        if weights is not None and len(weights) != 32 * m:
            raise ArgError("bp_ipp_verify_batch: need one 32-byte weight per proof")
        # weights None: the library draws them itself (bp_fr_random)
        _check(lib().bp_ipp_verify_batch(ctx.h, n, lg_n, G_factors.h, H_factors.h, G.h, H.h, ctypes.cast(arr, ctypes.c_void_p), m,
                                         bytes(weights) if weights is not None else None), "bp_ipp_verify_batch")

    @staticmethod
    def verification_scalars(curve, L_le, R_le, n, transcript):
        pb = 2 * curve_info(curve).fp_bytes
        lg_n = len(L_le) // pb
        us, uis, s = (ctypes.create_string_buffer(max(1, lg_n) * 32), ctypes.create_string_buffer(max(1, lg_n) * 32),
                      ctypes.create_string_buffer(max(1, n) * 32))
        _check(lib().bp_ipp_verification_scalars(curve, transcript.h, bytes(L_le), bytes(R_le), lg_n, n, us, uis, s), "bp_ipp_verification_scalars")
        return us.raw[: lg_n * 32], uis.raw[: lg_n * 32], s.raw[: n * 32]


# ---- vector polynomials (src/utils/vector_poly.rs) -------------------------------------------------------------------

def _handles(vs):
    arr = (ctypes.c_void_p * len(vs))()
    for i, v in enumerate(vs):
        arr[i] = v.h
    return arr


class VecPoly1:
    """A + B*X"""

    def __init__(self, a, b):
        self.v = (a, b)

    def inner_product(self, rhs):
        """-> Poly2 coefficients (t0, t1, t2) as 32-byte LE scalars"""
        ctx = self.v[0].ctx
        out = ctypes.create_string_buffer(96)
        _check(lib().bp_vecpoly1_inner_product(ctx.h, _handles(self.v), _handles(rhs.v), out), "bp_vecpoly1_inner_product")
        return out.raw[:32], out.raw[32:64], out.raw[64:96]

    def eval(self, x_le32):
        ctx = self.v[0].ctx
        h = ctypes.c_void_p()
        _check(lib().bp_vecpoly_eval(ctx.h, _handles(self.v), 1, bytes(x_le32), ctypes.byref(h)), "bp_vecpoly_eval")
        return FieldElementVector(ctx, h)


class VecPoly3:
    """A + B*X + C*X^2 + D*X^3"""

    def __init__(self, a, b, c, d):
        self.v = (a, b, c, d)

    @staticmethod
    def special_inner_product(lhs, rhs):
        """-> Poly6 coefficients (t1..t6); requires lhs.0 == 0 and rhs.2 == 0 (src/utils/vector_poly.rs:75-79)"""
        ctx = lhs.v[0].ctx
        out = ctypes.create_string_buffer(192)
        _check(lib().bp_vecpoly3_special_inner_product(ctx.h, _handles(lhs.v), _handles(rhs.v), out), "bp_vecpoly3_special_inner_product")
        return [out.raw[32 * i:32 * i + 32] for i in range(6)]

    def eval(self, x_le32):
        ctx = self.v[0].ctx
        h = ctypes.c_void_p()
        _check(lib().bp_vecpoly_eval(ctx.h, _handles(self.v), 3, bytes(x_le32), ctypes.byref(h)), "bp_vecpoly_eval")
        return FieldElementVector(ctx, h)


# ---- R1CS vector pipeline (src/r1cs/prover.rs:458-563, src/r1cs/verifier.rs:342-390) --------------------------------

VAR_MUL_LEFT, VAR_MUL_RIGHT, VAR_MUL_OUTPUT, VAR_COMMITTED, VAR_ONE = 0, 1, 2, 3, 4


class R1CSPlan:
    """The terms of a constraint system regrouped once for `flattened_constraints` (src/r1cs/prover.rs:142-184,
    src/r1cs/verifier.rs:149-193).  terms: iterable of (constraint q, kind VAR_*, index, coeff as int or 32-byte LE)."""

    def __init__(self, ctx, terms, n_constraints, n, m):
        terms = list(terms)
        T = len(terms)
        tq = (ctypes.c_uint32 * max(1, T))(*[t[0] for t in terms])
        kind = bytes(t[1] for t in terms)
        idx = (ctypes.c_uint32 * max(1, T))(*[t[2] for t in terms])
        coeff = b"".join(t[3] if isinstance(t[3], (bytes, bytearray)) else (t[3] % ctx.r).to_bytes(32, "little") for t in terms)
        h = ctypes.c_void_p()
        _check(lib().bp_r1cs_plan_create(ctx.h, T, ctypes.cast(tq, ctypes.c_void_p), kind or b"\0", ctypes.cast(idx, ctypes.c_void_p), coeff or b"\0",
                                         n_constraints, n, m, ctypes.byref(h)), "bp_r1cs_plan_create")
        self.ctx, self.h, self.n, self.m = ctx, h, n, m

    def flattened_constraints(self, z_le32, want_constant=True):
        """-> (wL, wR, wO, wV, wc): four FieldElementVectors and the 32-byte constant (None for the prover's form)."""
        out = (ctypes.c_void_p * 4)()
        wc = ctypes.create_string_buffer(32) if want_constant else None
        _check(lib().bp_r1cs_flattened_constraints(self.ctx.h, self.h, bytes(z_le32), out, wc), "bp_r1cs_flattened_constraints")
        vs = [FieldElementVector(self.ctx, ctypes.c_void_p(out[k])) for k in range(4)]
        return vs[0], vs[1], vs[2], vs[3], (wc.raw if wc is not None else None)

    def free(self):
        if self.h:
            lib().bp_r1cs_plan_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def r1cs_prove(ctx, transcript, plan, G, H, g_le, h_le, a_L, a_R, a_O, v_blinding, s_L, s_R, blindings_le32):
    """Prover::prove (src/r1cs/prover.rs:323-560) as one library call (bp_r1cs_prove) -> proof bytes.
    blindings_le32: i, o, s, t1, t3, t4, t5, t6 (8 x 32 bytes)."""
    n = len(a_L)
    size = lib().bp_r1cs_proof_bytes(ctx.curve, n)
    out = ctypes.create_string_buffer(size)
    _check(lib().bp_r1cs_prove(ctx.h, transcript.h, plan.h, G.h, H.h, bytes(g_le), bytes(h_le), a_L.h, a_R.h, a_O.h,
                               v_blinding.h if v_blinding is not None and len(v_blinding) else None, s_L.h, s_R.h, bytes(blindings_le32), out, size),
           "bp_r1cs_prove")
    return out.raw


def r1cs_verify(ctx, transcript, plan, G, H, g_le, h_le, V_le, n, proof, r_le32=None):
    """Verifier::verify (src/r1cs/verifier.rs:265-452) as one library call; raises VerificationError.
    r_le32: the verifier's random weight (verifier.rs:392); None = drawn inside the library, as the reference does."""
    m = len(V_le) // ctx.point_bytes
    _check(lib().bp_r1cs_verify(ctx.h, transcript.h, plan.h, G.h, H.h, bytes(g_le), bytes(h_le), bytes(V_le) or None, n, m, bytes(proof), len(proof),
                                bytes(r_le32) if r_le32 is not None else None), "bp_r1cs_verify")


def r1cs_prove_begin(ctx, transcript, G, H, h_le, m, a_L1, a_R1, a_O1, s_L1, s_R1, blindings3_le32):
    """Prover::prove up to create_randomized_constraints (prover.rs:323-369) for a system with second-phase constraints: "m", A_I1, A_O1,
    S1 and the 2-phase domain separator go onto the transcript; returns the opaque phase-1 bytes for r1cs_prove_finish.  The caller then
    plays the deferred callbacks (transcript.challenge_scalar = RandomizedConstraintSystem::challenge_scalar).  Vectors may be None / empty
    when there is no first-phase multiplier."""
    size = lib().bp_r1cs_phase1_bytes()
    out = ctypes.create_string_buffer(size)
    hv = lambda v: v.h if v is not None and len(v) else None
    _check(lib().bp_r1cs_prove_begin(ctx.h, transcript.h, G.h, H.h, bytes(h_le), m, hv(a_L1), hv(a_R1), hv(a_O1), hv(s_L1), hv(s_R1),
                                     bytes(blindings3_le32), out, size), "bp_r1cs_prove_begin")
    return out.raw


def r1cs_prove_finish(ctx, transcript, plan, G, H, g_le, h_le, phase1, a_L, a_R, a_O, v_blinding, s_L, s_R, blindings8_le32):
    """The rest of Prover::prove (prover.rs:371-593) over all n = n1 + n2 multipliers -> proof bytes.
    blindings8_le32: i2, o2, s2, t1, t3, t4, t5, t6."""
    n = len(a_L)
    size = lib().bp_r1cs_proof_bytes(ctx.curve, n)
    out = ctypes.create_string_buffer(size)
    _check(lib().bp_r1cs_prove_finish(ctx.h, transcript.h, plan.h, G.h, H.h, bytes(g_le), bytes(h_le), bytes(phase1), a_L.h, a_R.h, a_O.h,
                                      v_blinding.h if v_blinding is not None and len(v_blinding) else None, s_L.h, s_R.h, bytes(blindings8_le32), out, size),
           "bp_r1cs_prove_finish")
    return out.raw


def r1cs_verify_begin(ctx, transcript, m, proof):
    """Verifier::verify up to create_randomized_constraints (verifier.rs:276-287, 253); transcript only."""
    _check(lib().bp_r1cs_verify_begin(transcript.h, ctx.curve, m, bytes(proof), len(proof)), "bp_r1cs_verify_begin")


def r1cs_verify_finish(ctx, transcript, plan, G, H, g_le, h_le, V_le, n1, n, proof, r_le32=None):
    m = len(V_le) // ctx.point_bytes
    _check(lib().bp_r1cs_verify_finish(ctx.h, transcript.h, plan.h, G.h, H.h, bytes(g_le), bytes(h_le), bytes(V_le) or None, n1, n, m, bytes(proof), len(proof),
                                       bytes(r_le32) if r_le32 is not None else None), "bp_r1cs_verify_finish")


def r1cs_proof_compress(ctx, n, proof):
    size = lib().bp_r1cs_proof_compressed_bytes(ctx.curve, n)
    out = ctypes.create_string_buffer(size)
    _check(lib().bp_r1cs_proof_compress(ctx.h, n, bytes(proof), len(proof), out, size), "bp_r1cs_proof_compress")
    return out.raw


def r1cs_proof_decompress(ctx, n, data):
    size = lib().bp_r1cs_proof_bytes(ctx.curve, n)
    out = ctypes.create_string_buffer(size)
    _check(lib().bp_r1cs_proof_decompress(ctx.h, n, bytes(data), len(data), out, size), "bp_r1cs_proof_decompress")
    return out.raw


def r1cs_prover_polys(ctx, a_L, a_R, a_O, s_L, s_R, wL, wR, wO, y_le32):
    """-> (l_poly, r_poly) as VecPoly3 with the structurally-zero vectors filled by zeros"""
    out = (ctypes.c_void_p * 6)()
    _check(lib().bp_r1cs_prover_polys(ctx.h, _handles((a_L, a_R, a_O, s_L, s_R, wL, wR, wO)), bytes(y_le32), out), "bp_r1cs_prover_polys")
    v = [FieldElementVector(ctx, ctypes.c_void_p(out[i])) for i in range(6)]
    n = len(a_L)
    return VecPoly3(FieldElementVector.new(ctx, n), v[0], v[1], v[2]), VecPoly3(v[3], v[4], FieldElementVector.new(ctx, n), v[5])


def r1cs_ipp_inputs(ctx, l_eval, r_eval, y_le32, u_le32, n1, padded_n):
    """-> (l_vec, r_vec, G_factors, H_factors)"""
    out = (ctypes.c_void_p * 4)()
    _check(lib().bp_r1cs_ipp_inputs(ctx.h, l_eval.h, r_eval.h, bytes(y_le32), bytes(u_le32), n1, padded_n, out), "bp_r1cs_ipp_inputs")
    return tuple(FieldElementVector(ctx, ctypes.c_void_p(out[i])) for i in range(4))


def r1cs_verifier_scalars(ctx, transcript, L_le, R_le, padded_n, n1, wL, wR, wO, y_inv, x, u, a, b):
    """-> (u_sq, u_inv_sq, g_scalars, h_scalars)"""
    lg_n = len(L_le) // ctx.point_bytes
    us, uis = ctypes.create_string_buffer(max(1, lg_n) * 32), ctypes.create_string_buffer(max(1, lg_n) * 32)
    g, h = ctypes.c_void_p(), ctypes.c_void_p()
    _check(lib().bp_r1cs_verifier_scalars(ctx.h, transcript.h, bytes(L_le), bytes(R_le), lg_n, padded_n, n1, wL.h, wR.h, wO.h, bytes(y_inv), bytes(x),
                                          bytes(u), bytes(a), bytes(b), us, uis, ctypes.byref(g), ctypes.byref(h)), "bp_r1cs_verifier_scalars")
    return us.raw[: lg_n * 32], uis.raw[: lg_n * 32], FieldElementVector(ctx, g), FieldElementVector(ctx, h)
