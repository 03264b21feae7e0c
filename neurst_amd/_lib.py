"""ctypes binding of libneurst_hip.so (include/neurst_hip.h).

The library is the product: there is NO Python/CPU fallback.  If the shared
object is missing or a symbol is absent the import raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# NST_LIBRARY: another build of the SAME library (measurement twins such as lib/libneurst_hip_ablation.so); still a HIP shared
# object with the full ABI -- there is no other kind of back end to select
LIB_PATH = os.environ.get("NST_LIBRARY") or os.path.join(_HERE, "lib", "libneurst_hip.so")

NST_F32, NST_BF16 = 0, 1
NST_ABI_VERSION = 10
NST_COMM_F16, NST_COMM_U8 = 2, 3
NST_COMM_UNIQUE_ID_BYTES = 128


class NstGemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("trans_a", C.c_int), ("trans_b", C.c_int),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
        ("in_dtype", C.c_int), ("out_dtype", C.c_int),
        ("alpha", C.c_float),
        ("bias", C.c_void_p),
        ("relu", C.c_int),
        ("dropout_p", C.c_float),
        ("seed", C.c_uint64), ("stream_id", C.c_uint64),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("gate_src", C.c_void_p), ("ldg", C.c_int64),
        ("gate_scale", C.c_float),
        ("posenc", C.c_void_p), ("posenc_period", C.c_int),
        ("emb_scale", C.c_float),
        ("accumulate", C.c_int),
        ("split_k", C.c_int),
        ("workspace", C.c_void_p),
        ("workspace_bytes", C.c_int64),
        ("colsum", C.c_void_p), ("colsum_accumulate", C.c_int),
        ("reduce_job_out", C.c_void_p),
        ("rowdot_src", C.c_void_p), ("ldrs", C.c_int64), ("rowdot_dst", C.c_void_p), ("rowdot_rows", C.c_int),
        ("rowdot_heads", C.c_int),
    ]


class NstSplitkJob(C.Structure):
    _fields_ = [("slabs", C.c_void_p), ("C", C.c_void_p), ("ldc", C.c_int64), ("M", C.c_int), ("N", C.c_int),
                ("split", C.c_int), ("accumulate", C.c_int), ("cs_parts", C.c_void_p), ("cs_out", C.c_void_p),
                ("cs_accumulate", C.c_int), ("reserved", C.c_int)]


class NstLnFinalizeJob(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("nblocks", C.c_int), ("d", C.c_int),
                ("accumulate", C.c_int), ("reserved", C.c_int)]


class NstRowGemmDesc(C.Structure):
    _fields_ = [("rows", C.c_int64), ("n", C.c_int), ("k", C.c_int), ("trans_b", C.c_int), ("dtype", C.c_int),
                ("lda", C.c_int64), ("ldb", C.c_int64), ("dropout_p", C.c_float), ("eps", C.c_float),
                ("seed", C.c_uint64), ("stream_id", C.c_uint64)]


class NstAttnDesc(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("H", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("dh", C.c_int),
        ("ldq", C.c_int64), ("ldk", C.c_int64), ("ldv", C.c_int64), ("ldo", C.c_int64),
        ("dtype", C.c_int),
        ("scale", C.c_float),
        ("causal", C.c_int),
        ("float_min", C.c_float),
        ("dropout_p", C.c_float),
        ("seed", C.c_uint64), ("stream_id", C.c_uint64),
        ("dropout_mask", C.c_void_p), ("dropout_mask_bytes", C.c_int64),
        ("bsk", C.c_int64), ("bsv", C.c_int64),
        ("causal_offset", C.c_int), ("reserved0", C.c_int),
        ("ds_workspace", C.c_void_p), ("ds_workspace_bytes", C.c_int64),
    ]


class NstFfnDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_int64),
        ("d_model", C.c_int), ("filter_size", C.c_int), ("dtype", C.c_int),
        ("hidden_dropout_p", C.c_float),
        ("hidden_seed", C.c_uint64), ("hidden_stream_id", C.c_uint64),
        ("output_dropout_p", C.c_float),
        ("output_seed", C.c_uint64), ("output_stream_id", C.c_uint64),
        ("seed_offset", C.c_void_p),
        ("gate_bits", C.c_void_p), ("gate_bits_bytes", C.c_int64),
    ]


class NstTransposeJob(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("tiles_c", C.c_int),
                ("tile0", C.c_int)]


_P, _I, _L, _F, _U64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64

# name -> argtypes; every symbol declared in include/neurst_hip.h
SIGNATURES = {
    "nst_abi_version": [],
    "nst_last_error_string": [],
    "nst_grad_clip": [_P, _P, _I, _P, _I, _P, _L, _F, _F, _F, _P],
    "nst_crc32c": [_P, _L, C.c_uint32],
    "nst_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "nst_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _L, _P],
    "nst_layernorm_relu_fwd": [_P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "nst_layernorm_bwd_dropout": [_P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U64, _P, _P, _L, _I, _I, _I, _P, _L, _P],
    "nst_layernorm_relu_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _L, _P],
    "nst_layernorm_bwd_deferred": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U64, _P, _P, _L, _I, _I, _I, _P, _L, _P, _P],
    "nst_ln_finalize_multi": [_P, _I, _P],
    "nst_layernorm_relu_bwd_regate": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _I, _I, _I, _P, _L, _P, _P],
    "nst_add_layernorm_fwd": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _L, _I, _F, _I, _P],
    "nst_layernorm_bwd_mixed": [_P, _P, _I, _P, _P, _P, _P, _P, _P, _F, _U64, _U64, _P, _P, _L, _I, _I, _I, _P, _L, _P, _P],
    "nst_rowgemm_supported": [_I, _I, _I],
    "nst_gemm_add_layernorm_fwd": [C.POINTER(NstRowGemmDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nst_gemm_layernorm_bwd": [C.POINTER(NstRowGemmDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L, _P, _P],
    "nst_gemm_rowdot256": [C.POINTER(NstRowGemmDesc), _P, _P, _P, _P, _P, _I, _P],
    "nst_gemm": [C.POINTER(NstGemmDesc), _P, _P, _P, _P],
    "nst_gemm_wgrad_group": [_P, _P, _P, _P, _I, _P, _L, _P],
    "nst_splitk_reduce_multi": [_P, _I, _P],
    "nst_colsum": [_P, _P, _L, _I, _L, _I, _I, _P, _L, _P],
    "nst_seq_mask": [_P, _P, _I, _I, _I, _I, _F, _F, _P],
    "nst_xent_reduce": [_P, _P, _I, _I, _P, _P, _P, _P, _P],
    "nst_attention_dropout_mask_bytes": [C.POINTER(NstAttnDesc)],
    "nst_attention_fwd": [C.POINTER(NstAttnDesc), _P, _P, _P, _P, _P, _P, _P],
    "nst_attention_bwd": [C.POINTER(NstAttnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nst_conv1_ln_relu_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P],
    "nst_conv1_ln_relu_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P],
    "nst_conv2_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "nst_conv2_dgrad": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "nst_conv2_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P],
    "nst_embedding_fwd": [_P, _P, _P, _P, _L, _I, _I, _I, _F, _F, _U64, _U64, _I, _P],
    "nst_embedding_bwd": [_P, _P, _P, _L, _I, _I, _F, _F, _U64, _U64, _I, _P],
    "nst_scale_posenc_dropout_fwd": [_P, _P, _P, _L, _I, _I, _F, _F, _U64, _U64, _I, _P],
    "nst_scale_dropout_bwd": [_P, _P, _L, _F, _F, _U64, _U64, _I, _P],
    "nst_ls_xent_fwd": [_P, _P, _P, _P, _P, _L, _I, _L, _F, _I, _P],
    "nst_ls_xent_bwd": [_P, _P, _P, _P, _P, _L, _I, _L, _F, _F, _P, _I, _P],
    "nst_adam_update": [_P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P],
    "nst_adam_update_dev": [_P, _P, _P, _P, _P, _L, _F, _P, _F, _F, _F, _F, _P, _P],
    "nst_loss_scale_update": [_P, _L, _P, _F, _F, _P, _L, _P],
    "nst_cast_f32_to_bf16": [_P, _P, _L, _P],
    "nst_cast_bf16_to_f32": [_P, _P, _L, _P],
    "nst_probe_mfma": [_P, _P, _P, _P],
    "nst_probe_fetch": [_P, C.c_int64, C.c_int64, _I, _I, _I, _I, _I, _I, _P, _P],
    "nst_dropout_seed_offset_bind": [_P],
    "nst_dropout_seed_offset_set": [_U64, _P],
    "nst_dropout_seed_offset_add": [_U64, _P],
    "nst_ffn_supported": [_I, _I, _I],
    "nst_ffn_gate_bits_bytes": [C.POINTER(NstFfnDesc)],
    "nst_ffn_fwd": [C.POINTER(NstFfnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P],
    "nst_ffn_bwd": [C.POINTER(NstFfnDesc), _P, _P, _P, _P, _P, _P, _P, _P],
    "nst_ffn_ln_supported": [C.POINTER(NstFfnDesc)],
    "nst_ffn_ln_slab_bytes": [C.POINTER(NstFfnDesc)],
    "nst_ffn_add_layernorm_fwd": [C.POINTER(NstFfnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _L, _P],
    "nst_ffn_layernorm_bwd": [C.POINTER(NstFfnDesc), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _U64, _U64, _P, _P, _I, _P,
                              _L, _P, _P, _L, _P],
    "nst_transpose_bf16": [_P, _I, _I, _P],
    "nst_pack2d": [_P, _I, _I, _P],
    "nst_stream_create": [_I, C.POINTER(C.c_void_p)],
    "nst_stream_destroy": [_P],
    "nst_comm_unique_id": [_P, C.c_size_t],
    "nst_comm_init": [_P, C.c_size_t, _I, _I, C.POINTER(C.c_void_p)],
    "nst_comm_info": [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
    "nst_comm_allreduce_bucket": [_P, _P, _L, _I, C.POINTER(C.c_void_p), _I],
    "nst_comm_fence": [_P, _P],
    "nst_comm_broadcast": [_P, _P, _L, _I, _I, _P],
    "nst_comm_destroy": [_P],
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C neurst_amd/csrc -j8` (or `python -c 'import "
            f"__graft_entry__ as g; g.build()'`).  neurst_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud failure
        fn.argtypes = argtypes
        fn.restype = (C.c_char_p if name == "nst_last_error_string"
                      else C.c_int64 if name in ("nst_attention_dropout_mask_bytes", "nst_ffn_gate_bits_bytes", "nst_ffn_ln_slab_bytes")
                      else C.c_uint32 if name == "nst_crc32c" else C.c_int)
    ver = lib.nst_abi_version()
    if ver != NST_ABI_VERSION:
        raise ImportError(f"libneurst_hip.so ABI version {ver} != expected {NST_ABI_VERSION}")
    return lib


lib = _load()


class NstError(RuntimeError):
    pass


CALLS = [0]     # kernel-launching library calls so far (training/train_step.py: did a captured segment queue any kernel?)


def check(rc, what="", launches=True):
    """launches=False: a host-only entry point (it queues nothing on a stream)."""
    if launches:
        CALLS[0] += 1
    if rc != 0:
        msg = lib.nst_last_error_string()
        raise NstError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
