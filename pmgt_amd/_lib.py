"""ctypes bindings of include/pmgt_capi.h.

The HIP library is the ONLY compute path of this package: if it is missing or cannot be loaded the
import of anything that needs it raises — there is no eager/PyTorch or CPU fallback.
"""
import ctypes as C
import os

from . import _build

c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)


MAX_FEATS = 4      # PMGT_MAX_FEATS (include/pmgt_capi.h)


class PMGTConfigC(C.Structure):
    _fields_ = [("hidden_size", C.c_int), ("num_hidden_layers", C.c_int), ("num_attention_heads", C.c_int),
                ("intermediate_size", C.c_int), ("n_feats", C.c_int), ("feat_sizes", C.c_int * MAX_FEATS),
                ("max_position_embeddings", C.c_int), ("layer_norm_eps", C.c_float), ("beta", C.c_float),
                ("hidden_dropout_prob", C.c_float), ("attention_probs_dropout_prob", C.c_float), ("dtype", C.c_int)]


class TensorsC(C.Structure):
    _fields_ = [("params", C.c_void_p), ("grads", C.c_void_p), ("tables", C.c_void_p * MAX_FEATS),
                ("n_nodes", C.c_int64), ("rng_state", C.c_void_p), ("table_scales", C.c_float * MAX_FEATS)]


class BatchC(C.Structure):
    _fields_ = [("n_targets", C.c_int), ("n_pairs", C.c_int), ("seq_len", C.c_int),
                ("tgt_ids", C.c_void_p), ("tgt_mask", C.c_void_p), ("pair_ids", C.c_void_p), ("pair_mask", C.c_void_p),
                ("num_pairs", C.c_void_p), ("labels", C.c_void_p), ("nfr_masked_ids", C.c_void_p),
                ("nfr_targets", C.c_void_p), ("random_node_ratio", C.c_float), ("mask_node_ratio", C.c_float)]


class OutputsC(C.Structure):
    _fields_ = [("loss", C.c_void_p), ("logits", C.c_void_p), ("last_hidden", C.c_void_p), ("nfr_count", C.c_void_p)]


class AdamC(C.Structure):
    _fields_ = [("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("decay", C.c_void_p),
                ("lr", C.c_float), ("weight_decay", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("eps", C.c_float), ("max_grad_norm", C.c_float), ("step", C.c_void_p), ("scalars", C.c_void_p),
                ("scratch", C.c_void_p)]


DTYPE_F32, DTYPE_BF16, DTYPE_FP8 = 0, 1, 2
# void (*pmgt_grad_ready_fn)(void* user, int64_t offset, int64_t numel)  (include/pmgt_capi.h)
GRAD_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int64)
FLAG_TRAINING, FLAG_BACKWARD, FLAG_ACCUMULATE = 1, 2, 4
EPI_NONE, EPI_GELU, EPI_GELU_GRAD = 0, 1, 2

# every symbol include/pmgt_capi.h (product ABI) and include/pmgt_ops.h (single-kernel test entries) declare for libpmgt_hip.so
HIP_SYMBOLS = [
    "pmgt_last_error", "pmgt_abi_version", "pmgt_engine_create", "pmgt_engine_destroy", "pmgt_param_count",
    "pmgt_param_num_entries", "pmgt_param_entry", "pmgt_workspace_bytes", "pmgt_pretrain_step", "pmgt_encode_ids",
    "pmgt_encode_feats", "pmgt_encode_train", "pmgt_encode_backward", "pmgt_optimizer_step", "pmgt_profile_begin",
    "pmgt_profile_end", "pmgt_profile_sequence", "pmgt_profile_records", "pmgt_cast_from_f32", "pmgt_cast_to_f32", "pmgt_quantize_e4m3", "pmgt_dequantize_e4m3",
    "pmgt_engine_set_grad_ready_callback", "pmgt_engine_set_option", "pmgt_engine_get_option",
]
OPS_SYMBOLS = [
    "pmgt_op_gemm_nt", "pmgt_op_gemm_tn_slab_elems", "pmgt_op_gemm_tn", "pmgt_op_gemm_tn_bias", "pmgt_op_colsum",
    "pmgt_op_layernorm_fwd", "pmgt_op_layernorm_bwd", "pmgt_op_linear", "pmgt_op_linear_ln_bwd", "pmgt_op_attention_fwd", "pmgt_op_attention_bwd",
    "pmgt_op_qkvc_attention_fwd", "pmgt_op_attention_bwd_wgrad", "pmgt_op_attention_bwd_wgrad_parts",
    "pmgt_op_quant_rows_e4m3", "pmgt_op_gemm_nt_f8", "pmgt_op_gemm_tn_f8", "pmgt_op_qkvc_attention_fwd_f8",
    "pmgt_launch_trace_reset", "pmgt_launch_trace_count", "pmgt_op_seg_sort", "pmgt_op_seg_sort_temp_bytes", "pmgt_op_clock_probe", "pmgt_op_qkvc_attention_fwd_ex", "pmgt_op_attention_bwd_wgrad_vc2_parts",
]
# path options: pmgt_engine_set_option keys -> bit in the `path_opts` argument of the pmgt_op_* entries (include/pmgt_ops.h)
OPT = {k: 1 << i for i, k in enumerate((
    "tile_gemm", "valu_attention", "wave_attention_bwd", "no_shortcut", "no_fused_qkvc_attention", "no_head_major",
    "no_table_projection", "no_segment_sum", "consumer_quant", "no_fused_attention_bwd", "store_ln_input", "eager_reduce",
    "side_stream_reduce", "unfused_ln", "one_bucket", "small_arena", "no_role_split_ln", "no_tile_attention", "unfused_ln_bwd", "lockstep_attention_bwd", "side_stream_wgrad", "no_cls_only_attention_bwd", "no_beta_skip", "no_vc2_attention_bwd"))}
SAMPLER_SYMBOLS = [
    "pmgt_sampler_create", "pmgt_sampler_destroy", "pmgt_sampler_last_error", "pmgt_sampler_seed",
    "pmgt_sampler_context", "pmgt_sampler_batch", "pmgt_sampler_batch_mt", "pmgt_sampler_max_pairs",
    "pmgt_sampler_random_sample", "pmgt_sampler_randint", "pmgt_train_valid_split",
]

_hip = None
_sampler = None


def _load(path, what):
    if not os.path.exists(path):
        raise ImportError(f"{what} not built: {path} is missing. Run `python __graft_entry__.py` "
                          f"(or pmgt_amd._build.build_all()); pmgt_amd has no fallback path.")
    try:
        return C.CDLL(path)
    except OSError as e:   # fail loudly: no CPU/eager fallback exists
        raise ImportError(f"cannot load {what} ({path}): {e}") from e


def hip():
    """libpmgt_hip.so with argtypes set."""
    global _hip
    if _hip is not None:
        return _hip
    # PyTorch-ROCm first: it brings its own libamdhip64 / libhsa-runtime64.  If this library is loaded before torch, the
    # loader binds the system ROCm runtime instead and torch's later import mixes the two (device init then fails with
    # "no usable HIP device"); loaded after torch, the same SONAME resolves to the copy torch already mapped.
    import torch  # noqa: F401
    L = _load(_build.hip_lib_path(), "HIP engine library")
    vp, i, i64, f, u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32
    L.pmgt_last_error.restype = C.c_char_p
    L.pmgt_abi_version.restype = i
    L.pmgt_engine_create.restype = vp
    L.pmgt_engine_create.argtypes = [C.POINTER(PMGTConfigC)]
    L.pmgt_engine_destroy.argtypes = [vp]
    L.pmgt_engine_destroy.restype = None
    L.pmgt_param_count.restype = i64
    L.pmgt_param_count.argtypes = [vp]
    L.pmgt_param_num_entries.argtypes = [vp]
    L.pmgt_param_entry.argtypes = [vp, i, C.c_char_p, i, c_i64p, c_i64p, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    L.pmgt_workspace_bytes.restype = i64
    L.pmgt_workspace_bytes.argtypes = [vp, i, i, i, i]
    L.pmgt_pretrain_step.argtypes = [vp, C.POINTER(TensorsC), C.POINTER(BatchC), C.POINTER(OutputsC), vp, i64, i, vp]
    L.pmgt_encode_ids.argtypes = [vp, C.POINTER(TensorsC), vp, vp, i, i, vp, vp, vp, vp, i64, vp]
    featp = C.POINTER(C.c_void_p)      # const void* const* feats: host array of n_feats device pointers, or NULL
    L.pmgt_encode_feats.argtypes = [vp, C.POINTER(TensorsC), featp, vp, i, i, vp, vp, vp, vp, i64, vp]
    L.pmgt_encode_train.argtypes = [vp, C.POINTER(TensorsC), vp, featp, vp, i, i, vp, vp, i64, i, vp]
    L.pmgt_encode_backward.argtypes = [vp, C.POINTER(TensorsC), featp, vp, i, i, vp, i64, i, vp]
    L.pmgt_optimizer_step.argtypes = [vp, C.POINTER(TensorsC), C.POINTER(AdamC), vp]
    L.pmgt_profile_begin.argtypes = [vp]
    L.pmgt_profile_end.argtypes = [vp, C.c_char_p, i]
    L.pmgt_profile_sequence.argtypes = [vp, C.c_char_p, i]
    L.pmgt_profile_records.argtypes = [vp, C.c_char_p, i]
    L.pmgt_cast_from_f32.argtypes = [i, vp, vp, i64, vp]
    L.pmgt_cast_to_f32.argtypes = [i, vp, vp, i64, vp]
    L.pmgt_op_gemm_nt.argtypes = [i, vp, i64, vp, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, u32, vp]
    L.pmgt_op_gemm_tn_slab_elems.restype = i64
    L.pmgt_op_gemm_tn_slab_elems.argtypes = [i, i, i, i, u32]
    L.pmgt_op_gemm_tn.argtypes = [i, vp, i64, vp, i64, vp, i, i, i, vp, vp, i, vp, u32, vp]
    L.pmgt_op_colsum.argtypes = [i, vp, i64, i, i, vp, vp, vp]
    L.pmgt_op_layernorm_fwd.argtypes = [i, vp, vp, vp, vp, vp, i, i, f, f, u32, vp, vp]
    L.pmgt_op_layernorm_bwd.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, i, i, f, u32, f, u32, vp, vp]
    L.pmgt_op_qkvc_attention_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, vp]
    L.pmgt_engine_set_grad_ready_callback.argtypes = [vp, GRAD_READY_FN, vp]
    L.pmgt_engine_set_grad_ready_callback.restype = None
    L.pmgt_engine_set_option.argtypes = [vp, C.c_char_p, i]
    L.pmgt_engine_get_option.argtypes = [vp, C.c_char_p]
    L.pmgt_op_linear.argtypes = [i, vp, i64, vp, i64, vp, i64, i, i, i, vp, i, vp, i64, vp, i64, f, u32, vp, vp, vp, vp, vp, f, u32, vp]
    L.pmgt_launch_trace_count.argtypes = [C.c_char_p]
    L.pmgt_launch_trace_count.restype = C.c_int64
    L.pmgt_launch_trace_reset.restype = None
    L.pmgt_op_clock_probe.argtypes = [vp, i, vp]
    L.pmgt_op_qkvc_attention_fwd_ex.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, i, vp]
    L.pmgt_op_seg_sort_temp_bytes.argtypes = [i]
    L.pmgt_op_seg_sort_temp_bytes.restype = i64
    L.pmgt_op_seg_sort.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, i64, vp]
    L.pmgt_op_linear_ln_bwd.argtypes = [vp, i64, vp, i64, i, i, i, vp, i64, vp, vp, vp, vp, vp, vp, vp, f, u32, vp, vp, vp, u32, vp]
    L.pmgt_op_attention_fwd.argtypes = [i, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, u32, vp]
    L.pmgt_op_attention_bwd.argtypes = [i, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, u32, vp]
    L.pmgt_op_attention_bwd_wgrad.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, f, f, u32, u32, vp, i, vp]
    L.pmgt_op_attention_bwd_wgrad_parts.argtypes = [i]
    L.pmgt_op_attention_bwd_wgrad_vc2_parts.argtypes = [i]
    L.pmgt_op_gemm_tn_bias.argtypes = [i, vp, i64, vp, i64, i, i, i, vp, vp, vp, vp, i, i, u32, vp]
    L.pmgt_quantize_e4m3.argtypes = [vp, vp, i64, f, vp]
    L.pmgt_dequantize_e4m3.argtypes = [vp, vp, i64, f, vp]
    L.pmgt_op_quant_rows_e4m3.argtypes = [i, vp, i64, i, i, vp, i64, vp, vp]
    L.pmgt_op_gemm_nt_f8.argtypes = [vp, i64, vp, vp, f, vp, i64, vp, vp, i64, i, i, i, vp, vp, vp]
    L.pmgt_op_gemm_tn_f8.argtypes = [vp, i64, vp, i64, f, vp, i, i, i, vp, vp, i, vp, vp]
    L.pmgt_op_qkvc_attention_fwd_f8.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, u32, u32, vp, vp]
    _hip = L
    return L


class Ops:
    """The library as the per-kernel tests and profiling tools call it: every pmgt_op_* entry that takes a `path_opts` bit
    mask (include/pmgt_ops.h) gets it from `self.path` (an int, or option names through `use`), so call sites read as the
    kernel's own argument list.  Everything else passes straight through to the CDLL."""
    _BEFORE_STREAM = {"pmgt_op_gemm_nt", "pmgt_op_gemm_tn", "pmgt_op_gemm_tn_bias", "pmgt_op_linear", "pmgt_op_linear_ln_bwd", "pmgt_op_attention_fwd",
                      "pmgt_op_attention_bwd"}
    _LAST = {"pmgt_op_gemm_tn_slab_elems"}

    def __init__(self, lib=None):
        self._L = lib if lib is not None else hip()
        self.path = 0

    def use(self, *names):
        """Select path options by name for the following calls (no names = the product path); returns self."""
        self.path = 0
        for n in names:
            self.path |= OPT[n]
        return self

    def __getattr__(self, name):
        fn = getattr(self._L, name)
        if name in Ops._BEFORE_STREAM:
            return lambda *a: fn(*a[:-1], self.path, a[-1])
        if name in Ops._LAST:
            return lambda *a: fn(*a, self.path)
        return fn


def ops():
    """A fresh Ops view of libpmgt_hip.so (path options start at 0)."""
    return Ops()


def check(rc):
    if rc != 0:
        raise RuntimeError(f"pmgt HIP engine error {rc}: {hip().pmgt_last_error().decode()}")


def sampler():
    """libpmgt_sampler.so (host only) with argtypes set."""
    global _sampler
    if _sampler is not None:
        return _sampler
    L = _load(_build.sampler_lib_path(), "host sampler library")
    vp, i, i64 = C.c_void_p, C.c_int, C.c_int64
    L.pmgt_sampler_create.restype = vp
    L.pmgt_sampler_create.argtypes = [i64, vp, vp, vp, vp, i, i, i, i]
    L.pmgt_sampler_destroy.argtypes = [vp]
    L.pmgt_sampler_destroy.restype = None
    L.pmgt_sampler_last_error.restype = C.c_char_p
    L.pmgt_sampler_seed.argtypes = [vp, C.c_uint32]
    L.pmgt_sampler_seed.restype = None
    L.pmgt_sampler_context.argtypes = [vp, i64, vp, vp]
    L.pmgt_sampler_batch.argtypes = [vp, vp, i, i, vp, vp, vp, vp, vp, vp]
    L.pmgt_sampler_batch_mt.argtypes = [vp, vp, i, i, C.c_uint64, C.c_uint64, C.c_uint64, i, vp, vp, vp, vp, vp, vp]
    L.pmgt_sampler_max_pairs.argtypes = [vp, i]
    L.pmgt_sampler_random_sample.restype = C.c_double
    L.pmgt_sampler_random_sample.argtypes = [vp]
    L.pmgt_sampler_randint.restype = i64
    L.pmgt_sampler_randint.argtypes = [vp, i64]
    L.pmgt_train_valid_split.argtypes = [i64, C.c_double, C.c_uint32, vp, vp]
    _sampler = L
    return L
