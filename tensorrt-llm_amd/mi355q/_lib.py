"""ctypes binding of libmi355q.so (the C ABI in include/mi355q.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing the import
of any op fails loudly (build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C tensorrt-llm_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355Q_LIB") or os.path.normpath(os.path.join(_HERE, "..", "lib", "libmi355q.so"))

# nvinfer1::DataType codes
FLOAT, HALF, INT8, INT32, BOOL, UINT8, FP8, BF16, INT64 = range(9)
WEIGHT_INT8, WEIGHT_INT4 = 1, 2
LAYOUT_ROW_MAJOR, LAYOUT_REF_SM80, LAYOUT_REF_SM90, LAYOUT_GFX950 = range(4)
FIELD_FLOAT16, FIELD_FLOAT32, FIELD_FLOAT64, FIELD_INT8, FIELD_INT16, FIELD_INT32 = range(6)


class Dims(C.Structure):
    _fields_ = [("nbDims", C.c_int32), ("d", C.c_int32 * 8)]

    @classmethod
    def of(cls, shape):
        d = cls()
        d.nbDims = len(shape)
        for i, s in enumerate(shape):
            d.d[i] = int(s)
        return d

    def tolist(self):
        return [self.d[i] for i in range(self.nbDims)]


class TensorDesc(C.Structure):
    _fields_ = [("dims", Dims), ("type", C.c_int32), ("format", C.c_int32), ("scale", C.c_float)]

    @classmethod
    def of(cls, shape, dtype):
        t = cls()
        t.dims = Dims.of(shape)
        t.type = dtype
        t.format = 0
        t.scale = 1.0
        return t


class DynamicTensorDesc(C.Structure):
    _fields_ = [("desc", TensorDesc), ("min", Dims), ("max", Dims)]


class PluginField(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("type", C.c_int32), ("length", C.c_int32)]


class WeightOnlyParams(C.Structure):
    _fields_ = [
        ("qweight", C.c_void_p), ("scales", C.c_void_p), ("zeros", C.c_void_p), ("in_", C.c_void_p),
        ("act_scale", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
        ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32), ("group_size", C.c_int32),
        ("weight_type", C.c_int32), ("act_type", C.c_int32),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("alpha", C.c_void_p), ("act_fp8", C.c_int32),
    ]


class MoeGemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("weight_scales", C.c_void_p), ("biases", C.c_void_p), ("C", C.c_void_p),
        ("total_rows_before_expert", C.c_void_p),
        ("total_rows", C.c_int64), ("gemm_n", C.c_int64), ("gemm_k", C.c_int64),
        ("num_experts", C.c_int32), ("activation_type", C.c_int32), ("weight_type", C.c_int32), ("act_type", C.c_int32),
        ("splitk_workspace", C.c_void_p), ("splitk_workspace_bytes", C.c_size_t),
    ]


class MoeParams(C.Structure):
    _fields_ = [
        ("input", C.c_void_p), ("gating_output", C.c_void_p),
        ("fc1_expert_weights", C.c_void_p), ("fc1_scales", C.c_void_p), ("fc1_expert_biases", C.c_void_p),
        ("fc2_expert_weights", C.c_void_p), ("fc2_scales", C.c_void_p), ("fc2_expert_biases", C.c_void_p),
        ("finished", C.c_void_p), ("output", C.c_void_p), ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("num_rows", C.c_int32), ("hidden_size", C.c_int32), ("inter_size", C.c_int32), ("num_experts", C.c_int32),
        ("k", C.c_int32), ("activation_type", C.c_int32), ("act_type", C.c_int32), ("weight_type", C.c_int32),
        ("ep_size", C.c_int32), ("ep_rank", C.c_int32), ("tp_size", C.c_int32), ("tp_rank", C.c_int32),
        ("normalization_mode", C.c_int32),
    ]


class DecodeAttentionParams(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("qkv_bias", C.c_void_p), ("out", C.c_void_p),
        ("sequence_lengths", C.c_void_p), ("block_pointers", C.c_void_p),
        ("kv_scale_orig_quant", C.c_void_p), ("kv_scale_quant_orig", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("batch", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("head_size", C.c_int32),
        ("max_blocks_per_seq", C.c_int32), ("tokens_per_block", C.c_int32), ("max_seq_len", C.c_int32),
        ("act_type", C.c_int32), ("kv_type", C.c_int32),
        ("position_embedding_type", C.c_int32), ("rotary_embedding_dim", C.c_int32),
        ("rotary_embedding_base", C.c_float), ("rotary_embedding_scale_type", C.c_int32),
        ("rotary_embedding_scale", C.c_float), ("rotary_embedding_max_positions", C.c_int32),
        ("q_scaling", C.c_float), ("multi_block_mode", C.c_int32),
        ("cyclic_attention_window_size", C.c_int32), ("sink_token_length", C.c_int32),
        ("beam_width", C.c_int32), ("max_attention_window_size", C.c_int32),
        ("cache_indirection", C.c_void_p), ("input_lengths", C.c_void_p),
        ("kv_linear_buffer", C.c_void_p), ("linear_max_seq_len", C.c_int32),
        ("alibi_slopes", C.c_void_p),
        ("position_shift_enabled", C.c_int32),
        ("relative_attention_bias", C.c_void_p), ("relative_attention_bias_stride", C.c_int32), ("max_distance", C.c_int32),
    ]


class ContextAttentionParams(C.Structure):
    _fields_ = [
        ("qkv", C.c_void_p), ("qkv_bias", C.c_void_p), ("out", C.c_void_p),
        ("context_lengths", C.c_void_p), ("block_pointers", C.c_void_p),
        ("kv_scale_orig_quant", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("batch", C.c_int32), ("num_heads", C.c_int32), ("num_kv_heads", C.c_int32), ("head_size", C.c_int32),
        ("max_blocks_per_seq", C.c_int32), ("tokens_per_block", C.c_int32), ("max_context_length", C.c_int32),
        ("num_tokens", C.c_int32), ("remove_padding", C.c_int32),
        ("act_type", C.c_int32), ("kv_type", C.c_int32),
        ("position_embedding_type", C.c_int32), ("rotary_embedding_dim", C.c_int32),
        ("rotary_embedding_base", C.c_float), ("rotary_embedding_scale_type", C.c_int32),
        ("rotary_embedding_scale", C.c_float), ("rotary_embedding_max_positions", C.c_int32),
        ("q_scaling", C.c_float),
        ("kv_linear_buffer", C.c_void_p), ("linear_max_seq_len", C.c_int32),
        ("sequence_lengths", C.c_void_p), ("max_kv_length", C.c_int32),
        ("cyclic_attention_window_size", C.c_int32), ("sink_token_length", C.c_int32),
        ("alibi_slopes", C.c_void_p),
        ("position_shift_enabled", C.c_int32),
        ("relative_attention_bias", C.c_void_p), ("relative_attention_bias_stride", C.c_int32), ("max_distance", C.c_int32),
    ]


MAX_RANKS_PER_NODE, IPC_HANDLE_BYTES, COMM_UNIQUE_ID_BYTES = 8, 64, 128
AR_RING, AR_ONESHOT, AR_TWOSHOT, AR_AUTO = range(4)


class AllReduceParams(C.Structure):
    _fields_ = [
        ("elts_total", C.c_size_t), ("elts_per_rank", C.c_size_t), ("elts_per_block", C.c_size_t),
        ("rank_offset", C.c_size_t), ("ranks_per_node", C.c_size_t), ("rank", C.c_size_t), ("local_rank", C.c_size_t),
        ("barrier_flag", C.c_uint32),
        ("peer_barrier_ptrs_in", C.c_void_p * MAX_RANKS_PER_NODE),
        ("peer_barrier_ptrs_out", C.c_void_p * MAX_RANKS_PER_NODE),
        ("peer_comm_buffer_ptrs", C.c_void_p * MAX_RANKS_PER_NODE),
        ("local_output_buffer_ptr", C.c_void_p), ("timeout_flag", C.c_void_p),
        ("local_input_buffer_ptr", C.c_void_p), ("comm_buffer_bytes", C.c_size_t), ("timeout_ms", C.c_uint32),
        ("peer_comm_buffer_ptrs_ping", C.c_void_p * MAX_RANKS_PER_NODE),
        ("peer_comm_buffer_ptrs_pong", C.c_void_p * MAX_RANKS_PER_NODE),
        ("max_workgroups", C.c_uint32),
    ]


class GatedActParams(C.Structure):
    _fields_ = [
        ("act_in", C.c_void_p), ("gate_in", C.c_void_p), ("smoother", C.c_void_p), ("out", C.c_void_p),
        ("out_quant", C.c_void_p), ("scale", C.c_void_p), ("dynamic_scale", C.c_void_p),
        ("rows", C.c_int32), ("cols", C.c_int32), ("ld_act", C.c_int64), ("ld_gate", C.c_int64), ("type", C.c_int32),
    ]


class Mi355qError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"mi355q status {status}: {msg}")
        self.status = status


_lib = None

# every symbol include/mi355q.h declares: name -> (restype, argtypes)
_vp, _sz, _i32, _i64 = C.c_void_p, C.c_size_t, C.c_int32, C.c_int64
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    "mi355q_last_error": (C.c_char_p, []),
    "mi355q_version": (C.c_char_p, []),
    "mi355q_plugin_create": (_i32, [C.c_char_p, C.POINTER(PluginField), _i32, _PP]),
    "mi355q_plugin_deserialize": (_i32, [C.c_char_p, _vp, _sz, _PP]),
    "mi355q_plugin_destroy": (None, [_vp]),
    "mi355q_plugin_clone": (_i32, [_vp, _PP]),
    "mi355q_plugin_type": (C.c_char_p, [_vp]),
    "mi355q_plugin_nb_outputs": (_i32, [_vp]),
    "mi355q_plugin_get_output_dimensions": (_i32, [_vp, _i32, C.POINTER(Dims), _i32, C.POINTER(Dims)]),
    "mi355q_plugin_supports_format_combination": (_i32, [_vp, _i32, C.POINTER(TensorDesc), _i32, _i32]),
    "mi355q_plugin_get_output_data_type": (_i32, [_vp, _i32, C.POINTER(_i32), _i32]),
    "mi355q_plugin_configure": (_i32, [_vp, C.POINTER(DynamicTensorDesc), _i32, C.POINTER(DynamicTensorDesc), _i32]),
    "mi355q_plugin_get_workspace_size": (_sz, [_vp, C.POINTER(TensorDesc), _i32, C.POINTER(TensorDesc), _i32]),
    "mi355q_plugin_enqueue": (_i32, [_vp, C.POINTER(TensorDesc), C.POINTER(TensorDesc), _PP, _PP, _vp, _vp]),
    "mi355q_plugin_get_serialization_size": (_sz, [_vp]),
    "mi355q_plugin_serialize": (_i32, [_vp, _vp]),
    "mi355q_get_plugin_names": (C.POINTER(C.c_char_p), [C.POINTER(_i32)]),
    "mi355q_preprocess_weights_for_mixed_gemm": (_i32, [_vp, _vp, C.POINTER(_sz), _i32, _i32]),
    "mi355q_unprocess_weights_for_mixed_gemm": (_i32, [_vp, _vp, C.POINTER(_sz), _i32, _i32]),
    "mi355q_unpack_int4_packed_tensor_to_int8": (_i32, [_vp, _vp, _sz]),
    "mi355q_pack_int8_tensor_to_packed_int4": (_i32, [_vp, _vp, _sz]),
    "mi355q_symmetric_quantize": (_i32, [_vp, _vp, _vp, _vp, C.POINTER(_sz), _i32, _i32]),
    "mi355q_weight_only_gemm_workspace_size": (_sz, [_i32, _i32, _i32, _i32]),
    "mi355q_weight_only_gemm": (_i32, [C.POINTER(WeightOnlyParams), _vp]),
    "mi355q_weight_only_dequantize": (_i32, [C.POINTER(WeightOnlyParams), _vp]),
    "mi355q_apply_per_channel_scale": (_i32, [_vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    "mi355q_int8_gemm": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_per_token_quantization": (_i32, [_vp, _vp, _i64, _i64, _vp, _i32, _vp]),
    "mi355q_quantize_tensor": (_i32, [_vp, _vp, _i64, _vp, _i32, _vp]),
    "mi355q_gated_silu": (_i32, [_vp, _vp]),
    "mi355q_rms_norm": (_i32, [_vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mi355q_layer_norm": (_i32, [_vp, _vp, _vp, _vp, C.c_float, _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "mi355q_fp8_gemm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "mi355q_gemm8_workspace_size": (C.c_size_t, [_i32, _i32, _i32]),
    "mi355q_int8_gemm_ws": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, C.c_size_t, _vp]),
    "mi355q_fp8_gemm_ws": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _vp, C.c_size_t, _vp]),
    "mi355q_quantize_e4m3": (_i32, [_vp, _vp, _vp, _i64, _i64, _i32, _i32, _vp]),
    "mi355q_moe_gemm_bias_act": (_i32, [C.POINTER(MoeGemmParams), _vp]),
    "mi355q_moe_workspace_size": (_sz, [C.POINTER(MoeParams)]),
    "mi355q_run_moe": (_i32, [C.POINTER(MoeParams), _vp]),
    "mi355q_decode_attention_workspace_size": (_sz, [C.POINTER(DecodeAttentionParams)]),
    "mi355q_masked_multihead_attention": (_i32, [C.POINTER(DecodeAttentionParams), _vp]),
    "mi355q_decode_attention_static_scratch_size": (_sz, []),
    "mi355q_masked_multihead_attention_static": (_i32, [C.POINTER(DecodeAttentionParams), _vp]),
    "mi355q_context_attention_workspace_size": (_sz, [C.POINTER(ContextAttentionParams)]),
    "mi355q_context_attention": (_i32, [C.POINTER(ContextAttentionParams), _vp]),
    "mi355q_allreduce_select_implementation": (_i32, [_sz, _i32, _sz]),
    "mi355q_allreduce_max_workspace_size": (_sz, [_i32]),
    "mi355q_allreduce_params_deserialize": (_i32, [C.POINTER(AllReduceParams), _PP, _sz, _sz, C.c_uint32]),
    "mi355q_custom_all_reduce": (_i32, [C.POINTER(AllReduceParams), _vp, _sz, _i32, _i32, _vp]),
    "mi355q_custom_all_gather": (_i32, [C.POINTER(AllReduceParams), _vp, _vp, _sz, _vp]),
    "mi355q_multi_gpu_barrier": (_i32, [C.POINTER(AllReduceParams), _vp]),
    "mi355q_allreduce_barrier_bytes": (_sz, []),
    "mi355q_allreduce_alloc_timeout_flag": (_i32, [_PP]),
    "mi355q_allreduce_free_timeout_flag": (_i32, [_vp]),
    "mi355q_ipc_alloc": (_i32, [_sz, _PP]),
    "mi355q_ipc_free": (_i32, [_vp]),
    "mi355q_ipc_get_handle": (_i32, [_vp, _vp]),
    "mi355q_ipc_open_handle": (_i32, [_vp, _PP]),
    "mi355q_ipc_close_handle": (_i32, [_vp]),
    "mi355q_comm_get_unique_id": (_i32, [_vp]),
    "mi355q_comm_init_rank": (_i32, [C.POINTER(_i32), _i32, _vp, _i32]),
    "mi355q_comm_exists": (_i32, [C.POINTER(_i32), _i32]),
    "mi355q_comm_destroy": (_i32, [C.POINTER(_i32), _i32]),
}


def lib():
    """Load libmi355q.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP library is the only implementation. Build it with "
                "`make -C tensorrt-llm_amd/csrc` (or __graft_entry__.build()).")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status):
    if status != 0:
        raise Mi355qError(status, lib().mi355q_last_error().decode())
