"""Host-side mirror of the reference's Python callers of the hot path, executing eagerly.

Same names, argument order and meaning as
  tensorrt_llm/quantization/functional.py:27-359   (smooth_quant_gemm, weight_only_quant_matmul,
      weight_only_groupwise_quant_matmul, smooth_quant_layer_norm, smooth_quant_rms_norm, quantize_tensor,
      quantize_per_token)
  tensorrt_llm/functional.py:3487-3930                      (gpt_attention, generation phase)
  torch.ops.trtllm.* of cpp/tensorrt_llm/thop/weightOnlyQuantOp.cpp
so the parity tests read like the reference's own tests.  Every function forwards to the C ABI
through a plugin object exactly as the reference forwards to the TensorRT plugin of the same name.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .plugin import Plugin, str_dtype_to_trt, torch_dtype_to_trt, current_stream_ptr

BIAS, ZERO, PRE_QUANT_SCALE, FP8_ALPHA = 1, 2, 4, 8


# ----------------------------------------------------------------------------- thop ops (CPU tensors)
def _quant_type_id(quant_type):
    if quant_type in (torch.int8, L.WEIGHT_INT8, "int8"):
        return L.WEIGHT_INT8
    if quant_type in (getattr(torch, "quint4x2", None), L.WEIGHT_INT4, "int4"):
        return L.WEIGHT_INT4
    raise ValueError("Invalid quantization type: must be torch.int8 or torch.quint4x2")


def _shape3(t, quant_type_id):
    assert t.dim() in (2, 3), "Invalid dim. The dim of weight should be 2 or 3"
    experts = 1 if t.dim() == 2 else t.shape[0]
    rows = t.shape[-2]
    cols = t.shape[-1] * (2 if quant_type_id == L.WEIGHT_INT4 else 1)
    return (C.c_size_t * 3)(experts, rows, cols)


def preprocess_weights_for_mixed_gemm(row_major_quantized_weight, quant_type, layout=L.LAYOUT_GFX950):
    """torch.ops.trtllm.preprocess_weights_for_mixed_gemm (weightOnlyQuantOp.cpp:116-141).  The
    reference picks the layout from the GPU's SM version; here the default is the gfx950 layout and
    the reference's SM80 / SM90 layouts can be requested explicitly."""
    w = row_major_quantized_weight
    assert w.device.type == "cpu" and w.dtype == torch.int8 and w.is_contiguous(), \
        "Quantized tensor must be a contiguous int8 CPU tensor"
    qt = _quant_type_id(quant_type)
    out = torch.zeros_like(w)
    L.check(L.lib().mi355q_preprocess_weights_for_mixed_gemm(out.data_ptr(), w.data_ptr(), _shape3(w, qt), qt, layout))
    return out


def unprocess_weights_for_mixed_gemm(processed_weight, quant_type, layout):
    """Inverse of the above: any layout -> row-major (for ingesting reference-preprocessed weights)."""
    w = processed_weight
    assert w.device.type == "cpu" and w.dtype == torch.int8 and w.is_contiguous()
    qt = _quant_type_id(quant_type)
    out = torch.zeros_like(w)
    L.check(L.lib().mi355q_unprocess_weights_for_mixed_gemm(out.data_ptr(), w.data_ptr(), _shape3(w, qt), qt, layout))
    return out


def unpack_int4_packed_tensor_to_int8(weight):
    """torch.ops.trtllm.unpack_int4_packed_tensor_to_int8 (weightOnlyQuantOp.cpp:274-306)."""
    assert weight.device.type == "cpu" and weight.dtype == torch.int8 and weight.numel() != 0
    w = weight.contiguous()
    out = torch.zeros(list(w.shape[:-1]) + [w.shape[-1] * 2], dtype=torch.int8)
    L.check(L.lib().mi355q_unpack_int4_packed_tensor_to_int8(out.data_ptr(), w.data_ptr(), w.numel()))
    return out


def pack_int8_tensor_to_packed_int4(weight):
    """torch.ops.trtllm.pack_int8_tensor_to_packed_int4 (weightOnlyQuantOp.cpp:308-343)."""
    assert weight.device.type == "cpu" and weight.dtype == torch.int8 and weight.numel() != 0
    w = weight.contiguous()
    out = torch.zeros(list(w.shape[:-1]) + [(w.shape[-1] + 1) // 2], dtype=torch.int8)
    L.check(L.lib().mi355q_pack_int8_tensor_to_packed_int4(out.data_ptr(), w.data_ptr(), out.numel()))
    return out


def _symmetric_quantize_last_axis_of_batched_matrix(weight, quant_type, layout=L.LAYOUT_GFX950):
    """torch.ops.trtllm._symmetric_quantize_last_axis_of_batched_matrix (weightOnlyQuantOp.cpp:143-241):
    returns (unprocessed int8, processed int8, scales in weight.dtype)."""
    assert weight.device.type == "cpu" and weight.numel() != 0
    assert weight.dtype in (torch.float32, torch.float16, torch.bfloat16), "Weight must be FP16 or BF16"
    qt = _quant_type_id(quant_type)
    wf = weight.float().contiguous()
    experts = 1 if wf.dim() == 2 else wf.shape[0]
    rows, cols = wf.shape[-2], wf.shape[-1]
    bpc = cols * (4 if qt == L.WEIGHT_INT4 else 8) // 8
    qshape = [rows, bpc] if wf.dim() == 2 else [experts, rows, bpc]
    unprocessed = torch.empty(qshape, dtype=torch.int8)
    processed = torch.empty(qshape, dtype=torch.int8)
    scales = torch.empty([cols] if wf.dim() == 2 else [experts, cols], dtype=torch.float32)
    shape = (C.c_size_t * 3)(experts, rows, cols)
    L.check(L.lib().mi355q_symmetric_quantize(processed.data_ptr(), unprocessed.data_ptr(), scales.data_ptr(),
                                              wf.data_ptr(), shape, qt, layout))
    return unprocessed, processed, scales.to(weight.dtype)


def symmetric_quantize_last_axis_of_batched_matrix(weight, quant_type, layout=L.LAYOUT_GFX950):
    _, processed, scales = _symmetric_quantize_last_axis_of_batched_matrix(weight, quant_type, layout)
    return processed, scales


# ----------------------------------------------------------------------------- quantized GEMMs (device)
def _empty_like_opt(t, dtype):
    return t if t is not None and t.numel() else None


def weight_only_groupwise_quant_matmul(input, pre_quant_scale, weights, scales, zeros, biases, alpha,
                                       quant_algo, group_size, dtype="float16"):
    """quantization/functional.py:107-188.  `weights` is the preprocessed [K, N/4] tensor typed as
    the activation dtype (4 int4 per 16-bit element), exactly what the reference feeds the plugin."""
    plg = Plugin("WeightOnlyGroupwiseQuantMatmul", [
        ("type_id", str_dtype_to_trt(dtype), L.FIELD_INT32),
        ("quant_algo", quant_algo, L.FIELD_INT32),
        ("group_size", group_size, L.FIELD_INT32)])
    plug_inputs = [input]
    if quant_algo & PRE_QUANT_SCALE:
        plug_inputs += [pre_quant_scale]
    plug_inputs += [weights, scales]
    if quant_algo & ZERO:
        plug_inputs += [zeros]
    if quant_algo & BIAS:
        plug_inputs += [biases]
    if quant_algo & FP8_ALPHA:
        plug_inputs += [alpha]
    return plg(*plug_inputs)


ACT_GELU, ACT_RELU, ACT_SILU, ACT_SWIGLU, ACT_GEGLU, ACT_IDENTITY = range(6)  # ActivationType, moe_gemm_kernels.h:27-36


def moe_gemm_bias_act(a, expert_weights, weight_scales, biases, total_rows_before_expert, weight_type_id,
                      activation_type=ACT_IDENTITY):
    """MoeGemmRunner<T, uint8 | uint4>::moeGemmBiasAct (moe_gemm_kernels.h:51-58): a [rows, k] sorted by expert,
    expert_weights [E, ...] preprocessed per expert, weight_scales [E, n], biases [E, n] or None,
    total_rows_before_expert int64 [E] on the device.  Returns [rows, n]."""
    import ctypes as C
    E, n = weight_scales.shape
    rows, k = a.shape
    out = torch.empty(rows, n, dtype=a.dtype, device=a.device)
    p = L.MoeGemmParams()
    p.A, p.B, p.weight_scales = a.data_ptr(), expert_weights.data_ptr(), weight_scales.data_ptr()
    p.biases = biases.data_ptr() if biases is not None else None
    p.C, p.total_rows_before_expert = out.data_ptr(), total_rows_before_expert.data_ptr()
    p.total_rows, p.gemm_n, p.gemm_k, p.num_experts = rows, n, k, E
    p.activation_type, p.weight_type = activation_type, weight_type_id
    p.act_type = L.HALF if a.dtype == torch.float16 else L.BF16
    L.check(L.lib().mi355q_moe_gemm_bias_act(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    return out


def mixture_of_experts(hidden_states, routing, expert_weights_1, expert_weights_2, expert_scales_1, expert_scales_2,
                       top_k, activation_type, weight_type_id, expert_bias_1=None, expert_bias_2=None, finished=None,
                       normalization_mode=0, tp_size=1, tp_rank=0, parallelism_mode=0):
    """tensorrt_llm/layers/moe.py `_moe_plugin`: the MixtureOfExperts plugin with weight-only quantised experts.
    expert_weights_*: int8 [E_local, in, out] (int8) or [E_local, in, out / 2] (packed int4) already preprocessed per
    expert; routing: fp32 [tokens, E] logits.  Returns the hidden states' shape."""
    hidden = hidden_states.shape[-1]
    int4 = weight_type_id == L.WEIGHT_INT4
    E = routing.shape[-1]
    gated = activation_type in (ACT_SWIGLU, ACT_GEGLU)
    inter = expert_weights_1.shape[2] * (2 if int4 else 1) // (2 if gated else 1)
    fields = [("number_of_experts", E, L.FIELD_INT32), ("top_k", top_k, L.FIELD_INT32),
              ("expert_hidden_size", hidden, L.FIELD_INT32), ("expert_inter_size", inter, L.FIELD_INT32),
              ("activation_type", activation_type, L.FIELD_INT32),
              ("type_id", L.HALF if hidden_states.dtype == torch.float16 else L.BF16, L.FIELD_INT32),
              ("weight_type_id", L.INT8, L.FIELD_INT32), ("quant_mode", 1 if int4 else 2, L.FIELD_INT32),
              ("use_finished", int(finished is not None), L.FIELD_INT32),
              ("use_bias", int(expert_bias_1 is not None), L.FIELD_INT32),
              ("tp_size", tp_size, L.FIELD_INT32), ("tp_rank", tp_rank, L.FIELD_INT32),
              ("parallelism_mode", parallelism_mode, L.FIELD_INT32),
              ("normalization_mode", normalization_mode, L.FIELD_INT32)]
    inputs = [hidden_states, routing, expert_weights_1, expert_weights_2]
    if expert_bias_1 is not None:
        inputs += [expert_bias_1, expert_bias_2]
    if finished is not None:
        inputs += [finished]
    inputs += [expert_scales_1, expert_scales_2]
    return Plugin("MixtureOfExperts", fields)(*inputs)


def weight_only_quant_matmul(input, weights, scales, weightTypeId, dtype="float16"):
    """quantization/functional.py:68-104.  weights: int8 [K, N] (int8) or [K, N/2] (packed int4)."""
    plg = Plugin("WeightOnlyQuantMatmul", [
        ("type_id", str_dtype_to_trt(dtype), L.FIELD_INT32),
        ("weight_type_id", weightTypeId, L.FIELD_INT32)])
    return plg(input, weights, scales)


def smooth_quant_gemm(input, weights, scales_a, scales_b, per_token_scaling, per_channel_scaling,
                      dtype="float16"):
    """quantization/functional.py:27-65 (the output dtype is the plugin_config dtype there)."""
    plg = Plugin("SmoothQuantGemm", [
        ("has_per_channel_scaling", 1 if per_channel_scaling else 0, L.FIELD_INT32),
        ("has_per_token_scaling", 1 if per_token_scaling else 0, L.FIELD_INT32),
        ("type_id", str_dtype_to_trt(dtype), L.FIELD_INT32)])
    return plg(input, weights, scales_a, scales_b)


def quantize_per_token(x):
    """quantization/functional.py:332-359: returns (int8 tensor, fp32 per-token scales [.., 1])."""
    return Plugin("QuantizePerToken")(x)


def quantize_tensor(x, scale_factor):
    """quantization/functional.py:297-329 `quantize_tensor` (QuantizeTensor plugin): static per-tensor int8."""
    return Plugin("QuantizeTensor")(x, scale_factor)


def _norm_dtype(input, dtype):
    return dtype if dtype is not None else {torch.float16: "float16", torch.bfloat16: "bfloat16",
                                            torch.float32: "float32"}[input.dtype]


def smooth_quant_rms_norm(input, normalized_shape, weight=None, bias=None, scale=None, eps=1e-05,
                          dynamic_act_scaling=False, dtype=None):
    """quantization/functional.py:247-294: RmsnormQuantization plugin; weight / bias default to ones / zeros like
    there.  Returns int8, or (int8, per-token scales [.., 1]) with dynamic_act_scaling."""
    n = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[-1]
    dtype = _norm_dtype(input, dtype)
    plg = Plugin("RmsnormQuantization", [("eps", float(eps), L.FIELD_FLOAT32),
                                         ("dyn_act_scaling", int(dynamic_act_scaling), L.FIELD_INT32),
                                         ("type_id", str_dtype_to_trt(dtype), L.FIELD_INT32)])
    if weight is None:
        weight = torch.ones(n, dtype=input.dtype, device=input.device)
    if bias is None:
        bias = torch.zeros(n, dtype=input.dtype, device=input.device)
    return plg(input, weight, bias, scale)


def smooth_quant_layer_norm(input, normalized_shape, weight=None, bias=None, scale=None, eps=1e-05,
                            use_diff_of_squares=True, dynamic_act_scaling=False, dtype=None):
    """quantization/functional.py:191-244: LayernormQuantization plugin."""
    n = normalized_shape if isinstance(normalized_shape, int) else normalized_shape[-1]
    dtype = _norm_dtype(input, dtype)
    plg = Plugin("LayernormQuantization", [("eps", float(eps), L.FIELD_FLOAT32),
                                           ("use_diff_of_squares", int(use_diff_of_squares), L.FIELD_INT32),
                                           ("dyn_act_scaling", int(dynamic_act_scaling), L.FIELD_INT32),
                                           ("type_id", str_dtype_to_trt(dtype), L.FIELD_INT32)])
    if weight is None:
        weight = torch.ones(n, dtype=input.dtype, device=input.device)
    if bias is None:
        bias = torch.zeros(n, dtype=input.dtype, device=input.device)
    return plg(input, weight, bias, scale)


def rms_norm(input, weight, bias=None, eps=1e-05):
    """invokeGeneralRmsNorm without quantisation (the launcher below the plugins): T out."""
    x = input.contiguous()
    out = torch.empty_like(x)
    h = x.shape[-1]
    L.check(L.lib().mi355q_rms_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(),
                                    bias.data_ptr() if bias is not None else None, float(eps), x.numel() // h, h,
                                    None, None, None, torch_dtype_to_trt(x.dtype), current_stream_ptr()))
    return out


def layer_norm(input, weight, bias=None, eps=1e-05, use_diff_of_squares=True):
    """invokeGeneralLayerNorm without quantisation: T out."""
    x = input.contiguous()
    out = torch.empty_like(x)
    h = x.shape[-1]
    L.check(L.lib().mi355q_layer_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(),
                                      bias.data_ptr() if bias is not None else None, float(eps), x.numel() // h, h,
                                      int(use_diff_of_squares), None, None, None, torch_dtype_to_trt(x.dtype),
                                      current_stream_ptr()))
    return out


def gated_silu(act_in, gate_in, smoother=None, scale=None, dynamic_act_scaling=False):
    """SmoothQuantGatedMLP.forward between the gate/up GEMMs and the down projection (quantization/layers.py:922-941):
    y = T(T(T(silu(act_in)) * gate_in) / smoother), then quantize_tensor(y, scale) (static), quantize_per_token(y)
    (dynamic: returns (int8, scales [..., 1])), or y itself.  act_in / gate_in: [..., cols] views whose last dimension
    is contiguous (row strides may differ: the two halves of a fused gate|up GEMM output are fine)."""
    cols = act_in.shape[-1]
    a2, g2 = act_in.reshape(-1, cols), gate_in.reshape(-1, cols)
    assert a2.stride(-1) == 1 and g2.stride(-1) == 1
    rows = a2.shape[0]
    p = L.GatedActParams()
    p.act_in, p.gate_in = a2.data_ptr(), g2.data_ptr()
    p.ld_act, p.ld_gate = (a2.stride(0) if rows > 1 else cols), (g2.stride(0) if rows > 1 else cols)
    p.smoother = smoother.data_ptr() if smoother is not None else None
    p.rows, p.cols, p.type = rows, cols, torch_dtype_to_trt(act_in.dtype)
    out = q = scales = None
    if dynamic_act_scaling or scale is not None:
        q = torch.empty(list(act_in.shape), dtype=torch.int8, device=act_in.device)
        p.out_quant = q.data_ptr()
        if dynamic_act_scaling:
            scales = torch.empty(list(act_in.shape[:-1]) + [1], dtype=torch.float32, device=act_in.device)
            p.dynamic_scale = scales.data_ptr()
        else:
            scale = scale.to(torch.float32).contiguous()
            p.scale = scale.data_ptr()
    else:
        out = torch.empty(list(act_in.shape), dtype=act_in.dtype, device=act_in.device)
        p.out = out.data_ptr()
    L.check(L.lib().mi355q_gated_silu(C.byref(p), current_stream_ptr()))
    if dynamic_act_scaling:
        return q, scales
    return q if scale is not None else out


def swiglu(input):
    """functional.py:736-752: x, gate = chunk(input, 2, dim=-1); silu(gate) * x."""
    n = input.shape[-1] // 2
    return gated_silu(input[..., n:], input[..., :n])


def fp8_gemm(input, weights, scale_a, scale_b, dtype="float16"):
    """FP8Linear semantic (quantization/layers.py:777-879): e4m3 [M,K] x e4m3 [N,K]^T, per-tensor scales."""
    plg = Plugin("Fp8Gemm", [("type_id", str_dtype_to_trt(dtype), L.FIELD_INT32)])
    return plg(input, weights, scale_a, scale_b)


def quantize_e4m3(x, mode):
    """torch.ops.tensorrt_llm.quantize_e4m3_{activation,weight,per_tensor} (thop/fp8Op.cpp).
    mode: 'activation' (per row), 'weight' (per column), 'per_tensor'.  Returns (fp8 tensor, scales)."""
    m = {"activation": 0, "weight": 1, "per_tensor": 2}[mode]
    x2 = x.reshape(-1, x.shape[-1]).contiguous()
    rows, cols = x2.shape
    q = torch.empty(x2.shape, dtype=torch.uint8, device=x.device)
    nscale = rows if m == 0 else (cols if m == 1 else 2)
    scales = torch.empty(nscale, dtype=torch.float32, device=x.device)
    L.check(L.lib().mi355q_quantize_e4m3(q.data_ptr(), scales.data_ptr(), x2.data_ptr(), rows, cols,
                                         torch_dtype_to_trt(x.dtype), m, current_stream_ptr()))
    q = q.view(torch.float8_e4m3fn).reshape(x.shape)
    if m == 0:
        return q, scales.reshape(list(x.shape[:-1]) + [1])
    if m == 1:
        return q, scales.reshape(1, cols)
    return q, scales[:1].reshape(1, 1)


def dequantize_weights(weights, scales, zeros, k, n, group_size, weight_type, dtype):
    """Dense [k, n] dequantised weights from the GFX950 layout (the bit-exact gate; the reference
    tests recover them with an identity matmul, test_weight_only_quant_matmul.py:44-53)."""
    out = torch.empty(k, n, dtype=dtype, device=weights.device)
    p = L.WeightOnlyParams()
    p.qweight = weights.data_ptr()
    p.scales = scales.data_ptr()
    p.zeros = zeros.data_ptr() if zeros is not None and zeros.numel() else None
    p.out = out.data_ptr()
    p.m, p.n, p.k = 0, n, k
    p.group_size = group_size
    p.weight_type = weight_type
    p.act_type = torch_dtype_to_trt(dtype)
    L.check(L.lib().mi355q_weight_only_dequantize(C.byref(p), current_stream_ptr()))
    return out


# ----------------------------------------------------------------------------- attention
class PositionEmbeddingType:
    learned_absolute, rope_gptj, rope_gpt_neox, alibi, alibi_with_scale, relative = range(6)


class RotaryScalingType:
    none, linear, dynamic = range(3)


def make_gpt_attention_plugin(*, num_heads, num_kv_heads, head_size, dtype, layer_idx=0, q_scaling=1.0,
                              position_embedding_type=PositionEmbeddingType.learned_absolute,
                              rotary_embedding_dim=0, rotary_embedding_base=10000.0,
                              rotary_embedding_scale_type=RotaryScalingType.none, rotary_embedding_scale=1.0,
                              rotary_embedding_max_positions=1024, kv_cache_quant_mode=0,
                              remove_input_padding=False, paged_kv_cache=True, tokens_per_block=64,
                              max_context_length=0, qkv_bias_enabled=False, multi_block_mode=False,
                              enable_xqa=True, tp_size=1, tp_rank=0, use_paged_context_fmha=False, pos_shift_enabled=False, max_distance=0):
    """Field collection of tensorrt_llm/functional.py:3871-3925 / gptAttentionCommon.cpp:1561-1593."""
    I8, I32, F32 = L.FIELD_INT8, L.FIELD_INT32, L.FIELD_FLOAT32
    fields = [
        ("layer_idx", layer_idx, I32), ("num_heads", num_heads, I32), ("num_kv_heads", num_kv_heads, I32),
        ("head_size", head_size, I32), ("unidirectional", 1, I32), ("q_scaling", q_scaling, F32),
        ("position_embedding_type", position_embedding_type, I8),
        ("rotary_embedding_dim", rotary_embedding_dim, I32), ("rotary_embedding_base", rotary_embedding_base, F32),
        ("rotary_embedding_scale_type", rotary_embedding_scale_type, I8),
        ("rotary_embedding_scale", rotary_embedding_scale, F32),
        ("rotary_embedding_max_positions", rotary_embedding_max_positions, I32),
        ("tp_size", tp_size, I32), ("tp_rank", tp_rank, I32), ("unfuse_qkv_gemm", 0, I8),
        ("context_fmha_type", 0, I8), ("multi_block_mode", int(multi_block_mode), I8),
        ("enable_xqa", int(enable_xqa), I8), ("kv_cache_quant_mode", kv_cache_quant_mode, I32),
        ("remove_input_padding", int(remove_input_padding), I8), ("mask_type", 1, I32),
        ("paged_kv_cache", int(paged_kv_cache), I32), ("tokens_per_block", tokens_per_block, I32),
        ("type_id", str_dtype_to_trt(dtype), I32), ("max_context_length", max_context_length, I32),
        ("qkv_bias_enabled", int(qkv_bias_enabled), I8), ("do_cross_attention", 0, I8), ("max_distance", max_distance, I32),
        ("pos_shift_enabled", 1 if pos_shift_enabled else 0, I8), ("dense_context_fmha", 0, I8),
        ("use_paged_context_fmha", int(use_paged_context_fmha), I8),
        ("use_cache", 1, I32), ("is_medusa_enabled", 0, I8),
    ]
    return Plugin("GPTAttention", fields)


def gpt_attention(plugin, *, qkv, sequence_length, host_past_key_value_lengths, host_max_attention_window_sizes,
                  host_sink_token_length, context_lengths, cache_indirection, host_request_types,
                  kv_cache_block_pointers, host_kv_cache_block_pointers, kv_orig_quant_scale=None,
                  kv_quant_orig_scale=None, host_context_lengths=None, qkv_bias=None, alibi_slopes=None,
                  relative_attention_bias=None):
    """tensorrt_llm/functional.py:3487-3930 `gpt_attention` for a paged cache: assembles the plugin
    input list in the order of gptAttentionPlugin.h:43-68 and enqueues.  Returns the context tensor."""
    inputs = [qkv, sequence_length, host_past_key_value_lengths, host_max_attention_window_sizes,
              host_sink_token_length, context_lengths, cache_indirection, host_request_types,
              kv_cache_block_pointers, host_kv_cache_block_pointers]
    if kv_orig_quant_scale is not None:
        inputs += [kv_orig_quant_scale, kv_quant_orig_scale]
    if alibi_slopes is not None:
        inputs += [alibi_slopes]
    if relative_attention_bias is not None:
        inputs += [relative_attention_bias]
    if host_context_lengths is not None:
        inputs += [host_context_lengths]
    if qkv_bias is not None:
        inputs += [qkv_bias]
    return plugin(*inputs, device=qkv.device)
