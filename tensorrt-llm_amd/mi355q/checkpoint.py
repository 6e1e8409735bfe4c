"""Quantised checkpoint tensors -> the tensors the weight-only plugins take.
Mirror of the per-tensor transformations in tensorrt_llm/models/llama/weight.py:
  * GPTQ (`load_from_gptq_llama`, :832-925): qweight int32 [K/8, N] (8 nibbles along K per word), qzeros int32
    [K/g, N/8], scales [K/g, N]  ->  packed int4 in the GEMM layout, scales, zero = (-qzeros + 8 - 1) * scales;
  * AWQ / AMMO (`load_from_awq_llama`, :1253-1420): fp weight [N, K] + amax [N * K/g] + pre_quant_scale [K]  ->
    scale = amax / 8, q = clamp(round(W^T / scale), -8, 7), packed + laid out, pre-quant scale [1, K].
Host tensors in, host tensors out (the layout transform is the CPU routine of the C ABI); TP slicing follows
`torch_split` there: pass `tp_size`, `tp_rank`, `tp_dim`."""
import torch

from . import _lib as L
from . import functional as F


def unpack_int32_into_int8(w_packed):
    """weight.py:882-891: every int32 word holds 8 unsigned nibbles, lowest first, along the LAST dim."""
    b = w_packed.contiguous().view(torch.uint8)
    out = torch.empty(b.shape[0], b.shape[1] * 2, dtype=torch.int8)
    out[:, 0::2] = (b & 0x0F).to(torch.int8)
    out[:, 1::2] = (b >> 4).to(torch.int8)
    return out


def torch_split(v, dim, tp_size, tp_rank):
    if tp_size == 1:
        return v
    assert v.shape[dim] % tp_size == 0, "Invalid TP size"
    return v.split(v.shape[dim] // tp_size, dim=dim)[tp_rank]


def gptq_to_tllm(qweight_int32, qzeros_int32, scales, tp_size=1, tp_rank=0, tp_dim=-1, layout=L.LAYOUT_GFX950,
                 gptq_for_llama=True):
    """process_and_assign_weight (:893-925).  Returns dict(weight int8 [K, N/2] in `layout`, weights_scaling_factor
    [K/g, N], zero [K/g, N]) with scales' dtype."""
    if tp_dim != -1:
        qweight_int32, qzeros_int32, scales = (torch_split(t, tp_dim, tp_size, tp_rank) for t in
                                               (qweight_int32, qzeros_int32, scales))
    # nibbles run along K inside a word: unpack the transposed tensor, transpose back, re-centre to signed int4
    q = unpack_int32_into_int8(qweight_int32.T.contiguous()).T.contiguous() - 8
    weight = F.preprocess_weights_for_mixed_gemm(F.pack_int8_tensor_to_packed_int4(q), torch.quint4x2, layout)
    z = unpack_int32_into_int8(qzeros_int32).to(torch.int32)
    zero = ((-z + 8 - (1 if gptq_for_llama else 0)).to(scales.dtype) * scales).to(scales.dtype)
    return {"weight": weight, "weights_scaling_factor": scales.contiguous(), "zero": zero}


def awq_quantize_pack_preprocess(weight_kn, scale, group_size, layout=L.LAYOUT_GFX950):
    """AWQ_quantize_pack_preprocess (:1355-1359): weight [K, N], scale [K/g, N].  The division runs in the tensors' own
    dtype (`weight /= scale...` on the checkpoint's fp16 / bf16 tensors in the reference): in fp32 a quotient can land
    on the other side of a .5 tie and round() to a different code."""
    w = weight_kn / scale.repeat_interleave(group_size, dim=0).to(weight_kn.dtype)
    q = torch.clamp(torch.round(w).to(torch.int8), -8, 7)
    return F.preprocess_weights_for_mixed_gemm(F.pack_int8_tensor_to_packed_int4(q), torch.quint4x2, layout)


def awq_to_tllm(weight_nk, amax, pre_quant_scale, group_size, dtype=torch.float16, tp_size=1, tp_rank=0, tp_dim=0,
                layout=L.LAYOUT_GFX950):
    """get_tllm_weight_from_awq (:1361-1379).  weight_nk: [N, K] float; amax: N * K/g values; pre_quant_scale: K."""
    weight = weight_nk.T.contiguous()  # checkpoint dtype throughout, like the reference
    k, n = weight.shape
    weight = torch_split(weight, tp_dim, tp_size, tp_rank)
    amax = torch_split(amax.reshape(n, k // group_size).T.contiguous(), tp_dim, tp_size, tp_rank)
    pqs = pre_quant_scale.reshape(1, k)
    if tp_dim == 0:
        pqs = torch_split(pqs, 1, tp_size, tp_rank)
    scale = amax / 8.0
    return {"weight": awq_quantize_pack_preprocess(weight, scale, group_size, layout),
            "weights_scaling_factor": scale.to(dtype), "prequant_scaling_factor": pqs.to(dtype)}


def resmooth_and_get_scale(weight_kn, pre_quant_scale, avg_pre_quant_scale, group_size, resmooth=True):
    """reSmooth_and_get_scale (:1381-1399): fuse q/k/v (or gate/up) that share one input: move every weight to the
    averaged pre-quant scale, then the per-group scale amax / 8.  weight [K, N]; scales [1, K]."""
    k, n = weight_kn.shape
    w = weight_kn.clone()  # checkpoint dtype: `weight *= ...; weight /= ...` in place in the reference (:1385-1390)
    if resmooth:
        w = w * pre_quant_scale.reshape(k, 1).to(w.dtype)
        w = w / avg_pre_quant_scale.reshape(k, 1).to(w.dtype)
    amax = w.T.reshape(n, k // group_size, group_size).abs().amax(-1).T.contiguous()
    return w, amax / 8.0


# ----------------------------------------------------------------------------- SmoothQuant checkpoint tensors
def smooth_gemm(gemm_weights, act_scales, layernorm_weights=None, layernorm_bias=None, alpha=0.5, weight_scales=None):
    """SmoothQuant migration of a GEMM (or of several that share an input), models/llama/convert.py:148-176:
    s[k] = clamp(act_scales[k]^alpha / max_n|W[n, k]|^(1 - alpha), 1e-5); W[:, k] *= s[k] in place, and the layer norm
    in front of the GEMM is divided by s (weights and bias).  `gemm_weights`: [N, K] tensors ("transposed" in the
    reference's words).  Returns s."""
    ws = gemm_weights if isinstance(gemm_weights, (list, tuple)) else [gemm_weights]
    k = act_scales.numel()
    assert all(w.shape[1] == k for w in ws)
    dtype = ws[0].dtype
    if weight_scales is None:
        weight_scales = torch.stack([w.abs().amax(dim=0) for w in ws]).amax(dim=0)
    s = (act_scales.to(torch.float64).pow(alpha) / weight_scales.to(torch.float64).pow(1 - alpha)).clamp(min=1e-5)
    if layernorm_weights is not None:
        layernorm_weights.div_(s)
    if layernorm_bias is not None:
        layernorm_bias.div_(s)
    for w in ws:
        w.mul_(s.view(1, -1).to(w.dtype) if w.dtype != torch.float64 else s.view(1, -1))
        assert w.dtype == dtype
    return s


def generate_int8(weights_kn, act_range, is_qkv=False):
    """The SmoothQuant tensors of one linear layer from calibration ranges (models/llama/convert.py:21-126; the
    multi-query branch is left out).  weights_kn: fp [K, N]; act_range: {"x": per-input-channel |x| max, "y":
    per-output-channel |y| max, "w": per-output-channel |w| max [N]}.  Per-tensor (.t) and per-column (.col) variants:
      scale_w_orig_quant = 127 / max|w|; weight.int8 = clip(round(w * scale), -127, 127);
      scale_x_orig_quant = 127 / max|x|; scale_y_quant_orig = max|y| / 127;
      scale_y_accum_quant = (127 / max|y|) / (scale_x_orig_quant * scale_w_orig_quant)
    A fused QKV matrix counts as three matrices: its "per-tensor" scales are one per Q, K, V block, replicated along the
    block's columns.  The SmoothQuantGemm plugin consumes weight.int8[.col] transposed to [N, K], scale_w_quant_orig[.col]
    as scale_channels and (static activation scaling) scale_x_orig_quant for QuantizeTensor."""
    import numpy as np
    w = weights_kn.detach().cpu().double().numpy()
    rw = act_range["w"].detach().cpu().double().numpy().reshape(-1)
    n = w.shape[1]
    assert rw.size == n
    if is_qkv:
        per_block = rw.reshape(3, -1).max(axis=1, keepdims=True)
        w_oq_t = np.broadcast_to(127.0 / per_block, (3, n // 3)).reshape(n)
    else:
        w_oq_t = np.float64(127.0 / rw.max())
    w_oq_c = 127.0 / rw
    x_oq = 127.0 / float(act_range["x"].max())
    y_oq = 127.0 / float(act_range["y"].max())
    to_i8 = lambda a: np.clip(np.round(a), -127, 127).astype(np.int8)
    f32 = lambda a: np.asarray(a, dtype=np.float32)
    return {
        "weight.int8": to_i8(w * f32(w_oq_t)),
        "weight.int8.col": to_i8(w * f32(w_oq_c)),
        "scale_x_orig_quant": f32(x_oq),
        "scale_w_quant_orig": f32(1.0 / w_oq_t),
        "scale_w_quant_orig.col": f32(1.0 / w_oq_c),
        "scale_y_accum_quant": f32(y_oq / (x_oq * w_oq_t)),
        "scale_y_accum_quant.col": f32(y_oq / (x_oq * w_oq_c)),
        "scale_y_quant_orig": f32(1.0 / y_oq),
    }
