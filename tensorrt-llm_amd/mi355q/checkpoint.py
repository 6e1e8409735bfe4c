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
    """AWQ_quantize_pack_preprocess (:1355-1359): weight [K, N] float, scale [K/g, N]."""
    w = weight_kn / scale.repeat_interleave(group_size, dim=0)
    q = torch.clamp(torch.round(w).to(torch.int8), -8, 7)
    return F.preprocess_weights_for_mixed_gemm(F.pack_int8_tensor_to_packed_int4(q), torch.quint4x2, layout)


def awq_to_tllm(weight_nk, amax, pre_quant_scale, group_size, dtype=torch.float16, tp_size=1, tp_rank=0, tp_dim=0,
                layout=L.LAYOUT_GFX950):
    """get_tllm_weight_from_awq (:1361-1379).  weight_nk: [N, K] float; amax: N * K/g values; pre_quant_scale: K."""
    weight = weight_nk.float().T.contiguous()
    k, n = weight.shape
    weight = torch_split(weight, tp_dim, tp_size, tp_rank)
    amax = torch_split(amax.float().reshape(n, k // group_size).T.contiguous(), tp_dim, tp_size, tp_rank)
    pqs = pre_quant_scale.float().reshape(1, k)
    if tp_dim == 0:
        pqs = torch_split(pqs, 1, tp_size, tp_rank)
    scale = amax / 8.0
    return {"weight": awq_quantize_pack_preprocess(weight, scale, group_size, layout),
            "weights_scaling_factor": scale.to(dtype), "prequant_scaling_factor": pqs.to(dtype)}


def resmooth_and_get_scale(weight_kn, pre_quant_scale, avg_pre_quant_scale, group_size, resmooth=True):
    """reSmooth_and_get_scale (:1381-1399): fuse q/k/v (or gate/up) that share one input: move every weight to the
    averaged pre-quant scale, then the per-group scale amax / 8.  weight [K, N]; scales [1, K]."""
    k, n = weight_kn.shape
    w = weight_kn.float()
    if resmooth:
        w = w * pre_quant_scale.reshape(k, 1).float() / avg_pre_quant_scale.reshape(k, 1).float()
    amax = w.T.reshape(n, k // group_size, group_size).abs().amax(-1).T.contiguous()
    return w, amax / 8.0
