"""Tensor-parallel split of the quantized GEMMs and of decode attention.

Follows the reference's Python layers:
  tensorrt_llm/quantization/layers.py:470-476, 516-518   column-parallel weight-only linear (no collective,
                                                          optional all-gather of N when gather_output)
  tensorrt_llm/quantization/layers.py:552-559, 597-598   row-parallel: local GEMM on the K shard, all-reduce
                                                          of the [M, N] output, bias added after the reduction
  tensorrt_llm/layers/attention.py                        heads split across ranks (kv-heads and their q-heads)
One process per GPU; the collective is torch.distributed (backend "nccl" = RCCL over xGMI on the GPU
box, "gloo" in the CPU tests).  The local GEMM is passed in (`matmul`): on a GPU it is one of the
functional.* ops, in the CPU tests it is the checker, so the split / reduce logic is what is tested.
"""
from dataclasses import dataclass
from typing import Callable, Optional

import torch
import torch.distributed as dist

from . import _lib as L
from . import functional as F


def shard_range(total: int, tp_size: int, rank: int, multiple: int = 1):
    """[lo, hi) of `total` owned by `rank`; every shard has the same size (the reference asserts
    divisibility, layers.py:427-431)."""
    if total % (tp_size * multiple) != 0:
        raise ValueError(f"cannot split {total} over tp_size={tp_size} in multiples of {multiple}")
    per = total // tp_size
    return rank * per, (rank + 1) * per


@dataclass
class WoqShard:
    """One rank's slice of a weight-only (AWQ/GPTQ group-wise or per-channel) linear layer."""
    qweight: torch.Tensor            # int8 CPU tensor, row-major packed [K_local, N_local * bits / 8]
    scales: torch.Tensor             # [K_local / g, N_local] or [N_local]
    zeros: Optional[torch.Tensor]
    bias: Optional[torch.Tensor]     # column: [N_local]; row: full [N], applied after the all-reduce
    pre_quant_scale: Optional[torch.Tensor]  # [1, K_local]
    group_size: int                  # 0 = per-channel
    bits: int
    k: int
    n: int
    parallel: str                    # "column" | "row"

    def processed(self, layout=L.LAYOUT_GFX950):
        qt = torch.int8 if self.bits == 8 else torch.quint4x2
        return F.preprocess_weights_for_mixed_gemm(self.qweight.contiguous(), qt, layout)


def column_split(qweight, scales, zeros, bias, pre_quant_scale, group_size, bits, tp_size, rank) -> WoqShard:
    """Shard N (layers.py:470-476): weight[:, n0:n1], scales/zeros[..., n0:n1], bias[n0:n1]; the
    activation and its pre-quant scale are replicated."""
    K = qweight.shape[0]
    N = qweight.shape[1] * 8 // bits
    n0, n1 = shard_range(N, tp_size, rank, multiple=16)
    b0, b1 = n0 * bits // 8, n1 * bits // 8
    return WoqShard(qweight[:, b0:b1].contiguous(), scales[..., n0:n1].contiguous(),
                    None if zeros is None else zeros[..., n0:n1].contiguous(),
                    None if bias is None else bias[..., n0:n1].contiguous(),
                    pre_quant_scale, group_size, bits, K, n1 - n0, "column")


def row_split(qweight, scales, zeros, bias, pre_quant_scale, group_size, bits, tp_size, rank) -> WoqShard:
    """Shard K (layers.py:552-559): weight[k0:k1], group-wise scales/zeros[k0/g:k1/g] (per-channel
    scales are replicated), pre_quant_scale[:, k0:k1]; the bias stays whole and is added once,
    after the all-reduce."""
    K = qweight.shape[0]
    N = qweight.shape[1] * 8 // bits
    k0, k1 = shard_range(K, tp_size, rank, multiple=max(group_size, 64))
    if group_size > 0:
        g0, g1 = k0 // group_size, k1 // group_size
        s = scales[g0:g1].contiguous()
        z = None if zeros is None else zeros[g0:g1].contiguous()
    else:
        s, z = scales, zeros
    pq = None if pre_quant_scale is None else pre_quant_scale[..., k0:k1].contiguous()
    return WoqShard(qweight[k0:k1].contiguous(), s, z, bias, pq, group_size, bits, k1 - k0, N, "row")


def all_reduce_sum(t: torch.Tensor, group=None) -> torch.Tensor:
    """AllreducePlugin::enqueue (allreducePlugin.cpp:141-203): in-place sum over the TP group, 16-bit
    on the wire when the tensor is 16-bit.  World size 1 (or no process group): identity."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def all_gather_last_dim(t: torch.Tensor, group=None, helper=None) -> torch.Tensor:
    """allgather plugin + the transpose the reference does for gather_output (layers.py:516-518).  `helper`: a
    mi355q.ipc.CustomAllReduceHelper, whose AllGather plugin (the push kernel over the peer mappings) then does the
    gather along dim 0; without one the gather is torch.distributed's (RCCL on the GPU box, gloo on the CPU)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    if helper is not None:
        g = helper.allgather(t.contiguous().unsqueeze(0))           # [world, ..., N_local]
        return torch.cat(list(g.unbind(0)), dim=-1)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t.contiguous(), group=group)
    return torch.cat(parts, dim=-1)


def column_parallel_linear(x, shard: WoqShard, matmul: Callable, gather_output=False, group=None, helper=None):
    """y_local[M, N_local] = matmul(x, shard) (+ local bias); no collective unless gather_output."""
    assert shard.parallel == "column"
    y = matmul(x, shard, shard.bias)
    return all_gather_last_dim(y, group, helper) if gather_output else y


def row_parallel_linear(x_local, shard: WoqShard, matmul: Callable, group=None):
    """y[M, N] = all_reduce(matmul(x_local[M, K_local], shard)) + bias (layers.py:597-598)."""
    assert shard.parallel == "row"
    y = all_reduce_sum(matmul(x_local, shard, None), group)
    if shard.bias is not None:
        y = y + shard.bias.to(y.device).reshape(1, -1)
    return y


def gpu_woq_matmul(dtype_str="bfloat16", layout=L.LAYOUT_GFX950):
    """The `matmul` for a GPU rank: weight_only_groupwise_quant_matmul / weight_only_quant_matmul
    on the shard's processed weights (cached on the shard)."""
    def run(x, shard: WoqShard, bias):
        dev = x.device
        cache = shard.__dict__.setdefault("_dev", {})
        if dev not in cache:
            dt = x.dtype
            w = shard.processed(layout)
            cache[dev] = dict(
                w=w.view(dt).to(dev) if shard.group_size > 0 else w.to(dev),
                s=shard.scales.to(dev), z=None if shard.zeros is None else shard.zeros.to(dev),
                pq=None if shard.pre_quant_scale is None else shard.pre_quant_scale.to(dev))
        c = cache[dev]
        if shard.group_size > 0:
            algo = (F.BIAS if bias is not None else 0) | (F.ZERO if c["z"] is not None else 0) \
                | (F.PRE_QUANT_SCALE if c["pq"] is not None else 0)
            return F.weight_only_groupwise_quant_matmul(x, c["pq"], c["w"], c["s"], c["z"],
                                                        None if bias is None else bias.to(dev), None, algo,
                                                        shard.group_size, dtype_str)
        y = F.weight_only_quant_matmul(x, c["w"], c["s"], L.WEIGHT_INT8 if shard.bits == 8 else L.WEIGHT_INT4,
                                       dtype_str)
        return y if bias is None else y + bias.to(dev).reshape(1, -1)
    return run


# ----------------------------------------------------------------------------- SmoothQuant
def sq_column_split(weight_i8, scale_channels, tp_size, rank):
    """SmoothQuantLinear (layers.py:86-150): B[N, K] -> B[n0:n1], per-channel scales[n0:n1]."""
    n0, n1 = shard_range(weight_i8.shape[0], tp_size, rank)
    sc = scale_channels if scale_channels.numel() == 1 else scale_channels.reshape(-1)[n0:n1].reshape(1, -1)
    return weight_i8[n0:n1].contiguous(), sc.contiguous()


def sq_row_split(weight_i8, tp_size, rank):
    """SmoothQuantRowLinear (layers.py:153-221): B[N, K] -> B[:, k0:k1]; per-channel scales are
    replicated, the dequantised 16-bit outputs are all-reduced."""
    k0, k1 = shard_range(weight_i8.shape[1], tp_size, rank)
    return weight_i8[:, k0:k1].contiguous()


# ----------------------------------------------------------------------------- attention
def attention_head_split(num_heads, num_kv_heads, tp_size, rank):
    """Heads owned by `rank`: kv-heads [r*Hkv/tp, (r+1)*Hkv/tp) and their Hq/Hkv query heads; when
    tp_size > Hkv the kv-heads are replicated (attention.py: num_kv_heads = ceil(Hkv / tp)).
    Returns (q_lo, q_hi, kv_lo, kv_hi).  No collective: every rank's output columns are disjoint
    and feed the row-parallel o_proj directly."""
    q0, q1 = shard_range(num_heads, tp_size, rank)
    if num_kv_heads >= tp_size:
        kv0, kv1 = shard_range(num_kv_heads, tp_size, rank)
    else:
        if tp_size % num_kv_heads != 0:
            raise ValueError("tp_size must be a multiple of num_kv_heads when it exceeds it")
        kv0 = rank // (tp_size // num_kv_heads)
        kv1 = kv0 + 1
    return q0, q1, kv0, kv1


def select_allreduce_strategy(message_bytes, world_size, max_workspace_bytes=8 * 1000 * 1000 * 8):
    """AllreducePlugin::selectImplementation (allreducePlugin.cpp:111-139) through the C ABI:
    0 = RING (RCCL), 1 = ONESHOT, 2 = TWOSHOT."""
    return int(L.lib().mi355q_allreduce_select_implementation(message_bytes, world_size, max_workspace_bytes))
