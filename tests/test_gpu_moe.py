"""GPU parity: the expert GEMMs of a mixture-of-experts layer with weight-only quantised experts, one grouped launch
through the C ABI (mi355q_moe_gemm_bias_act) against oracle.moe_gemm_bias_act (MoeGemmRunner<T, uint8 | uint4>,
kernels/cutlass_kernels/moe_gemm/moe_gemm_kernels.h:51-58).  Gate: the reference's weight-only tolerance
max|ref| * 2^-(bits-1) * 1.5 (tests/quantization/_utils.py:99-109) and the much tighter 2 eps max|ref32| + 1e-3 of
test_gpu_gemm.py (same dequantised weights, fp32 accumulation).  No stored vectors of the reference cover the grouped GEMM
(cpp/tests/kernels/mixtureOfExpertsTest.cu computes its reference on the fly): parity unpinned beyond the cited semantics."""
import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

pytestmark = pytest.mark.gpu

DT = {"float16": torch.float16, "bfloat16": torch.bfloat16}
ACT = {"identity": F.ACT_IDENTITY, "relu": F.ACT_RELU, "silu": F.ACT_SILU, "gelu": F.ACT_GELU}


def _case(dev, rows_per_expert, n, k, dtype_str, wtype, activation, with_bias, seed=0):
    torch.manual_seed(seed)
    dt = DT[dtype_str]
    E = len(rows_per_expert)
    total = sum(rows_per_expert)
    tq = torch.int8 if wtype == L.WEIGHT_INT8 else torch.quint4x2
    q_all, proc_all, sc_all = [], [], []
    for e in range(E):
        weight = torch.rand(k, n, dtype=dt) * 2 - 1.0
        unproc, proc, scales = F._symmetric_quantize_last_axis_of_batched_matrix(weight, tq)
        q_all.append(unproc.numpy() if wtype == L.WEIGHT_INT8 else oracle.unpack_int4_packed_tensor_to_int8(unproc.numpy()))
        proc_all.append(proc)
        sc_all.append(scales.reshape(-1))
    q = np.stack(q_all)
    proc = torch.stack(proc_all).contiguous()
    scales = torch.stack(sc_all).contiguous()
    bias = (torch.randn(E, n) * 0.5).to(dt) if with_bias else None
    a = (torch.rand(max(total, 1), k, dtype=dt) * 2 - 1.0)[:total]
    cum = torch.tensor(np.cumsum(rows_per_expert), dtype=torch.int64)
    out = F.moe_gemm_bias_act(a.to(dev), proc.to(dev), scales.to(dev), None if bias is None else bias.to(dev), cum.to(dev),
                              wtype, ACT[activation])
    torch.cuda.synchronize()
    ref16, ref32 = oracle.moe_gemm_bias_act(a, q, scales, bias, cum.tolist(), activation)
    if total == 0:
        assert out.shape == (0, n)
        return
    atol = oracle.woq_tolerance(ref16, wtype, dt)
    torch.testing.assert_close(out.cpu().float(), ref16.float(), atol=atol, rtol=1e-7)
    tight = 2.0 * torch.finfo(dt).eps * ref32.abs().max().item() + 2e-3
    assert (out.cpu().float() - ref32).abs().max().item() <= tight


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("wtype", [L.WEIGHT_INT8, L.WEIGHT_INT4])
def test_moe_gemm_ragged_experts(cuda, dtype_str, wtype):
    # ragged rows per expert incl. empty experts, a single row, rows that straddle tile heights
    _case(cuda, [5, 0, 33, 1, 128, 0, 70, 19], 256, 512, dtype_str, wtype, "identity", True)
    _case(cuda, [0, 0, 300, 0], 384, 256, dtype_str, wtype, "identity", False)


@pytest.mark.parametrize("activation", ["relu", "silu", "gelu"])
def test_moe_gemm_activations(cuda, activation):
    _case(cuda, [17, 40, 3, 64], 256, 384, "float16", L.WEIGHT_INT8, activation, True)
    _case(cuda, [100, 28], 128, 256, "bfloat16", L.WEIGHT_INT4, activation, True)


def test_moe_gemm_decode_and_prefill_row_counts(cuda):
    # Mixtral-like: 8 experts, top-2; decode (16 rows per expert on average) and a prefill-sized group (tile height 128)
    _case(cuda, [20, 12, 16, 9, 31, 14, 10, 16], 1024, 1024, "bfloat16", L.WEIGHT_INT4, "identity", False)
    _case(cuda, [400, 130, 0, 257, 64, 512, 1, 300], 512, 512, "float16", L.WEIGHT_INT8, "silu", True)


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
def test_moe_gemm_prefill_sized_experts_big_tile(cuda, dtype_str):
    # 768 .. 1023 rows per expert on average, int4, n large enough for prefer_tile256: the 256 x 256 / 8-wave tile of the
    # grouped launch in one part; ragged experts (one empty, one of a single row, tiles straddled), bias + activation
    # epilogue.  Then the sizes this test used to run (256 - 356 rows per expert), which take 64-row tiles now.
    _case(cuda, [900, 0, 1, 2300], 8192, 128, dtype_str, L.WEIGHT_INT4, "identity", False)
    _case(cuda, [780, 1020, 700], 8192, 192, dtype_str, L.WEIGHT_INT4, "silu", True)
    _case(cuda, [600, 0, 1, 423], 8192, 128, dtype_str, L.WEIGHT_INT4, "identity", False)
    _case(cuda, [257, 511, 300], 8192, 192, dtype_str, L.WEIGHT_INT4, "silu", True)


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
def test_moe_gemm_two_part_launch(cuda, dtype_str):
    # >= 1024 rows per expert on average: the 256-row launch takes the experts' whole tiles only and a second launch of
    # 64-row tiles their ragged ends (an exact multiple of 256: no tail; a single row: tail only; 6 and 20 rows past a
    # multiple; a tail of 255)
    _case(cuda, [1030, 1536, 1, 1556], 8192, 128, dtype_str, L.WEIGHT_INT4, "silu", True)
    _case(cuda, [1024, 511, 2048], 8192, 192, dtype_str, L.WEIGHT_INT4, "identity", False)


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("rows", [16, 48])
def test_moe_gemm_identity_rows_recover_dequantised_weights(cuda, dtype_str, rows):
    # the reference's identity-activation gate (test_weight_only_quant_matmul.py:44-53) through the grouped launch's
    # integer-domain tiles (int4 experts, <= 64 rows per expert): one-hot rows must return round16(q * scale) bit for bit
    torch.manual_seed(3)
    dt = DT[dtype_str]
    E, k, n = 3, 256, 384
    q_all, proc_all, sc_all = [], [], []
    for e in range(E):
        weight = torch.rand(k, n, dtype=dt) * 2 - 1.0
        unproc, proc, scales = F._symmetric_quantize_last_axis_of_batched_matrix(weight, torch.quint4x2)
        q_all.append(oracle.unpack_int4_packed_tensor_to_int8(unproc.numpy()))
        proc_all.append(proc)
        sc_all.append(scales.reshape(-1))
    proc = torch.stack(proc_all).contiguous()
    scales = torch.stack(sc_all).contiguous()
    picks = [torch.randperm(k)[:rows] for _ in range(E)]
    a = torch.cat([torch.eye(k, dtype=dt)[pk] for pk in picks], 0)
    cum = torch.tensor([rows * (e + 1) for e in range(E)], dtype=torch.int64)
    out = F.moe_gemm_bias_act(a.to(cuda), proc.to(cuda), scales.to(cuda), None, cum.to(cuda), L.WEIGHT_INT4, ACT["identity"])
    torch.cuda.synchronize()
    for e in range(E):
        w = (torch.from_numpy(q_all[e]).float() * scales[e].float()[None, :]).to(dt)       # round16(q * s)
        got = out[e * rows:(e + 1) * rows].cpu()
        assert torch.equal(got.view(torch.int16), w[picks[e]].view(torch.int16)), f"expert {e}"


def test_moe_gemm_no_rows(cuda):
    _case(cuda, [0, 0, 0], 128, 128, "float16", L.WEIGHT_INT8, "identity", False)
