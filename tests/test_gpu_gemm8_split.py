"""GPU parity: the persistent / split-K-remainder form of the 8-bit GEMMs (csrc/gemm8.hip gemm8_sk_kernel), taken when a
workspace is given and the shape's 256 x 256 tiles leave the last round of CUs part-filled AND the cost model says the cut
pays (cfg3's down projection 4096 x 5120 x 13824 is one; tests/test_gpu_parity_round2.py covers that shape through the
plugin).  int8: accumulators and fp16 outputs exact against the oracle (tests/quantization/test_smooth_quant_gemm.py:123
asserts exact equality), and bit-identical to the call without a workspace; repeated calls on one workspace (the counting
words must be back at zero); a workspace full of garbage.  fp8: against the oracle at the tolerance of the existing fp8
test, and the fixed summation order makes repeated calls bit-identical."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L

pytestmark = pytest.mark.gpu

OUT = {torch.float16: L.HALF, torch.int32: L.INT32, torch.bfloat16: L.BF16, torch.float32: L.FLOAT}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _int8(a, b, sa, sb, out, ws, per_token=1, per_channel=1):
    m, k = a.shape
    L.check(L.lib().mi355q_int8_gemm_ws(a.data_ptr(), b.data_ptr(), per_token, per_channel, sb.data_ptr(), sa.data_ptr(),
                                        out.data_ptr(), OUT[out.dtype], m, b.shape[0], k,
                                        ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, _stream()))


@pytest.mark.parametrize("m,n,k,rows", [(256, 4352, 8192, None),      # less than one round: 17 tiles x 4 pieces of 16 K-tiles
                                        (1024, 5120, 13824, 48),      # 80 tiles x 3 pieces of 36 K-tiles
                                        (4096, 5120, 13824, 32)])     # cfg3 down: one whole round + 64 tiles x 4 pieces
def test_int8_split_exact(cuda, m, n, k, rows):
    need = L.lib().mi355q_gemm8_workspace_size(m, n, k)
    assert need > 0, "the shape is expected to take the split kernel"
    torch.manual_seed(1)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    sa = torch.randint(1, 10, (m, 1)).float() * 1e-2
    sb = torch.randint(1, 10, (1, n)).float() * 1e-2
    ad, bd, sad, sbd = a.to(cuda), b.to(cuda), sa.to(cuda), sb.to(cuda)
    one = torch.ones(1, device=cuda)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda).fill_(0xC7)       # garbage: nothing in it may be trusted
    sel = torch.arange(m) if rows is None else torch.cat(
        [torch.tensor([0, 255, m - 256, m - 1]), torch.randint(0, m, (rows,), generator=torch.Generator().manual_seed(2))]).unique()
    ref, acc_ref = oracle.smooth_quant_gemm(a[sel].numpy(), b.numpy(), sa[sel].numpy(), sb.numpy(), True, True, torch.float16)
    plain16 = torch.zeros(m, n, dtype=torch.float16, device=cuda)
    _int8(ad, bd, sad, sbd, plain16, None)
    for rep in range(3):                                                   # the same workspace, call after call
        acc = torch.zeros(m, n, dtype=torch.int32, device=cuda)
        out = torch.zeros(m, n, dtype=torch.float16, device=cuda)
        _int8(ad, bd, one, one, acc, ws, 0, 0)
        _int8(ad, bd, sad, sbd, out, ws)
        torch.cuda.synchronize()
        assert torch.equal(acc.cpu()[sel], torch.from_numpy(acc_ref)), f"int32 accumulators differ (call {rep})"
        assert torch.equal(out.cpu()[sel], ref), f"fp16 outputs differ from the oracle (call {rep})"
        assert torch.equal(out, plain16), f"split and one-tile-per-workgroup results differ (call {rep})"


def test_int8_small_workspace_falls_back(cuda):
    m, n, k = 256, 4352, 8192
    need = L.lib().mi355q_gemm8_workspace_size(m, n, k)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=cuda)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=cuda)
    one = torch.ones(1, device=cuda)
    full, small = torch.zeros(m, n, dtype=torch.int32, device=cuda), torch.zeros(m, n, dtype=torch.int32, device=cuda)
    _int8(a, b, one, one, full, torch.empty(need, dtype=torch.uint8, device=cuda), 0, 0)
    _int8(a, b, one, one, small, torch.empty(need // 2, dtype=torch.uint8, device=cuda), 0, 0)   # too small: not touched
    torch.cuda.synchronize()
    assert torch.equal(full, small)
    assert L.lib().mi355q_gemm8_workspace_size(4096, 8192, 8192) == 0     # whole rounds: no workspace wanted
    assert L.lib().mi355q_gemm8_workspace_size(100, 4352, 8192) == 0      # ragged m: the small-tile kernel


def test_fp8_split_against_oracle(cuda):
    m, n, k = 1024, 5120, 13824
    need = L.lib().mi355q_gemm8_workspace_size(m, n, k)
    assert need > 0
    torch.manual_seed(3)
    a = torch.randn(m, k).to(torch.float8_e4m3fn)
    b = torch.randn(n, k).to(torch.float8_e4m3fn)
    sa, sb = torch.tensor([0.5]), torch.tensor([0.25])
    ad, bd, sad, sbd = a.to(cuda), b.to(cuda), sa.to(cuda), sb.to(cuda)
    ws = torch.empty(need, dtype=torch.uint8, device=cuda).fill_(0x5A)
    outs = []
    for rep in range(2):
        out = torch.zeros(m, n, dtype=torch.float32, device=cuda)
        L.check(L.lib().mi355q_fp8_gemm_ws(ad.data_ptr(), bd.data_ptr(), sad.data_ptr(), sbd.data_ptr(), out.data_ptr(), L.FLOAT,
                                           m, n, k, ws.data_ptr(), need, _stream()))
        torch.cuda.synchronize()
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]), "the pieces are added in a fixed order: repeated calls must agree bit for bit"
    sel = torch.randint(0, m, (24,), generator=torch.Generator().manual_seed(4)).unique()
    ref, _ = oracle.fp8_gemm(a[sel], b, sa.item(), sb.item(), torch.float32)
    err = (outs[0][sel] - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() * np.sqrt(k), f"fp8 split GEMM: max err {err}"   # fp32 re-association
