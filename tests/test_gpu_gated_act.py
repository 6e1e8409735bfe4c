"""GPU parity: the gated activation of the SmoothQuant MLP ("SwiGLU between gate/up and down") and
the one-pass per-token quantisation, through the C ABI vs the CPU oracle (oracle.gated_silu, restating
quantization/layers.py:922-941 + functional.py:736-752 with one rounding to T per layer of the reference's graph).
Bars: T outputs within two ulp of T (the silu's exp and the division by the smoother go through v_exp / v_rcp: a
result can land on the other side of a rounding boundary) and on < 1 % of the elements; int8 codes within 1 on < 1 %;
dynamic scales within one ulp of T.  No golden vector of the reference covers the fused op (it is built from TRT
layers there): parity is anchored on the layer semantics -- "parity unpinned" beyond the oracle."""
import numpy as np
import pytest
import torch

import oracle
from mi355q import functional as F

pytestmark = pytest.mark.gpu

ULP = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2e-6}


def check_close(got, ref, dtype):
    g, r = got.float(), ref.float()
    err = (g - r).abs()
    # a 1-ulp difference after the silu's rounding can move the product (and the quotient) by one more
    tol = ULP[dtype] * r.abs().clamp_min(1e-30) * 2.05 + 1e-30
    assert (err <= tol).all(), f"max rel err {(err / r.abs().clamp_min(1e-12)).max().item()}"
    if dtype != torch.float32:  # fp32: v_exp / v_rcp are 1-ulp approximations, most results differ in the last bit
        assert (err > 0).float().mean().item() < 0.01


def check_codes(got, ref):
    d = (got.to(torch.int32) - ref.to(torch.int32)).abs()
    assert d.max().item() <= 1
    assert (d > 0).float().mean().item() < 0.01


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(1, 8), (7, 1000), (33, 5120), (16, 13824), (3, 28672)])
def test_swiglu_fused_layout(cuda, dtype, rows, cols):
    g = torch.Generator().manual_seed(rows * 131 + cols)
    x = (torch.randn(rows, 2 * cols, generator=g) * 2).to(dtype)
    out = F.swiglu(x.to(cuda))
    torch.cuda.synchronize()
    ref, _ = oracle.gated_silu(x[:, cols:], x[:, :cols])
    assert out.shape == (rows, cols) and out.dtype == dtype
    check_close(out.cpu(), ref, dtype)


@pytest.mark.parametrize("mode", ["plain", "static", "dynamic"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("cols", [11008, 13824])
def test_smooth_quant_gated_mlp_chain(cuda, dtype, mode, cols):
    g = torch.Generator().manual_seed(cols + len(mode))
    rows = 24
    a = (torch.randn(2, rows // 2, cols, generator=g) * 1.5).to(dtype)
    gate = torch.randn(2, rows // 2, cols, generator=g).to(dtype)
    smoother = (torch.rand(cols, generator=g) + 0.5).to(dtype)
    scale = torch.tensor([13.0])
    kw = {}
    if mode == "static":
        kw["scale"] = scale.to(cuda)
    if mode == "dynamic":
        kw["dynamic_act_scaling"] = True
    out = F.gated_silu(a.to(cuda), gate.to(cuda), smoother.to(cuda), **kw)
    torch.cuda.synchronize()
    ref, ref_s = oracle.gated_silu(a, gate, smoother, scale=13.0 if mode == "static" else None, dynamic=mode == "dynamic")
    if mode == "plain":
        check_close(out.cpu(), ref, dtype)
    elif mode == "static":
        assert out.dtype == torch.int8
        check_codes(out.cpu(), ref)
    else:
        q, s = out
        assert s.shape == (2, rows // 2, 1)
        rel = ((s.cpu() - ref_s) / ref_s).abs()
        assert rel.max().item() <= ULP[dtype] * 1.01
        check_codes(q.cpu(), ref)


def test_gated_silu_edges(cuda):
    dt = torch.float16
    # empty batch
    out = F.gated_silu(torch.zeros(0, 64, dtype=dt, device=cuda), torch.zeros(0, 64, dtype=dt, device=cuda))
    assert out.shape == (0, 64)
    # zeros stay zeros; a huge negative input gives -0 * gate = 0, a huge positive one passes through
    a = torch.tensor([[0.0, -60000.0, 30000.0, 1.0] * 2], dtype=dt)
    gate = torch.ones(1, 8, dtype=dt)
    out = F.gated_silu(a.to(cuda), gate.to(cuda)).cpu()
    ref, _ = oracle.gated_silu(a, gate)
    assert torch.equal(out.float().abs(), ref.float().abs()) or (out.float() - ref.float()).abs().max() <= 32
    # errors: columns not a multiple of 8
    with pytest.raises(Exception, match="multiple of 8"):
        F.gated_silu(torch.zeros(2, 12, dtype=dt, device=cuda), torch.zeros(2, 12, dtype=dt, device=cuda))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(5, 64), (4096, 5120), (64, 13824), (9, 28672), (37, 1001), (3, 40000)])
def test_per_token_quantization_one_pass(cuda, dtype, rows, cols):
    # the row kernel serves quantize_per_token whenever cols % 8 == 0 and cols <= 32768 (the last two shapes take the
    # two-pass kernel): bit-exact with the oracle either way (a10)
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 3).to(dtype)
    q, s = F.quantize_per_token(x.to(cuda))
    torch.cuda.synchronize()
    ref_q, ref_s = oracle.quantize_per_token(x)
    assert np.array_equal(s.cpu().numpy().reshape(-1, 1), ref_s)
    assert (q.cpu().numpy() == ref_q).all()
