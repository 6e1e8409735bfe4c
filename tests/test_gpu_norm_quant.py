"""GPU parity: the SmoothQuant feeders through the plugin C ABI vs the CPU oracle --
RmsnormQuantization / LayernormQuantization (static and dynamic per-token scaling) and QuantizeTensor.
Bars: the reference's own is `atol = 1` on the int8 codes and `atol = 1e-2` on the dynamic scales
(tests/quantization/test_smooth_quant_rms_norm.py:103-113, test_smooth_quant_layer_norm.py); here, tighter:
codes differ by at most 1 AND on fewer than 0.2 % of the elements (a difference needs the fp32 statistics' summation
order to flip a rounding), dynamic scales within one ulp of the activation type; QuantizeTensor and the
un-quantised outputs of a row-constant input are exact."""
import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F
from mi355q.plugin import Plugin

pytestmark = pytest.mark.gpu

# one ulp of the 16-bit types; fp32: a few ulp (E[x^2] - E[x]^2 cancels)
ULP = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7, torch.float32: 2e-6}


def check_codes(got, ref):
    d = (got.to(torch.int32) - ref.to(torch.int32)).abs()
    assert d.max().item() <= 1, f"max code difference {d.max().item()}"
    frac = (d > 0).float().mean().item()
    assert frac < 2e-3, f"{frac:.4%} of the codes differ"


def run_case(kind, dtype, shape, dynamic, dev, use_diff=True, eps=1e-5, seed=0, gamma_bias=True):
    g = torch.Generator().manual_seed(seed)
    H = shape[-1]
    x = torch.randn(*shape, generator=g).to(dtype)
    gamma = (torch.rand(H, generator=g) + 0.5).to(dtype) if gamma_bias else None
    beta = (torch.randn(H, generator=g) * 0.1).to(dtype) if gamma_bias else None
    scale = torch.randint(2, 32, (1,), generator=g).float()  # test_smooth_quant_rms_norm.py:42
    fn = F.smooth_quant_rms_norm if kind == "rms" else F.smooth_quant_layer_norm
    kw = dict(weight=None if gamma is None else gamma.to(dev), bias=None if beta is None else beta.to(dev),
              scale=scale.to(dev), eps=eps, dynamic_act_scaling=dynamic)
    if kind == "layer":
        kw["use_diff_of_squares"] = use_diff
    out = fn(x.to(dev), H, **kw)
    torch.cuda.synchronize()
    gam = gamma if gamma is not None else torch.ones(H, dtype=dtype)
    bet = beta if beta is not None else torch.zeros(H, dtype=dtype)
    ref_q, ref_s = oracle.norm_quant(x, gam, bet, eps, kind=kind, scale=float(scale), dynamic=dynamic,
                                     use_diff_of_squares=use_diff)
    if dynamic:
        q, s = out
        assert s.shape == ref_s.shape and s.dtype == torch.float32
        rel = ((s.cpu() - ref_s) / ref_s).abs()
        assert rel.max().item() <= ULP[dtype] * 1.01, f"dynamic scale rel err {rel.max().item()}"
        assert (rel > 2e-6).float().mean().item() < 0.02
    else:
        q = out
    assert q.dtype == torch.int8 and q.shape == x.shape
    check_codes(q.cpu(), ref_q)


@pytest.mark.parametrize("dynamic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_rmsnorm_quantization_reference_grid(cuda, dtype, dynamic):
    # the reference's case: shape [2, 5, 10, 10] (hidden 10: the generic kernel), float16 / float32 x static / dynamic
    run_case("rms", dtype, [2, 5, 10, 10], dynamic, cuda, eps=1e-6, gamma_bias=False)
    run_case("rms", dtype, [2, 5, 10, 10], dynamic, cuda)


@pytest.mark.parametrize("dynamic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("H", [768, 4096, 5120, 8192, 13824])
def test_rmsnorm_quantization_llama_widths(cuda, dtype, dynamic, H):
    run_case("rms", dtype, [3, 7, H], dynamic, cuda)


@pytest.mark.parametrize("use_diff", [True, False])
@pytest.mark.parametrize("dynamic", [False, True])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_layernorm_quantization(cuda, dtype, dynamic, use_diff):
    run_case("layer", dtype, [2, 5, 10, 10], dynamic, cuda, use_diff=use_diff)
    run_case("layer", dtype, [4, 9, 768], dynamic, cuda, use_diff=use_diff)
    run_case("layer", dtype, [17, 5120], dynamic, cuda, use_diff=use_diff)


def test_norm_edge_shapes(cuda):
    # one row, one element short of / past a register-path boundary, > 16384 (re-reading kernel), empty batch
    for H in (8, 2040, 2048, 2056, 16384, 16392, 20000):
        run_case("rms", torch.float16, [2, H], True, cuda)
        run_case("layer", torch.bfloat16, [1, H], False, cuda)
    q, s = F.smooth_quant_rms_norm(torch.empty(0, 64, dtype=torch.float16, device=cuda), 64,
                                   scale=torch.ones(1, device=cuda), dynamic_act_scaling=True)
    assert q.shape == (0, 64) and s.shape == (0, 1)


def test_norm_without_quantisation_and_exact_rows(cuda):
    # un-quantised launcher (invokeGeneralRmsNorm with scale = dynamic_scale = NULL) vs the oracle, within one ulp of T;
    # a row of equal values x gives exactly T(x * rsqrt(x^2 + eps) * gamma) whatever the summation order
    g = torch.Generator().manual_seed(3)
    for dtype in (torch.float16, torch.bfloat16, torch.float32):
        x = torch.randn(5, 4096, generator=g).to(dtype)
        gamma = (torch.rand(4096, generator=g) + 0.5).to(dtype)
        beta = (torch.randn(4096, generator=g) * 0.1).to(dtype)
        for kind, fn in (("rms", F.rms_norm), ("layer", F.layer_norm)):
            out = fn(x.to(cuda), gamma.to(cuda), beta.to(cuda), eps=1e-5).cpu()
            ref, _ = oracle.norm_quant(x, gamma, beta, 1e-5, kind=kind)
            err = (out.float() - ref.float()).abs().max().item()
            assert err <= ULP[dtype] * 8, (kind, dtype, err)
            if dtype != torch.float32:  # the rounding to T hides the last-ulp difference of rstd almost everywhere
                assert (out != ref).float().mean().item() < 0.01
    xc = torch.full((3, 1024), 0.75, dtype=torch.float16)
    gam = torch.ones(1024, dtype=torch.float16)
    out = F.rms_norm(xc.to(cuda), gam.to(cuda), None, eps=0.0).cpu()
    assert torch.equal(out, torch.ones(3, 1024, dtype=torch.float16))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
def test_quantize_tensor_exact(cuda, dtype):
    g = torch.Generator().manual_seed(1)
    for shape in ([2, 16, 768], [3, 20], [4096, 5120]):
        x = (torch.randn(*shape, generator=g) * 3).to(dtype)
        scale = torch.tensor([[11.5]], dtype=torch.float32)
        q = F.quantize_tensor(x.to(cuda), scale.to(cuda)).cpu()
        assert torch.equal(q, oracle.quantize_tensor(x, 11.5))
    with pytest.raises(L.Mi355qError, match="multiple of 4"):
        F.quantize_tensor(torch.zeros(3, 3, dtype=dtype, device=cuda), torch.ones(1, 1, device=cuda))


def test_rmsnorm_quant_feeds_smooth_quant_gemm(cuda):
    # the chain the reference builds (models/llama: smooth_quant_rms_norm -> SmoothQuantLinear): dynamic per-token
    # codes + scales straight into the int8 GEMM; against the oracle's chain
    g = torch.Generator().manual_seed(5)
    M, K, N = 48, 1024, 768
    x = torch.randn(M, K, generator=g).to(torch.float16)
    gamma = (torch.rand(K, generator=g) + 0.5).to(torch.float16)
    w = torch.randint(-128, 128, (N, K), generator=g, dtype=torch.int8)
    s_ch = (torch.randint(1, 10, (1, N), generator=g).float() * 1e-2)
    q, s_tok = F.smooth_quant_rms_norm(x.to(cuda), K, weight=gamma.to(cuda), scale=torch.ones(1, device=cuda),
                                       dynamic_act_scaling=True)
    out = F.smooth_quant_gemm(q, w.to(cuda), s_tok, s_ch.to(cuda), True, True, "float16").cpu()
    rq, rs = oracle.norm_quant(x, gamma, torch.zeros(K, dtype=torch.float16), 1e-5, kind="rms", dynamic=True)
    ref = oracle.smooth_quant_gemm(rq, w, rs, s_ch, True, True, torch.float16)
    ref = ref[0] if isinstance(ref, tuple) else ref
    # a one-code difference in q moves an output by |w| * s_tok * s_ch <= 128 * 0.04 * 0.09
    assert (out.float() - ref.float()).abs().max().item() <= 0.5
    assert ((out.float() - ref.float()).abs() > 1e-2).float().mean().item() < 0.05


def test_norm_quant_llama13b_size_properties(cuda):
    # cfg3's feeder at full size (M = 4096 tokens, hidden 5120) through size-independent properties:
    # scaling the input by 2 leaves the RMSNorm codes unchanged (eps = 0); codes saturate at |127|; every row's
    # dynamic scale * 127 equals the largest |normalised value|, i.e. max |code| == 127 in every row
    g = torch.Generator().manual_seed(9)
    x = torch.randn(4096, 5120, generator=g).to(torch.float16).to(cuda)
    gamma = torch.ones(5120, dtype=torch.float16, device=cuda)
    one = torch.ones(1, device=cuda)
    q1, s1 = F.smooth_quant_rms_norm(x, 5120, weight=gamma, scale=one, eps=0.0, dynamic_act_scaling=True)
    q2, s2 = F.smooth_quant_rms_norm(x * 2, 5120, weight=gamma, scale=one, eps=0.0, dynamic_act_scaling=True)
    assert (q1.to(torch.int32) - q2.to(torch.int32)).abs().max().item() <= 1
    assert ((q1 != q2).float().mean().item()) < 2e-3
    assert torch.equal(q1.to(torch.int32).abs().amax(-1), torch.full((4096,), 127, dtype=torch.int32, device=cuda))
    assert torch.allclose(s1, s2, rtol=2e-3)
