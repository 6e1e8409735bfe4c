"""GPU parity of the quantized GEMMs through the C ABI (plugin enqueue) vs the CPU oracle, on the
reference tests' own parameter grids and seeds:
  tests/quantization/test_weight_only_groupwise_quant_matmul.py:212-311 (torch.manual_seed(0))
  tests/quantization/test_weight_only_quant_matmul.py
  tests/quantization/test_smooth_quant_gemm.py:125-146
Gates: dequantised weights and int32 accumulators bit-exact; scaled outputs within the reference's
tolerance max|ref| * 2^-(bits-1) * 1.5 (_utils.py:99-109); SmoothQuant outputs exact."""
import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

pytestmark = pytest.mark.gpu

DT = {"float16": torch.float16, "bfloat16": torch.bfloat16}


def _groupwise_case(m, n, k, dtype_str, has_pre_quant, has_zero, has_bias, group_size, dev, use_w4a8_awq=False):
    """_woq_groupwise_matmul (test_weight_only_groupwise_quant_matmul.py:129-210) with our plugin."""
    torch.manual_seed(0)
    dt = DT[dtype_str]
    total_groups = (k + group_size - 1) // group_size
    activation = torch.randn(m, k, dtype=dt)
    bias = torch.randn(1, n, dtype=dt) if has_bias else None
    zero = torch.randn(total_groups, n, dtype=dt) if has_zero else None
    scale = torch.rand(total_groups, n, dtype=dt)
    pre_quant_scale = torch.rand(1, k, dtype=dt)
    unprocessed_int_weight = torch.randint(-2 ** 31, 2 ** 31, (k, n // 8), dtype=torch.int32)
    unprocessed_weight = unprocessed_int_weight.view(torch.int8)
    ref_q_weight = F.unpack_int4_packed_tensor_to_int8(unprocessed_weight)
    assert (ref_q_weight.numpy() == oracle.unpack_int4_packed_tensor_to_int8(unprocessed_weight.numpy())).all()
    cuda_q_weight = F.preprocess_weights_for_mixed_gemm(unprocessed_weight, torch.quint4x2).view(dt)
    fp8_alpha = torch.rand(1, dtype=torch.float32) if use_w4a8_awq else None  # :156-157
    quant_algo = use_w4a8_awq * F.FP8_ALPHA + has_pre_quant * F.PRE_QUANT_SCALE + has_zero * F.ZERO + has_bias * F.BIAS

    d = lambda t: None if t is None else t.to(dev)
    out = F.weight_only_groupwise_quant_matmul(d(activation), d(pre_quant_scale), d(cuda_q_weight), d(scale), d(zero),
                                               d(bias), d(fp8_alpha), quant_algo, group_size, dtype_str)
    torch.cuda.synchronize()
    ref16, ref32 = oracle.woq_groupwise_matmul(activation, pre_quant_scale if has_pre_quant else None,
                                               ref_q_weight.numpy(), scale, zero, bias, group_size,
                                               alpha=fp8_alpha, act_fp8=use_w4a8_awq)
    if use_w4a8_awq:
        # the reference's own gate for this mode (:202-210): the UN-rounded activation x alpha x pre-quant scale
        # against the plugin, woq_assert_near_eq(ref, output, 2) -- wTypeId 2 = the int4 tolerance
        a = activation.float() * fp8_alpha
        if has_pre_quant:
            a = a * pre_quant_scale.float()
        w = oracle.dequant_fma(ref_q_weight.numpy(), scale, zero, group_size).float()
        gt = a.double() @ w.double() + (bias.double() if has_bias else 0.0)
        atol_ref = gt.abs().max().item() / (1 << 3) * 1.5 * (3.0 if dt == torch.bfloat16 else 1.0)
        assert (out.cpu().double() - gt).abs().max().item() <= atol_ref
    atol = oracle.woq_tolerance(ref16, oracle.INT4_W)
    torch.testing.assert_close(out.cpu().float(), ref16.float(), atol=atol, rtol=1e-7)
    # much tighter than the reference gate: the same dequantised weights, fp32 accumulation
    tight = 2.0 * torch.finfo(dt).eps * ref32.abs().max().item() + 1e-3
    assert (out.cpu().float() - ref32).abs().max().item() <= tight

    # bit-exact gate: dequantised weights recovered from the device layout
    w_dev = F.dequantize_weights(d(cuda_q_weight), d(scale), d(zero), k, n, group_size, L.WEIGHT_INT4, dt)
    torch.cuda.synchronize()
    w_ref = oracle.dequant_fma(ref_q_weight.numpy(), scale, zero, group_size)
    assert torch.equal(w_dev.cpu().view(torch.int16), w_ref.view(torch.int16)), "dequantised weights not bit-exact"


GROUPWISE_GRID = [
    # (m, n, k, has_pre_quant, has_zero, has_bias, group_size): int4 grids :212-219 and prequant grids :259-266
    (1, 1024, 64, False, True, True, 64), (16, 1024, 256, False, True, False, 64),
    (32, 2048, 384, False, False, True, 64), (64, 2048, 1024, False, False, False, 64),
    (2, 1024, 128, False, True, True, 128), (8, 1024, 256, False, True, False, 128),
    (48, 2048, 384, False, False, True, 128), (96, 2048, 1024, False, False, False, 128),
    (3, 1024, 64, True, True, True, 64), (128, 1024, 256, True, True, False, 64),
    (192, 2048, 384, True, False, True, 64), (4, 1024, 128, True, True, True, 128),
    (64, 1024, 256, True, True, False, 128), (256, 2048, 1024, True, False, False, 128),
]


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("m,n,k,pre,zero,bias,gs", GROUPWISE_GRID)
def test_weight_only_groupwise_quant_matmul(cuda, dtype_str, m, n, k, pre, zero, bias, gs):
    _groupwise_case(m, n, k, dtype_str, pre, zero, bias, gs, cuda)


# W4A8-AWQ (FP8_ALPHA): the reference's Hopper-only grid (:318-330, :360-372 at the group sizes built here) plus the
# decode and mid-size row counts, both activation types
W4A8_GRID = [(256, 2048, 1024, True, False, False, 128), (32, 1024, 128, True, True, True, 128),
             (32, 2048, 1024, True, False, True, 128), (64, 2048, 2048, True, False, False, 128),
             (1, 4096, 4096, True, True, False, 128), (8, 1280, 8192, True, True, True, 128),
             (16, 1024, 512, False, True, True, 64), (100, 1024, 1024, False, False, True, 64)]


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("m,n,k,pre,zero,bias,gs", W4A8_GRID)
def test_w4a8_awq_groupwise_quant_matmul(cuda, dtype_str, m, n, k, pre, zero, bias, gs):
    _groupwise_case(m, n, k, dtype_str, pre, zero, bias, gs, cuda, use_w4a8_awq=True)


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("m", [1, 2, 4, 8, 16])
def test_groupwise_baseline_cfg1_shape(cuda, dtype_str, m):
    # BASELINE cfg1: N = K = 4096, g = 128, zeros + pre-quant scale
    _groupwise_case(m, 4096, 4096, dtype_str, True, True, False, 128, cuda)


@pytest.mark.parametrize("pre", [True, False])
@pytest.mark.parametrize("m", [17, 24, 31, 32])
def test_groupwise_two_row_chunks_large_k(cuda, m, pre):
    # 16 < m <= 32.  With a pre-quant scale the call stays on the decode kernel (two 16-row chunks per block; at k = 8192
    # its activation slab needs split-K, so the workspace query must cover every m' <= m, not only multiples of 16 --
    # m = 24 used to fail); without one it takes the skinny form of the GEMM (ragged second m-block, 8 pieces per tile)
    _groupwise_case(m, 2048, 8192, "bfloat16", pre, True, True, 128, cuda)


@pytest.mark.parametrize("m,n,k,pre,zero,bias,gs,dtype_str", [
    (64, 3584, 8192, False, True, False, 128, "bfloat16"), (33, 1280, 8192, True, True, True, 128, "float16"),
    (40, 1024, 4096, False, False, True, 64, "bfloat16"), (64, 8192, 1024, False, True, True, 64, "float16"),
    (49, 2560, 2048, True, False, False, 128, "bfloat16"), (64, 128, 128, False, True, True, 128, "float16")])
def test_groupwise_skinny_form(cuda, m, n, k, pre, zero, bias, gs, dtype_str):
    # 16 < m <= 64 at decode-sized weights (the tensor-parallel slices of cfg4 at a few dozen sequences): the skinny form
    # of the GEMM -- one burst of loads per workgroup, integer-domain int4 arithmetic, split-K closed inside the launch
    # (woq_skinny_kernel).  Ragged last m-block, both group sizes, 1 .. 16 pieces per tile, one-piece tiles (no partials)
    _groupwise_case(m, n, k, dtype_str, pre, zero, bias, gs, cuda)


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("m,n,k,pre,zero,bias,gs", [(512, 32768, 256, False, True, False, 128),
                                                    (300, 32768, 320, True, True, True, 64),
                                                    (1024, 16384, 192, False, False, True, 64),
                                                    (2304, 8192, 192, False, True, True, 64)])
def test_groupwise_prefill_big_tile(cuda, dtype_str, m, n, k, pre, zero, bias, gs):
    # m >= 256 with enough 256 x 256 tiles to fill the chip: the 8-wave tile of the prefill kernel (ragged last m-tile,
    # int4 half k-tile, both scale-group forms; 9 m-tiles: the tile walk's groups of 4 m-tiles end in a group of one)
    _groupwise_case(m, n, k, dtype_str, pre, zero, bias, gs, cuda)


def test_groupwise_empty_batch(cuda):
    dt = torch.float16
    w = torch.zeros(64, 16, dtype=dt, device=cuda)
    out = F.weight_only_groupwise_quant_matmul(torch.zeros(0, 64, dtype=dt, device=cuda), None, w,
                                               torch.ones(1, 64, dtype=dt, device=cuda), None, None, None, 0, 64)
    assert out.shape == (0, 64)


@pytest.mark.parametrize("dtype_str", ["float16", "bfloat16"])
@pytest.mark.parametrize("wtype", [1, 2])
@pytest.mark.parametrize("m,n,k", [(1, 1024, 4096), (4, 4096, 1024), (16, 1024, 1024), (64, 3072, 768)])
def test_weight_only_quant_matmul(cuda, dtype_str, wtype, m, n, k):
    # test_weight_only_quant_matmul.py: weights rand*2-1 quantised per column, output vs (A @ q) * scale
    torch.manual_seed(0)
    dt = DT[dtype_str]
    tq = torch.int8 if wtype == 1 else torch.quint4x2
    weight = (torch.rand(k, n, dtype=dt) * 2 - 1.0)
    unproc, proc, scales = F._symmetric_quantize_last_axis_of_batched_matrix(weight, tq)
    q_ref, s_ref = oracle.symmetric_quantize(weight, wtype)
    assert (unproc.numpy() == q_ref).all() and torch.equal(scales, s_ref)
    q_unpacked = q_ref if wtype == 1 else oracle.unpack_int4_packed_tensor_to_int8(q_ref)
    act = torch.rand(m, k, dtype=dt) * 2 - 1.0
    out = F.weight_only_quant_matmul(act.to(cuda), proc.to(cuda), scales.to(cuda), wtype, dtype_str)
    torch.cuda.synchronize()
    ref16, _ = oracle.woq_matmul(act, q_unpacked, scales)
    torch.testing.assert_close(out.cpu().float(), ref16.float(), atol=oracle.woq_tolerance(ref16, wtype), rtol=1e-7)
    # _unconvert_weights gate (test_weight_only_quant_matmul.py:44-53,151-167): dequantised weights bit-exact
    w_dev = F.dequantize_weights(proc.to(cuda), scales.to(cuda), None, k, n, 0, wtype, dt)
    w_ref = oracle.dequant_fma(q_unpacked, scales.reshape(1, n), None, k)
    assert torch.equal(w_dev.cpu().view(torch.int16), w_ref.view(torch.int16))


def test_reference_preprocessed_weights_can_be_ingested(cuda):
    # a checkpoint preprocessed by the reference for SM80 -> row-major -> gfx950 layout -> same result
    torch.manual_seed(0)
    k, n, dt = 256, 512, torch.float16
    raw = torch.randint(-128, 128, (k, n // 2), dtype=torch.int8)
    sm80 = torch.from_numpy(oracle.preprocess_weights_for_mixed_gemm(raw.numpy(), oracle.INT4_W, arch=80))
    rm = F.unprocess_weights_for_mixed_gemm(sm80, torch.quint4x2, L.LAYOUT_REF_SM80)
    assert torch.equal(rm, raw)
    w = F.preprocess_weights_for_mixed_gemm(rm, torch.quint4x2)
    scale = torch.rand(k // 128, n, dtype=dt)
    w_dev = F.dequantize_weights(w.to(cuda), scale.to(cuda), None, k, n, 128, L.WEIGHT_INT4, dt)
    w_ref = oracle.dequant_fma(oracle.unpack_int4_packed_tensor_to_int8(raw.numpy()), scale, None, 128)
    assert torch.equal(w_dev.cpu().view(torch.int16), w_ref.view(torch.int16))


# ----------------------------------------------------------------------------- SmoothQuant
def _sq_case(m_shape, n, k, per_token, per_channel, dtype, dev):
    """_sq_gemm (test_smooth_quant_gemm.py:43-123)."""
    torch.manual_seed(0)
    m = int(np.prod(m_shape))
    a = torch.randint(-128, 128, (*m_shape, k), dtype=torch.int8)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    shape_a = (m, 1) if per_token else (1, 1)
    shape_b = (1, n) if per_channel else (1, 1)
    sa = torch.randint(1, 10, shape_a).float() * 1e-2
    sb = torch.randint(1, 10, shape_b).float() * 1e-2
    out = F.smooth_quant_gemm(a.to(dev), b.to(dev), sa.to(dev), sb.to(dev), per_token, per_channel,
                              {torch.float16: "float16", torch.float32: "float32", torch.int32: "int32",
                               torch.bfloat16: "bfloat16"}[dtype])
    torch.cuda.synchronize()
    ref, acc = oracle.smooth_quant_gemm(a.reshape(m, k).numpy(), b.numpy(), sa.numpy(), sb.numpy(), per_token,
                                        per_channel, dtype)
    # np.testing.assert_allclose(ref, out) with default rtol 1e-7 (:123) == exact for these dtypes
    assert torch.equal(out.cpu().reshape(m, n), ref), "SmoothQuant output differs from the reference formula"
    # int32 accumulators bit-exact: unit scales, int32 output
    one = torch.ones(1, 1)
    acc_dev = F.smooth_quant_gemm(a.to(dev), b.to(dev), one.to(dev), one.to(dev), False, False, "int32")
    assert torch.equal(acc_dev.cpu().reshape(m, n), torch.from_numpy(acc))


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.int32])
@pytest.mark.parametrize("per_token,per_channel", [(False, False), (False, True), (True, False), (True, True)])
def test_smooth_quant_gemm_reference_grid(cuda, dtype, per_token, per_channel):
    bs, inseq, hidden = 2, 16, 768
    _sq_case((bs, inseq), 3 * hidden, hidden, per_token, per_channel, dtype, cuda)   # qkv_gemm
    _sq_case((bs, inseq), 4 * hidden, hidden, per_token, per_channel, dtype, cuda)   # mlp_gemm_1


def test_smooth_quant_gemm_ragged_and_large(cuda):
    _sq_case((1,), 160, 48, True, True, torch.float16, cuda)          # tiles with row / col / k tails
    _sq_case((257,), 1000, 784, True, True, torch.float32, cuda)
    _sq_case((512,), 2560, 5120, True, True, torch.float16, cuda)     # a slice of the 13B MLP shape
    _sq_case((0,), 128, 64, True, True, torch.float16, cuda)          # empty batch


@pytest.mark.parametrize("m,n,k", [(256, 256, 128), (256, 512, 640), (512, 768, 1024), (768, 1280, 256)])
def test_smooth_quant_gemm_phased_256_tiles(cuda, m, n, k):
    # multiples of 256: the phased 256 x 256 kernel, and for n = 1280 with 3 m-tiles no whole round -> single shape;
    # (512, 768): tiles do not fill a round either; all must be exact, int32 accumulators included
    _sq_case((m,), n, k, True, True, torch.float16, cuda)
    _sq_case((m,), n, k, False, False, torch.int32, cuda)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_quantize_per_token(cuda, dtype):
    torch.manual_seed(0)
    x = torch.randn(37, 1000, dtype=dtype)
    x[3] = 0  # an all-zero row exercises the 1e-6 floor (quantization.cu:99)
    q, s = F.quantize_per_token(x.to(cuda))
    torch.cuda.synchronize()
    q_ref, s_ref = oracle.quantize_per_token(x)
    assert (q.cpu().numpy() == q_ref).all()
    assert np.array_equal(s.cpu().numpy().reshape(-1, 1), s_ref)


# ----------------------------------------------------------------------------- FP8
@pytest.mark.parametrize("mode", ["activation", "weight", "per_tensor"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
def test_quantize_e4m3(cuda, mode, dtype):
    a = torch.tensor([[1, 2, 3], [2, 4, 6]], dtype=dtype)  # test_fp8_quantization.py:45-80
    _, s = F.quantize_e4m3(a.to(cuda), mode)
    _, s_ref = oracle.quantize_e4m3(a, mode)
    assert torch.equal(s.cpu().reshape(-1), s_ref.reshape(-1))
    torch.manual_seed(42)
    x = torch.randn(64, 256, dtype=dtype)
    q, s = F.quantize_e4m3(x.to(cuda), mode)
    q_ref, s_ref = oracle.quantize_e4m3(x, mode)
    assert torch.equal(s.cpu().reshape(-1), s_ref.reshape(-1))
    assert torch.equal(q.cpu().view(torch.uint8), q_ref.view(torch.uint8))
    assert (q.cpu().float() * s.cpu() - x.float()).abs().max().item() <= 0.2  # atol of :113-115


@pytest.mark.parametrize("out_dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("m,n,k", [(64, 256, 128), (200, 520, 1008), (512, 1024, 4096), (256, 256, 128), (768, 512, 640)])
def test_fp8_gemm(cuda, out_dtype, m, n, k):
    # parity unpinned by the reference (closed libraries); semantic of quantization/layers.py:777-879
    torch.manual_seed(0)
    a, b = torch.randn(m, k), torch.randn(n, k)
    qa, sa = oracle.quantize_e4m3(a, "per_tensor")
    qb, sb = oracle.quantize_e4m3(b, "per_tensor")
    name = {torch.float16: "float16", torch.bfloat16: "bfloat16", torch.float32: "float32"}[out_dtype]
    out = F.fp8_gemm(qa.to(cuda), qb.to(cuda), sa.reshape(1).to(cuda), sb.reshape(1).to(cuda), name)
    torch.cuda.synchronize()
    ref, _ = oracle.fp8_gemm(qa, qb, sa.item(), sb.item(), torch.float32)
    # products of e4m3 values are exact in fp32; only the fp32 accumulation order differs
    tol = 1e-5 * ref.abs().max().item() * (k ** 0.5) + torch.finfo(out_dtype).eps * ref.abs().max().item()
    assert (out.cpu().float() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rows,cols", [(1, 4096), (5, 11008), (64, 4096), (300, 1024), (4096, 5120)])
def test_apply_per_channel_scale_direct(cuda, dtype, rows, cols):
    # the stand-alone smoothing pass (preQuantScaleKernel.cu:26-136 apply_per_channel_scale_kernel_launcher): out = act *
    # scale[col] in T, one rounding -- bit-exact against the oracle; in place too (the plugin smooths into its workspace)
    import ctypes as C
    from mi355q import _lib as L
    g = torch.Generator().manual_seed(rows + cols)
    act = torch.randn(rows, cols, generator=g).to(dtype)
    scale = (torch.rand(1, cols, generator=g) + 0.5).to(dtype)
    ref = oracle.apply_per_channel_scale(act, scale)
    d_act, d_scale = act.to(cuda), scale.to(cuda)
    out = torch.zeros_like(d_act)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    code = L.HALF if dtype == torch.float16 else L.BF16
    L.check(L.lib().mi355q_apply_per_channel_scale(out.data_ptr(), d_act.data_ptr(), d_scale.data_ptr(), rows, cols, code, stream))
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    L.check(L.lib().mi355q_apply_per_channel_scale(d_act.data_ptr(), d_act.data_ptr(), d_scale.data_ptr(), rows, cols, code, stream))
    torch.cuda.synchronize()
    assert torch.equal(d_act.cpu().view(torch.int16), ref.view(torch.int16))
