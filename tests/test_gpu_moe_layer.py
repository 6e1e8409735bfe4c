"""GPU parity: the whole mixture-of-experts layer with weight-only quantised experts through the MixtureOfExperts plugin
and mi355q_run_moe, against oracle.run_moe (CutlassMoeFCRunner::runMoe, kernels/mixtureOfExperts/moe_kernels.cu:922-1035).
Routing (scales, expert choice, permutation, per-expert row counts) is compared exactly through the workspace; the
output within the weight-only tolerance of the two chained GEMMs.  The reference's own test
(cpp/tests/kernels/mixtureOfExpertsTest.cu) computes its expectations on the fly from unquantised experts: no stored
vector pins the quantised layer ("parity unpinned" beyond the cited semantics)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

pytestmark = pytest.mark.gpu

ACT = {"identity": F.ACT_IDENTITY, "relu": F.ACT_RELU, "silu": F.ACT_SILU, "gelu": F.ACT_GELU, "swiglu": F.ACT_SWIGLU,
       "geglu": F.ACT_GEGLU}


def make_experts(E, k_in, n_out, dt, wtype, seed):
    torch.manual_seed(seed)
    tq = torch.int8 if wtype == L.WEIGHT_INT8 else torch.quint4x2
    q_all, proc_all, sc_all = [], [], []
    for e in range(E):
        weight = (torch.rand(k_in, n_out, dtype=dt) * 2 - 1.0) * 0.2
        unproc, proc, scales = F._symmetric_quantize_last_axis_of_batched_matrix(weight, tq)
        q_all.append(unproc.numpy() if wtype == L.WEIGHT_INT8 else oracle.unpack_int4_packed_tensor_to_int8(unproc.numpy()))
        proc_all.append(proc)
        sc_all.append(scales.reshape(-1))
    return np.stack(q_all), torch.stack(proc_all).contiguous(), torch.stack(sc_all).contiguous()


def run_case(dev, rows, hidden, inter, E, k, dt, wtype, activation, with_bias=False, renorm=False, finished=None,
             parallel=(0, 1, 0), seed=0):
    gated = activation in ("swiglu", "geglu")
    pmode, psize, prank = parallel
    ep_size, ep_rank = (psize, prank) if pmode == 1 else (1, 0)
    tp_rank = prank if pmode == 2 else 0
    epn = E // ep_size
    q1, w1, s1 = make_experts(epn, hidden, inter * (2 if gated else 1), dt, wtype, seed)
    q2, w2, s2 = make_experts(epn, inter, hidden, dt, wtype, seed + 1)
    g = torch.Generator().manual_seed(seed + 2)
    x = (torch.randn(rows, hidden, generator=g) * 0.5).to(dt)
    gating = torch.randn(rows, E, generator=g) * 2.0
    b1 = (torch.randn(epn, q1.shape[2], generator=g) * 0.1).to(dt) if with_bias else None
    b2 = (torch.randn(epn, hidden, generator=g) * 0.1).to(dt) if with_bias else None
    fin = None if finished is None else torch.tensor(finished, dtype=torch.bool)
    out = F.mixture_of_experts(
        x.to(dev), gating.to(dev), w1.to(dev), w2.to(dev), s1.to(dev), s2.to(dev), k, ACT[activation], wtype,
        None if b1 is None else b1.to(dev), None if b2 is None else b2.to(dev), None if fin is None else fin.to(dev),
        normalization_mode=int(renorm), tp_size=psize, tp_rank=prank, parallelism_mode=pmode)
    torch.cuda.synchronize()
    ref, mid = oracle.run_moe(x, gating, q1, s1, b1, q2, s2, b2, k, activation, renorm, fin, ep_size, ep_rank, tp_rank)
    assert out.shape == ref.shape
    # two chained weight-only GEMMs: the reference tolerance of the second on top of the first one's rounding
    atol = 2.0 * oracle.woq_tolerance(mid["fc2"] if mid["fc2"].numel() else ref, wtype, dt) + 2e-3
    err = (out.cpu().float() - ref.float()).abs().max().item()
    assert err <= atol, f"max abs err {err} > {atol}"
    tight = 8.0 * torch.finfo(dt).eps * max(ref.float().abs().max().item(), 1.0)
    assert err <= tight, f"max abs err {err} > {tight} (rounding-level gate)"
    return out.cpu(), ref, mid


@pytest.mark.parametrize("dt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("wtype", [L.WEIGHT_INT8, L.WEIGHT_INT4])
@pytest.mark.parametrize("activation", ["swiglu", "relu"])
def test_moe_layer_mixtral_like(cuda, dt, wtype, activation):
    # 8 experts, top-2, renormalised scales (Mixtral): decode-sized (one sort workgroup) and prefill-sized (multi-block sort)
    run_case(cuda, 40, 256, 384, 8, 2, dt, wtype, activation, renorm=True)
    run_case(cuda, 1500, 256, 256, 8, 2, dt, wtype, activation, renorm=True, seed=3)


@pytest.mark.parametrize("activation", ["identity", "silu", "gelu", "geglu"])
def test_moe_layer_activations_bias_and_scale_modes(cuda, activation):
    run_case(cuda, 77, 128, 256, 4, 1, torch.float16, L.WEIGHT_INT8, activation, with_bias=True)             # k = 1, scale
    run_case(cuda, 77, 128, 256, 4, 1, torch.float16, L.WEIGHT_INT8, activation, with_bias=True, renorm=True)  # k = 1: no scale
    run_case(cuda, 33, 256, 128, 16, 4, torch.bfloat16, L.WEIGHT_INT4, activation, with_bias=True)            # k = 4, plain scales


def test_moe_layer_many_experts_and_edge_rows(cuda):
    run_case(cuda, 1, 128, 128, 8, 2, torch.float16, L.WEIGHT_INT8, "swiglu", renorm=True)        # a single token
    run_case(cuda, 300, 128, 128, 64, 2, torch.bfloat16, L.WEIGHT_INT8, "silu")                  # 64 experts, most nearly empty
    run_case(cuda, 50, 128, 128, 128, 8, torch.float16, L.WEIGHT_INT4, "relu", renorm=True)      # 128 experts: 2 per lane in the router
    run_case(cuda, 1024, 128, 128, 8, 2, torch.float16, L.WEIGHT_INT8, "identity")               # n = 2048: exactly one chunk
    run_case(cuda, 1025, 128, 128, 8, 2, torch.float16, L.WEIGHT_INT8, "identity")               # one pair into a second chunk


@pytest.mark.parametrize("rows,wtype", [(60, L.WEIGHT_INT4), (400, L.WEIGHT_INT4), (400, L.WEIGHT_INT8), (900, L.WEIGHT_INT4)])
def test_moe_layer_w2_split_k(cuda, rows, wtype):
    # a long inter_size against a narrow hidden_size: the w2 GEMM has few n-tiles and >= 16 k-tiles, so it splits K over
    # workgroups through the fp32 slabs of the workspace (8 / 4 / 2 splits allowed up to 512 / 1024 / 2048 expanded rows)
    run_case(cuda, rows, 512, 4096, 8, 2, torch.bfloat16, wtype, "swiglu", renorm=True, seed=5)


def test_moe_layer_two_part_expert_gemms(cuda):
    # 1050 expanded rows per expert on average, int4, wide enough for the 256 x 256 tile in both expert GEMMs: each runs as a
    # launch over the experts' whole 256-row tiles plus a launch of 64-row tiles over their ragged ends -- with the row
    # gather and the gated activation fused into the first, the finalize step reading the second's rows
    run_case(cuda, 2100, 4096, 2048, 4, 2, torch.bfloat16, L.WEIGHT_INT4, "swiglu", renorm=True, seed=7)


def test_moe_layer_finished_rows_and_parallelism(cuda):
    fin = [i % 3 == 0 for i in range(90)]
    out, _, _ = run_case(cuda, 90, 128, 256, 8, 2, torch.float16, L.WEIGHT_INT8, "swiglu", finished=fin, with_bias=True)
    assert (out[::3] == 0).all()  # finished rows are routed nowhere
    # expert parallelism: every rank computes its experts' share; the shares sum to the unsplit layer
    full, ref_full, _ = run_case(cuda, 64, 128, 256, 8, 2, torch.bfloat16, L.WEIGHT_INT8, "silu")
    for rank in range(2):
        run_case(cuda, 64, 128, 256, 8, 2, torch.bfloat16, L.WEIGHT_INT8, "silu", parallel=(1, 2, rank))
    # tensor parallelism: bias 2 only on rank 0
    run_case(cuda, 64, 128, 256, 8, 2, torch.float16, L.WEIGHT_INT4, "gelu", with_bias=True, parallel=(2, 2, 1))


@pytest.mark.parametrize("rows,E,k", [(333, 8, 2), (40, 8, 2), (100, 6, 3), (64, 64, 4), (1100, 8, 2), (30, 128, 2)])
def test_moe_routing_exact_through_c_abi(cuda, rows, E, k):
    # mi355q_run_moe directly; the routing state it leaves in the workspace equals the oracle's, entry by entry.
    # One-wave and four-wave forms of the small sort, k = 3 over 6 experts, 64 experts with k = 4, the multi-block sort
    # (1100 rows), 128 experts (two per lane in the router)
    hidden, inter, dt, wtype = 128, 128, torch.float16, L.WEIGHT_INT8
    q1, w1, s1 = make_experts(E, hidden, inter, dt, wtype, 0)
    q2, w2, s2 = make_experts(E, inter, hidden, dt, wtype, 1)
    g = torch.Generator().manual_seed(9)
    x = (torch.randn(rows, hidden, generator=g) * 0.5).to(dt)
    gating = torch.randn(rows, E, generator=g)
    gating[5] = 0.25      # all experts tie: the two lowest indices win
    gating[6, 3] = gating[6, 1] = 4.0
    dev = cuda
    xd, gd, out = x.to(dev), gating.to(dev), torch.empty(rows, hidden, dtype=dt, device=dev)
    w1d, w2d, s1d, s2d = w1.to(dev), w2.to(dev), s1.to(dev), s2.to(dev)
    p = L.MoeParams()
    p.input, p.gating_output, p.output = xd.data_ptr(), gd.data_ptr(), out.data_ptr()
    p.fc1_expert_weights, p.fc1_scales = w1d.data_ptr(), s1d.data_ptr()
    p.fc2_expert_weights, p.fc2_scales = w2d.data_ptr(), s2d.data_ptr()
    p.num_rows, p.hidden_size, p.inter_size, p.num_experts, p.k = rows, hidden, inter, E, k
    p.activation_type, p.act_type, p.weight_type = F.ACT_RELU, L.HALF, wtype
    need = L.lib().mi355q_moe_workspace_size(C.byref(p))
    assert need > 0
    ws = torch.zeros(need, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need
    L.check(L.lib().mi355q_run_moe(C.byref(p), None))
    torch.cuda.synchronize()
    ref, mid = oracle.run_moe(x, gating, q1, s1, None, q2, s2, None, k, "relu")
    n = rows * k
    al = lambda b: (b + 255) // 256 * 256
    w = ws.cpu()
    off = 0
    scales = w[off:off + n * 4].view(torch.float32); off += al(n * 4)
    experts = w[off:off + n * 4].view(torch.int32); off += al(n * 4)
    src_to_dest = w[off:off + n * 4].view(torch.int32); off += al(n * 4)
    off += al(n * 4)                                   # dest_to_src
    off += al((E + 1) * max(1, (n + 2047) // 2048) * 4)  # block counts [buckets, sort chunks of 2048 pairs]
    total_rows = w[off:off + E * 8].view(torch.int64)
    assert torch.equal(experts.reshape(rows, k).long(), mid["experts"])
    assert experts.reshape(rows, k)[5].tolist()[:2] == [0, 1] and experts.reshape(rows, k)[6].tolist()[:2] == [1, 3]
    assert (scales.reshape(rows, k) - mid["scales"]).abs().max().item() <= 1e-6
    assert torch.equal(total_rows, mid["total_rows"])
    # src_to_dest is indexed by the expanded source row j * rows + token
    pair = torch.arange(n)
    exp_src = (pair % k) * rows + pair // k
    assert torch.equal(src_to_dest[exp_src].long(), mid["dest_of_pair"])
    assert (out.cpu().float() - ref.float()).abs().max().item() <= 8 * torch.finfo(dt).eps * max(ref.float().abs().max().item(), 1.0)
    # errors: short workspace, dense experts
    p.workspace_bytes = need - 1
    assert L.lib().mi355q_run_moe(C.byref(p), None) != 0 and b"workspace" in L.lib().mi355q_last_error()
    p.workspace_bytes = need
    p.weight_type = 0
    assert L.lib().mi355q_run_moe(C.byref(p), None) != 0
