"""CPU tests of the host-side pieces either side of the kernels: the paged-KV block manager
(mirror of tensorrt_llm/runtime/kv_cache_manager.py) and the GPTQ / AWQ checkpoint-tensor conversions
(tensorrt_llm/models/llama/weight.py).  The layout transform they call is the C ABI's host routine."""
import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import checkpoint as ck
from mi355q import functional as F
from mi355q.kv_cache_manager import BlocksManager, GenerationSequence, KVCacheManager


def make_pools(layers, blocks, tpb, heads=2, dh=8, dtype=torch.float16):
    return [torch.zeros(2 * blocks * heads * tpb * dh, dtype=dtype) for _ in range(layers)]


def test_blocks_manager_pointer_layout_and_reuse():
    blocks, tpb = 6, 4
    pools = make_pools(2, blocks, tpb)
    bm = BlocksManager(pools, blocks, max_blocks_per_seq=3)
    epb = pools[0].nelement() // (2 * blocks)
    s0, s1 = GenerationSequence(10, 0), GenerationSequence(11, 1)
    bm.allocate(s0)
    bm.allocate(s1)
    bm.allocate(s0)
    arr = bm.get_pointer_array(1, 1)
    assert arr.shape == (2, 1, 2, 3) and arr.dtype == torch.int64
    base = pools[1].data_ptr()
    assert arr[0, 0, 0].tolist() == [base, base + 2 * epb * 2, 0]              # blocks 0 and 2 (FIFO order)
    assert arr[1, 0, 0].tolist() == [base + 1 * epb * 2, 0, 0]
    assert arr[0, 0, 1, 0].item() == base + (blocks + 0) * epb * 2             # V half follows all K blocks
    assert bm.get_number_blocks(s0) == 2
    bm.free(s0)
    assert len(bm.free_blocks) == 5 and [b.idx for b in bm.free_blocks] == [3, 4, 5, 0, 2]
    for i in range(5):
        bm.allocate(GenerationSequence(20 + i, 1 + i))
    assert not bm.has_free_block()
    with pytest.raises(RuntimeError, match="Can't allocate new block"):
        bm.allocate(GenerationSequence(30, 6))


def test_kv_cache_manager_context_and_steps():
    blocks, tpb, max_blocks = 16, 4, 4
    pools = make_pools(3, blocks, tpb)
    mgr = KVCacheManager(pools, blocks, tpb, max_blocks, max_attention_window_size=16, sink_token_len=0)
    a, b = GenerationSequence(0, 0), GenerationSequence(1, 1)
    mgr.add_sequence(a, context_len=5)    # 2 blocks
    mgr.add_sequence(b, context_len=8)    # exactly 2 blocks: the 9th token opens a third at the first step
    assert mgr.blocks_manager.get_number_blocks(a) == 2 and mgr.blocks_manager.get_number_blocks(b) == 2
    ptrs = mgr.get_block_pointers(1)
    assert ptrs.shape == (3, 2, 1, 2, max_blocks)
    mgr.step([False, False])
    assert mgr.blocks_manager.get_number_blocks(a) == 2 and mgr.blocks_manager.get_number_blocks(b) == 3
    assert mgr.lens == [6, 9]
    for _ in range(2):
        mgr.step([False, False])
    mgr.step([False, False])              # a: length 8 -> token 9 starts block 3
    assert mgr.blocks_manager.get_number_blocks(a) == 3
    mgr.step([True, False])               # a finishes: its blocks return, b becomes batch index 0
    assert len(mgr.sequences) == 1 and mgr.sequences[0].get_batch_idx() == 0 and mgr.lens == [13]
    assert len(mgr.blocks_manager.free_blocks) == blocks - mgr.blocks_manager.get_number_blocks(b)
    # cyclic window: once the window is full no further blocks are taken
    for _ in range(10):
        mgr.step([False])
    assert mgr.blocks_manager.get_number_blocks(b) == 4 == max_blocks


def test_kv_cache_manager_sink_bubble_and_beams():
    pools = make_pools(1, 32, 4)
    mgr = KVCacheManager(pools, 32, 4, 8, max_attention_window_size=16, sink_token_len=2, beam_width=2)
    assert mgr.bubble_len == 2 and mgr.sink_block_token_num == 4 and mgr.max_token_num == 18
    s = GenerationSequence(7, 0)
    mgr.add_sequence(s, context_len=5)    # 5 + bubble 2 = 7 tokens -> 2 blocks, the last one not shared across beams
    beams = mgr.blocks_manager.allocated_blocks[s]
    assert beams[0][0].idx == beams[1][0].idx and beams[0][0].is_shared()
    assert beams[0][1].idx != beams[1][1].idx and not beams[0][1].is_shared()
    ptr = mgr.get_block_pointers(2)
    assert ptr.shape == (1, 1, 2, 2, 8) and ptr[0, 0, 0, 0, 0] == ptr[0, 0, 1, 0, 0] and ptr[0, 0, 0, 0, 1] != ptr[0, 0, 1, 0, 1]


def _gptq_pack(q_u4_kn, z_u4, ):
    k, n = q_u4_kn.shape
    qw = torch.zeros(k // 8, n, dtype=torch.int32)
    for i in range(8):
        qw |= q_u4_kn[i::8].to(torch.int32) << (4 * i)
    g, _ = z_u4.shape
    qz = torch.zeros(g, n // 8, dtype=torch.int32)
    for i in range(8):
        qz |= z_u4[:, i::8].to(torch.int32) << (4 * i)
    return qw, qz


def test_gptq_tensors_to_plugin_tensors():
    # a GPTQ checkpoint's dequantisation is scale * (q - (z + 1)); the converted tensors must dequantise to the same
    g0 = torch.Generator().manual_seed(0)
    k, n, gs = 256, 128, 64
    q = torch.randint(0, 16, (k, n), generator=g0)
    z = torch.randint(0, 15, (k // gs, n), generator=g0)
    scales = (torch.rand(k // gs, n, generator=g0) * 0.02 + 0.005).half()
    qw, qz = _gptq_pack(q, z)
    t = ck.gptq_to_tllm(qw, qz, scales, layout=L.LAYOUT_ROW_MAJOR)
    assert t["weight"].shape == (k, n // 2) and t["weight"].dtype == torch.int8
    unpacked = F.unpack_int4_packed_tensor_to_int8(t["weight"])          # signed int4 values, [K, N]
    assert torch.equal(unpacked.to(torch.int64), q - 8)
    want_zero = ((-(z) + 8 - 1).half() * scales)
    assert torch.equal(t["zero"], want_zero)
    deq = oracle.dequant_fma(unpacked, t["weights_scaling_factor"], t["zero"], gs).float()
    ref = (scales.float().repeat_interleave(gs, 0) * (q - (z.repeat_interleave(gs, 0) + 1)).float())
    assert (deq - ref).abs().max().item() <= 2e-3                         # one fp16 rounding of the zero term
    # the production layout is the same values behind the GFX950 permutation
    t2 = ck.gptq_to_tllm(qw, qz, scales)
    back = F.unprocess_weights_for_mixed_gemm(t2["weight"], torch.quint4x2, L.LAYOUT_GFX950)
    assert torch.equal(back, t["weight"])
    # TP: column split (dim 1) of every tensor
    t_r1 = ck.gptq_to_tllm(qw, qz, scales, tp_size=2, tp_rank=1, tp_dim=1, layout=L.LAYOUT_ROW_MAJOR)
    assert torch.equal(F.unpack_int4_packed_tensor_to_int8(t_r1["weight"]), unpacked[:, n // 2:])
    assert torch.equal(t_r1["zero"], want_zero[:, n // 2:])


def test_awq_tensors_to_plugin_tensors():
    g0 = torch.Generator().manual_seed(1)
    n, k, gs = 64, 256, 128
    w = torch.randn(n, k, generator=g0)
    amax = w.reshape(n, k // gs, gs).abs().amax(-1).reshape(-1)           # AMMO stores amax per (n, group)
    pqs = torch.rand(k, generator=g0) + 0.5
    t = ck.awq_to_tllm(w, amax, pqs, gs, layout=L.LAYOUT_ROW_MAJOR)
    scale = t["weights_scaling_factor"]
    assert scale.shape == (k // gs, n) and t["prequant_scaling_factor"].shape == (1, k)
    q = F.unpack_int4_packed_tensor_to_int8(t["weight"]).float()
    assert q.min().item() >= -8 and q.max().item() <= 7
    deq = q * (amax.reshape(n, k // gs).T / 8.0).repeat_interleave(gs, 0)
    # half a step everywhere, a whole step where +amax rounds to 8 and is clamped to 7
    assert (deq - w.T).abs().max().item() <= amax.max().item() / 8.0 + 1e-6
    # row-parallel split (tp_dim 0) also splits the pre-quant scale
    t1 = ck.awq_to_tllm(w, amax, pqs, gs, tp_size=2, tp_rank=1, tp_dim=0, layout=L.LAYOUT_ROW_MAJOR)
    assert t1["weight"].shape == (k // 2, n // 2) and torch.equal(t1["prequant_scaling_factor"],
                                                                  t["prequant_scaling_factor"][:, k // 2:])
    # qkv fusion: re-smoothing to the averaged pre-quant scale keeps (x * s_avg) @ w' == (x * s) @ w
    w_kn = w.T.contiguous()
    avg = (pqs + 0.25)
    w2, sc2 = ck.resmooth_and_get_scale(w_kn, pqs.reshape(1, k), avg.reshape(1, k), gs)
    x = torch.randn(3, k, generator=g0)
    assert torch.allclose((x * avg) @ w2, (x * pqs) @ w_kn, rtol=1e-4, atol=1e-4)
    assert sc2.shape == (k // gs, n)


def test_awq_division_runs_in_the_checkpoint_dtype():
    """`weight /= scale...` runs on the checkpoint's fp16 tensors in the reference (models/llama/weight.py:1355-1359): a
    quotient that fp16 rounds onto a .5 tie gives another int4 code than the fp32 quotient would."""
    from mi355q import checkpoint as ck
    gs = 64
    # w / s = 2.4996 in fp32 (-> 2), but fp16(2.4996...) = 2.5 -> round-half-even 2 ... pick a value that differs:
    s = torch.tensor(0.1, dtype=torch.float16)
    w16 = torch.tensor(0.35, dtype=torch.float16)      # fp16(0.35) / fp16(0.1) = 3.5003 in fp32 -> 4; in fp16 3.5 -> 4
    cands = torch.arange(0.05, 0.75, 0.0001).to(torch.float16).unique()
    q16 = torch.round(cands / s)
    q32 = torch.round(cands.float() / s.float())
    differ = cands[q16 != q32.to(torch.float16)]
    assert differ.numel() > 0, "no fp16 / fp32 rounding difference in the probe range"
    val = differ[0]
    weight_nk = torch.zeros(64, gs, dtype=torch.float16)
    weight_nk[:, 0] = val
    amax = torch.full((64,), float(s) * 8.0, dtype=torch.float16)          # scale = amax / 8 = s
    t = ck.awq_to_tllm(weight_nk, amax, torch.ones(gs, dtype=torch.float16), gs, layout=L.LAYOUT_ROW_MAJOR)
    q = F.unpack_int4_packed_tensor_to_int8(t["weight"])
    want = int(torch.clamp(torch.round(val / (amax[0] / 8.0)), -8, 7).item())        # the reference's arithmetic
    other = int(torch.clamp(torch.round(val.float() / (amax[0].float() / 8.0)), -8, 7).item())
    assert want != other
    assert (q[0] == want).all(), "the quotient must be formed in fp16 like the reference"


def test_smoothquant_checkpoint_tensors():
    """models/llama/convert.py:21-176 restated in mi355q.checkpoint: the migration leaves x @ W^T unchanged when the
    activations are divided by the same scales, and the int8 tensors + scales reproduce the fp GEMM through the
    oracle's SmoothQuant GEMM (a8) within the int8 quantisation error."""
    from mi355q import checkpoint as ck
    g = torch.Generator().manual_seed(5)
    m, k, n = 16, 96, 48
    x = torch.randn(m, k, generator=g, dtype=torch.float64)
    x[:, 7] *= 30  # an outlier channel: what SmoothQuant is for
    w_nk = torch.randn(n, k, generator=g, dtype=torch.float64) * 0.1
    ln = torch.ones(k, dtype=torch.float64)
    y_ref = x @ w_nk.t()
    w2 = w_nk.clone()
    s = ck.smooth_gemm(w2, x.abs().amax(0), layernorm_weights=ln, alpha=0.5)
    assert s.shape == (k,) and (s >= 1e-5).all() and s[7] == s.max()
    assert torch.allclose((x / s) @ w2.t(), y_ref, rtol=1e-12, atol=1e-12)
    assert torch.allclose(ln, 1.0 / s)
    xs = x / s
    # the migrated problem is better conditioned for int8: the outlier channel no longer dominates the range
    assert (xs.abs().amax(0).max() / xs.abs().amax(0).median()) < (x.abs().amax(0).max() / x.abs().amax(0).median())

    rng = {"x": xs.abs().amax(0), "y": y_ref.abs().amax(0), "w": w2.abs().amax(1)}
    t = ck.generate_int8(w2.t().contiguous(), rng)
    assert t["weight.int8"].shape == (k, n) and t["weight.int8"].dtype == np.int8
    assert np.abs(t["weight.int8.col"]).max() == 127 and np.abs(t["weight.int8.col"]).max(axis=0).min() >= 126
    assert t["scale_w_quant_orig.col"].shape == (n,) and t["scale_w_quant_orig"].shape == ()
    np.testing.assert_allclose(t["scale_y_accum_quant.col"],
                               (127.0 / rng["y"].max().item()) * t["scale_w_quant_orig.col"] / t["scale_x_orig_quant"], rtol=1e-6)
    # static per-tensor activation scale + per-channel weights through the oracle GEMM
    a_i8 = np.clip(np.round(xs.numpy() * t["scale_x_orig_quant"]), -128, 127).astype(np.int8)
    out, _ = oracle.smooth_quant_gemm(torch.from_numpy(a_i8), torch.from_numpy(t["weight.int8.col"].T.copy()),
                                   torch.tensor([1.0 / float(t["scale_x_orig_quant"])]),
                                   torch.from_numpy(t["scale_w_quant_orig.col"]).reshape(1, n), False, True, torch.float32)
    err = (torch.as_tensor(out).double() - y_ref).abs().max().item()
    assert err < 0.03 * y_ref.abs().max().item(), err

    # fused QKV: "per-tensor" means one scale per Q, K, V block, replicated along the block's columns
    wq = torch.randn(k, 3 * 16, generator=g, dtype=torch.float64) * torch.tensor([1.0, 4.0, 0.25]).repeat_interleave(16)
    tq = ck.generate_int8(wq, {"x": xs.abs().amax(0), "y": torch.ones(48), "w": wq.abs().amax(0)}, is_qkv=True)
    sw = tq["scale_w_quant_orig"]
    assert sw.shape == (48,) and len(set(sw[:16])) == 1 and len(set(sw[16:32])) == 1 and sw[16] > sw[0] > sw[32]
    assert np.abs(tq["weight.int8"][:, 16:32]).max() == 127 and np.abs(tq["weight.int8"][:, 32:]).max() == 127
