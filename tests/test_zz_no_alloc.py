"""The entry point that allocates nothing (mi355q_masked_multihead_attention_static): same generation step, split-KV
partials in the library's static device array instead of a caller workspace.  Checked through the C ABI against the
existing entry point (mi355q_masked_multihead_attention) on the same inputs: bit-identical outputs (the same kernels
run), on two calls in a row, the second with a smaller input -- the array keeps the first call's contents."""
import ctypes as C

import pytest
import torch

from mi355q import _lib as L

pytestmark = pytest.mark.gpu


def _case(dev, B, Hq, Hkv, D, ctx, tpb, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    blocks_per_seq = (ctx + tpb - 1) // tpb
    nblk = B * blocks_per_seq
    pool = (torch.randn(nblk, 2, Hkv, tpb, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    perm = torch.randperm(nblk, generator=g).reshape(B, blocks_per_seq)
    be = Hkv * tpb * D * 2
    ptrs = (torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + pool.data_ptr()).to(torch.int64).to(dev)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, generator=g).to(torch.bfloat16).to(dev)
    lens = torch.randint(ctx // 2, ctx + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    return pool, ptrs, qkv, lens, blocks_per_seq


def _params(qkv, out, lens, ptrs, B, Hq, Hkv, D, blocks_per_seq, tpb, ctx):
    p = L.DecodeAttentionParams()
    p.qkv, p.qkv_bias, p.out = qkv.data_ptr(), None, out.data_ptr()
    p.sequence_lengths, p.block_pointers = lens.data_ptr(), ptrs.data_ptr()
    p.batch, p.num_heads, p.num_kv_heads, p.head_size = B, Hq, Hkv, D
    p.max_blocks_per_seq, p.tokens_per_block, p.max_seq_len = blocks_per_seq, tpb, ctx
    p.act_type = p.kv_type = L.BF16
    p.position_embedding_type, p.rotary_embedding_dim, p.rotary_embedding_base = 2, D, 10000.0
    p.rotary_embedding_scale_type, p.rotary_embedding_scale, p.rotary_embedding_max_positions = 0, 1.0, 4096
    p.q_scaling, p.multi_block_mode = 1.0, 0
    return p


def _both(dev, B, Hq, Hkv, D, ctx, tpb, seed):
    pool, ptrs, qkv, lens, bps = _case(dev, B, Hq, Hkv, D, ctx, tpb, seed)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out_ws = torch.zeros(B, Hq * D, dtype=torch.bfloat16, device=dev)
    out_st = torch.zeros_like(out_ws)
    # existing entry point, caller workspace
    p = _params(qkv, out_ws, lens, ptrs, B, Hq, Hkv, D, bps, tpb, ctx)
    need = L.lib().mi355q_decode_attention_workspace_size(C.byref(p))
    assert need > 0, "the case must split the KV range (otherwise no scratch is used)"
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need
    L.check(L.lib().mi355q_masked_multihead_attention(C.byref(p), stream))
    # the static-scratch entry point: no workspace handed over
    q = _params(qkv, out_st, lens, ptrs, B, Hq, Hkv, D, bps, tpb, ctx)
    q.workspace, q.workspace_bytes = None, 0
    L.check(L.lib().mi355q_masked_multihead_attention_static(C.byref(q), stream))
    torch.cuda.synchronize()
    assert out_ws.abs().max().item() > 0
    assert torch.equal(out_ws.view(torch.int16), out_st.view(torch.int16)), "static-scratch output differs"


def test_static_scratch_entry_matches_workspace_entry(cuda):
    assert L.lib().mi355q_decode_attention_static_scratch_size() >= 1 << 20
    _both(cuda, B=8, Hq=8, Hkv=1, D=128, ctx=2048, tpb=64, seed=11)   # 8 kv-head workgroups -> many splits
    _both(cuda, B=3, Hq=8, Hkv=1, D=128, ctx=640, tpb=64, seed=12)    # smaller, right after: stale partials around
    _both(cuda, B=4, Hq=4, Hkv=4, D=64, ctx=1024, tpb=32, seed=13)    # the VALU kernel's split path


def test_static_scratch_rejects_oversize(cuda):
    B, Hq, Hkv, D, ctx, tpb = 512, 32, 32, 128, 4096, 64
    p = L.DecodeAttentionParams()
    p.batch, p.num_heads, p.num_kv_heads, p.head_size = B, Hq, Hkv, D
    p.max_blocks_per_seq, p.tokens_per_block, p.max_seq_len = ctx // tpb, tpb, ctx
    p.act_type = p.kv_type = L.BF16
    p.multi_block_mode = 64   # 512 x 32 heads x 64 splits x 130 floats >> 8 MiB
    one = torch.zeros(16, dtype=torch.int64, device=cuda)
    p.qkv = p.out = p.sequence_lengths = p.block_pointers = one.data_ptr()
    rc = L.lib().mi355q_masked_multihead_attention_static(C.byref(p), None)
    assert rc == -1 and b"static scratch" in L.lib().mi355q_last_error()
