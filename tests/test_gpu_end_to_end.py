"""GPU: the pieces either side of the attention kernels driven the way the reference's runtime drives them --
KVCacheManager hands out blocks and the pointer tensor, GPTAttention runs the context phase and then generation
steps that cross block boundaries, a GPTQ-format checkpoint tensor feeds the weight-only GEMM.  Outputs against the
oracles (which keep their own cache), so a wrong block pointer or a block allocated too late shows up as a mismatch."""
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import checkpoint as ck
from mi355q import functional as F
from mi355q.kv_cache_manager import GenerationSequence, KVCacheManager

pytestmark = pytest.mark.gpu


def test_manager_driven_prefill_and_decode(cuda):
    dev = cuda
    layers, layer_idx = 2, 1
    Hq, Hkv, D, tpb, dtype = 8, 2, 128, 32, torch.float16
    blocks, max_blocks = 24, 4
    pools = [torch.zeros(2 * blocks * Hkv * tpb * D, dtype=dtype, device=dev) for _ in range(layers)]
    mgr = KVCacheManager(pools, blocks, tpb, max_blocks, max_attention_window_size=max_blocks * tpb, sink_token_len=0)
    lens = [30, 63]
    seqs = [GenerationSequence(100 + i, i) for i in range(2)]
    for s, n in zip(seqs, lens):
        mgr.add_sequence(s, n)
    plg = F.make_gpt_attention_plugin(num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16", layer_idx=layer_idx,
                                      position_embedding_type=2, rotary_embedding_dim=D, remove_input_padding=True,
                                      paged_kv_cache=True, tokens_per_block=tpb)
    g = torch.Generator().manual_seed(11)
    hidden = (Hq + 2 * Hkv) * D
    rows = [(torch.randn(n, hidden, generator=g) * 0.5).to(dtype) for n in lens]
    ref_cache = oracle.PagedKVCache(2, Hkv, D, max_blocks * tpb, tpb, dtype)
    B = 2
    window = torch.full((layers,), max_blocks * tpb, dtype=torch.int32)

    def call(qkv, seq_len, past, ctx, req):
        table = mgr.get_block_pointers(1).reshape(layers, B, 2, max_blocks)
        return F.gpt_attention(
            plg, qkv=qkv.to(dev), sequence_length=torch.tensor(seq_len, dtype=torch.int32, device=dev),
            host_past_key_value_lengths=torch.tensor(past, dtype=torch.int32), host_max_attention_window_sizes=window,
            host_sink_token_length=torch.zeros(1, dtype=torch.int32),
            context_lengths=torch.tensor(ctx, dtype=torch.int32, device=dev),
            cache_indirection=torch.zeros(B, 1, max_blocks * tpb, dtype=torch.int32, device=dev),
            host_request_types=torch.tensor(req, dtype=torch.int32), kv_cache_block_pointers=table.to(dev),
            host_kv_cache_block_pointers=table, host_context_lengths=torch.tensor(ctx, dtype=torch.int32)).cpu()

    out = call(torch.cat(rows, 0), lens, [0, 0], lens, [0, 0])
    refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=ref_cache, rope="neox", rot_dim=D)
    assert (out.float() - torch.cat(refs, 0).float()).abs().max().item() <= 5e-3
    cur = list(lens)
    for step in range(4):                  # 30 -> 34 crosses the 32-token block, 63 -> 67 the 64-token one
        mgr.step([False, False])           # the runtime allocates the block the new token needs BEFORE the step
        qkv = (torch.randn(B, hidden, generator=g) * 0.5).to(dtype)
        cur = [c + 1 for c in cur]
        out = call(qkv, cur, [c - 1 for c in cur], lens, [1, 1])
        ref = oracle.decode_attention(qkv, ref_cache, cur, Hq, Hkv, D, rope="neox", rot_dim=D)
        err = (out.float() - ref.float()).abs().max().item()
        assert err <= 2e-3, f"step {step}: {err}"
    assert mgr.blocks_manager.get_number_blocks(seqs[0]) == 2 and mgr.blocks_manager.get_number_blocks(seqs[1]) == 3
    # the other layer's pool was never touched
    assert pools[0].abs().max().item() == 0
    # and the cache the kernels wrote, seen through the manager's debug view, is the oracle's cache
    cont = mgr.blocks_manager.get_continuous_caches(layer_idx).cpu().reshape(B, 2, max_blocks, Hkv, tpb, D)
    for b in range(B):
        for kv in range(2):
            for h in range(Hkv):
                got = cont[b, kv, :, h].reshape(max_blocks * tpb, D)[:cur[b]]
                want = ref_cache.gather(b, kv, h, cur[b])
                assert (got.float() - want.float()).abs().max().item() <= 2e-2


def test_gptq_checkpoint_tensor_through_the_gemm(cuda):
    g = torch.Generator().manual_seed(3)
    k, n, gs, m = 512, 256, 128, 5
    q = torch.randint(0, 16, (k, n), generator=g)
    z = torch.randint(0, 15, (k // gs, n), generator=g)
    scales = (torch.rand(k // gs, n, generator=g) * 0.02 + 0.005).half()
    qw = torch.zeros(k // 8, n, dtype=torch.int32)
    for i in range(8):
        qw |= q[i::8].to(torch.int32) << (4 * i)
    qz = torch.zeros(k // gs, n // 8, dtype=torch.int32)
    for i in range(8):
        qz |= z[:, i::8].to(torch.int32) << (4 * i)
    t = ck.gptq_to_tllm(qw, qz, scales)
    x = torch.randn(m, k, generator=g).half()
    out = F.weight_only_groupwise_quant_matmul(
        x.to(cuda), None, t["weight"].view(torch.float16).to(cuda), t["weights_scaling_factor"].to(cuda), t["zero"].to(cuda), None, None,
        quant_algo=F.ZERO, group_size=gs, dtype="float16").cpu()
    w_ref = scales.float().repeat_interleave(gs, 0) * (q - (z.repeat_interleave(gs, 0) + 1)).float()
    ref = x.float() @ w_ref
    assert (out.float() - ref).abs().max().item() <= ref.abs().max().item() * 2 ** -3 * 1.5  # the reference's int4 gate
    assert (out.float() - ref).abs().max().item() <= 3e-2
