"""GPU parity: paged decode attention (GPTAttention generation phase) through the C ABI vs the
CPU oracle.  Tolerance = the reference's own generation tolerance, atol 2e-3
(tests/attention/test_gpt_attention.py:1290-1293), applied to fp16 / bf16 outputs of O(1) magnitude;
the appended K/V cache lines are compared exactly (bit-exact for 16-bit caches, exact integer codes
for int8 / fp8 caches)."""
import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

pytestmark = pytest.mark.gpu

# the reference gate is a flat 2e-3 (fp16); bf16: one bf16 ulp of the largest output (outputs are O(1): ulp(1.0) = 2^-7)
ATOL = {torch.float16: 2e-3, torch.bfloat16: 2.0 ** -7}


def build_case(B, Hq, Hkv, D, seq_lens, tpb, dtype, kv_dtype, seed=42, max_seq=None):
    torch.manual_seed(seed)
    max_seq = max_seq or (max(seq_lens) + 24)  # max_seq_len = in_len + 24 (test_gpt_attention.py:665-666)
    cache = oracle.PagedKVCache(B, Hkv, D, max_seq, tpb, kv_dtype, seed=seed)
    kv_scale = 1.0
    if kv_dtype == torch.int8:
        kv_scale = 4.0 / 127.0  # kv_dequant = absmax / 127 (test_gpt_attention.py:941-956)
    for b in range(B):
        n = seq_lens[b] - 1
        if n <= 0:
            continue
        kv = torch.randn(2, Hkv, n, D) * 0.5
        q = oracle.quantize_kv(kv.to(dtype), kv_dtype, 1.0 / kv_scale)
        nb = (n + tpb - 1) // tpb
        pad = torch.zeros(2, Hkv, nb * tpb, D, dtype=kv_dtype)
        pad[:, :, :n] = q
        blocks = pad.reshape(2, Hkv, nb, tpb, D).permute(2, 0, 1, 3, 4)
        cache.pool[cache.block_ids[b, :nb]] = blocks
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D).to(dtype)
    return cache, qkv, kv_scale


def run_plugin(cache, qkv, seq_lens, Hq, Hkv, D, tpb, dtype, kv_dtype, dev, rope=None, bias=None, kv_scale=1.0,
               remove_padding=False, num_layers=1, layer_idx=0, alibi=None, pe=None, rel_bias=None, max_distance=0,
               q_scaling=1.0):
    B = qkv.shape[0]
    qm = 0
    if kv_dtype == torch.int8:
        qm = 1 << 6
    elif kv_dtype == torch.float8_e4m3fn:
        qm = 1 << 7
    pe = {None: 0, "gptj": 1, "neox": 2}[rope] if pe is None else pe
    plg = F.make_gpt_attention_plugin(
        num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16" if dtype == torch.float16 else "bfloat16",
        layer_idx=layer_idx, position_embedding_type=pe, rotary_embedding_dim=D if rope else 0,
        kv_cache_quant_mode=qm, remove_input_padding=remove_padding, paged_kv_cache=True, tokens_per_block=tpb,
        qkv_bias_enabled=bias is not None, max_distance=max_distance, q_scaling=q_scaling)
    pool = cache.pool.to(dev)
    esz = pool.element_size()
    offs = cache.offsets()  # [B, 2, max_blocks] element offsets
    ptrs = (offs * esz + pool.data_ptr()).to(torch.int64)
    # [num_layers, B, 2, max_blocks]; other layers get poisoned pointers that must never be touched
    table = torch.zeros(num_layers, *ptrs.shape, dtype=torch.int64)
    table[layer_idx] = ptrs
    seq = torch.tensor(seq_lens, dtype=torch.int32)
    max_len = cache.max_blocks * tpb
    q_in = qkv.to(dev)
    q_in = q_in if remove_padding else q_in.reshape(B, 1, -1)
    kwargs = dict(
        qkv=q_in, sequence_length=seq.to(dev), host_past_key_value_lengths=seq - 1,
        host_max_attention_window_sizes=torch.full((num_layers,), max_len, dtype=torch.int32),
        host_sink_token_length=torch.zeros(1, dtype=torch.int32), context_lengths=(seq - 1).to(dev),
        cache_indirection=torch.zeros(B, 1, max_len, dtype=torch.int32, device=dev),
        host_request_types=torch.ones(B, dtype=torch.int32),
        kv_cache_block_pointers=table.to(dev), host_kv_cache_block_pointers=table)
    if qm:
        kwargs["kv_orig_quant_scale"] = torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev)
        kwargs["kv_quant_orig_scale"] = torch.tensor([kv_scale], dtype=torch.float32, device=dev)
    if remove_padding:
        kwargs["host_context_lengths"] = seq - 1
    if bias is not None:
        kwargs["qkv_bias"] = bias.to(dev)
    if alibi is not None:
        kwargs["alibi_slopes"] = alibi.to(dev)
    if rel_bias is not None:
        kwargs["relative_attention_bias"] = rel_bias.to(dev)
    out = F.gpt_attention(plg, **kwargs)
    torch.cuda.synchronize()
    return out.reshape(B, -1).cpu(), pool.cpu()


def check(B, Hq, Hkv, D, seq_lens, tpb, dtype, kv_dtype, dev, rope=None, with_bias=False, **kw):
    cache, qkv, kv_scale = build_case(B, Hq, Hkv, D, seq_lens, tpb, dtype, kv_dtype)
    bias = (torch.randn((Hq + 2 * Hkv) * D) * 0.1).to(dtype) if with_bias else None
    out, pool_after = run_plugin(cache, qkv, seq_lens, Hq, Hkv, D, tpb, dtype, kv_dtype, dev, rope=rope, bias=bias,
                                 kv_scale=kv_scale, **kw)
    ref = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, rope=rope, rot_dim=D if rope else 0,
                                  qkv_bias=bias, kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale)
    err = (out.float() - ref.float()).abs().max().item()
    assert err <= ATOL[dtype], f"max abs err {err}"
    # the oracle appended the new token into cache.pool; the kernel must have written the same lines
    same = pool_after.view(torch.uint8) == cache.pool.view(torch.uint8)
    if rope is None:
        assert same.all(), "appended K/V cache lines differ from the oracle"
    else:  # RoPE goes through sin/cos: allow 1 code of difference on the appended K line
        a = pool_after.float() if kv_dtype != torch.int8 else pool_after.to(torch.int32).float()
        r = cache.pool.float() if kv_dtype != torch.int8 else cache.pool.to(torch.int32).float()
        tol = 1.01 if kv_dtype == torch.int8 else (0.13 if kv_dtype == torch.float8_e4m3fn else 2e-2)
        assert (a - r).abs().max().item() <= tol
    return err


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D", [(4, 4, 64), (4, 4, 128), (8, 2, 128), (8, 1, 128), (8, 4, 64)])
def test_decode_16bit_cache(cuda, dtype, Hq, Hkv, D):
    # grids of tests/attention/test_gpt_attention.py: MHA H4 d64, GQA H8 Hkv 2|4, MQA; ragged lengths
    check(2, Hq, Hkv, D, [130, 5], 64, dtype, dtype, cuda)


@pytest.mark.parametrize("kv_dtype", [torch.int8, torch.float8_e4m3fn])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D,tpb", [(4, 4, 64, 64), (16, 2, 128, 64), (16, 2, 128, 128), (4, 4, 128, 64)])
def test_decode_8bit_cache(cuda, dtype, kv_dtype, Hq, Hkv, D, tpb):
    # 8-bit KV: gpt2 H4 d64 (:108-112); XQA grid H16 Hkv2 d128 in_len 165 (:201-243)
    check(2, Hq, Hkv, D, [166, 41], tpb, dtype, kv_dtype, cuda)


@pytest.mark.parametrize("rope", ["neox", "gptj"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_decode_rope_and_bias(cuda, dtype, rope):
    check(2, 8, 2, 128, [77, 300], 64, dtype, dtype, cuda, rope=rope, with_bias=True)
    check(2, 4, 4, 64, [77, 300], 64, dtype, torch.int8, cuda, rope=rope)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D,kv_dtype", [(8, 8, 64, None), (8, 8, 128, torch.int8), (16, 2, 128, None), (12, 4, 32, None)])
def test_decode_alibi(cuda, dtype, Hq, Hkv, D, kv_dtype):
    # kALIBI (3) / kALIBI_WITH_SCALE (4): score += slope[h] * (t - tlength) (Template.h:2026), slopes in T, any head
    # ratio (GQA 8 takes the dot-product kernel instead of the MFMA one; 12:4 at d32 the general form); the reference's
    # test grid has bloom / falcon style ALiBi in test_gpt_attention.py only at generation ("parity unpinned" vs its
    # goldens: the HF models are not loadable here)
    kv_dtype = kv_dtype or dtype
    seq_lens = [300, 17]
    for pe, scale in ((3, 1.0), (4, 1.0 / np.sqrt(D))):
        slopes = (torch.tensor([2.0 ** (-8.0 * (i + 1) / Hq) for i in range(Hq)]) * scale).to(dtype)
        cache, qkv, kv_scale = build_case(2, Hq, Hkv, D, seq_lens, 64, dtype, kv_dtype, seed=pe)
        out, pool_after = run_plugin(cache, qkv, seq_lens, Hq, Hkv, D, 64, dtype, kv_dtype, cuda, kv_scale=kv_scale,
                                     alibi=slopes, pe=pe)
        ref = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, kv_scale_orig_quant=1.0 / kv_scale,
                                      kv_scale_quant_orig=kv_scale, alibi_slopes=slopes)
        noal = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, kv_scale_orig_quant=1.0 / kv_scale,
                                       kv_scale_quant_orig=kv_scale, update_cache=False)
        assert (out.float() - ref.float()).abs().max().item() <= ATOL[dtype]
        assert (noal.float() - ref.float()).abs().max().item() > 10 * ATOL[dtype]  # the bias matters in this case
        assert torch.equal(pool_after.view(torch.uint8), cache.pool.view(torch.uint8))


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D,kv_dtype", [(8, 8, 64, None), (8, 2, 128, torch.int8), (12, 4, 32, None)])
def test_decode_relative_attention_bias(cuda, dtype, Hq, Hkv, D, kv_dtype):
    # kRELATIVE (5): T5-style bias added to the scaled score (Template.h:1797, :2026), q_scaling = 1 / sqrt(Dh) so that the
    # score is q.k itself.  Explicit tensor [Hq, S, S] (max_distance 0, entry [h][tlength][t], :1775-1790) and the
    # implicit bucket table [Hq, num_buckets] with max_distance 128 (:1960-1977).  The reference's attention test has no
    # relative-bias case for this plugin ("parity unpinned": the oracle restates the cited lines); split-KV lengths.
    kv_dtype = kv_dtype or dtype
    seq_lens = [300, 17, 1200]
    g = torch.Generator().manual_seed(9)
    for max_distance, shape in ((0, (Hq, 1224, 1224)), (128, (Hq, 32))):
        rel = (torch.randn(shape, generator=g) * 0.7).to(dtype)
        cache, qkv, kv_scale = build_case(3, Hq, Hkv, D, seq_lens, 64, dtype, kv_dtype, seed=max_distance + 1)
        qkv = (qkv.float() * 0.3).to(dtype)     # un-scaled scores: keep them O(1)
        out, pool_after = run_plugin(cache, qkv, seq_lens, Hq, Hkv, D, 64, dtype, kv_dtype, cuda, kv_scale=kv_scale,
                                     rel_bias=rel, pe=5, max_distance=max_distance, q_scaling=1.0 / np.sqrt(D))
        ref = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, q_scaling=1.0 / np.sqrt(D),
                                      kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale,
                                      relative_attention_bias=rel, max_distance=max_distance)
        plain = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, q_scaling=1.0 / np.sqrt(D),
                                        kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale, update_cache=False)
        assert (out.float() - ref.float()).abs().max().item() <= ATOL[dtype]
        assert (plain.float() - ref.float()).abs().max().item() > 10 * ATOL[dtype]  # the bias matters in this case
        assert torch.equal(pool_after.view(torch.uint8), cache.pool.view(torch.uint8))


@pytest.mark.parametrize("D", [48, 80, 96, 104, 112, 144, 160, 192, 224])
def test_decode_other_head_sizes(cuda, D):
    # the head sizes decoderMaskedMultiheadAttention.cu dispatches on besides 32 / 64 / 128 / 256: run padded to the next
    # power of two in the general kernel (cache rows and the qkv / output rows keep their real size)
    check(2, 4, 2, D, [150, 9], 64, torch.float16, torch.float16, cuda)
    check(2, 6, 6, D, [70, 131], 16, torch.bfloat16, torch.bfloat16, cuda, rope="neox" if D % 16 == 0 else "gptj", with_bias=True)
    if D % 16 == 0:  # an 8-bit cache row is read 16 bytes per lane
        check(2, 8, 1, D, [300, 33], 32, torch.float16, torch.int8, cuda)
        check(1, 4, 4, D, [2100], 64, torch.bfloat16, torch.float8_e4m3fn, cuda)  # split-KV merge of padded records


def test_decode_edge_lengths(cuda):
    # first generated token (no cached keys), exact block boundaries, one past a boundary
    for sl in ([1, 1], [64, 65], [128, 129], [2, 257]):
        check(2, 4, 4, 128, sl, 64, torch.float16, torch.float16, cuda)
        check(2, 8, 1, 128, sl, 64, torch.bfloat16, torch.int8, cuda)


def test_decode_long_multi_block(cuda):
    # multi-block grids (:95-106): B2 in2048 H4 d64; B1 in4096 H1 d128 -- few (b, kv-head) pairs force kv splits
    check(2, 4, 4, 64, [2049, 1500], 64, torch.float16, torch.float16, cuda)
    check(1, 8, 1, 128, [4097], 128, torch.bfloat16, torch.int8, cuda)
    check(1, 1, 1, 128, [4097], 64, torch.float16, torch.float16, cuda)


def test_decode_cfg5_instantiation(cuda):
    # the exact kernel BASELINE cfg5 times: bf16 activations, fp8 e4m3 cache, 4 query heads per kv head (Mixtral 32q / 8kv),
    # split-KV lengths; and the same group with an int8 cache
    check(2, 32, 8, 128, [2049, 700], 64, torch.bfloat16, torch.float8_e4m3fn, cuda)
    check(2, 32, 8, 128, [2049, 700], 64, torch.bfloat16, torch.float8_e4m3fn, cuda, rope="neox")
    check(2, 32, 8, 128, [1030, 65], 64, torch.bfloat16, torch.int8, cuda)


def test_decode_remove_padding_and_layer_offset(cuda):
    # packed input layout + block-pointer slicing by layer_idx (gptAttentionPlugin.cpp:506-519)
    check(3, 4, 2, 128, [9, 200, 70], 64, torch.float16, torch.float16, cuda, remove_padding=True, num_layers=3,
          layer_idx=2)


def test_decode_llama7b_shape_property(cuda):
    # BASELINE cfg2 at full size through a size-independent property: with all V rows of a head equal
    # to a constant vector c, the output must be c * sum(p)/(sum(p)+1e-6) ~= c whatever K and q are
    # (softmax weights sum to one), and the cache append must land at position seq-1.
    B, H, D, ctx, tpb = 64, 32, 128, 2048, 64
    dev = cuda
    g = torch.Generator(device="cpu").manual_seed(42)
    nblk = B * (ctx // tpb)
    pool = torch.empty(nblk, 2, H, tpb, D, dtype=torch.bfloat16, device=dev)
    pool[:, 0] = (torch.randn(nblk, H, tpb, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    cvec = torch.randn(H, D, generator=g).to(torch.bfloat16)
    pool[:, 1] = cvec.to(dev)[None, :, None, :]
    perm = torch.randperm(nblk, generator=g).reshape(B, ctx // tpb)
    be = H * tpb * D * 2
    ptrs = torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + pool.data_ptr()
    qkv = torch.randn(B, 3 * H * D, generator=g).to(torch.bfloat16)
    qkv[:, 2 * H * D:] = cvec.reshape(1, -1)  # the new token's v is c as well
    seq = torch.full((B,), ctx, dtype=torch.int32)
    plg = F.make_gpt_attention_plugin(num_heads=H, num_kv_heads=H, head_size=D, dtype="bfloat16",
                                      position_embedding_type=2, rotary_embedding_dim=D, tokens_per_block=tpb)
    out = F.gpt_attention(
        plg, qkv=qkv.to(dev).reshape(B, 1, -1), sequence_length=seq.to(dev), host_past_key_value_lengths=seq - 1,
        host_max_attention_window_sizes=torch.tensor([ctx], dtype=torch.int32),
        host_sink_token_length=torch.zeros(1, dtype=torch.int32), context_lengths=(seq - 1).to(dev),
        cache_indirection=torch.zeros(B, 1, ctx, dtype=torch.int32, device=dev),
        host_request_types=torch.ones(B, dtype=torch.int32),
        kv_cache_block_pointers=ptrs.reshape(1, B, 2, -1).to(dev),
        host_kv_cache_block_pointers=ptrs.reshape(1, B, 2, -1))
    torch.cuda.synchronize()
    got = out.reshape(B, H, D).float().cpu()
    assert (got - cvec.float()[None]).abs().max().item() <= 2e-2  # one bf16 ulp at |c| <= 4
    last = pool[perm[:, -1].to(dev), 1, :, tpb - 1].cpu()  # V line of token ctx-1 of every sequence
    assert torch.equal(last, cvec[None].expand(B, H, D))


@pytest.mark.parametrize("Hq,Hkv,kv_dtype,label", [(8, 1, torch.int8, "cfg4: Llama-2-70B 64q/8kv int8 KV, the TP = 8 per-GPU slice"),
                                                   (64, 8, torch.int8, "cfg4 without the split: 64q/8kv int8 KV"),
                                                   (32, 8, torch.float8_e4m3fn, "cfg5: Mixtral 32q/8kv fp8 KV")])
def test_decode_gqa_8bit_full_size_property(cuda, Hq, Hkv, kv_dtype, label):
    # BASELINE cfg4 / cfg5 at full size (batch 64, ctx 2048, d = 128, the MFMA GQA kernel) through the same
    # size-independent property as cfg2 above: every cached V line of a kv head holds quant(c), so whatever K and q are the
    # output is a convex mix of dequant(quant(c)) (cached tokens) and c (the new token, still un-quantised in registers);
    # the appended K / V lines of the new token must equal the oracle's quantisation of them, bit for bit.
    B, D, ctx, tpb, dt = 64, 128, 2048, 64, torch.bfloat16
    dev = cuda
    g = torch.Generator(device="cpu").manual_seed(7)
    kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else 0.5       # dequantisation scale (KV_CACHE_DEQUANTIZATION_SCALE)
    nblk = B * (ctx // tpb)
    cvec = torch.randn(Hkv, D, generator=g).to(dt)
    c8 = oracle.quantize_kv(cvec, kv_dtype, 1.0 / kv_scale)           # [Hkv, D]
    cq = c8.float() * kv_scale
    pool = torch.empty(nblk, 2, Hkv, tpb, D, dtype=kv_dtype, device=dev)
    k8 = oracle.quantize_kv(torch.randn(nblk, Hkv, tpb, D, generator=g) * 0.5, kv_dtype, 1.0 / kv_scale)
    pool[:, 0] = k8.to(dev)
    pool[:, 1] = c8.to(dev)[None, :, None, :]
    perm = torch.randperm(nblk, generator=g).reshape(B, ctx // tpb)
    be = Hkv * tpb * D                                                # one byte per element
    ptrs = torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + pool.data_ptr()
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, generator=g).to(dt)
    qkv[:, (Hq + Hkv) * D:] = cvec.reshape(1, -1)
    seq = torch.full((B,), ctx, dtype=torch.int32)
    plg = F.make_gpt_attention_plugin(num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="bfloat16", tokens_per_block=tpb,
                                      kv_cache_quant_mode=(1 << 6) if kv_dtype == torch.int8 else (1 << 7))
    out = F.gpt_attention(
        plg, qkv=qkv.to(dev).reshape(B, 1, -1), sequence_length=seq.to(dev), host_past_key_value_lengths=seq - 1,
        host_max_attention_window_sizes=torch.tensor([ctx], dtype=torch.int32),
        host_sink_token_length=torch.zeros(1, dtype=torch.int32), context_lengths=(seq - 1).to(dev),
        cache_indirection=torch.zeros(B, 1, ctx, dtype=torch.int32, device=dev),
        host_request_types=torch.ones(B, dtype=torch.int32),
        kv_cache_block_pointers=ptrs.reshape(1, B, 2, -1).to(dev),
        host_kv_cache_block_pointers=ptrs.reshape(1, B, 2, -1),
        kv_orig_quant_scale=torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev),
        kv_quant_orig_scale=torch.tensor([kv_scale], dtype=torch.float32, device=dev))
    torch.cuda.synchronize()
    got = out.reshape(B, Hq, D).float().cpu()
    grp = Hq // Hkv
    ref = cq.repeat_interleave(grp, 0)[None]                          # [1, Hq, D]
    slack = (cvec.float() - cq).abs().max().item()                    # the new token's share pulls towards c
    assert (got - ref).abs().max().item() <= slack + 3e-2, label
    # the new token's lines at position ctx - 1
    last = perm[:, -1].to(dev)
    v_line = pool[last, 1, :, tpb - 1].cpu()
    assert torch.equal(v_line.view(torch.uint8), c8[None].expand(B, Hkv, D).contiguous().view(torch.uint8)), label
    k_new = qkv[:, Hq * D:(Hq + Hkv) * D].reshape(B, Hkv, D)
    k_line = pool[last, 0, :, tpb - 1].cpu()
    assert torch.equal(k_line.view(torch.uint8), oracle.quantize_kv(k_new, kv_dtype, 1.0 / kv_scale).view(torch.uint8)), label
