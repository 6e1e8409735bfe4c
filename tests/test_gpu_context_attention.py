"""GPU parity: the context (prefill) phase of GPTAttention through the plugin C ABI vs the CPU oracle
(oracle.context_attention).  Tolerance = the reference's own context tolerance, atol 5e-3 on fp16
(tests/attention/test_gpt_attention.py:1134-1145), x4 for bf16's 3 fewer mantissa bits; the K / V lines
appended to the paged cache are compared exactly without RoPE (bit-exact for 16-bit caches, exact codes for
int8 / fp8) and within one code with RoPE (sin / cos go through different libm's)."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

pytestmark = pytest.mark.gpu

ATOL = {torch.float16: 5e-3, torch.bfloat16: 2e-2}
QM = {torch.int8: 1 << 6, torch.float8_e4m3fn: 1 << 7}


def make_inputs(lens, Hq, Hkv, D, dtype, seed=42):
    g = torch.Generator().manual_seed(seed)
    return [(torch.randn(n, (Hq + 2 * Hkv) * D, generator=g) * 0.5).to(dtype) for n in lens]


def make_plugin(Hq, Hkv, D, dtype, kv_dtype, tpb, rope, bias, remove_padding, layer_idx=0, max_context_length=0,
                rot_dim=None):
    pe = {None: 0, "gptj": 1, "neox": 2}[rope]
    return F.make_gpt_attention_plugin(
        num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16" if dtype == torch.float16 else "bfloat16",
        layer_idx=layer_idx, position_embedding_type=pe, rotary_embedding_dim=(rot_dim or D) if rope else 0,
        kv_cache_quant_mode=QM.get(kv_dtype, 0), remove_input_padding=remove_padding, paged_kv_cache=True,
        tokens_per_block=tpb, qkv_bias_enabled=bias is not None, max_context_length=max_context_length)


def block_table(cache, pool, num_layers, layer_idx):
    ptrs = (cache.offsets() * pool.element_size() + pool.data_ptr()).to(torch.int64)
    table = torch.zeros(num_layers, *ptrs.shape, dtype=torch.int64)  # other layers: null pointers, never touched
    table[layer_idx] = ptrs
    return table


def run_context(plg, rows, cache, pool, dev, kv_scale, remove_padding, bias=None, num_layers=1, layer_idx=0,
                gen_rows=None, gen_seq_lens=None, **extra):
    """Enqueue the plugin with len(rows) context requests (+ optional generation requests after them)."""
    lens = [r.shape[0] for r in rows]
    nctx, ngen = len(rows), 0 if gen_rows is None else gen_rows.shape[0]
    B = nctx + ngen
    hidden = (rows[0] if rows else gen_rows).shape[1]
    if remove_padding:
        parts = list(rows) + ([gen_rows] if ngen else [])
        qkv = torch.cat(parts, 0).to(dev)
    else:
        S = max(lens)
        qkv = torch.full((B, S, hidden), float("nan"), dtype=rows[0].dtype)  # padding rows must never be read into results
        for b, r in enumerate(rows):
            qkv[b, :lens[b]] = r
        qkv = qkv.to(dev)
    ctx = torch.tensor(lens + ([0] * ngen if gen_seq_lens is None else [s - 1 for s in gen_seq_lens]), dtype=torch.int32)
    seq = torch.tensor(lens + (list(gen_seq_lens) if ngen else []), dtype=torch.int32)
    past = torch.tensor([0] * nctx + ([s - 1 for s in gen_seq_lens] if ngen else []), dtype=torch.int32)
    max_len = cache.max_blocks * cache.tpb
    table = block_table(cache, pool, num_layers, layer_idx)
    kwargs = dict(
        qkv=qkv, sequence_length=seq.to(dev), host_past_key_value_lengths=past,
        host_max_attention_window_sizes=torch.full((num_layers,), max_len, dtype=torch.int32),
        host_sink_token_length=torch.zeros(1, dtype=torch.int32), context_lengths=ctx.to(dev),
        cache_indirection=torch.zeros(B, 1, max_len, dtype=torch.int32, device=dev),
        host_request_types=torch.tensor([0] * nctx + [1] * ngen, dtype=torch.int32),
        kv_cache_block_pointers=table.to(dev), host_kv_cache_block_pointers=table)
    if cache.dtype in QM:
        kwargs["kv_orig_quant_scale"] = torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev)
        kwargs["kv_quant_orig_scale"] = torch.tensor([kv_scale], dtype=torch.float32, device=dev)
    if remove_padding:
        kwargs["host_context_lengths"] = ctx
    if bias is not None:
        kwargs["qkv_bias"] = bias.to(dev)
    kwargs.update(extra)
    out = F.gpt_attention(plg, **kwargs)
    torch.cuda.synchronize()
    out = out.cpu()
    if remove_padding:
        offs = [0] + [int(v) for v in np.cumsum(lens)]
        outs = [out[offs[b]:offs[b + 1]] for b in range(nctx)]
        gen_out = out[offs[-1]:] if ngen else None
    else:
        outs = [out[b, :lens[b]] for b in range(nctx)]
        gen_out = None
    return outs, gen_out


def compare_cache(pool_after, cache, rope):
    kv_dtype = cache.dtype
    if rope is None:
        assert torch.equal(pool_after.view(torch.uint8), cache.pool.view(torch.uint8)), "appended K/V lines differ"
        return
    a = pool_after.to(torch.int32).float() if kv_dtype == torch.int8 else pool_after.float()
    r = cache.pool.to(torch.int32).float() if kv_dtype == torch.int8 else cache.pool.float()
    tol = 1.01 if kv_dtype == torch.int8 else (0.26 if kv_dtype == torch.float8_e4m3fn else 2e-2)
    assert (a - r).abs().max().item() <= tol
    # V never goes through RoPE: exact
    assert torch.equal(pool_after[:, 1].contiguous().view(torch.uint8), cache.pool[:, 1].contiguous().view(torch.uint8))


def check(lens, Hq, Hkv, D, tpb, dtype, kv_dtype, dev, rope=None, with_bias=False, remove_padding=False,
          num_layers=1, layer_idx=0, seed=42, rot_dim=None):
    rows = make_inputs(lens, Hq, Hkv, D, dtype, seed)
    bias = (torch.randn((Hq + 2 * Hkv) * D) * 0.1).to(dtype) if with_bias else None
    kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else 1.0
    cache = oracle.PagedKVCache(len(lens), Hkv, D, max(lens) + 24, tpb, kv_dtype, seed=seed)
    pool = cache.pool.to(dev)
    plg = make_plugin(Hq, Hkv, D, dtype, kv_dtype, tpb, rope, bias, remove_padding, layer_idx, rot_dim=rot_dim)
    outs, _ = run_context(plg, rows, cache, pool, dev, kv_scale, remove_padding, bias, num_layers, layer_idx)
    refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, rope=rope, rot_dim=(rot_dim or D) if rope else 0,
                                    qkv_bias=bias, kv_scale_orig_quant=1.0 / kv_scale)
    for b, (o, r) in enumerate(zip(outs, refs)):
        assert o.shape == r.shape
        err = (o.float() - r.float()).abs().max().item()
        assert err <= ATOL[dtype], f"sequence {b}: max abs err {err}"
    compare_cache(pool.cpu(), cache, rope)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D", [(4, 4, 64), (4, 4, 128), (8, 2, 128), (8, 1, 128), (8, 4, 64)])
def test_context_16bit_cache(cuda, dtype, Hq, Hkv, D):
    # grids of tests/attention/test_gpt_attention.py: MHA H4 d64 / d128, GQA, MQA; ragged lengths, padded input
    check([130, 37], Hq, Hkv, D, 64, dtype, dtype, cuda)


@pytest.mark.parametrize("kv_dtype", [torch.int8, torch.float8_e4m3fn])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_context_8bit_cache(cuda, dtype, kv_dtype):
    check([166, 41], 4, 4, 64, 64, dtype, kv_dtype, cuda)
    check([166, 41], 16, 2, 128, 128, dtype, kv_dtype, cuda)


@pytest.mark.parametrize("rope", ["neox", "gptj"])
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_context_rope_and_bias(cuda, dtype, rope):
    check([77, 300], 8, 2, 128, 64, dtype, dtype, cuda, rope=rope, with_bias=True)
    check([77, 300], 4, 4, 64, 64, dtype, torch.int8, cuda, rope=rope)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D,kv_dtype", [(4, 4, 64, None), (8, 2, 128, torch.int8), (6, 2, 32, None)])
def test_context_relative_attention_bias(cuda, dtype, Hq, Hkv, D, kv_dtype):
    # kRELATIVE (5) in the context phase: the bias is added to the scaled scores before the masked softmax
    # (gptAttentionCommon.cpp:816-817, 1006-1016; addRelativeAttentionBiasUnaligned, unfusedAttentionKernels.cu:1688-1737,
    # bidirectional = false).  Explicit tensor [Hq, S', S'] (max_distance 0) and the implicit bucket table [Hq, num_buckets]
    # (max_distance 128); then ONE generation step of every sequence with the same bias input, over the cache this call
    # appended.  The reference's attention test has no relative-bias case for this plugin: "parity unpinned", the oracle
    # restates the cited lines.
    kv_dtype = kv_dtype or dtype
    lens, tpb, dev = [150, 9, 333], 64, cuda
    kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else 1.0
    qs = 1.0 / np.sqrt(D)  # score = q.k itself
    g = torch.Generator().manual_seed(5)
    for max_distance, shape in ((0, (Hq, 384, 384)), (128, (Hq, 32))):
        rel = (torch.randn(shape, generator=g) * 0.7).to(dtype)
        rows = [(r.float() * 0.3).to(dtype) for r in make_inputs(lens, Hq, Hkv, D, dtype, seed=max_distance + 3)]
        cache = oracle.PagedKVCache(len(lens), Hkv, D, max(lens) + 24, tpb, kv_dtype, seed=1)
        pool = cache.pool.to(dev)
        plg = F.make_gpt_attention_plugin(
            num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16" if dtype == torch.float16 else "bfloat16",
            position_embedding_type=5, kv_cache_quant_mode=QM.get(kv_dtype, 0), paged_kv_cache=True, tokens_per_block=tpb,
            max_distance=max_distance, q_scaling=qs, remove_input_padding=True)
        outs, _ = run_context(plg, rows, cache, pool, dev, kv_scale, True, relative_attention_bias=rel.to(dev))
        refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, q_scaling=qs, kv_scale_orig_quant=1.0 / kv_scale,
                                        relative_attention_bias=rel, max_distance=max_distance)
        plain = oracle.context_attention(rows, lens, Hq, Hkv, D, q_scaling=qs)
        for b, (o, r) in enumerate(zip(outs, refs)):
            err = (o.float() - r.float()).abs().max().item()
            assert err <= ATOL[dtype], f"max_distance {max_distance}, sequence {b}: max abs err {err}"
        assert max((p.float() - r.float()).abs().max().item() for p, r in zip(plain, refs)) > 5 * ATOL[dtype]  # the bias matters here
        compare_cache(pool.cpu(), cache, None)
        # the generation step that follows
        gen = (torch.randn(len(lens), (Hq + 2 * Hkv) * D, generator=g) * 0.15).to(dtype)
        seq = [n + 1 for n in lens]
        _, gout = run_context(plg, [], cache, pool, dev, kv_scale, True, gen_rows=gen, gen_seq_lens=seq,
                              relative_attention_bias=rel.to(dev))
        gref = oracle.decode_attention(gen, cache, seq, Hq, Hkv, D, q_scaling=qs, kv_scale_orig_quant=1.0 / kv_scale,
                                       kv_scale_quant_orig=kv_scale, relative_attention_bias=rel, max_distance=max_distance)
        assert (gout.float() - gref.float()).abs().max().item() <= ATOL[dtype]


def test_context_partial_rotary(cuda):
    # rotary_embedding_dim < head_size (GPT-J / GPT-NeoX style rotary_pct): NeoX takes the generic prepare kernel,
    # GPT-J the vectorised one with pass-through elements
    for rope in ("neox", "gptj"):
        check([70, 33], 4, 2, 128, 64, torch.float16, torch.float16, cuda, rope=rope, rot_dim=64, with_bias=True)
        check([70, 33], 4, 4, 64, 64, torch.bfloat16, torch.int8, cuda, rope=rope, rot_dim=32)


def test_context_edge_lengths(cuda):
    # one token, step / wave-tile / workgroup-tile boundaries and one either side, cache block boundaries
    for lens in ([1, 2], [31, 32], [33, 64], [127, 128], [129, 65], [256, 257]):
        check(lens, 4, 4, 128, 64, torch.float16, torch.float16, cuda)
        check(lens, 8, 1, 64, 32, torch.bfloat16, torch.int8, cuda)


def test_context_large_logits(cuda):
    # scores of +-30: the running maximum must be the column's true maximum (the 4 lanes that share a query column
    # all-reduce it), else the fp16 probabilities overflow.  Regression test for the swap-based all-reduce.
    for dtype in (torch.float16, torch.bfloat16):
        for D in (64, 128):
            lens = [150, 64]
            g = torch.Generator().manual_seed(5)
            rows = [(torch.randn(n, 3 * 4 * D, generator=g) * 3.0).to(dtype) for n in lens]
            cache = oracle.PagedKVCache(2, 4, D, 200, 64, dtype)
            pool = cache.pool.to(cuda)
            plg = make_plugin(4, 4, D, dtype, dtype, 64, None, None, True)
            outs, _ = run_context(plg, rows, cache, pool, cuda, 1.0, True)
            refs = oracle.context_attention(rows, lens, 4, 4, D, cache=cache)
            for o, r in zip(outs, refs):
                assert torch.isfinite(o.float()).all()
                assert (o.float() - r.float()).abs().max().item() <= ATOL[dtype] * 8  # outputs of magnitude ~8


@pytest.mark.parametrize("D", [32, 48, 80, 96, 112, 160, 192, 256])
def test_context_other_head_sizes(cuda, D):
    # head sizes besides 64 / 128: zero-padded to the next of the two in the workspace (the cache and the output keep the
    # real size); RoPE (generic prepare kernel), bias, 16-bit and 8-bit caches
    check([130, 37], 4, 2, D, 64, torch.float16, torch.float16, cuda, rope="neox", with_bias=True)
    check([70, 200], 6, 6, D, 16, torch.bfloat16, torch.bfloat16, cuda, remove_padding=True)
    if D % 16 == 0:
        check([166, 41], 8, 1, D, 32, torch.float16, torch.int8, cuda, rope="gptj")


def test_context_remove_padding_and_layer_offset(cuda):
    check([9, 200, 70], 4, 2, 128, 64, torch.float16, torch.float16, cuda, remove_padding=True, num_layers=3, layer_idx=2)
    check([5, 1, 131], 8, 8, 64, 64, torch.bfloat16, torch.float8_e4m3fn, cuda, remove_padding=True, rope="neox")


def test_context_then_generation_step(cuda):
    # prefill, then one generation step over the cache the prefill wrote: both phases against their oracles
    Hq, Hkv, D, tpb, dtype, dev = 8, 2, 128, 64, torch.float16, cuda
    for kv_dtype in (dtype, torch.int8):
        lens = [70, 129]
        kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else 1.0
        rows = make_inputs(lens, Hq, Hkv, D, dtype)
        cache = oracle.PagedKVCache(2, Hkv, D, max(lens) + 24, tpb, kv_dtype)
        pool = cache.pool.to(dev)
        plg = make_plugin(Hq, Hkv, D, dtype, kv_dtype, tpb, "neox", None, True)
        run_context(plg, rows, cache, pool, dev, kv_scale, True)
        cache.pool = pool.cpu().clone()  # continue the oracle from the cache the GPU wrote
        step = (torch.randn(2, (Hq + 2 * Hkv) * D) * 0.5).to(dtype)
        seq = [n + 1 for n in lens]
        # generation-only enqueue (no context rows)
        B = 2
        table = block_table(cache, pool, 1, 0)
        max_len = cache.max_blocks * tpb
        kwargs = dict(
            qkv=step.to(dev), sequence_length=torch.tensor(seq, dtype=torch.int32, device=dev),
            host_past_key_value_lengths=torch.tensor(lens, dtype=torch.int32),
            host_max_attention_window_sizes=torch.tensor([max_len], dtype=torch.int32),
            host_sink_token_length=torch.zeros(1, dtype=torch.int32),
            context_lengths=torch.tensor(lens, dtype=torch.int32, device=dev),
            cache_indirection=torch.zeros(B, 1, max_len, dtype=torch.int32, device=dev),
            host_request_types=torch.ones(B, dtype=torch.int32),
            kv_cache_block_pointers=table.to(dev), host_kv_cache_block_pointers=table,
            host_context_lengths=torch.tensor(lens, dtype=torch.int32))
        if kv_dtype in QM:
            kwargs["kv_orig_quant_scale"] = torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev)
            kwargs["kv_quant_orig_scale"] = torch.tensor([kv_scale], dtype=torch.float32, device=dev)
        out = F.gpt_attention(plg, **kwargs).cpu()
        ref = oracle.decode_attention(step, cache, seq, Hq, Hkv, D, rope="neox", rot_dim=D,
                                      kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale)
        assert (out.float() - ref.float()).abs().max().item() <= 2e-3
        # and the prefill's last row equals what a generation step over the first len-1 tokens would give
        # (same math, different kernel): the causal row t only sees keys <= t


def test_mixed_context_and_generation_batch(cuda):
    # in-flight batching (gptAttentionPlugin.cpp:363-390): context requests first, then generation requests,
    # packed input; every request against its own oracle
    Hq, Hkv, D, tpb, dtype, dev = 4, 2, 128, 64, torch.bfloat16, cuda
    ctx_lens, gen_seq = [50, 140], [90, 7]
    B = 4
    cache = oracle.PagedKVCache(B, Hkv, D, 200, tpb, dtype)
    g = torch.Generator().manual_seed(1)
    # pre-fill the generation requests' history (sequences 2, 3)
    for i, s in enumerate(gen_seq):
        b = 2 + i
        for kv in range(2):
            for h in range(Hkv):
                for t in range(s - 1):
                    cache.write(b, kv, h, t, (torch.randn(D, generator=g) * 0.5).to(dtype))
    pool = cache.pool.to(dev)
    rows = make_inputs(ctx_lens, Hq, Hkv, D, dtype)
    gen_rows = (torch.randn(2, (Hq + 2 * Hkv) * D, generator=g) * 0.5).to(dtype)
    plg = make_plugin(Hq, Hkv, D, dtype, dtype, tpb, None, None, True)
    outs, gen_out = run_context(plg, rows, cache, pool, dev, 1.0, True, gen_rows=gen_rows, gen_seq_lens=gen_seq)
    refs = oracle.context_attention(rows, ctx_lens, Hq, Hkv, D, cache=cache)
    for o, r in zip(outs, refs):
        assert (o.float() - r.float()).abs().max().item() <= ATOL[dtype]

    class View:  # sequences 2, 3 of the cache as a 2-sequence cache for the decode oracle
        dtype = cache.dtype

        def gather(self, b, kv, head, length):
            return cache.gather(b + 2, kv, head, length)

        def write(self, b, kv, head, tok, vec):
            cache.write(b + 2, kv, head, tok, vec)

    ref_gen = oracle.decode_attention(gen_rows, View(), gen_seq, Hq, Hkv, D)
    assert (gen_out.float() - ref_gen.float()).abs().max().item() <= 2e-3 * 8
    assert torch.equal(pool.cpu().view(torch.uint8), cache.pool.view(torch.uint8))


def test_context_c_abi_without_cache_and_errors(cuda):
    # mi355q_context_attention directly, block_pointers NULL (use_cache = false: attention only)
    Hq, Hkv, D, dtype, dev = 4, 4, 64, torch.float16, cuda
    lens = [40, 97]
    rows = make_inputs(lens, Hq, Hkv, D, dtype)
    qkv = torch.cat(rows, 0).to(dev)
    out = torch.zeros(sum(lens), Hq * D, dtype=dtype, device=dev)
    cl = torch.tensor(lens, dtype=torch.int32, device=dev)
    p = L.ContextAttentionParams()
    p.qkv, p.out, p.context_lengths = qkv.data_ptr(), out.data_ptr(), cl.data_ptr()
    p.batch, p.num_heads, p.num_kv_heads, p.head_size = 2, Hq, Hkv, D
    p.max_context_length, p.num_tokens, p.remove_padding = max(lens), sum(lens), 1
    p.act_type = p.kv_type = L.HALF
    p.q_scaling = 1.0
    need = L.lib().mi355q_context_attention_workspace_size(C.byref(p))
    assert need >= sum(lens) * 3 * Hq * D * 2
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need
    L.check(L.lib().mi355q_context_attention(C.byref(p), None))
    torch.cuda.synchronize()
    refs = torch.cat(oracle.context_attention(rows, lens, Hq, Hkv, D), 0)
    assert (out.cpu().float() - refs.float()).abs().max().item() <= ATOL[dtype]
    # error behaviour: short workspace, unsupported head size
    p.workspace_bytes = need - 1
    assert L.lib().mi355q_context_attention(C.byref(p), None) != 0
    assert b"workspace" in L.lib().mi355q_last_error()
    p.workspace_bytes = need
    p.head_size = 260
    assert L.lib().mi355q_context_attention(C.byref(p), None) != 0


def test_context_llama7b_shape_properties(cuda):
    # BASELINE-sized prefill (llama-7B heads, 2048 tokens) through size-independent properties:
    #  * constant V rows -> every output row equals that constant (softmax weights sum to one);
    #  * causality: the first 1000 output rows do not change when the sequence is cut to 1000 tokens.
    H, D, S, tpb, dev = 32, 128, 2048, 64, cuda
    g = torch.Generator().manual_seed(7)
    qkv = (torch.randn(1, S, 3 * H * D, generator=g) * 0.5).to(torch.bfloat16)
    cvec = torch.randn(H * D, generator=g).to(torch.bfloat16)
    qkv_c = qkv.clone()
    qkv_c[:, :, 2 * H * D:] = cvec
    out = torch.empty(1, S, H * D, dtype=torch.bfloat16, device=dev)
    ws = None

    def run(x, n):
        nonlocal ws
        xd = x.to(dev)
        cl = torch.tensor([n], dtype=torch.int32, device=dev)
        p = L.ContextAttentionParams()
        p.qkv, p.out, p.context_lengths = xd.data_ptr(), out.data_ptr(), cl.data_ptr()
        p.batch, p.num_heads, p.num_kv_heads, p.head_size = 1, H, H, D
        p.max_context_length, p.num_tokens, p.remove_padding = S, S, 0
        p.act_type = p.kv_type = L.BF16
        p.position_embedding_type, p.rotary_embedding_dim, p.rotary_embedding_base = 2, D, 10000.0
        p.rotary_embedding_scale, p.rotary_embedding_max_positions = 1.0, 4096
        p.q_scaling = 1.0
        need = L.lib().mi355q_context_attention_workspace_size(C.byref(p))
        if ws is None:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = ws.data_ptr(), need
        L.check(L.lib().mi355q_context_attention(C.byref(p), None))
        torch.cuda.synchronize()
        return out.cpu().clone()

    got = run(qkv_c, S)
    assert (got[0].float() - cvec.float()[None]).abs().max().item() <= 2e-2
    full = run(qkv, S)
    cut = run(qkv, 1000)
    assert torch.equal(full[0, :1000], cut[0, :1000])
