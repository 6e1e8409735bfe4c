"""GPU parity: the context phase beyond the plain causal prefill -- chunked context over a non-empty cache
(use_paged_context_fmha, gptAttentionCommon.cpp:825-864), the sliding-window mask + cyclic append (gptKernels.cu:174-177,
unfusedAttentionKernels_2_template.h:336-340) and ALiBi (unfusedAttentionKernels.cu:212-218), through the plugin and the
C ABI, against oracle.context_attention.  The reference's own tests cover none of the three in the context phase
("parity unpinned": the oracle restates the cited lines); tolerances as tests/test_gpu_context_attention.py."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F
from test_gpu_context_attention import ATOL, QM, block_table, compare_cache, make_inputs

pytestmark = pytest.mark.gpu


def plugin(Hq, Hkv, D, dtype, kv_dtype, tpb, rope=None, pe=None, paged_ctx=False):
    pe = {None: 0, "gptj": 1, "neox": 2}[rope] if pe is None else pe
    return F.make_gpt_attention_plugin(
        num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16" if dtype == torch.float16 else "bfloat16",
        position_embedding_type=pe, rotary_embedding_dim=D if rope else 0, kv_cache_quant_mode=QM.get(kv_dtype, 0),
        remove_input_padding=True, paged_kv_cache=True, tokens_per_block=tpb, use_paged_context_fmha=paged_ctx)


def enqueue(plg, rows, kv_lens, cache, pool, dev, window=None, sink=0, kv_scale=1.0, alibi=None):
    lens = [r.shape[0] for r in rows]
    B = len(rows)
    max_len = cache.max_blocks * cache.tpb
    table = block_table(cache, pool, 1, 0)
    kwargs = dict(
        qkv=torch.cat(rows, 0).to(dev), sequence_length=torch.tensor(kv_lens, dtype=torch.int32, device=dev),
        host_past_key_value_lengths=torch.tensor(kv_lens, dtype=torch.int32),
        host_max_attention_window_sizes=torch.tensor([window or max_len], dtype=torch.int32),
        host_sink_token_length=torch.tensor([sink], dtype=torch.int32),
        context_lengths=torch.tensor(lens, dtype=torch.int32, device=dev),
        cache_indirection=torch.zeros(B, 1, max_len, dtype=torch.int32, device=dev),
        host_request_types=torch.zeros(B, dtype=torch.int32),
        kv_cache_block_pointers=table.to(dev), host_kv_cache_block_pointers=table,
        host_context_lengths=torch.tensor(lens, dtype=torch.int32))
    if cache.dtype in QM:
        kwargs["kv_orig_quant_scale"] = torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev)
        kwargs["kv_quant_orig_scale"] = torch.tensor([kv_scale], dtype=torch.float32, device=dev)
    if alibi is not None:
        kwargs["alibi_slopes"] = alibi.to(dev)
    out = F.gpt_attention(plg, **kwargs)
    torch.cuda.synchronize()
    out = out.cpu()
    offs = np.concatenate([[0], np.cumsum(lens)])
    return [out[offs[b]:offs[b + 1]] for b in range(B)]


def assert_close(outs, refs, dtype, what=""):
    for b, (o, r) in enumerate(zip(outs, refs)):
        assert o.shape == r.shape
        err = (o.float() - r.float()).abs().max().item()
        assert err <= ATOL[dtype], f"{what} sequence {b}: max abs err {err}"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Hq,Hkv,D,tpb", [(4, 4, 128, 64), (8, 2, 128, 16), (8, 1, 64, 32), (4, 4, 64, 128)])
def test_chunked_context_equals_one_shot(cuda, dtype, Hq, Hkv, D, tpb):
    # a prompt fed in three chunks (ragged, tile and block boundaries crossed mid-chunk); every chunk against the oracle
    # continuing over the same cache, and the concatenation against the one-shot prefill of the whole prompt
    total = [300, 171]
    cuts = [[130, 64, 106], [1, 127, 43]]
    rows_all = make_inputs(total, Hq, Hkv, D, dtype)
    cache = oracle.PagedKVCache(2, Hkv, D, max(total) + 24, tpb, dtype)
    pool = cache.pool.to(cuda)
    plg = plugin(Hq, Hkv, D, dtype, dtype, tpb, rope="neox", paged_ctx=True)
    done = [0, 0]
    got = [[], []]
    for step in range(3):
        lens = [cuts[b][step] for b in range(2)]
        rows = [rows_all[b][done[b]:done[b] + lens[b]] for b in range(2)]
        kv = [done[b] + lens[b] for b in range(2)]
        outs = enqueue(plg, rows, kv, cache, pool, cuda)
        refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, rope="neox", rot_dim=D, past_lens=done)
        assert_close(outs, refs, dtype, f"chunk {step}")
        compare_cache(pool.cpu(), cache, "neox")
        cache.pool = pool.cpu().clone()  # the oracle continues over the cache the GPU wrote
        for b in range(2):
            got[b].append(outs[b])
            done[b] += lens[b]
    whole = oracle.context_attention(rows_all, total, Hq, Hkv, D, rope="neox", rot_dim=D)
    assert_close([torch.cat(g, 0) for g in got], whole, dtype, "chunks vs one shot")


def test_chunked_context_c_abi_linear_cache_and_errors(cuda):
    # the same through mi355q_context_attention over a KVLinearBuffer; 8-bit caches and wrapped windows are refused
    Hq, Hkv, D, dtype, dev = 4, 2, 128, torch.float16, cuda
    total, first = [200, 90], [136, 37]
    rows_all = make_inputs(total, Hq, Hkv, D, dtype, seed=3)
    max_seq = 256
    lin = torch.zeros(2, 2, Hkv, max_seq, D, dtype=dtype, device=dev)

    class LinCache:  # the oracle's view of the linear buffer
        dtype = torch.float16

        def __init__(self):
            self.buf = torch.zeros(2, 2, Hkv, max_seq, D, dtype=dtype)

        def write(self, b, kv, head, tok, vec):
            self.buf[b, kv, head, tok] = vec

        def gather(self, b, kv, head, length):
            return self.buf[b, kv, head, :length]

    cache = LinCache()
    done = [0, 0]
    outs_all = [[], []]
    for lens in (first, [total[b] - first[b] for b in range(2)]):
        rows = [rows_all[b][done[b]:done[b] + lens[b]] for b in range(2)]
        kv = [done[b] + lens[b] for b in range(2)]
        qkv = torch.cat(rows, 0).to(dev)
        out = torch.zeros(sum(lens), Hq * D, dtype=dtype, device=dev)
        cl = torch.tensor(lens, dtype=torch.int32, device=dev)
        sl = torch.tensor(kv, dtype=torch.int32, device=dev)
        p = L.ContextAttentionParams()
        p.qkv, p.out, p.context_lengths = qkv.data_ptr(), out.data_ptr(), cl.data_ptr()
        p.batch, p.num_heads, p.num_kv_heads, p.head_size = 2, Hq, Hkv, D
        p.max_context_length, p.num_tokens, p.remove_padding = max(lens), sum(lens), 1
        p.act_type = p.kv_type = L.HALF
        p.q_scaling = 1.0
        p.kv_linear_buffer, p.linear_max_seq_len = lin.data_ptr(), max_seq
        p.sequence_lengths, p.max_kv_length = sl.data_ptr(), max(kv)
        need = L.lib().mi355q_context_attention_workspace_size(C.byref(p))
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = ws.data_ptr(), need
        L.check(L.lib().mi355q_context_attention(C.byref(p), None))
        torch.cuda.synchronize()
        refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, past_lens=done)
        o = out.cpu()
        offs = np.concatenate([[0], np.cumsum(lens)])
        assert_close([o[offs[b]:offs[b + 1]] for b in range(2)], refs, dtype)
        assert torch.equal(lin.cpu().view(torch.uint8), cache.buf.view(torch.uint8))
        for b in range(2):
            done[b] += lens[b]
    # refused: 8-bit cache, kv length beyond the window
    p.kv_type = L.INT8
    assert L.lib().mi355q_context_attention(C.byref(p), None) != 0 and b"16-bit" in L.lib().mi355q_last_error()
    p.kv_type = L.HALF
    p.cyclic_attention_window_size = 64
    assert L.lib().mi355q_context_attention(C.byref(p), None) != 0 and b"wrapped" in L.lib().mi355q_last_error()


@pytest.mark.parametrize("dtype,kv_dtype", [(torch.float16, torch.float16), (torch.bfloat16, torch.int8)])
@pytest.mark.parametrize("window,sink,tpb,lens", [(64, 0, 64, [300, 40]), (100, 0, 32, [333, 101]), (96, 4, 16, [250, 97]),
                                                   (128, 0, 64, [129, 128]), (17, 0, 16, [70, 200])])
def test_sliding_window_context(cuda, dtype, kv_dtype, window, sink, tpb, lens):
    # contexts longer than the attention window: the mask of gptKernels.cu:174-177, and a cache that keeps the sink tokens
    # + the last window - sink tokens at their cyclic slots; then one generation step continues over that cache
    Hq, Hkv, D = 4, 2, 128
    kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else 1.0
    rows = make_inputs(lens, Hq, Hkv, D, dtype, seed=window)
    cap_blocks = (window + tpb - 1) // tpb + 2
    cache = oracle.PagedKVCache(2, Hkv, D, cap_blocks * tpb, tpb, kv_dtype)
    cmap = oracle.CyclicMap(window, sink, tpb, cache.max_blocks)
    pool = cache.pool.to(cuda)
    plg = plugin(Hq, Hkv, D, dtype, kv_dtype, tpb)
    outs = enqueue(plg, rows, lens, cache, pool, cuda, window=window, sink=sink, kv_scale=kv_scale)
    refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, window=window, cmap=cmap,
                                    kv_scale_orig_quant=1.0 / kv_scale)
    assert_close(outs, refs, dtype)
    compare_cache(pool.cpu(), cache, None)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("pe", [3, 4])
def test_alibi_context(cuda, dtype, pe):
    # kALIBI / kALIBI_WITH_SCALE: the slopes tensor carries the scale (functional.py generate_alibi_slopes), the kernel adds
    # slope[h] * (t - a) to the scaled score
    Hq, Hkv, D, tpb = 8, 8, 64, 64
    lens = [200, 77]
    rows = make_inputs(lens, Hq, Hkv, D, dtype, seed=pe)
    scale = 1.0 if pe == 3 else 1.0 / np.sqrt(D)
    slopes = (torch.tensor([2.0 ** (-8.0 * (i + 1) / Hq) for i in range(Hq)]) * scale).to(dtype)
    cache = oracle.PagedKVCache(2, Hkv, D, max(lens) + 24, tpb, dtype)
    pool = cache.pool.to(cuda)
    plg = plugin(Hq, Hkv, D, dtype, dtype, tpb, pe=pe)
    outs = enqueue(plg, rows, lens, cache, pool, cuda, alibi=slopes)
    refs = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=cache, alibi_slopes=slopes)
    assert_close(outs, refs, dtype)
    compare_cache(pool.cpu(), cache, None)
    # chunked + ALiBi + window together (16-bit, D = 128): second chunk over the first
    Hq, Hkv, D = 4, 2, 128
    total, first = [260], [150]
    rows_all = make_inputs(total, Hq, Hkv, D, dtype, seed=11)
    slopes = (torch.tensor([2.0 ** (-2.0 * (i + 1)) for i in range(Hq)]) * scale).to(dtype)
    cache = oracle.PagedKVCache(1, Hkv, D, 320, tpb, dtype)
    pool = cache.pool.to(cuda)
    plg = plugin(Hq, Hkv, D, dtype, dtype, tpb, pe=pe, paged_ctx=True)
    done = [0]
    for lens in (first, [total[0] - first[0]]):
        r = [rows_all[0][done[0]:done[0] + lens[0]]]
        outs = enqueue(plg, r, [done[0] + lens[0]], cache, pool, cuda, alibi=slopes)
        refs = oracle.context_attention(r, lens, Hq, Hkv, D, cache=cache, alibi_slopes=slopes, past_lens=done)
        assert_close(outs, refs, dtype, "chunked alibi")
        cache.pool = pool.cpu().clone()
        done[0] += lens[0]
