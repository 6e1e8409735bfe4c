"""GPU parity: cache addressing of the generation step beyond the identity map, through the C ABI against the oracle's
restatement (oracle.decode_attention_slots / oracle.CyclicMap: kvCacheUtils.h:36-153 KVBlockArray, :241-338
KVLinearBuffer, decoderMaskedMultiheadAttentionTemplate.h:1480-1500, 1873-1879, 1929, 2076, 2441):
  * cyclic attention window + sink tokens (slot map, kv_loop_length, the overwritten slot is still read),
  * the contiguous KVLinearBuffer cache,
  * beam width > 1 through cache_indirection,
  * the general kernel's head ratios and head sizes (one workgroup per query head).
Outputs: the reference's generation tolerance (see test_gpu_attention.py); appended cache lines: exact.
No fixture of the reference covers these cases with stored outputs (its tests compute a torch reference on the fly):
the oracle's formulas are pinned line by line to the cited source, the numerics by test_attention_hf_pin.py."""
import ctypes as C

import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

pytestmark = pytest.mark.gpu

ATOL = {torch.float16: 2e-3, torch.bfloat16: 2.0 ** -7}
CODE = {torch.float16: L.HALF, torch.bfloat16: L.BF16, torch.int8: L.INT8, torch.float8_e4m3fn: L.FP8}


def _fill(slots, seq_lens, cmap, dtype, kv_dtype, kv_scale, gen):
    """Random history: token t of row r sits in slot cmap.slot(t); later tokens overwrite earlier ones."""
    R, _, H, cap, D = slots.shape
    for r in range(R):
        n = seq_lens[r] - 1
        kv = (torch.randn(2, H, n, D, generator=gen) * 0.5).to(dtype)
        q = oracle.quantize_kv(kv, kv_dtype, 1.0 / kv_scale)
        for t in range(n):
            slots[r, :, :, cmap.slot(t)] = q[:, :, t]


def _run(dev, B, Hq, Hkv, D, seq_lens, dtype, kv_dtype=None, tpb=64, max_blocks=None, window=0, sink=0, linear=False,
         beam_width=1, rope="neox", seed=3, multi_block=0, via_plugin=False, pos_shift=False):
    kv_dtype = kv_dtype or dtype
    gen = torch.Generator().manual_seed(seed)
    capacity = (max_blocks * tpb) if not linear else max_blocks * tpb
    win = window if window > 0 else capacity
    cmap = oracle.CyclicMap(win, sink, tpb, max_blocks, linear=linear)
    kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else (0.5 if kv_dtype == torch.float8_e4m3fn else 1.0)
    slots = torch.zeros(B, 2, Hkv, capacity, D, dtype=kv_dtype)
    _fill(slots, seq_lens, cmap, dtype, kv_dtype, kv_scale, gen)
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D, generator=gen).to(dtype)
    indir = input_lengths = None
    if beam_width > 1:
        # every beam of a batch entry shares the context in beam 0; generated slots point at random beams
        input_lengths = torch.tensor([max(1, (seq_lens[b] - 1) // 2) for b in range(B)], dtype=torch.int32)
        indir = torch.randint(0, beam_width, (B, win), generator=gen, dtype=torch.int32)
    # ---- device side
    esz = slots.element_size()
    if linear:
        dcache = slots.clone().to(dev)  # [B, 2, H, S, D] IS the KVLinearBuffer layout
        ptrs = None
    else:
        # pool of blocks [nblk, 2, H, tpb, D] in a shuffled order
        nblk = B * max_blocks
        perm = torch.randperm(nblk, generator=gen).reshape(B, max_blocks)
        pool = torch.zeros(nblk, 2, Hkv, tpb, D, dtype=kv_dtype)
        blocks = slots.reshape(B, 2, Hkv, max_blocks, tpb, D).permute(0, 3, 1, 2, 4, 5)  # [B, blk, 2, H, tpb, D]
        pool[perm.reshape(-1)] = blocks.reshape(nblk, 2, Hkv, tpb, D)
        dcache = pool.to(dev)
        be = Hkv * tpb * D * esz
        ptrs = (torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + dcache.data_ptr()).to(torch.int64).to(dev)
    d_qkv, d_lens = qkv.to(dev), torch.tensor(seq_lens, dtype=torch.int32, device=dev)
    out = torch.zeros(B, Hq * D, dtype=dtype, device=dev)
    if via_plugin:
        # the same step through the GPTAttention plugin's input list (gptAttentionPlugin.h:43-68)
        qm = {torch.int8: 1 << 6, torch.float8_e4m3fn: 1 << 7}.get(kv_dtype, 0)
        plg = F.make_gpt_attention_plugin(
            num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16" if dtype == torch.float16 else "bfloat16",
            position_embedding_type={None: 0, "gptj": 1, "neox": 2}[rope], rotary_embedding_dim=D if rope else 0,
            kv_cache_quant_mode=qm, paged_kv_cache=not linear, tokens_per_block=tpb, pos_shift_enabled=pos_shift)
        lens = torch.tensor(seq_lens, dtype=torch.int32)
        ins = [d_qkv.reshape(B, 1, -1), d_lens, lens - 1, torch.tensor([win], dtype=torch.int32),
               torch.tensor([sink], dtype=torch.int32),
               (input_lengths if input_lengths is not None else lens - 1).to(dev),
               (indir if indir is not None else torch.zeros(B, win, dtype=torch.int32)).reshape(
                   B // beam_width, beam_width, win).to(dev),
               torch.ones(B, dtype=torch.int32)]
        ins += [dcache] if linear else [ptrs.reshape(1, B, 2, max_blocks), ptrs.reshape(1, B, 2, max_blocks).cpu()]
        if qm:
            ins += [torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev),
                    torch.tensor([kv_scale], dtype=torch.float32, device=dev)]
        outs = [out.reshape(B, 1, -1)] + ([dcache] if linear else [])  # present_key_value aliases past_key_value
        plg.configure(ins)
        nbytes = plg.get_workspace_size(ins, outs)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        plg.enqueue(ins, outs, ws)
        torch.cuda.synchronize()
    else:
        p = L.DecodeAttentionParams()
        p.qkv, p.qkv_bias, p.out = d_qkv.data_ptr(), None, out.data_ptr()
        p.sequence_lengths = d_lens.data_ptr()
        p.block_pointers = ptrs.data_ptr() if ptrs is not None else None
        p.batch, p.num_heads, p.num_kv_heads, p.head_size = B, Hq, Hkv, D
        p.max_blocks_per_seq, p.tokens_per_block, p.max_seq_len = max_blocks, tpb, max(seq_lens)
        p.act_type, p.kv_type = CODE[dtype], CODE[kv_dtype]
        p.position_embedding_type = {None: 0, "gptj": 1, "neox": 2}[rope]
        p.rotary_embedding_dim, p.rotary_embedding_base = (D if rope else 0), 10000.0
        p.rotary_embedding_scale_type, p.rotary_embedding_scale, p.rotary_embedding_max_positions = 0, 1.0, 4096
        p.q_scaling, p.multi_block_mode = 1.0, multi_block
        p.cyclic_attention_window_size, p.sink_token_length = window, sink
        p.position_shift_enabled = 1 if pos_shift else 0
        keep = []
        if kv_dtype != dtype:
            so = torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=dev)
            sq = torch.tensor([kv_scale], dtype=torch.float32, device=dev)
            p.kv_scale_orig_quant, p.kv_scale_quant_orig = so.data_ptr(), sq.data_ptr()
            keep += [so, sq]
        if beam_width > 1:
            d_ind, d_il = indir.to(dev), input_lengths.to(dev)
            p.beam_width, p.max_attention_window_size = beam_width, win
            p.cache_indirection, p.input_lengths = d_ind.data_ptr(), d_il.data_ptr()
            keep += [d_ind, d_il]
        if linear:
            p.kv_linear_buffer, p.linear_max_seq_len = dcache.data_ptr(), capacity
        need = L.lib().mi355q_decode_attention_workspace_size(C.byref(p))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
        p.workspace, p.workspace_bytes = ws.data_ptr(), need
        L.check(L.lib().mi355q_masked_multihead_attention(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        torch.cuda.synchronize()
    # ---- oracle
    ref_slots = slots.clone()
    ref = oracle.decode_attention_slots(qkv, ref_slots, seq_lens, Hq, Hkv, D, cmap, rope=rope, rot_dim=D if rope else 0,
                                        kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale,
                                        beam_width=beam_width, cache_indir=indir, input_lengths=input_lengths,
                                        pos_shift=pos_shift)
    err = (out.cpu().float() - ref.float()).abs().max().item()
    assert err <= ATOL[dtype], f"output differs from the oracle by {err}"
    # the cache after the step: exactly the oracle's (the new token in its slot, nothing else touched)
    if linear:
        got = dcache.cpu()
    else:
        got = dcache.cpu()[perm.reshape(-1)].reshape(B, max_blocks, 2, Hkv, tpb, D).permute(0, 2, 3, 1, 4, 5).reshape(
            B, 2, Hkv, capacity, D)
    # V lines (and K without RoPE): exact.  K through RoPE goes through sin / cos: one code of difference on the
    # appended line, the allowance of test_gpu_attention.py
    def codes(x):
        return x.view(torch.uint8) if kv_dtype == torch.float8_e4m3fn else x
    exact = slice(None) if (rope is None or pos_shift) else slice(1, 2)   # POS_SHIFT stores the key un-rotated: exact too
    a, b = codes(got[:, exact]), codes(ref_slots[:, exact])
    if not torch.equal(a, b):
        bad = (a != b).any(-1).nonzero()[:8].tolist()
        raise AssertionError(f"cache lines differ at (row, k|v, head, slot) {bad}; expected writes at slots "
                             f"{[cmap.slot(l - 1) for l in seq_lens]}")
    if rope is not None:
        gk = got[:, 0].to(torch.int32).float() if kv_dtype == torch.int8 else got[:, 0].float()
        rk = ref_slots[:, 0].to(torch.int32).float() if kv_dtype == torch.int8 else ref_slots[:, 0].float()
        tol = 1.01 if kv_dtype == torch.int8 else (0.13 if kv_dtype == torch.float8_e4m3fn else 2e-2)
        diff = (gk - rk).abs()
        assert diff.max().item() <= tol, f"K lines differ by {diff.max().item()}"
        # and only the appended slots may differ at all
        touched = diff.amax(dim=(1, 3)) > 0  # [row, slot]
        for r in range(B):
            others = touched[r].clone()
            others[cmap.slot(seq_lens[r] - 1)] = False
            assert not others.any(), f"row {r}: K slots other than the appended one changed"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_cyclic_window_mha(cuda, dtype):
    # window 128 over 3 blocks of 64: one-more-block mode (192 cyclic slots); lengths below, at and past the wrap
    _run(cuda, 5, 4, 4, 128, [40, 129, 130, 200, 300], dtype, tpb=64, max_blocks=3, window=128)
    # window = capacity: plain cyclic cache of 128 slots, wrapped several times
    _run(cuda, 4, 4, 4, 64, [100, 129, 257, 400], dtype, tpb=32, max_blocks=4, window=128, rope="gptj")


def test_cyclic_window_with_sink_tokens(cuda):
    # 4 sink tokens in a block of 64: bubble 60, 124 cyclic slots behind it (kvCacheUtils.h:89-93)
    _run(cuda, 4, 4, 4, 128, [3, 100, 131, 380], torch.float16, tpb=64, max_blocks=3, window=128, sink=4)
    # a whole block of sink tokens: no bubble
    _run(cuda, 3, 8, 8, 64, [70, 200, 500], torch.bfloat16, tpb=64, max_blocks=4, window=256, sink=64)


@pytest.mark.parametrize("kv_dtype", [torch.int8, torch.float8_e4m3fn])
def test_cyclic_window_gqa_8bit_cache(cuda, kv_dtype):
    # the MFMA GQA kernel (8 query heads per kv head), wrapped window, splits forced so that the overwritten slot is
    # read by one split and written by it after its loop
    _run(cuda, 3, 8, 1, 128, [90, 300, 777], torch.bfloat16, kv_dtype=kv_dtype, tpb=64, max_blocks=4, window=256,
         multi_block=4)
    _run(cuda, 2, 16, 4, 128, [257, 600], torch.float16, kv_dtype=kv_dtype, tpb=32, max_blocks=8, window=256)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_linear_buffer(cuda, dtype):
    # KVLinearBuffer: [B, 2, H, S, D], S not a power of two; the three kernels (VALU, GQA, general)
    _run(cuda, 4, 4, 4, 128, [1, 50, 199, 200], dtype, tpb=200, max_blocks=1, linear=True)
    _run(cuda, 3, 8, 2, 128, [33, 150, 300], dtype, tpb=300, max_blocks=1, linear=True)
    _run(cuda, 3, 6, 2, 64, [33, 150, 300], dtype, tpb=300, max_blocks=1, linear=True)  # ratio 3: general kernel
    _run(cuda, 3, 4, 4, 64, [90, 250, 333], dtype, kv_dtype=torch.int8, tpb=120, max_blocks=1, linear=True, window=100)


@pytest.mark.parametrize("beam_width", [2, 4])
def test_beam_search_cache_indirection(cuda, beam_width):
    B = 3 * beam_width
    lens = [l for l in (37, 90, 161) for _ in range(beam_width)]
    _run(cuda, B, 4, 4, 128, lens, torch.float16, tpb=64, max_blocks=3, beam_width=beam_width)
    _run(cuda, B, 8, 2, 64, lens, torch.bfloat16, kv_dtype=torch.int8, tpb=32, max_blocks=6, beam_width=beam_width,
         rope="gptj", multi_block=2)


def test_general_kernel_head_ratios_and_sizes(cuda):
    _run(cuda, 3, 6, 2, 128, [20, 130, 257], torch.float16, tpb=64, max_blocks=5)       # ratio 3
    _run(cuda, 2, 12, 1, 64, [64, 300], torch.bfloat16, tpb=64, max_blocks=5)           # ratio 12
    _run(cuda, 3, 4, 2, 256, [20, 130, 257], torch.float16, tpb=64, max_blocks=5)       # head size 256
    _run(cuda, 3, 4, 4, 32, [20, 130, 257], torch.bfloat16, tpb=64, max_blocks=5, rope="gptj")  # head size 32
    _run(cuda, 2, 4, 4, 32, [99, 300], torch.float16, kv_dtype=torch.int8, tpb=16, max_blocks=20)  # 16 < 32 rows / load


def test_through_the_plugin(cuda):
    # window and sink tokens from host_max_attention_window_sizes / host_sink_token_length, beams from the shape of
    # cache_indirection, the contiguous cache as past / present_key_value (gptAttentionPlugin.cpp:476-537, 612-616)
    _run(cuda, 4, 4, 4, 128, [100, 129, 257, 400], torch.float16, tpb=64, max_blocks=3, window=128, via_plugin=True)
    _run(cuda, 3, 8, 1, 128, [90, 300, 777], torch.bfloat16, kv_dtype=torch.int8, tpb=64, max_blocks=5, window=256,
         sink=64, via_plugin=True)
    _run(cuda, 6, 4, 4, 64, [37, 37, 90, 90, 161, 161], torch.float16, tpb=64, max_blocks=3, beam_width=2, via_plugin=True)
    _run(cuda, 3, 8, 2, 128, [33, 150, 300], torch.bfloat16, tpb=300, max_blocks=1, linear=True, via_plugin=True)


class _LinearCache:
    """The KVLinearBuffer layout [B, 2, Hkv, S, Dh] behind the accessor the oracle's attention functions use."""

    def __init__(self, B, H, S, D, dtype):
        self.data = torch.zeros(B, 2, H, S, D, dtype=dtype)
        self.dtype = dtype

    def write(self, b, kv, head, tok, vec):
        self.data[b, kv, head, tok] = vec

    def gather(self, b, kv, head, length):
        return self.data[b, kv, head, :length]


@pytest.mark.parametrize("kv_dtype", [None, torch.int8])
def test_linear_cache_context_then_generation_through_the_plugin(cuda, kv_dtype):
    # paged_kv_cache = 0: the context phase fills present_key_value, a generation step continues on it
    dtype, Hq, Hkv, D, S = torch.float16, 8, 2, 128, 150
    kv_dtype = kv_dtype or dtype
    lens = [37, 64, 5]
    B = len(lens)
    g = torch.Generator().manual_seed(5)
    rows = [(torch.randn(n, (Hq + 2 * Hkv) * D, generator=g) * 0.5).to(dtype) for n in lens]
    kv_scale = 4.0 / 127.0 if kv_dtype == torch.int8 else 1.0
    qm = {torch.int8: 1 << 6}.get(kv_dtype, 0)
    plg = F.make_gpt_attention_plugin(num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16",
                                      position_embedding_type=2, rotary_embedding_dim=D, kv_cache_quant_mode=qm,
                                      remove_input_padding=True, paged_kv_cache=False, tokens_per_block=64)
    dcache = torch.zeros(B, 2, Hkv, S, D, dtype=kv_dtype, device=cuda)
    scales = [torch.tensor([1.0 / kv_scale], dtype=torch.float32, device=cuda),
              torch.tensor([kv_scale], dtype=torch.float32, device=cuda)] if qm else []

    def enqueue(qkv, seq, past, ctx, req):
        n = qkv.shape[0]
        out = torch.zeros(n, Hq * D, dtype=dtype, device=cuda)
        ins = [qkv.to(cuda), seq.to(cuda), past, torch.tensor([S], dtype=torch.int32), torch.zeros(1, dtype=torch.int32),
               ctx.to(cuda), torch.zeros(B, 1, S, dtype=torch.int32, device=cuda), req, dcache] + scales + [ctx]
        outs = [out, dcache]
        plg.configure(ins)
        ws = torch.empty(max(plg.get_workspace_size(ins, outs), 1), dtype=torch.uint8, device=cuda)
        plg.enqueue(ins, outs, ws)
        torch.cuda.synchronize()
        return out.cpu()

    ref_cache = _LinearCache(B, Hkv, S, D, kv_dtype)
    L32 = torch.tensor(lens, dtype=torch.int32)
    out = enqueue(torch.cat(rows, 0), L32, torch.zeros(B, dtype=torch.int32), L32, torch.zeros(B, dtype=torch.int32))
    ref = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=ref_cache, rope="neox", rot_dim=D,
                                   kv_scale_orig_quant=1.0 / kv_scale)
    err = max((out[sum(lens[:b]):sum(lens[:b + 1])].float() - ref[b].float()).abs().max().item() for b in range(B))
    assert err <= 5e-3, f"context output differs by {err}"
    got = dcache.cpu()
    assert torch.equal(got[:, 1], ref_cache.data[:, 1]), "V lines of the context differ"
    kd = (got[:, 0].to(torch.int32).float() - ref_cache.data[:, 0].to(torch.int32).float()).abs().max().item() \
        if kv_dtype == torch.int8 else (got[:, 0].float() - ref_cache.data[:, 0].float()).abs().max().item()
    assert kd <= (1.01 if kv_dtype == torch.int8 else 2e-2)
    ref_cache.data.copy_(got)  # continue from the device's cache (K lines may differ by one code)
    # one generation step
    qkv = (torch.randn(B, (Hq + 2 * Hkv) * D, generator=g) * 0.5).to(dtype)
    seq = L32 + 1
    out = enqueue(qkv, seq, L32, L32, torch.ones(B, dtype=torch.int32))
    ref = oracle.decode_attention(qkv, ref_cache, seq.tolist(), Hq, Hkv, D, rope="neox", rot_dim=D,
                                  kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale)
    err = (out.float() - ref.float()).abs().max().item()
    assert err <= 2e-3, f"generation output differs by {err}"
    assert torch.equal(dcache.cpu()[:, 1], ref_cache.data[:, 1]), "V line of the new token differs"


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("rope", ["neox", "gptj"])
def test_position_shift_streamingllm(cuda, dtype, rope):
    """position_shift_enabled (StreamingLLM): the cache holds un-rotated keys, the step rotates the live keys to their
    position inside the cache, q / the new key to position min(kv length, window), and stores the new key un-rotated
    (oracle.decode_attention_slots pos_shift: invokeShiftKCache + the POS_SHIFT branches of the masked-MHA kernel).  The
    reference's own StreamingLLM cases (tests/attention/test_gpt_attention.py:188-192: llama, sink 4, pos shift on / off)
    run with the window at the sequence length; here also past the wrap, with sink tokens (bubble: 4 % 64 != 0), in
    one-more-block mode, on a linear cache, GQA, forced kv splits and through the plugin.  Parity unpinned beyond that grid."""
    # window >= every length, no sink: positions are the token indices (the reference's own cases)
    _run(cuda, 3, 4, 4, 64, [40, 129, 100], dtype, tpb=64, max_blocks=3, rope=rope, pos_shift=True)
    # window 128 over 2 blocks (no spare block): lengths below, at and past the wrap, sink 4 with a bubble
    _run(cuda, 4, 4, 4, 128, [40, 129, 130, 300], dtype, tpb=64, max_blocks=3, window=100, sink=4, rope=rope, pos_shift=True, seed=11)
    # one-more-block mode (window 128 in 3 blocks), GQA 8:2, forced splits
    _run(cuda, 3, 8, 2, 128, [129, 200, 450], dtype, tpb=64, max_blocks=3, window=128, rope=rope, pos_shift=True, multi_block=3, seed=12)
    # contiguous cache, window with sink tokens
    _run(cuda, 2, 4, 2, 64, [150, 60], dtype, tpb=1, max_blocks=96, window=80, sink=3, linear=True, rope=rope, pos_shift=True, seed=13)
    # and through the plugin's input list
    _run(cuda, 2, 4, 4, 128, [260, 77], dtype, tpb=64, max_blocks=4, window=192, sink=4, rope=rope, pos_shift=True, via_plugin=True, seed=14)


def test_position_shift_refusals(cuda):
    p = L.DecodeAttentionParams()
    p.batch, p.num_heads, p.num_kv_heads, p.head_size = 1, 4, 4, 64
    p.max_blocks_per_seq, p.tokens_per_block, p.max_seq_len = 2, 64, 100
    p.act_type, p.kv_type = L.HALF, L.INT8
    p.position_embedding_type, p.rotary_embedding_dim, p.position_shift_enabled = 2, 64, 1
    assert L.lib().mi355q_masked_multihead_attention(C.byref(p), None) == -2          # 8-bit cache: MI355Q_ERR_UNSUPPORTED
    p.kv_type, p.position_embedding_type = L.HALF, 0
    assert L.lib().mi355q_masked_multihead_attention(C.byref(p), None) == -1          # no rotary embedding to shift


def test_position_shift_context_then_generation_through_the_plugin(cuda):
    """pos_shift_enabled through the plugin: the context phase stores K without the rotation (bit-exact against the
    oracle: no sin / cos is involved), its outputs are those of the plain context phase; a generation step then continues
    on that cache.  With the window at the sequence length and no sink tokens every key gets the position it has in the
    sequence, so the step also has to agree with the un-shifted pipeline (the reference's StreamingLLM grid compares
    exactly these two, tests/attention/test_gpt_attention.py:188-192)."""
    dtype, Hq, Hkv, D, S = torch.float16, 8, 2, 128, 150
    lens = [37, 64, 5]
    B = len(lens)
    g = torch.Generator().manual_seed(7)
    rows = [(torch.randn(n, (Hq + 2 * Hkv) * D, generator=g) * 0.5).to(dtype) for n in lens]
    qkv = (torch.randn(B, (Hq + 2 * Hkv) * D, generator=g) * 0.5).to(dtype)
    L32 = torch.tensor(lens, dtype=torch.int32)
    outs_gen = {}
    for shift in (True, False):
        plg = F.make_gpt_attention_plugin(num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="float16",
                                          position_embedding_type=2, rotary_embedding_dim=D, remove_input_padding=True,
                                          paged_kv_cache=False, tokens_per_block=64, pos_shift_enabled=shift)
        dcache = torch.zeros(B, 2, Hkv, S, D, dtype=dtype, device=cuda)

        def enqueue(x, seq, past, ctx, req):
            out = torch.zeros(x.shape[0], Hq * D, dtype=dtype, device=cuda)
            ins = [x.to(cuda), seq.to(cuda), past, torch.tensor([S], dtype=torch.int32), torch.zeros(1, dtype=torch.int32),
                   ctx.to(cuda), torch.zeros(B, 1, S, dtype=torch.int32, device=cuda), req, dcache, ctx]
            plg.configure(ins)
            ws = torch.empty(max(plg.get_workspace_size(ins, [out, dcache]), 1), dtype=torch.uint8, device=cuda)
            plg.enqueue(ins, [out, dcache], ws)
            torch.cuda.synchronize()
            return out.cpu()

        ref_cache = _LinearCache(B, Hkv, S, D, dtype)
        out = enqueue(torch.cat(rows, 0), L32, torch.zeros(B, dtype=torch.int32), L32, torch.zeros(B, dtype=torch.int32))
        ref = oracle.context_attention(rows, lens, Hq, Hkv, D, cache=ref_cache, rope="neox", rot_dim=D, pos_shift=shift)
        err = max((out[sum(lens[:b]):sum(lens[:b + 1])].float() - ref[b].float()).abs().max().item() for b in range(B))
        assert err <= 5e-3, f"context output differs by {err} (pos_shift {shift})"
        got = dcache.cpu()
        if shift:
            assert torch.equal(got, ref_cache.data), "the context phase must store K un-rotated, bit for bit"
            slots = got.clone()
            cmap = oracle.CyclicMap(S, 0, linear=True)
            seq = (L32 + 1).tolist()
            ref_gen = oracle.decode_attention_slots(qkv, slots, seq, Hq, Hkv, D, cmap, rope="neox", rot_dim=D, pos_shift=True)
        outs_gen[shift] = enqueue(qkv, L32 + 1, L32, L32, torch.ones(B, dtype=torch.int32))
        if shift:
            err = (outs_gen[True].float() - ref_gen.float()).abs().max().item()
            assert err <= 2e-3, f"generation output with pos_shift differs from the oracle by {err}"
            assert torch.equal(dcache.cpu(), slots), "the generation step must append the un-rotated key / the value"
    err = (outs_gen[True].float() - outs_gen[False].float()).abs().max().item()
    assert err <= 4e-3, f"pos_shift and plain pipelines disagree by {err} where every position equals the token index"
