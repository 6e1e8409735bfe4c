"""Pin the CPU oracle against the known-answer vectors the reference itself spells out
(SURVEY 8c): permutation maps, register relayouts, pack/unpack conventions, tolerances, fp8 scale KAT.
CPU only."""
import numpy as np
import pytest
import torch

import oracle

# cutlass_preprocessors.cpp:157-160 (mirrored in kernels/weightOnlyBatchedGemv/kernel.h:84-90,134-140)
PERM_INT8 = [0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15]
PERM_INT4 = [0, 1, 8, 9, 16, 17, 24, 25, 2, 3, 10, 11, 18, 19, 26, 27,
             4, 5, 12, 13, 20, 21, 28, 29, 6, 7, 14, 15, 22, 23, 30, 31]


def test_row_permutation_map_int8():
    rows, cols = 32, 8
    w = np.repeat(np.arange(rows, dtype=np.uint8)[:, None], cols, axis=1)  # every byte = its row index
    out = oracle.permute_B_rows_for_mixed_gemm(w, rows, cols, oracle.INT8_W).reshape(rows, cols)
    assert out[:16, 0].tolist() == PERM_INT8
    assert out[16:, 0].tolist() == [16 + p for p in PERM_INT8]


def test_row_permutation_map_int4():
    rows, cols = 64, 8
    vals = (np.arange(rows) % 16).astype(np.int8) - 8  # int4 can't hold 0..31: check via two passes
    # use the group-local row index split into low/high 4 bits over two matrices
    for shift in (0, 4):
        idx = ((np.arange(rows) % 32) >> shift) & 0xF
        unpacked = np.repeat(((idx ^ 8) - 8).astype(np.int8)[:, None], cols, axis=1)  # signed nibble with bits == idx
        packed = oracle.pack_int8_tensor_to_packed_int4(unpacked)
        out = oracle.permute_B_rows_for_mixed_gemm(packed, rows, cols, oracle.INT4_W).reshape(rows, cols // 2)
        got = oracle.unpack_int4_packed_tensor_to_int8(out.view(np.int8))[:, 0].astype(np.int32) & 0xF
        want = [(p >> shift) & 0xF for p in PERM_INT4] * 2
        assert got.tolist() == want
    del vals


def test_register_relayout_int8():
    # [e3 e2 e1 e0] -> [e3 e1 e2 e0] after +128 (cutlass_preprocessors.cpp:393-401)
    w = np.array([0, 1, 2, 3, -128, -1, 127, 5], dtype=np.int8)
    out = oracle.add_bias_and_interleave_int8s(w)
    assert out.tolist() == [128, 130, 129, 131, 0, 255, 127, 133]


def test_register_relayout_int4():
    # [e7 .. e0] -> [e7 e5 e3 e1 e6 e4 e2 e0] after +8 (cutlass_preprocessors.cpp:434-441)
    elts = np.array([-8, -7, -6, -5, -4, -3, -2, -1], dtype=np.int8)  # +8 -> 0..7
    packed = oracle.pack_int8_tensor_to_packed_int4(elts[None, :])
    out = oracle.add_bias_and_interleave_int4s(packed)
    word = int(out.view(np.uint32).ravel()[0])
    nibbles = [(word >> (4 * i)) & 0xF for i in range(8)]
    assert nibbles == [0, 2, 4, 6, 1, 3, 5, 7]


def test_pack_unpack_convention():
    # low nibble = even element, sign extension (weightOnlyQuantOp.cpp:296-302)
    packed = np.array([[0x8F, 0x70, 0x1E]], dtype=np.uint8).view(np.int8)
    assert oracle.unpack_int4_packed_tensor_to_int8(packed).tolist() == [[-1, -8, 0, 7, -2, 1]]
    rt = oracle.pack_int8_tensor_to_packed_int4(oracle.unpack_int4_packed_tensor_to_int8(packed))
    assert (rt == packed).all()
    with pytest.raises(ValueError):
        oracle.pack_int8_tensor_to_packed_int4(np.array([[8, 0]], dtype=np.int8))


def test_interleave_description_int4():
    # kernel.h:71-79: with interleave 4 and 64-row tiles, 4 adjacent columns' 64-row tiles alternate
    rows, cols = 128, 8
    # column-major input where every element of column c, row-tile t carries the tag (c, t)
    cm = np.zeros((cols, rows), dtype=np.int8)
    for c in range(cols):
        for r in range(rows):
            cm[c, r] = (c % 4) * 2 + (r // 64) - 8  # 0..7 tags as signed nibbles
    packed = oracle.pack_int8_tensor_to_packed_int4(cm)
    out = oracle.interleave_column_major_tensor(packed, rows, cols, oracle.INT4_W)
    got = oracle.unpack_int4_packed_tensor_to_int8(out.view(np.int8).reshape(cols // 4, -1)) + 8
    # one "row" of the interleaved tensor = 4 columns x 128 rows: tile0 of c0..c3, then tile1 of c0..c3
    want = np.repeat(np.array([0, 2, 4, 6, 1, 3, 5, 7]), 64)
    assert (got[0] == want).all() and (got[1] == want).all()


def test_preprocess_pipeline_is_composition_and_sm90_is_transpose():
    rng = np.random.default_rng(0)
    K, N = 128, 64
    for qt in (oracle.INT8_W, oracle.INT4_W):
        bits = 8 if qt == oracle.INT8_W else 4
        w = rng.integers(-128, 128, size=(K, N * bits // 8), dtype=np.int8)
        full = oracle.preprocess_weights_for_mixed_gemm(w, qt, arch=80)
        a = oracle.permute_B_rows_for_mixed_gemm(w, K, N, qt)
        b = oracle.subbyte_transpose(a, K, N, qt)
        c = oracle.interleave_column_major_tensor(b, K, N, qt)
        d = (oracle.add_bias_and_interleave_int8s if qt == oracle.INT8_W else oracle.add_bias_and_interleave_int4s)(c)
        assert (full.view(np.uint8).ravel() == d.ravel()).all()
        sm90 = oracle.preprocess_weights_for_mixed_gemm(w, qt, arch=90)
        if qt == oracle.INT8_W:
            assert (sm90.reshape(N, K) == w.T).all()
        else:
            un = oracle.unpack_int4_packed_tensor_to_int8(w)
            assert (oracle.unpack_int4_packed_tensor_to_int8(sm90.reshape(N, K // 2)) == un.T).all()


def test_symmetric_quantize_known_answers():
    # cutlass_preprocessors.cpp:646-717: scale = absmax / 2^(bits-1); round half away; clip
    w = torch.tensor([[1.0, -2.0], [0.5, 2.0], [-1.0, 0.25], [0.0625, -0.75]] * 16)  # [64, 2] -> pad cols
    w = torch.cat([w] * 32, dim=1)  # [64, 64]
    q8, s8 = oracle.symmetric_quantize(w, oracle.INT8_W)
    assert s8[0].item() == 1.0 / 128 and s8[1].item() == 2.0 / 128
    assert q8[0, 0] == 127 and q8[0, 1] == -128 and q8[1, 0] == 64 and q8[2, 0] == -128 and q8[3, 0] == 8
    q4, s4 = oracle.symmetric_quantize(w, oracle.INT4_W)
    u4 = oracle.unpack_int4_packed_tensor_to_int8(q4)
    assert s4[0].item() == 1.0 / 8 and u4[0, 0] == 7 and u4[0, 1] == -8 and u4[1, 0] == 4 and u4[3, 0] == 1  # 0.5 -> 1


def test_fp8_scale_known_answers():
    # tests/quantization/test_fp8_quantization.py:45-80: A=[[1,2,3],[2,4,6]]
    a = torch.tensor([[1, 2, 3], [2, 4, 6]], dtype=torch.float32)
    q, s = oracle.quantize_e4m3(a, "activation")
    assert torch.equal(s, torch.tensor([[3.0], [6.0]]) / 448.0)
    q, s = oracle.quantize_e4m3(a, "weight")
    assert torch.equal(s, torch.tensor([[2.0, 4.0, 6.0]]) / 448.0)
    q, s = oracle.quantize_e4m3(a, "per_tensor")
    assert torch.equal(s, torch.tensor([[6.0]]) / 448.0)
    # exact round trip on small integers whose row max is 7 (scale = 1/64), :117-123
    torch.manual_seed(42)
    b = torch.randint(0, 8, (16, 64)).float()
    b[:, 0] = 7
    qb, sb = oracle.quantize_e4m3(b, "activation")
    assert torch.equal(qb.float() * sb, b)
    assert oracle.fp8_scale(0.0) == np.float32(1.0 / (448 * 512))  # cudaFp8Utils.cu:215


def test_woq_tolerance_formula():
    ref = torch.tensor([[-4.0, 2.0]])
    assert oracle.woq_tolerance(ref, oracle.INT4_W) == 4.0 / 8 * 1.5      # _utils.py:99-109
    assert oracle.woq_tolerance(ref, oracle.INT8_W) == 4.0 / 128 * 1.5
    assert oracle.woq_tolerance(ref, oracle.INT4_W, torch.bfloat16) == 4.0 / 8 * 1.5 * 3


def test_dequant_single_vs_double_rounding():
    torch.manual_seed(0)
    K, N, g = 128, 64, 64
    q = np.random.default_rng(1).integers(-8, 8, size=(K, N), dtype=np.int8)
    for dt in (torch.float16, torch.bfloat16):
        scale = torch.rand(K // g, N, dtype=dt)
        zero = torch.randn(K // g, N, dtype=dt)
        w1 = oracle.dequant_fma(q, scale, zero, g)
        w2 = oracle.dequant_torch(q, scale, zero, g)
        # torch's own two-rounding reference (test_weight_only_groupwise_quant_matmul.py:190-195)
        ref = torch.from_numpy(q).to(dt) * scale.repeat_interleave(g, 0) + zero.repeat_interleave(g, 0)
        assert torch.equal(w2, ref)
        # the two definitions differ by at most one rounding of the intermediate q*scale
        qs = (torch.from_numpy(q).float() * scale.float().repeat_interleave(g, 0)).abs()
        ulp = torch.finfo(dt).eps * torch.maximum(qs, ref.float().abs()).clamp_min(1e-3)
        assert ((w1.float() - ref.float()).abs() <= ulp).all()
        # scale-only: both definitions coincide exactly (one rounding of an exact product)
        assert torch.equal(oracle.dequant_fma(q, scale, None, g), torch.from_numpy(q).to(dt) * scale.repeat_interleave(g, 0))


def test_smooth_quant_reference_matches_torch_formula():
    # tests/quantization/_utils.py:112-142 on the reference's own small shape family
    torch.manual_seed(0)
    m, n, k = 32, 96, 64
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    sa = torch.randint(1, 10, (m, 1)).float() * 1e-2
    sb = torch.randint(1, 10, (1, n)).float() * 1e-2
    for dt in (torch.float16, torch.float32, torch.int32):
        out, acc = oracle.smooth_quant_gemm(a.numpy(), b.numpy(), sa.numpy(), sb.numpy(), True, True, dt)
        ref_acc = a.to(torch.int32) @ b.to(torch.int32).t()
        assert (torch.from_numpy(acc) == ref_acc).all()
        ref = ref_acc.float() * (sa @ sb)
        ref = torch.round(ref).to(dt) if dt == torch.int32 else ref.to(dt)
        assert torch.equal(out, ref)


def test_quantize_per_token_matches_torch_reference():
    # _utils.py:145-150 (x*127/max).round().clip  vs kernel order x*(127/max): equal up to ties
    torch.manual_seed(0)
    x = torch.randn(8, 256, dtype=torch.float16)
    q, s = oracle.quantize_per_token(x)
    xf = x.float()
    xmax = xf.abs().amax(-1, keepdim=True)
    ref = (xf * 127.0 / xmax).round().clip(-128, 127).to(torch.int8)
    assert (np.abs(q.astype(np.int32) - ref.numpy().astype(np.int32)) <= 1).all()
    assert (q == ref.numpy()).mean() > 0.999
    assert np.allclose(s, (xmax / 127.0).numpy(), rtol=1e-6)


def test_decode_attention_reference_against_plain_softmax():
    # the oracle equals softmax(q K^T / sqrt(d)) V up to the +1e-6 in the denominator
    torch.manual_seed(42)
    B, Hq, Hkv, D, tpb = 2, 4, 2, 64, 16
    seq = [5, 37]
    cache = oracle.PagedKVCache(B, Hkv, D, 64, tpb, torch.float16)
    for b in range(B):
        for t in range(seq[b] - 1):
            for h in range(Hkv):
                cache.write(b, 0, h, t, torch.randn(D).half())
                cache.write(b, 1, h, t, torch.randn(D).half())
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D).half()
    out = oracle.decode_attention(qkv, cache, seq, Hq, Hkv, D)
    for b in range(B):
        k_new = qkv[b, Hq * D:(Hq + Hkv) * D].reshape(Hkv, D)
        v_new = qkv[b, (Hq + Hkv) * D:].reshape(Hkv, D)
        for h in range(Hq):
            hk = h // (Hq // Hkv)
            # cache was updated in place by the oracle: position seq-1 now holds the new token
            K = cache.gather(b, 0, hk, seq[b]).float()
            V = cache.gather(b, 1, hk, seq[b]).float()
            assert torch.equal(K[-1], k_new[hk].float()) and torch.equal(V[-1], v_new[hk].float())
            p = torch.softmax((K @ qkv[b, h * D:(h + 1) * D].float()) / D ** 0.5, 0)
            ref = p @ V
            assert torch.allclose(out[b, h * D:(h + 1) * D].float(), ref, atol=2e-3)


def test_c_decode_attention_agrees_with_python_oracle():
    # bench.py's cpu_baseline leg times the C / OpenMP loop; it must be the same math as oracle.decode_attention
    torch.manual_seed(0)
    B, H, Hkv, D, ctx, tpb = 3, 8, 2, 128, 300, 64
    for dt in (torch.bfloat16, torch.float16):
        cache = oracle.PagedKVCache(B, Hkv, D, ctx, tpb, dt)
        cache.pool.copy_((torch.randn(cache.pool.shape) * 0.5).to(dt))
        qkv = torch.randn(B, (H + 2 * Hkv) * D).to(dt)
        lens = [300, 17, 1]
        for rope in (None, "neox"):
            a = oracle.decode_attention(qkv, cache, lens, H, Hkv, D, rope=rope, rot_dim=D if rope else 0, update_cache=False)
            b = oracle.decode_attention_c(qkv, cache, lens, H, Hkv, D, rope=rope)
            assert (a.float() - b.float()).abs().max().item() <= (2e-3 if dt == torch.float16 else 8e-3)


def test_cyclic_map_values_from_the_header():
    """CyclicMap against values worked out by hand from kvCacheUtils.h:89-94, 109-117, 291-294, 308-316."""
    m = oracle.CyclicMap(128, 0, 64, 3)               # (3 - 1) * 64 >= 128: one more block, 128 + 64 cyclic slots
    assert (m.bubble, m.one_more, m.cyc_len) == (0, True, 192)
    assert [m.slot(t) for t in (0, 127, 128, 191, 192, 300)] == [0, 127, 128, 191, 0, 108]
    m = oracle.CyclicMap(128, 4, 64, 3)               # 4 sink tokens: bubble 60, no extra block (128 < 188)
    assert (m.bubble, m.one_more, m.cyc_len) == (60, False, 124)
    assert [m.slot(t) for t in (0, 3, 4, 127, 128, 130)] == [0, 3, 64, 187, 64, 66]
    m = oracle.CyclicMap(100, 10, linear=True)        # KVLinearBuffer: no bubble, window - sink cyclic slots
    assert (m.bubble, m.one_more, m.cyc_len) == (0, False, 90)
    assert [m.slot(t) for t in (9, 10, 99, 100, 190)] == [9, 10, 99, 10, 10]


def test_decode_attention_slots_is_decode_attention_without_a_window():
    """With window = capacity, no sink tokens and one beam the slot-space restatement must reproduce decode_attention
    bit for bit (same arithmetic, identity addressing)."""
    torch.manual_seed(1)
    B, Hq, Hkv, D, tpb, lens = 2, 4, 2, 64, 16, [9, 40]
    cache = oracle.PagedKVCache(B, Hkv, D, 64, tpb, torch.float16)
    slots = torch.zeros(B, 2, Hkv, 64, D, dtype=torch.float16)
    for b in range(B):
        for t in range(lens[b] - 1):
            for kv in range(2):
                for h in range(Hkv):
                    v = (torch.randn(D) * 0.5).half()
                    cache.write(b, kv, h, t, v)
                    slots[b, kv, h, t] = v
    qkv = torch.randn(B, (Hq + 2 * Hkv) * D).half()
    a = oracle.decode_attention(qkv, cache, lens, Hq, Hkv, D, rope="neox", rot_dim=D)
    m = oracle.CyclicMap(64, 0, tpb, 4)
    b_ = oracle.decode_attention_slots(qkv, slots, lens, Hq, Hkv, D, m, rope="neox", rot_dim=D)
    assert torch.equal(a, b_)
    for b in range(B):
        for h in range(Hkv):
            assert torch.equal(cache.gather(b, 0, h, lens[b]), slots[b, 0, h, :lens[b]])


def test_w4a8_awq_oracle_known_answers():
    """W4A8-AWQ restatement on values worked out by hand: e4m3 rounding of the smoothed activation (1.0625 -> 1.0,
    1.1875 -> 1.25 (tie to even mantissa), 500 -> 448 saturated), alpha before the bias."""
    act = torch.tensor([[1.0625, 1.1875, 500.0, -3.0] + [0.0] * 60], dtype=torch.float16)
    q = np.zeros((64, 16), np.int8)
    q[0, 0], q[1, 0], q[2, 0], q[3, 0] = 1, 1, 1, 2
    scale = torch.ones(1, 16, dtype=torch.float16)
    bias = torch.full((1, 16), 0.5, dtype=torch.float16)
    out16, out32 = oracle.woq_groupwise_matmul(act, None, q, scale, None, bias, 64, alpha=torch.tensor([0.5]), act_fp8=True)
    assert out32[0, 0].item() == (1.0 + 1.25 + 448.0 - 6.0) * 0.5 + 0.5
    assert out32[0, 1].item() == 0.5


def test_moe_route_ties_and_known_answers():
    # moeTopK (moe_kernels.cu:120-178): arg-max with the lower index winning ties (cub::ArgMax); scales are the softmax
    # probabilities of the chosen experts; experts of other ranks / finished rows get the sentinel
    g = torch.tensor([[0.25, 0.25, 0.25, 0.25], [1.0, 3.0, 3.0, 0.0], [0.0, 0.0, 5.0, 1.0]])
    s, e = oracle.moe_route(g, 2)
    assert e.tolist() == [[0, 1], [1, 2], [2, 3]]
    p = torch.softmax(g, dim=1)
    assert torch.allclose(s[0], p[0, :2]) and torch.allclose(s[2], torch.stack([p[2, 2], p[2, 3]]))
    # expert parallelism, rank 1 of 2: experts 2, 3 are local (0, 1), the others the sentinel 2; finished row 0
    s2, e2 = oracle.moe_route(g, 2, finished=torch.tensor([True, False, False]), start_expert=2, experts_per_node=2)
    assert e2.tolist() == [[2, 2], [2, 0], [0, 1]] and torch.equal(s, s2)


def test_run_moe_properties():
    # (i) identical experts + renormalised scales: the layer equals ONE expert's FFN whatever the routing says
    # (ii) top-1 with renormalisation: scale 1 (finalizeMoeRoutingKernelLauncherSelectBias, moe_kernels.cu:735-739)
    torch.manual_seed(0)
    E, h, inter, rows = 4, 32, 64, 24
    rng = np.random.default_rng(0)
    q1 = np.repeat(rng.integers(-127, 128, (1, h, inter), dtype=np.int8), E, 0)
    q2 = np.repeat(rng.integers(-127, 128, (1, inter, h), dtype=np.int8), E, 0)
    s1 = (torch.rand(1, inter) * 0.02).half().repeat(E, 1)
    s2 = (torch.rand(1, h) * 0.02).half().repeat(E, 1)
    x = (torch.randn(rows, h) * 0.5).half()
    gate = torch.randn(rows, E)
    one, _ = oracle.run_moe(x, torch.zeros(rows, 1), q1[:1], s1[:1], None, q2[:1], s2[:1], None, 1, "relu", True)
    for k in (1, 2, 4):
        out, mid = oracle.run_moe(x, gate, q1, s1, None, q2, s2, None, k, "relu", True)
        tol = 4 * 2.0 ** -10 * max(one.float().abs().max().item(), 1.0)  # the k-way sum is held in fp16
        assert (out.float() - one.float()).abs().max().item() <= tol
        assert int(mid["total_rows"][-1]) == rows * k
    # (iii) the sort is stable: within an expert the rows keep pair order token * k + j
    out, mid = oracle.run_moe(x, gate, q1, s1, None, q2, s2, None, 2, "relu", False)
    dest = mid["dest_of_pair"]
    keys = mid["experts"].reshape(-1)
    for e in range(E):
        d = dest[keys == e]
        assert torch.equal(d, torch.sort(d).values)


def test_context_attention_oracle_chunks_window_alibi():
    # chunked context == one shot; a window at least as long as the context == none; zero ALiBi slopes == none;
    # a sliding window really drops keys
    torch.manual_seed(1)
    Hq, Hkv, D, S = 4, 2, 32, 50
    rows = [(torch.randn(S, (Hq + 2 * Hkv) * D) * 0.5).half()]
    whole = oracle.context_attention(rows, [S], Hq, Hkv, D, rope="neox", rot_dim=D)[0]
    cache = oracle.PagedKVCache(1, Hkv, D, 64, 16, torch.float16)
    a = oracle.context_attention([rows[0][:20]], [20], Hq, Hkv, D, cache=cache, rope="neox", rot_dim=D)[0]
    b = oracle.context_attention([rows[0][20:]], [30], Hq, Hkv, D, cache=cache, rope="neox", rot_dim=D, past_lens=[20])[0]
    assert (torch.cat([a, b]).float() - whole.float()).abs().max().item() <= 2e-3
    same = oracle.context_attention(rows, [S], Hq, Hkv, D, rope="neox", rot_dim=D, window=S,
                                    alibi_slopes=torch.zeros(Hq).half())[0]
    assert torch.equal(same, whole)
    win = oracle.context_attention(rows, [S], Hq, Hkv, D, rope="neox", rot_dim=D, window=8)[0]
    assert torch.equal(win[:9], whole[:9]) and not torch.equal(win[9:], whole[9:])  # rows 0..8 see every earlier key
    # the cyclic append keeps the last `window` tokens at their slots (unfusedAttentionKernels_2_template.h:336-340)
    cache2 = oracle.PagedKVCache(1, Hkv, D, 32, 16, torch.float16)
    cmap = oracle.CyclicMap(16, 0, 16, cache2.max_blocks)
    oracle.context_attention(rows, [S], Hq, Hkv, D, cache=cache2, window=16, cmap=cmap)
    ref = oracle.PagedKVCache(1, Hkv, D, 64, 16, torch.float16)
    oracle.context_attention(rows, [S], Hq, Hkv, D, cache=ref)
    for t in range(S - 16, S):
        assert torch.equal(cache2.gather(0, 0, 0, 32)[cmap.slot(t)], ref.gather(0, 0, 0, S)[t])
