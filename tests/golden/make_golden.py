"""Generates the committed fixtures of tests/golden/ with the CPU oracle (oracle/), using the
seeds and input constructions of the reference's own tests (the reference holds no
numeric golden files for these ops -- its tests recompute every expected value with torch -- and it
cannot be imported here, so the fixtures are made by the pinned restatement).

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz

16-bit tensors are stored as their int16 bit patterns, fp8 as uint8.  The GFX950 weight layout is
not a reference layout; its fixture comes from the independent numpy restatement below
(`gfx950_layout_numpy`), not from the product code.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def bits16(t):
    return t.contiguous().view(torch.int16).numpy()


def gfx950_layout_numpy(q_unpacked, bits):
    """DESIGN.md section 3, restated: q_unpacked int8 [K, N] -> layout bytes (uint8, K*N*bits/8)."""
    K, N = q_unpacked.shape
    out = np.zeros(K * N * bits // 8, np.uint8)
    run = K * 16 * bits // 8
    for k in range(K):
        for n in range(N):
            nb, c = divmod(n, 16)
            v = int(q_unpacked[k, n])
            if bits == 4:
                kt, r = divmod(k, 128)
                s, h, j = r // 32, (r % 32) // 8, r % 8
                lane = h * 16 + c
                half = kt == K // 128
                byte = nb * run + kt * 1024 + (lane * 8 if half else lane * 16) + s * 4
                pos = 2 * j if j < 4 else 2 * (j - 4) + 1
                out[byte + pos // 2] |= ((v + 8) & 0xF) << (4 * (pos & 1))
            else:
                kt, r = divmod(k, 64)
                s, h, j = r // 32, (r % 32) // 8, r % 8
                lane = h * 16 + c
                out[nb * run + kt * 1024 + lane * 16 + (s * 2 + j // 4) * 4 + j % 4] = v & 0xFF
    return out


def layouts():
    rng = np.random.default_rng(20231205)  # weightOnlyKernelTest.cpp:432 seeds mt19937(20231205)
    out = {}
    for name, qt, bits, K, N in (("int8", oracle.INT8_W, 8, 128, 64), ("int4", oracle.INT4_W, 4, 192, 64)):
        w = rng.integers(-128, 128, size=(K, N * bits // 8), dtype=np.int8)
        un = oracle.unpack_int4_packed_tensor_to_int8(w) if bits == 4 else w
        p = oracle.permute_B_rows_for_mixed_gemm(w, K, N, qt)
        t = oracle.subbyte_transpose(p, K, N, qt)
        il = oracle.interleave_column_major_tensor(t, K, N, qt)
        fin = oracle.add_bias_and_interleave_int4s(il) if bits == 4 else oracle.add_bias_and_interleave_int8s(il)
        out.update({f"{name}_w": w, f"{name}_unpacked": un, f"{name}_permuted": p, f"{name}_transposed": t,
                    f"{name}_interleaved": il, f"{name}_sm80": fin.reshape(w.shape).view(np.int8),
                    f"{name}_sm80_direct": oracle.preprocess_weights_for_mixed_gemm(w, qt, arch=80),
                    f"{name}_sm90": oracle.preprocess_weights_for_mixed_gemm(w, qt, arch=90),
                    f"{name}_gfx950": gfx950_layout_numpy(un, bits).view(np.int8).reshape(w.shape)})
    np.savez_compressed(os.path.join(HERE, "layouts.npz"), **out)


def woq():
    """test_weight_only_groupwise_quant_matmul.py:141-210 construction, torch.manual_seed(0)."""
    out = {}
    cases = [(1, 256, 128, 64, True, True, True), (3, 256, 384, 128, True, False, True),
             (16, 256, 128, 128, False, True, False), (5, 256, 192, 64, True, True, False)]
    for dt_name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        for i, (m, n, k, g, has_zero, has_bias, has_pre) in enumerate(cases):
            torch.manual_seed(0)
            act = torch.randn(m, k, dtype=dt)
            bias = torch.randn(1, n, dtype=dt) if has_bias else None
            zero = torch.randn(k // g, n, dtype=dt) if has_zero else None
            scale = torch.rand(k // g, n, dtype=dt)
            pre = torch.rand(1, k, dtype=dt) if has_pre else None
            w = torch.randint(-2 ** 31, 2 ** 31, (k, n // 8), dtype=torch.int32).view(torch.int8)
            q = oracle.unpack_int4_packed_tensor_to_int8(w.numpy())
            ref16, ref32 = oracle.woq_groupwise_matmul(act, pre, q, scale, zero, bias, g)
            key = f"{dt_name}_{i}"
            out[key + "_meta"] = np.array([m, n, k, g, has_zero, has_bias, has_pre], np.int32)
            out[key + "_act"] = bits16(act)
            out[key + "_w"] = w.numpy()
            out[key + "_scale"] = bits16(scale)
            if has_zero:
                out[key + "_zero"] = bits16(zero)
            if has_bias:
                out[key + "_bias"] = bits16(bias)
            if has_pre:
                out[key + "_pre"] = bits16(pre)
            out[key + "_dequant"] = bits16(oracle.dequant_fma(q, scale, zero, g))
            out[key + "_ref16"] = bits16(ref16)
            out[key + "_ref32"] = ref32.numpy()
        # per-channel int8 (test_weight_only_quant_matmul.py:99-119)
        torch.manual_seed(0)
        m, n, k = 4, 128, 256
        act = torch.randn(m, k, dtype=dt)
        wf = (torch.rand(k, n) * 2 - 1).to(dt)
        q8, s8 = oracle.symmetric_quantize(wf, oracle.INT8_W)
        r16, r32 = oracle.woq_matmul(act, q8, s8)
        out.update({f"{dt_name}_pc_act": bits16(act), f"{dt_name}_pc_wf": bits16(wf), f"{dt_name}_pc_q": q8,
                    f"{dt_name}_pc_scale": bits16(s8), f"{dt_name}_pc_ref16": bits16(r16), f"{dt_name}_pc_ref32": r32.numpy()})
    np.savez_compressed(os.path.join(HERE, "woq.npz"), **out)


def smooth_quant():
    """test_smooth_quant_gemm.py:43-62 construction (scaled-down qkv shape), seed 0."""
    torch.manual_seed(0)
    m, n, k = 32, 256, 768
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    st = torch.randint(1, 10, (m, 1)).float() * 1e-2
    sc = torch.randint(1, 10, (1, n)).float() * 1e-2
    out = dict(a=a.numpy(), b=b.numpy(), scale_tokens=st.numpy(), scale_channels=sc.numpy())
    for pt in (0, 1):
        for pc in (0, 1):
            s_t = st if pt else st[:1]
            s_c = sc if pc else sc[:, :1]
            o16, acc = oracle.smooth_quant_gemm(a.numpy(), b.numpy(), s_t.numpy(), s_c.numpy(), pt, pc, torch.float16)
            out[f"out_f16_pt{pt}_pc{pc}"] = bits16(o16)
    out["acc"] = acc
    x = torch.randn(16, 512).to(torch.float16)
    q, s = oracle.quantize_per_token(x)
    out.update(ptq_x=bits16(x), ptq_q=q, ptq_scale=s)
    # fp8 KAT of test_fp8_quantization.py:45-80 plus a random activation
    kat = torch.tensor([[1., 2., 3.], [2., 4., 6.]])
    for mode in ("activation", "weight", "per_tensor"):
        qf, sf = oracle.quantize_e4m3(kat, mode)
        out[f"fp8_kat_{mode}_q"] = qf.view(torch.uint8).numpy()
        out[f"fp8_kat_{mode}_scale"] = sf.numpy()
    xr = torch.randn(8, 64).to(torch.float16)
    qf, sf = oracle.quantize_e4m3(xr, "activation")
    out.update(fp8_x=bits16(xr), fp8_q=qf.view(torch.uint8).numpy(), fp8_scale=sf.numpy())
    np.savez_compressed(os.path.join(HERE, "smooth_quant.npz"), **out)


def attention():
    """B=2, ragged ctx {130, 5}, Dh 128, tokens_per_block 64 (test_gpt_attention.py:660-727 shapes,
    int8 scale :941-956)."""
    out = {}
    cases = [("mha_bf16", 4, 4, torch.bfloat16, torch.bfloat16, None), ("gqa_int8", 4, 2, torch.float16, torch.int8, "neox"),
             ("mqa_fp8", 4, 1, torch.bfloat16, torch.float8_e4m3fn, None)]
    B, D, tpb, seq_lens = 2, 128, 64, [130, 5]
    for name, Hq, Hkv, dt, kvdt, rope in cases:
        torch.manual_seed(42)
        cache = oracle.PagedKVCache(B, Hkv, D, max(seq_lens) + 24, tpb, kvdt, seed=42)
        kv_scale = 4.0 / 127.0 if kvdt == torch.int8 else 1.0
        for b in range(B):
            n = seq_lens[b] - 1
            kv = torch.randn(2, Hkv, n, D) * 0.5
            q = oracle.quantize_kv(kv.to(dt), kvdt, 1.0 / kv_scale)
            nb = (n + tpb - 1) // tpb
            pad = torch.zeros(2, Hkv, nb * tpb, D, dtype=kvdt)
            pad[:, :, :n] = q
            cache.pool[cache.block_ids[b, :nb]] = pad.reshape(2, Hkv, nb, tpb, D).permute(2, 0, 1, 3, 4)
        qkv = torch.randn(B, (Hq + 2 * Hkv) * D).to(dt)
        pool_before = cache.pool.clone()
        ref = oracle.decode_attention(qkv, cache, seq_lens, Hq, Hkv, D, rope=rope, rot_dim=D if rope else 0,
                                      kv_scale_orig_quant=1.0 / kv_scale, kv_scale_quant_orig=kv_scale)
        raw = (lambda t: t.view(torch.uint8).numpy()) if kvdt != torch.bfloat16 else bits16
        out.update({f"{name}_meta": np.array([Hq, Hkv, D, tpb] + seq_lens, np.int32),
                    f"{name}_kv_scale": np.float32(kv_scale),
                    f"{name}_qkv": bits16(qkv), f"{name}_block_ids": cache.block_ids.numpy(),
                    f"{name}_pool_before": raw(pool_before), f"{name}_pool_after": raw(cache.pool),
                    f"{name}_out": bits16(ref)})
    np.savez_compressed(os.path.join(HERE, "attention.npz"), **out)


if __name__ == "__main__":
    oracle.build()
    layouts()
    woq()
    smooth_quant()
    attention()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
