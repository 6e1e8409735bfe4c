"""Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py with the reference
tests' seeds and constructions).  CPU: the oracle still reproduces them bit for bit and the host
side of the C ABI produces the fixture layouts.  GPU: the HIP path, through the C ABI, against the
same committed data."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import oracle
from mi355q import _lib as L
from mi355q import functional as F

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def t16(a, dt):
    return torch.from_numpy(np.ascontiguousarray(a)).view(dt)


def test_oracle_reproduces_fixtures(tmp_path):
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    mg.HERE = str(tmp_path)
    for fn in (mg.layouts, mg.woq, mg.smooth_quant, mg.attention):
        fn()
    for f in ("layouts.npz", "woq.npz", "smooth_quant.npz", "attention.npz"):
        want, got = load(f), np.load(os.path.join(str(tmp_path), f))
        assert sorted(want.files) == sorted(got.files)
        for k in want.files:
            assert np.array_equal(want[k], got[k]), f"{f}:{k} drifted"


def test_layout_fixtures_internal_consistency():
    g = load("layouts.npz")
    for name in ("int8", "int4"):
        assert np.array_equal(g[f"{name}_sm80"], g[f"{name}_sm80_direct"])  # step pipeline == one-shot entry


@pytest.mark.parametrize("name,tq", [("int8", torch.int8), ("int4", torch.quint4x2)])
def test_host_abi_layouts_match_fixtures(name, tq):
    g = load("layouts.npz")
    w = torch.from_numpy(g[f"{name}_w"])
    for layout, key in ((L.LAYOUT_REF_SM80, "sm80"), (L.LAYOUT_REF_SM90, "sm90"), (L.LAYOUT_GFX950, "gfx950")):
        got = F.preprocess_weights_for_mixed_gemm(w, tq, layout).numpy()
        assert np.array_equal(got, g[f"{name}_{key}"]), f"{name} layout {key}"
        assert np.array_equal(F.unprocess_weights_for_mixed_gemm(torch.from_numpy(got), tq, layout).numpy(), g[f"{name}_w"])
    if name == "int4":
        assert np.array_equal(F.unpack_int4_packed_tensor_to_int8(w).numpy(), g["int4_unpacked"])


# ----------------------------------------------------------------------------- GPU: HIP path vs fixtures
@pytest.mark.gpu
@pytest.mark.parametrize("dt_name,dt", [("f16", torch.float16), ("bf16", torch.bfloat16)])
def test_gpu_woq_vs_fixtures(cuda, dt_name, dt):
    g = load("woq.npz")
    dstr = "float16" if dt == torch.float16 else "bfloat16"
    d = lambda t: None if t is None else t.to(cuda)
    i = 0
    while f"{dt_name}_{i}_meta" in g.files:
        key = f"{dt_name}_{i}"
        m, n, k, gs, hz, hb, hp = [int(v) for v in g[key + "_meta"]]
        act, scale = t16(g[key + "_act"], dt), t16(g[key + "_scale"], dt)
        zero = t16(g[key + "_zero"], dt) if hz else None
        bias = t16(g[key + "_bias"], dt) if hb else None
        pre = t16(g[key + "_pre"], dt) if hp else None
        w = torch.from_numpy(g[key + "_w"])
        wp = F.preprocess_weights_for_mixed_gemm(w, torch.quint4x2).view(dt)
        algo = hp * F.PRE_QUANT_SCALE + hz * F.ZERO + hb * F.BIAS
        out = F.weight_only_groupwise_quant_matmul(d(act), d(pre), d(wp), d(scale), d(zero), d(bias), None, algo, gs, dstr)
        wd = F.dequantize_weights(d(wp), d(scale), d(zero), k, n, gs, L.WEIGHT_INT4, dt)
        torch.cuda.synchronize()
        assert np.array_equal(wd.cpu().view(torch.int16).numpy(), g[key + "_dequant"]), "dequantised weights not bit-exact"
        ref32 = torch.from_numpy(g[key + "_ref32"])
        ref16 = t16(g[key + "_ref16"], dt)
        torch.testing.assert_close(out.cpu().float(), ref16.float(), atol=oracle.woq_tolerance(ref16, oracle.INT4_W), rtol=1e-7)
        assert (out.cpu().float() - ref32).abs().max().item() <= 2.0 * torch.finfo(dt).eps * ref32.abs().max().item() + 1e-3
        i += 1
    assert i == 4
    # per-channel int8
    act = t16(g[f"{dt_name}_pc_act"], dt)
    q8 = torch.from_numpy(g[f"{dt_name}_pc_q"])
    s8 = t16(g[f"{dt_name}_pc_scale"], dt)
    wp = F.preprocess_weights_for_mixed_gemm(q8, torch.int8)
    out = F.weight_only_quant_matmul(d(act), d(wp), d(s8), L.WEIGHT_INT8, dstr)
    torch.cuda.synchronize()
    ref32 = torch.from_numpy(g[f"{dt_name}_pc_ref32"])
    assert (out.cpu().float() - ref32).abs().max().item() <= 2.0 * torch.finfo(dt).eps * ref32.abs().max().item() + 1e-3


@pytest.mark.gpu
def test_gpu_smooth_quant_and_quantizers_vs_fixtures(cuda):
    g = load("smooth_quant.npz")
    a, b = torch.from_numpy(g["a"]).to(cuda), torch.from_numpy(g["b"]).to(cuda)
    st, sc = torch.from_numpy(g["scale_tokens"]), torch.from_numpy(g["scale_channels"])
    for pt in (0, 1):
        for pc in (0, 1):
            s_t = (st if pt else st[:1]).to(cuda)
            s_c = (sc if pc else sc[:, :1]).to(cuda)
            out = F.smooth_quant_gemm(a, b, s_t, s_c, bool(pt), bool(pc), "float16")
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().view(torch.int16).numpy(), g[f"out_f16_pt{pt}_pc{pc}"]), (pt, pc)
    acc = F.smooth_quant_gemm(a, b, torch.ones(1, 1, device=cuda), torch.ones(1, 1, device=cuda), False, False, "int32")
    torch.cuda.synchronize()
    assert np.array_equal(acc.cpu().numpy(), g["acc"]), "int32 accumulators must be exact"
    q, s = F.quantize_per_token(t16(g["ptq_x"], torch.float16).to(cuda))
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), g["ptq_q"]) and np.array_equal(s.cpu().numpy().reshape(-1, 1), g["ptq_scale"])
    qf, sf = F.quantize_e4m3(t16(g["fp8_x"], torch.float16).to(cuda), "activation")
    torch.cuda.synchronize()
    assert np.array_equal(qf.cpu().view(torch.uint8).numpy(), g["fp8_q"])
    assert np.array_equal(sf.cpu().numpy().reshape(g["fp8_scale"].shape), g["fp8_scale"])
    kat = torch.tensor([[1., 2., 3.], [2., 4., 6.]], dtype=torch.float16, device=cuda)
    for mode in ("activation", "weight", "per_tensor"):
        qk, sk = F.quantize_e4m3(kat, mode)
        torch.cuda.synchronize()
        assert np.array_equal(sk.cpu().numpy().reshape(-1), g[f"fp8_kat_{mode}_scale"].reshape(-1)), mode
        assert np.array_equal(qk.cpu().view(torch.uint8).numpy(), g[f"fp8_kat_{mode}_q"]), mode


@pytest.mark.gpu
@pytest.mark.parametrize("name,dt,kvdt,rope", [("mha_bf16", torch.bfloat16, torch.bfloat16, None),
                                                ("gqa_int8", torch.float16, torch.int8, "neox"),
                                                ("mqa_fp8", torch.bfloat16, torch.float8_e4m3fn, None)])
def test_gpu_attention_vs_fixtures(cuda, name, dt, kvdt, rope):
    from test_gpu_attention import run_plugin, ATOL
    g = load("attention.npz")
    Hq, Hkv, D, tpb, *seq_lens = [int(v) for v in g[f"{name}_meta"]]
    kv_scale = float(g[f"{name}_kv_scale"])
    cache = oracle.PagedKVCache(len(seq_lens), Hkv, D, max(seq_lens) + 24, tpb, kvdt, seed=42)
    assert np.array_equal(cache.block_ids.numpy(), g[f"{name}_block_ids"])
    raw = torch.from_numpy(np.ascontiguousarray(g[f"{name}_pool_before"]))
    cache.pool = raw.view(kvdt).reshape(cache.pool.shape).clone()
    qkv = t16(g[f"{name}_qkv"], dt)
    out, pool_after = run_plugin(cache, qkv, seq_lens, Hq, Hkv, D, tpb, dt, kvdt, cuda, rope=rope, kv_scale=kv_scale)
    ref = t16(g[f"{name}_out"], dt)
    assert (out.float() - ref.float()).abs().max().item() <= ATOL[dt]
    want_pool = torch.from_numpy(np.ascontiguousarray(g[f"{name}_pool_after"])).view(torch.uint8).reshape(-1)
    got_pool = pool_after.contiguous().view(torch.uint8).reshape(-1)
    if rope is None:
        assert torch.equal(got_pool, want_pool), "appended K/V cache lines differ from the fixture"
    else:  # RoPE through sin/cos: at most one int8 code on the appended K line
        assert (got_pool.view(torch.int8).int() - want_pool.view(torch.int8).int()).abs().max().item() <= 1
