#!/usr/bin/env python3
"""bench.py -- headline benchmark of the quantized-inference hot path on MI355X.

Workload at N=1 (BASELINE.json configs[1], the config the metric is quoted on): masked-MHA decode,
Llama-2-7B shape (32 q heads = 32 kv heads, Dh 128), batch 64, ctx 2048, paged bf16 KV
(tokens_per_block 64), RoPE + cache append fused, through the GPTAttention plugin enqueue of the
C ABI.  One "step" = one enqueue over the whole batch (one layer's decode attention).  K/V for one
step is 2.15 GB, far beyond the 256 MB Infinity Cache, so the rate is an HBM rate.
`value` = algorithmic bytes of all ranks / max-over-ranks time; inputs are
resident in HBM before the timed region.  Secondary numbers of the same metric family (W4A16 GEMV
GB/s with >= 64 rotating weight sets, SmoothQuant int8 GEMM TOP/s) ride along under "secondary".

N>1: attention shards by KV head / sequence with no collective -> every rank runs the
same per-GPU workload on its own cache, weak scaling, no data-path collective in the headline.  The path's one real
exchange step -- the all-reduce behind a row-parallel GEMM -- rides along as `secondary.tp_row_parallel_gemm_allreduce`
(custom push all-reduce and RCCL, all ranks take part).
"""
import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (_ROOT, os.path.join(_ROOT, "tensorrt-llm_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured copy)
INT8_PEAK_TOPS = 5000.0    # dense int8 MFMA = 2x bf16 (2.5 PF)


def build_attention_case(dev, B=64, H=32, D=128, ctx=2048, tpb=64, seed=42):
    from mi355q import functional as F
    g = torch.Generator(device="cpu").manual_seed(seed)
    nblk = B * (ctx // tpb)
    pool = (torch.randn(nblk, 2, H, tpb, D, generator=g, dtype=torch.float32) * 0.5).to(torch.bfloat16).to(dev)
    perm = torch.randperm(nblk, generator=g).reshape(B, ctx // tpb)   # shuffled block table
    be = H * tpb * D * 2
    ptrs = (torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + pool.data_ptr()).reshape(1, B, 2, -1)
    qkv = torch.randn(B, 1, 3 * H * D, generator=g).to(torch.bfloat16).to(dev)
    seq = torch.full((B,), ctx, dtype=torch.int32)
    plg = F.make_gpt_attention_plugin(num_heads=H, num_kv_heads=H, head_size=D, dtype="bfloat16",
                                      position_embedding_type=2, rotary_embedding_dim=D, tokens_per_block=tpb)
    inputs = [qkv, seq.to(dev), seq - 1, torch.tensor([ctx], dtype=torch.int32), torch.zeros(1, dtype=torch.int32),
              (seq - 1).to(dev), torch.zeros(B, 1, ctx, dtype=torch.int32, device=dev),
              torch.ones(B, dtype=torch.int32), ptrs.to(dev), ptrs]
    out = torch.empty(B, 1, H * D, dtype=torch.bfloat16, device=dev)
    plg.configure(inputs)
    ws_bytes = plg.get_workspace_size(inputs, [out])
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    # K+V read + qkv in + out + appended k,v
    algo_bytes = 2 * B * H * ctx * D * 2 + B * 3 * H * D * 2 + B * H * D * 2 + 2 * B * H * D * 2
    keep = (pool, ptrs)
    return plg, inputs, out, ws, algo_bytes, keep


def bench_decode_attention(dev, steps, B=64, Hq=8, Hkv=1, D=128, ctx=2048, tpb=64, kv="int8", ragged=False, sets=1):
    """Other decode-attention configs of the metric family: cfg4 (Llama-2-70B TP=8 per-GPU slice: 8 q heads on one
    int8 kv head), cfg5 (fp8 KV, 32 q / 8 kv heads), GQA bf16, and cfg2's ragged variant (lengths in [1024, 2048]).
    `sets` independent caches are cycled when one step's K/V would fit the 256 MB Infinity Cache."""
    from mi355q import functional as F
    g = torch.Generator(device="cpu").manual_seed(7)
    kvdt = {"bf16": torch.bfloat16, "int8": torch.int8, "fp8": torch.float8_e4m3fn}[kv]
    esz = 2 if kv == "bf16" else 1
    nblk = B * (ctx // tpb)
    lens = torch.randint(1024, ctx + 1, (B,), generator=g, dtype=torch.int32) if ragged else torch.full((B,), ctx, dtype=torch.int32)
    cases = []
    for _ in range(sets):
        if kv == "bf16":
            pool = (torch.randn(nblk, 2, Hkv, tpb, D, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        elif kv == "int8":
            pool = torch.randint(-127, 128, (nblk, 2, Hkv, tpb, D), generator=g, dtype=torch.int8).to(dev)
        else:
            pool = (torch.randn(nblk, 2, Hkv, tpb, D, generator=g) * 0.5).to(torch.float8_e4m3fn).to(dev)
        perm = torch.randperm(nblk, generator=g).reshape(B, ctx // tpb)
        be = Hkv * tpb * D * esz
        ptrs = (torch.stack([perm * 2 * be, perm * 2 * be + be], 1) + pool.data_ptr()).reshape(1, B, 2, -1)
        cases.append((pool, ptrs.to(dev), ptrs))
    qkv = torch.randn(B, 1, (Hq + 2 * Hkv) * D, generator=g).to(torch.bfloat16).to(dev)
    qm = {"bf16": 0, "int8": 1 << 6, "fp8": 1 << 7}[kv]
    plg = F.make_gpt_attention_plugin(num_heads=Hq, num_kv_heads=Hkv, head_size=D, dtype="bfloat16",
                                      position_embedding_type=2, rotary_embedding_dim=D, tokens_per_block=tpb,
                                      kv_cache_quant_mode=qm)
    sc = [torch.tensor([32.0], device=dev), torch.tensor([1 / 32.0], device=dev)] if qm else []
    out = torch.empty(B, 1, Hq * D, dtype=torch.bfloat16, device=dev)

    def inputs_of(c):
        return [qkv, lens.to(dev), lens - 1, torch.tensor([ctx], dtype=torch.int32), torch.zeros(1, dtype=torch.int32),
                (lens - 1).to(dev), torch.zeros(B, 1, ctx, dtype=torch.int32, device=dev),
                torch.ones(B, dtype=torch.int32), c[1], c[2]] + sc
    ins = [inputs_of(c) for c in cases]
    plg.configure(ins[0])
    ws_bytes = plg.get_workspace_size(ins[0], [out])
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    it = [0]

    def step():
        plg.enqueue(ins[it[0] % sets], [out], ws)
        it[0] += 1
    # one hipGraph of `sets` enqueues (as for the GEMV): at 15-60 us per kernel the Python enqueue (~25 us) would
    # otherwise be what is measured; falls back to eager launches if the capture is refused
    launches = "eager"
    for _ in range(max(2, sets)):
        step()
    torch.cuda.synchronize()
    try:
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            per_graph = sets * max(1, -(-8 // sets))  # >= 8 enqueues per graph: a graph launch costs ~25 us itself
            with torch.cuda.graph(graph, stream=side):
                for _ in range(per_graph):
                    step()
        reps = max(1, steps // per_graph)
        _, dev_s = time_steps(graph.replay, reps, 2, lambda: None)
        steps = reps * per_graph
        launches = "hipGraph"
    except Exception:
        torch.cuda.synchronize()
        _, dev_s = time_steps(step, steps, max(2, sets), lambda: None)
    tok = int(lens.sum())
    algo = 2 * Hkv * tok * D * esz + B * (Hq + 2 * Hkv) * D * 2 + B * Hq * D * 2 + 2 * B * Hkv * D * esz
    return {"batch": B, "q_heads": Hq, "kv_heads": Hkv, "head_dim": D, "ctx": "1024..2048 ragged" if ragged else ctx,
            "kv_cache": kv, "us_per_call": dev_s / steps * 1e6, "GBps": algo / (dev_s / steps) / 1e9,
            "frac_of_8TBps": algo / (dev_s / steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo, "cache_sets": sets,
            "launches": launches}


def trace(msg):
    """Stage markers on stderr (MI355Q_BENCH_TRACE=1): where a multi-rank run stands."""
    if os.environ.get("MI355Q_BENCH_TRACE") == "1":
        print(f"[bench rank {os.environ.get('RANK', '0')}] {msg}", file=sys.stderr, flush=True)


def time_steps(fn, steps, warmup, barrier):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    barrier()
    wall = time.perf_counter() - t0
    return wall, e0.elapsed_time(e1) / 1e3   # seconds: host wall, device (HIP events on the launch stream)


def bench_gemv(dev, steps, m=1, n=4096, k=4096, g=128, sets=72, pre_quant=False):
    """cfg1: W4A16 AWQ GEMV with zeros + pre-quant scale; `sets` independent weight sets (> 256 MB
    in total) are cycled so the rate is HBM, not Infinity Cache."""
    from mi355q import _lib as L
    from mi355q import functional as F
    import ctypes as C
    torch.manual_seed(0)
    dt = torch.bfloat16
    w = torch.randint(-128, 128, (sets, k, n // 2), dtype=torch.int8, device=dev)   # any bytes are valid int4 tiles
    scale = torch.rand(sets, k // g, n, dtype=dt, device=dev)
    zero = torch.randn(sets, k // g, n, dtype=dt, device=dev)
    pre = torch.rand(1, k, dtype=dt, device=dev)
    act = torch.randn(m, k, dtype=dt, device=dev)
    out = torch.empty(m, n, dtype=dt, device=dev)
    ws_bytes = L.lib().mi355q_weight_only_gemm_workspace_size(m, n, k, L.WEIGHT_INT4)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    params = []
    for s in range(sets):
        p = L.WeightOnlyParams()
        p.qweight, p.scales, p.zeros = w[s].data_ptr(), scale[s].data_ptr(), zero[s].data_ptr()
        p.in_, p.act_scale, p.bias, p.out = act.data_ptr(), (pre.data_ptr() if pre_quant else None), None, out.data_ptr()
        p.m, p.n, p.k, p.group_size = m, n, k, g
        p.weight_type, p.act_type = L.WEIGHT_INT4, L.BF16
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes
        params.append(p)
    stream = torch.cuda.current_stream().cuda_stream
    fn = L.lib().mi355q_weight_only_gemm
    state = {"i": 0}

    def step():
        rc = fn(C.byref(params[state["i"] % sets]), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        state["i"] += 1
        assert rc == 0, L.lib().mi355q_last_error()

    # one hipGraph holding `sets` back-to-back launches (one per weight set): removes the host launch
    # cost (~3.5 us per launch, more than the kernel itself) from the measurement
    for _ in range(sets):
        step()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        stream = side.cuda_stream
        with torch.cuda.graph(graph, stream=side):
            for _ in range(sets):
                step()
    reps = max(1, steps // sets)
    _, dev_s = time_steps(graph.replay, reps, 2, lambda: None)
    steps = reps * sets
    algo = k * n // 2 + 2 * (k // g) * n * 2 + (k * 2 if pre_quant else 0) + 2 * m * (k + n)
    return {"m": m, "n": n, "k": k, "flags": "ZERO+PRE_QUANT_SCALE (scale fused into the activation staging)" if pre_quant else "ZERO", "us_per_call": dev_s / steps * 1e6, "GBps": algo / (dev_s / steps) / 1e9,
            "frac_of_8TBps": algo / (dev_s / steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo,
            "weight_sets": sets}


def bench_int8_gemm(dev, steps, m=4096, n=13824, k=5120):
    """cfg3: SmoothQuant W8A8 GEMM, Llama-2-13B gate/up shape, per-token x per-channel, fp16 out."""
    from mi355q import functional as F
    torch.manual_seed(0)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sa = (torch.randint(1, 10, (m, 1), device=dev).float() * 1e-2)
    sb = (torch.randint(1, 10, (1, n), device=dev).float() * 1e-2)
    plg = F.Plugin("SmoothQuantGemm", [("has_per_channel_scaling", 1, 5), ("has_per_token_scaling", 1, 5),
                                       ("type_id", 1, 5)])
    out = torch.empty(m, n, dtype=torch.float16, device=dev)
    ins = [a, b, sa, sb]
    plg.configure(ins)
    ws_bytes = plg.get_workspace_size(ins, [out])      # the split-K-remainder kernel's partial tiles (TensorRT's scratch)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev) if ws_bytes else None
    _, dev_s = time_steps(lambda: plg.enqueue(ins, [out], ws), steps, 2, lambda: None)
    ops = 2.0 * m * n * k
    return {"m": m, "n": n, "k": k, "ms_per_call": dev_s / steps * 1e3, "TOPS": ops / (dev_s / steps) / 1e12,
            "frac_of_int8_peak": ops / (dev_s / steps) / 1e12 / INT8_PEAK_TOPS}


def bench_fp8_gemm(dev, steps, m=4096, n=14336, k=4096):
    """cfg5: FP8 e4m3 GEMM, per-tensor scales, Mixtral-8x7B expert shapes (w1/w3 and w2), bf16 out."""
    from mi355q import functional as F
    torch.manual_seed(0)
    a = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn)
    b = torch.randn(n, k, device=dev).to(torch.float8_e4m3fn)
    sa = torch.tensor([0.5], device=dev)
    sb = torch.tensor([0.25], device=dev)
    _, dev_s = time_steps(lambda: F.fp8_gemm(a, b, sa, sb, "bfloat16"), steps, 2, lambda: None)
    ops = 2.0 * m * n * k
    return {"m": m, "n": n, "k": k, "ms_per_call": dev_s / steps * 1e3, "TFLOPS": ops / (dev_s / steps) / 1e12,
            "frac_of_fp8_peak": ops / (dev_s / steps) / 1e12 / INT8_PEAK_TOPS}


def bench_moe_gemm(dev, steps, tokens=64, top_k=2, experts=8, n=28672, k=4096):
    """§8(f) rank 4: the expert GEMMs of a Mixtral-8x7B MoE layer (w1|w3 fused: 4096 -> 2 x 14336) with int4
    weight-only experts, one grouped launch; rows spread evenly over the experts (synthetic routing)."""
    from mi355q import functional as F
    from mi355q import _lib as L
    torch.manual_seed(0)
    dt = torch.bfloat16
    rows = tokens * top_k
    per = [rows // experts + (1 if e < rows % experts else 0) for e in range(experts)]
    cum = torch.tensor(per, dtype=torch.int64).cumsum(0).to(dev)
    w = torch.randint(-128, 128, (experts, k, n // 2), dtype=torch.int8, device=dev)  # preprocessed bytes: any pattern
    sc = (torch.rand(experts, n, device=dev) * 0.01).to(dt)
    a = (torch.randn(rows, k, device=dev) * 0.5).to(dt)
    _, dev_s = time_steps(lambda: F.moe_gemm_bias_act(a, w, sc, None, cum, L.WEIGHT_INT4), steps, 2, lambda: None)
    t = dev_s / steps
    flops = 2.0 * rows * n * k
    wbytes = experts * n * k // 2 + experts * n * 2
    algo = wbytes + rows * k * 2 + rows * n * 2
    return {"tokens": tokens, "top_k": top_k, "experts": experts, "rows": rows, "n": n, "k": k, "weights": "int4 per-column",
            "us_per_call": t * 1e6, "TFLOPS": flops / t / 1e12, "GBps": algo / t / 1e9,
            "frac_of_8TBps": algo / t / 1e9 / HBM_PEAK_GBS, "frac_of_bf16_peak": flops / t / 1e12 / (INT8_PEAK_TOPS / 2),
            "algorithmic_bytes": algo}


def bench_moe_layer(dev, steps, tokens=64, top_k=2, experts=8, hidden=4096, inter=14336):
    """§8(f) rank 4: a whole Mixtral-8x7B MoE layer with int4 weight-only experts through mi355q_run_moe: routing
    (softmax + top-2, renormalised), sort by expert, gather, w1|w3 GEMM (4096 -> 2 x 14336), SwiGLU, w2 GEMM
    (14336 -> 4096), finalize.  Random router logits.  Flops: the two GEMMs; bytes: the expert weights + scales (every
    expert is hit at these token counts) + the activations in and out of each step."""
    from mi355q import _lib as L
    import ctypes as C
    torch.manual_seed(0)
    dt = torch.bfloat16
    w1 = torch.randint(-128, 128, (experts, hidden, inter), dtype=torch.int8, device=dev)       # [E, K, 2 inter / 2]
    w2 = torch.randint(-128, 128, (experts, inter, hidden // 2), dtype=torch.int8, device=dev)
    s1 = (torch.rand(experts, 2 * inter, device=dev) * 0.01).to(dt)
    s2 = (torch.rand(experts, hidden, device=dev) * 0.01).to(dt)
    x = (torch.randn(tokens, hidden, device=dev) * 0.5).to(dt)
    gating = torch.randn(tokens, experts, device=dev)
    out = torch.empty(tokens, hidden, dtype=dt, device=dev)
    p = L.MoeParams()
    p.input, p.gating_output, p.output = x.data_ptr(), gating.data_ptr(), out.data_ptr()
    p.fc1_expert_weights, p.fc1_scales = w1.data_ptr(), s1.data_ptr()
    p.fc2_expert_weights, p.fc2_scales = w2.data_ptr(), s2.data_ptr()
    p.num_rows, p.hidden_size, p.inter_size, p.num_experts, p.k = tokens, hidden, inter, experts, top_k
    p.activation_type, p.act_type, p.weight_type, p.normalization_mode = 3, L.BF16, L.WEIGHT_INT4, 1
    need = L.lib().mi355q_moe_workspace_size(C.byref(p))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need

    def step():
        rc = L.lib().mi355q_run_moe(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.lib().mi355q_last_error()
    _, dev_s = time_steps(step, steps, 2, lambda: None)
    t = dev_s / steps
    rows = tokens * top_k
    flops = 2.0 * rows * hidden * 2 * inter + 2.0 * rows * inter * hidden
    wbytes = experts * (hidden * inter + inter * hidden // 2) + experts * (2 * inter + hidden) * 2
    # (gather and SwiGLU fused into the w1 GEMM: its rows are read once, the [rows, 2 inter] buffer does not exist)
    abytes = tokens * hidden * 2 * 2 + rows * hidden * 2 * 3 + rows * inter * 2 * 2
    return {"tokens": tokens, "top_k": top_k, "experts": experts, "hidden": hidden, "inter": inter,
            "weights": "int4 per-column", "activation": "swiglu", "us_per_call": t * 1e6, "TFLOPS": flops / t / 1e12,
            "GBps": (wbytes + abytes) / t / 1e9, "frac_of_8TBps": (wbytes + abytes) / t / 1e9 / HBM_PEAK_GBS,
            "frac_of_bf16_peak": flops / t / 1e12 / (INT8_PEAK_TOPS / 2), "algorithmic_bytes": wbytes + abytes}


def bench_prefill_w4a16(dev, steps, m=4096, n=11008, k=4096, g=128):
    """a4: prefill-M W4A16 AWQ GEMM (dequant in registers -> bf16 MFMA), Llama-2-7B gate/up shape."""
    from mi355q import _lib as L
    import ctypes as C
    torch.manual_seed(0)
    dt = torch.bfloat16
    w = torch.randint(-128, 128, (k, n // 2), dtype=torch.int8, device=dev)
    scale = torch.rand(k // g, n, dtype=dt, device=dev)
    zero = torch.randn(k // g, n, dtype=dt, device=dev)
    act = torch.randn(m, k, dtype=dt, device=dev)
    out = torch.empty(m, n, dtype=dt, device=dev)
    ws_bytes = L.lib().mi355q_weight_only_gemm_workspace_size(m, n, k, L.WEIGHT_INT4)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    p = L.WeightOnlyParams()
    p.qweight, p.scales, p.zeros = w.data_ptr(), scale.data_ptr(), zero.data_ptr()
    p.in_, p.act_scale, p.bias, p.out = act.data_ptr(), None, None, out.data_ptr()
    p.m, p.n, p.k, p.group_size = m, n, k, g
    p.weight_type, p.act_type = L.WEIGHT_INT4, L.BF16
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes

    def step():
        rc = L.lib().mi355q_weight_only_gemm(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.lib().mi355q_last_error()
    _, dev_s = time_steps(step, steps, 2, lambda: None)
    ops = 2.0 * m * n * k
    return {"m": m, "n": n, "k": k, "ms_per_call": dev_s / steps * 1e3, "TFLOPS": ops / (dev_s / steps) / 1e12,
            "frac_of_bf16_peak": ops / (dev_s / steps) / 1e12 / 2500.0}


def bench_prefill_w8a16(dev, steps, m=4096, n=11008, k=4096):
    """a2 at prefill size: W8A16 per-channel GEMM through the WeightOnlyQuantMatmul plugin (int8 weights, bf16)."""
    from mi355q import functional as F
    torch.manual_seed(0)
    w = torch.randint(-128, 128, (k, n), dtype=torch.int8, device=dev)
    sc = torch.rand(n, dtype=torch.bfloat16, device=dev)
    x = torch.randn(m, k, dtype=torch.bfloat16, device=dev)
    _, dev_s = time_steps(lambda: F.weight_only_quant_matmul(x, w, sc, 1, "bfloat16"), steps, 2, lambda: None)
    ops = 2.0 * m * n * k
    return {"m": m, "n": n, "k": k, "weights": "int8 per-channel", "ms_per_call": dev_s / steps * 1e3,
            "TFLOPS": ops / (dev_s / steps) / 1e12, "frac_of_bf16_peak": ops / (dev_s / steps) / 1e12 / 2500.0}


def bench_rmsnorm_quant(dev, steps, tokens=4096, hidden=5120, sets=24):
    """8f rank 2: RMSNorm + dynamic per-token int8 quantisation, the cfg3 GEMM's feeder (Llama-2-13B hidden).
    Algorithmic bytes per call: tokens*hidden*(2 in + 1 out) + hidden*2*2 (gamma, beta) + tokens*4 (scales);
    rotating input sets (> 256 MB in total) keep the Infinity Cache out of the number."""
    from mi355q import _lib as L
    import ctypes as C
    torch.manual_seed(0)
    xs = [torch.randn(tokens, hidden, device=dev).to(torch.float16) for _ in range(sets)]
    gamma = torch.rand(hidden, device=dev).to(torch.float16)
    beta = torch.zeros(hidden, device=dev, dtype=torch.float16)
    one = torch.ones(1, device=dev)
    q = torch.empty(tokens, hidden, dtype=torch.int8, device=dev)
    sc = torch.empty(tokens, 1, dtype=torch.float32, device=dev)
    it = [0]
    fn = L.lib().mi355q_rms_norm
    ptrs = [x.data_ptr() for x in xs]
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def step():  # the launcher below the plugin (invokeGeneralRmsNorm's signature): the kernel is 20 us, keep the host out
        rc = fn(None, ptrs[it[0] % sets], gamma.data_ptr(), beta.data_ptr(), 1e-5, tokens, hidden, one.data_ptr(),
                sc.data_ptr(), q.data_ptr(), L.HALF, stream)
        assert rc == 0
        it[0] += 1
    _, dev_s = time_steps(step, steps, sets, lambda: None)
    algo = tokens * hidden * 3 + hidden * 4 + tokens * 4
    return {"tokens": tokens, "hidden": hidden, "us_per_call": dev_s / steps * 1e6, "GBps": algo / (dev_s / steps) / 1e9,
            "frac_of_8TBps": algo / (dev_s / steps) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": algo}


def bench_gated_act(dev, steps, tokens=4096, inter=13824, sets=6):
    """8f rank 2: silu(fc) * gate / smoother + dynamic per-token int8 quantisation in one pass (the feeder of the cfg3
    down projection, Llama-2-13B intermediate size), and the plain one-pass per-token quantisation (a10).
    Algorithmic bytes: tokens*inter*(2 + 2 in + 1 out) resp. tokens*inter*(2 + 1)."""
    from mi355q import _lib as L
    import ctypes as C
    torch.manual_seed(0)
    gu = [torch.randn(tokens, 2 * inter, device=dev).to(torch.float16) for _ in range(sets)]
    smoother = (torch.rand(inter, device=dev) + 0.5).to(torch.float16)
    q = torch.empty(tokens, inter, dtype=torch.int8, device=dev)
    sc = torch.empty(tokens, 1, dtype=torch.float32, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    params = []
    for t in gu:
        p = L.GatedActParams()
        p.act_in, p.gate_in, p.smoother = t.data_ptr(), t.data_ptr() + inter * 2, smoother.data_ptr()
        p.out_quant, p.dynamic_scale = q.data_ptr(), sc.data_ptr()
        p.rows, p.cols, p.ld_act, p.ld_gate, p.type = tokens, inter, 2 * inter, 2 * inter, L.HALF
        params.append(p)
    it = [0]

    def step_gated():
        assert L.lib().mi355q_gated_silu(C.byref(params[it[0] % sets]), stream) == 0
        it[0] += 1

    def step_ptq():  # the first `inter` columns of every row: a [tokens, inter] matrix needs a dense source
        assert L.lib().mi355q_per_token_quantization(q.data_ptr(), dense[it[0] % sets].data_ptr(), tokens, inter,
                                                     sc.data_ptr(), L.HALF, stream) == 0
        it[0] += 1
    _, dev_s = time_steps(step_gated, steps, sets, lambda: None)
    algo = tokens * inter * 5 + inter * 2 + tokens * 4
    out = [{"op": "silu(fc)*gate/smoother + per-token int8", "tokens": tokens, "cols": inter, "us_per_call": dev_s / steps * 1e6,
            "GBps": algo / (dev_s / steps) / 1e9, "frac_of_8TBps": algo / (dev_s / steps) / 1e9 / HBM_PEAK_GBS,
            "algorithmic_bytes": algo}]
    dense = [t.view(2 * tokens, inter)[:tokens] for t in gu]
    _, dev_s = time_steps(step_ptq, steps, sets, lambda: None)
    algo = tokens * inter * 3 + tokens * 4
    out.append({"op": "per-token int8 quantisation", "tokens": tokens, "cols": inter, "us_per_call": dev_s / steps * 1e6,
                "GBps": algo / (dev_s / steps) / 1e9, "frac_of_8TBps": algo / (dev_s / steps) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes": algo})
    return out


def bench_context_attention(dev, steps, B=4, H=32, Hkv=32, D=128, S=2048, tpb=64, kv="bf16", past=0, cached=False):
    """8f rank 1: context (prefill) attention, Llama-2-7B heads: bias-less RoPE + paged KV append + causal
    attention; flops counted on the causal half (2 * 2 * S^2/2 * D per head).  past > 0: chunked context -- the
    last S - past tokens of the sequence attend to all S keys through the paged cache."""
    from mi355q import _lib as L
    import ctypes as C
    torch.manual_seed(0)
    dt = torch.bfloat16
    Sq = S - past                      # this call's tokens
    qkv = (torch.randn(B, Sq, (H + 2 * Hkv) * D, device=dev) * 0.5).to(dt)
    out = torch.empty(B, Sq, H * D, dtype=dt, device=dev)
    cl = torch.full((B,), Sq, dtype=torch.int32, device=dev)
    sl = torch.full((B,), S, dtype=torch.int32, device=dev)
    kvdt = {"bf16": dt, "int8": torch.int8, "fp8": torch.float8_e4m3fn}[kv]
    nblk = B * (S // tpb)
    pool = torch.zeros(nblk, 2, Hkv, tpb, D, dtype=kvdt, device=dev)
    if past:
        pool.copy_((torch.randn(pool.shape, device=dev) * 0.5).to(kvdt))
    be = Hkv * tpb * D * pool.element_size()
    ids = torch.randperm(nblk).reshape(B, S // tpb)
    ptrs = (torch.stack([ids * 2 * be, ids * 2 * be + be], 1) + pool.data_ptr()).to(torch.int64).to(dev)
    sc = torch.tensor([8.0], device=dev)
    p = L.ContextAttentionParams()
    p.qkv, p.out, p.context_lengths, p.block_pointers = qkv.data_ptr(), out.data_ptr(), cl.data_ptr(), ptrs.data_ptr()
    p.kv_scale_orig_quant = sc.data_ptr()
    p.batch, p.num_heads, p.num_kv_heads, p.head_size = B, H, Hkv, D
    p.max_blocks_per_seq, p.tokens_per_block = S // tpb, tpb
    p.max_context_length, p.num_tokens, p.remove_padding = Sq, B * Sq, 0
    if past or cached:   # (cached with past == 0: lab form -- K / V streamed back from the cache just written)
        p.sequence_lengths, p.max_kv_length = sl.data_ptr(), S
    p.act_type = L.BF16
    p.kv_type = {"bf16": L.BF16, "int8": L.INT8, "fp8": L.FP8}[kv]
    p.position_embedding_type, p.rotary_embedding_dim, p.rotary_embedding_base = 2, D, 10000.0
    p.rotary_embedding_scale, p.rotary_embedding_max_positions, p.q_scaling = 1.0, 4096, 1.0
    need = L.lib().mi355q_context_attention_workspace_size(C.byref(p))
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    p.workspace, p.workspace_bytes = ws.data_ptr(), need

    def step():
        rc = L.lib().mi355q_context_attention(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, L.lib().mi355q_last_error()
    _, dev_s = time_steps(step, steps, 2, lambda: None)
    flops = 4.0 * B * H * D * (Sq * past + Sq * (Sq + 1) / 2)
    return {"batch": B, "heads": H, "kv_heads": Hkv, "head_dim": D, "seq": S, "past_in_cache": past, "kv_cache": kv,
            "ms_per_call": dev_s / steps * 1e3, "TFLOPS_causal": flops / (dev_s / steps) / 1e12,
            "frac_of_bf16_peak": flops / (dev_s / steps) / 1e12 / 2500.0}


def bench_tp_row_parallel(dev, world, rank, iters=40):
    """cfg4's exchange step: the row-parallel W4A16 GEMM of Llama-2-70B's down projection under TP = 8
    (K_local 3584 -> N 8192, g = 128) followed by the all-reduce of its [M, 8192] bf16 output, M in {1, 8, 64}: through
    the custom push all-reduce (csrc/allreduce.hip, AUTO strategy) and through RCCL (torch.distributed).  Every rank
    takes part; times are the max over ranks.  bus GB/s = 2 (N - 1) / N * bytes / time (per-link bound 153 GB/s for a
    ring, 7 links in parallel for the direct exchange).  World 1: the GEMM alone (the all-reduce is the identity)."""
    from mi355q import _lib as L
    from mi355q import functional as F
    import ctypes as C
    torch.manual_seed(100 + rank)
    dt, k, n, g = torch.bfloat16, 3584, 8192, 128
    w = torch.randint(-128, 128, (k, n // 2), dtype=torch.int8, device=dev)
    scale = torch.rand(k // g, n, dtype=dt, device=dev)
    zero = torch.randn(k // g, n, dtype=dt, device=dev)

    def agree(ok):  # every rank learns whether all ranks got through the stage (keeps the collectives matched)
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    helper, err = None, None
    trace("tp leg: start")
    if world > 1:
        try:
            from mi355q.ipc import CustomAllReduceHelper
            helper = CustomAllReduceHelper()
        except Exception as e:  # noqa: BLE001
            err = repr(e)
        if not agree(helper is not None):
            helper, err = None, err or "a peer rank could not map the IPC buffers"
    out = []
    trace(f"tp leg: helper {'ready' if helper is not None else err}")
    for m in (1, 8, 64):
        trace(f"tp leg: m = {m}")
        act = torch.randn(m, k, dtype=dt, device=dev)
        y = torch.empty(m, n, dtype=dt, device=dev)
        ws_bytes = L.lib().mi355q_weight_only_gemm_workspace_size(m, n, k, L.WEIGHT_INT4)
        ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        p = L.WeightOnlyParams()
        p.qweight, p.scales, p.zeros = w.data_ptr(), scale.data_ptr(), zero.data_ptr()
        p.in_, p.act_scale, p.bias, p.out = act.data_ptr(), None, None, y.data_ptr()
        p.m, p.n, p.k, p.group_size = m, n, k, g
        p.weight_type, p.act_type = L.WEIGHT_INT4, L.BF16
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes

        def gemm():
            rc = L.lib().mi355q_weight_only_gemm(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, L.lib().mi355q_last_error()

        def timed(fn):
            bar = (lambda: dist.barrier()) if world > 1 else (lambda: None)
            _, dev_s = time_steps(fn, iters, 3, bar)
            t = torch.tensor([dev_s], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return t.item() / iters * 1e6

        nbytes = m * n * 2
        line = {"m": m, "k_local": k, "n": n, "message_bytes": nbytes, "gemm_us": timed(gemm)}
        trace("tp leg: gemm timed")
        if world == 1:
            line["allreduce"] = "identity (world 1)"
        else:
            bus = lambda us: 2.0 * (world - 1) / world * nbytes / (us * 1e-6) / 1e9
            red = torch.empty_like(y)

            def rccl():
                gemm()
                red.copy_(y)
                dist.all_reduce(red)
            def rccl_only():
                dist.all_reduce(red)
            if dist.get_backend() == "nccl":
                line["gemm_plus_rccl_us"] = timed(rccl)
                trace("tp leg: rccl timed")
                t = timed(rccl_only)
                line["rccl_allreduce_us"], line["rccl_bus_GBps"] = t, bus(t)
            else:
                line["rccl"] = "skipped: one-GPU rehearsal on gloo"
            if helper is not None:
                ok = True
                try:
                    def custom():
                        gemm()
                        helper.allreduce(y, out=red, direct=True)
                    def custom_only():
                        helper.allreduce(y, out=red, direct=True)
                    line["gemm_plus_custom_us"] = timed(custom)
                    t = timed(custom_only)
                    line["custom_allreduce_us"], line["custom_bus_GBps"] = t, bus(t)
                    line["custom_strategy"] = {1: "oneshot", 2: "twoshot", 0: "ring"}[
                        L.lib().mi355q_allreduce_select_implementation(nbytes, world, helper.size)]
                    # the AllGather plugin's kernel on the same buffers (gather_output of a column-parallel layer):
                    # every rank's [m, n] block back to back; checked against RCCL's all-gather
                    gath = torch.empty(world * m, n, dtype=dt, device=dev)
                    gbus = lambda us: (world - 1) * nbytes / (us * 1e-6) / 1e9

                    def custom_gather():
                        helper.allgather(y, out=gath, direct=True)
                    t = timed(custom_gather)
                    line["custom_allgather_us"], line["custom_allgather_bus_GBps"] = t, gbus(t)
                    if dist.get_backend() == "nccl":
                        ref = torch.empty_like(gath)

                        def rccl_gather():
                            dist.all_gather_into_tensor(ref, y)
                        t = timed(rccl_gather)
                        line["rccl_allgather_us"], line["rccl_allgather_bus_GBps"] = t, gbus(t)
                        torch.cuda.synchronize()
                        line["allgather_matches_rccl"] = bool(torch.equal(ref, gath))
                    ok = not helper.timed_out()
                except Exception as e:  # noqa: BLE001
                    ok, line["custom_error"] = False, repr(e)
                if not agree(ok):
                    line.setdefault("custom_error", "a wait for a peer rank timed out")
                    helper = None   # the flags are no longer in step: RCCL only from here on
            elif err:
                line["custom_error"] = err
        out.append(line)
    # the prefill message of cfg4 (M = 4096: 64 MiB, beyond the comm buffers) through the AllReduce plugin's RING strategy:
    # ncclAllReduce issued inside libmi355q on its own communicator (mi355q_comm_init_rank = initCommMap), next to
    # torch.distributed's all-reduce of the same tensor
    if world > 1 and dist.get_backend() == "nccl":
        line = {"m": 4096, "n": n, "message_bytes": 4096 * n * 2, "allreduce": "AllReduce plugin, strategy RING (RCCL inside the library)"}
        comm_err = None
        try:
            from mi355q.ipc import init_comm
            from mi355q.plugin import Plugin, torch_dtype_to_trt
            init_comm()
        except Exception as e:  # noqa: BLE001
            comm_err = repr(e)
        try:
            if not agree(comm_err is None):  # every rank or none goes on to the collective
                raise RuntimeError(comm_err or "a peer rank could not create the communicator")
            x = torch.randn(4096, n, dtype=dt, device=dev)
            red, ref = torch.empty_like(x), torch.empty_like(x)
            plg = Plugin("AllReduce", [("group", list(range(world)), L.FIELD_INT32), ("type_id", torch_dtype_to_trt(dt), L.FIELD_INT32),
                                       ("strategy", L.AR_RING, L.FIELD_INT8), ("counter", 0, L.FIELD_INT32)])
            bus = lambda us: 2.0 * (world - 1) / world * x.numel() * 2 / (us * 1e-6) / 1e9

            def timed_big(fn):
                _, dev_s = time_steps(fn, 10, 2, lambda: dist.barrier())
                t = torch.tensor([dev_s], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                return t.item() / 10 * 1e6
            t = timed_big(lambda: plg.enqueue([x], [red]))
            line["plugin_ring_us"], line["plugin_ring_bus_GBps"] = t, bus(t)

            def torch_ar():
                ref.copy_(x)
                dist.all_reduce(ref)
            t = timed_big(torch_ar)
            line["torch_copy_plus_allreduce_us"] = t
            torch.cuda.synchronize()
            line["max_abs_diff_vs_torch"] = (red.float() - ref.float()).abs().max().item()
        except Exception as e:  # noqa: BLE001
            line["error"] = repr(e)
        out.append(line)
    return out


def cpu_baseline_attention(H=32, D=128, ctx=2048, tpb=64, seqs=64, budget_s=12.0):
    """The oracle's C / OpenMP port of the MMHA math (oracle.c orc_decode_attention_16) on the same workload (all 64
    sequences x 32 heads x ctx 2048 = one bench step), timed on the host cores of the GPU box (one GPU's share: 16):
    whole steps repeated for about `budget_s` seconds of wall clock; the value is the median pass."""
    import oracle
    torch.manual_seed(42)
    cores = min(16, os.cpu_count() or 1)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    cache = oracle.PagedKVCache(seqs, H, D, ctx, tpb, torch.bfloat16)
    cache.pool.copy_((torch.randn(cache.pool.shape) * 0.5).to(torch.bfloat16))
    qkv = torch.randn(seqs, 3 * H * D).to(torch.bfloat16)
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 2000):
        t0 = time.perf_counter()
        oracle.decode_attention_c(qkv, cache, [ctx] * seqs, H, H, D, rope="neox")
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    nbytes = seqs * (2 * H * ctx * D * 2 + 3 * H * D * 2 + H * D * 2 + 2 * H * D * 2)
    return {"value": nbytes / med / 1e9, "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{seqs} of 64 sequences x {H} heads x ctx {ctx} (one whole step), oracle.c orc_decode_attention_16 "
                      f"(C, OpenMP, {cores} threads): {len(times)} passes in {sum(times):.1f} s, median {med * 1e3:.1f} ms, "
                      f"best {times[0] * 1e3:.1f} ms"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-secondary", action="store_true", help="skip the GEMV / int8 GEMM side measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tp", action="store_true", help="skip the row-parallel GEMM + all-reduce leg")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library is the only implementation (no CPU fallback)")
    # rehearsal on a one-GPU box (MI355Q_BENCH_ONE_GPU=1): every rank on device 0, gloo instead of RCCL -- exercises the
    # multi-rank code paths (IPC mapping, flags, matched collectives); its numbers mean nothing
    one_gpu = os.environ.get("MI355Q_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        barrier = lambda: dist.barrier()
    else:
        barrier = lambda: None

    B, H, D, ctx, tpb = 64, 32, 128, 2048, 64
    trace("headline: building the cache")
    plg, inputs, out, ws, algo_bytes, keep = build_attention_case(dev, B, H, D, ctx, tpb, seed=42 + rank)
    trace("headline: timing")
    step = lambda: plg.enqueue(inputs, [out], ws)
    wall, dev_s = time_steps(step, args.steps, args.warmup, barrier)

    t = torch.tensor([wall, dev_s], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall_max, dev_max = t.tolist()

    trace("headline: done")
    # the path's one exchange step (row-parallel GEMM + all-reduce): every rank takes part, rank 0 reports
    tp_lines = None
    if not args.no_tp:
        try:
            tp_lines = bench_tp_row_parallel(dev, world, rank)
        except Exception as e:  # noqa: BLE001 -- never takes the headline down
            tp_lines = {"error": repr(e)}

    if rank == 0:
        per_launch = dev_s / args.steps
        achieved = algo_bytes / per_launch / 1e9
        res = {
            "metric": "W4A16 GEMV HBM GB/s + paged decode-attn GB/s + int8 GEMM TFLOPS, Llama shapes",
            "value": world * algo_bytes * args.steps / wall_max / 1e9,
            "unit": "GB/s (paged decode attention, algorithmic bytes, whole job)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16 (KV + q/out), fp32 accumulate", "data": "synthetic",
            "config": {"workload": "masked-MHA decode, Llama-2-7B shape: 32 heads x d128, batch 64, ctx 2048, "
                                   "paged bf16 KV (tokens_per_block 64), RoPE + KV append fused; one step = one "
                                   "GPTAttention enqueue (one layer)",
                       "batch_per_gpu": B, "heads": H, "head_dim": D, "ctx": ctx, "tokens_per_block": tpb,
                       "parallelism": f"{world} independent KV-head/sequence shards, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "decode_attn_kernel<bf16, kv=bf16, Dh=128, G=1, U=8>",
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_us": per_launch * 1e6},
        }
        # HBM traffic per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
        # this same command; KiB units, read side doubled on gfx950 as MI355X_MICROARCH.md prescribes):
        # recorded by tools/run_round.sh into profiles/, read back here so the JSON line carries it
        tfile = os.path.join(_ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                tj = json.load(f)
            res["roofline"]["traffic"] = tj.get("decode_attn_kernel_bytes_per_launch")
            # NOT measured in this run: PMC passes are separate rocprofv3 runs of this same command (tools/run_round.sh)
            res["roofline"]["traffic_from"] = "profiles/hbm_traffic.json (earlier builder-side PMC run: " + str(tj.get("source")) + ")"
        if not args.no_secondary and world == 1:  # side measurements and the CPU baseline: N = 1 only
            sec = {}
            try:
                # launch floor (tools/lab_skel.hip): an empty 256-workgroup launch in a stream of launches is 1.65 us and
                # streaming 8 MiB + an LDS reduce + the store 3.0 us, so cfg1 (8.9 MB) cannot exceed ~37 % of 8 TB/s per
                # launch; the fused shapes (qkv 12288, gate|up 22016: GEMVs that share their input, N concatenated) amortise it
                sec["w4a16_gemv"] = [bench_gemv(dev, 4 * 72, m=1), bench_gemv(dev, 4 * 72, m=1, pre_quant=True),
                                     bench_gemv(dev, 4 * 72, m=4), bench_gemv(dev, 4 * 72, m=16),
                                     bench_gemv(dev, 2 * 72, m=1, n=12288, k=4096, sets=24),
                                     bench_gemv(dev, 2 * 72, m=1, n=22016, k=4096, sets=14),
                                     bench_gemv(dev, 2 * 72, m=1, n=11008, k=4096, sets=24),
                                     bench_gemv(dev, 2 * 72, m=1, n=4096, k=11008, sets=24),
                                     bench_gemv(dev, 48, m=1, n=28672, k=8192, sets=6),
                                     bench_gemv(dev, 48, m=16, n=28672, k=8192, sets=6),
                                     bench_gemv(dev, 48, m=32, n=28672, k=8192, sets=6)]
                sec["decode_attention_other"] = [
                    bench_decode_attention(dev, 48, Hq=32, Hkv=32, kv="bf16", ragged=True),          # cfg2 ragged
                    bench_decode_attention(dev, 96, Hq=8, Hkv=1, kv="int8", sets=12),                 # cfg4 per-GPU slice
                    bench_decode_attention(dev, 48, Hq=32, Hkv=8, kv="fp8", sets=2),                  # cfg5 fp8 KV
                    bench_decode_attention(dev, 48, Hq=64, Hkv=8, kv="bf16"),                         # 70B GQA, bf16 KV
                    bench_decode_attention(dev, 48, Hq=64, Hkv=8, kv="int8", sets=2)]                 # 70B GQA, int8 KV
                sec["w4a16_gemm_tp8"] = [bench_gemv(dev, 144, m=mm, n=nn, k=kk, sets=ss)                # cfg4 per-GPU GEMMs
                                         for (mm, nn, kk, ss) in ((1, 1280, 8192, 64), (1, 3584, 8192, 24),
                                                                  (1, 8192, 1024, 72), (1, 8192, 3584, 24),
                                                                  (8, 3584, 8192, 24), (64, 3584, 8192, 24))]
                sec["int8_gemm"] = [bench_int8_gemm(dev, 10), bench_int8_gemm(dev, 10, n=5120, k=13824)]
                sec["fp8_gemm"] = [bench_fp8_gemm(dev, 10), bench_fp8_gemm(dev, 10, n=4096, k=14336)]
                sec["rmsnorm_quant"] = [bench_rmsnorm_quant(dev, 96), bench_rmsnorm_quant(dev, 96, hidden=8192, sets=16)]
                sec["gated_act_quant"] = bench_gated_act(dev, 48)
                sec["context_attention"] = [bench_context_attention(dev, 10),
                                            bench_context_attention(dev, 10, B=2, H=64, Hkv=8, S=4096, kv="fp8"),
                                            bench_context_attention(dev, 10, B=4, S=4096, past=2048)]
                sec["w4a16_prefill_gemm"] = [bench_prefill_w4a16(dev, 10), bench_prefill_w4a16(dev, 10, n=4096, k=11008)]
                sec["w8a16_prefill_gemm"] = [bench_prefill_w8a16(dev, 10), bench_prefill_w8a16(dev, 10, n=4096, k=11008)]
                sec["moe_w4_grouped_gemm"] = [bench_moe_gemm(dev, 10, tokens=64), bench_moe_gemm(dev, 10, tokens=4096)]
                sec["moe_w4_layer"] = [bench_moe_layer(dev, 20, tokens=64), bench_moe_layer(dev, 20, tokens=4096)]
            except Exception as e:  # secondary numbers must never take the headline down
                sec["error"] = repr(e)
            res["secondary"] = sec
        if tp_lines is not None:
            res.setdefault("secondary", {})["tp_row_parallel_gemm_allreduce"] = tp_lines
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline_attention(H, D, ctx, tpb)
        # the driver keeps the TAIL of stdout: the long secondary object goes first, the roofline, the CPU baseline and a
        # short summary of the metric's three legs (cfg1 GEMV, cfg2 decode attention, cfg3 int8 GEMM) last
        sec = res.pop("secondary", None)
        ordered = {k: res[k] for k in res if k not in ("roofline", "cpu_baseline")}
        if sec is not None:
            ordered["secondary"] = sec
        ordered["roofline"] = res["roofline"]
        if "cpu_baseline" in res:
            ordered["cpu_baseline"] = res["cpu_baseline"]
        legs = {"decode_attn_cfg2_GBps": round(achieved, 1), "decode_attn_cfg2_frac_of_8TBps": round(achieved / HBM_PEAK_GBS, 4)}
        try:
            g = sec["w4a16_gemv"][0]
            legs["w4a16_gemv_cfg1_m1_GBps"] = round(g["GBps"], 1)
            legs["w4a16_gemv_cfg1_m1_frac_of_8TBps"] = round(g["frac_of_8TBps"], 4)
            legs["w4a16_gemv_28672x8192_m1_frac_of_8TBps"] = round(sec["w4a16_gemv"][8]["frac_of_8TBps"], 4)
            legs["int8_gemm_cfg3_TOPS"] = [round(x["TOPS"], 1) for x in sec["int8_gemm"]]
            legs["int8_gemm_cfg3_frac_of_5POPS"] = [round(x["frac_of_int8_peak"], 4) for x in sec["int8_gemm"]]
        except (TypeError, KeyError, IndexError):
            pass
        ordered["legs"] = legs
        print(json.dumps(ordered))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
