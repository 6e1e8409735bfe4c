"""Lab: performance sanity sweep over plugin modes the bench does not time (looking for silently slow instantiations:
the W8A16 128x256 tile ran 8x slow for a whole round without a test noticing).  Prints rate per case."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorrt-llm_amd"))
import torch, bench
from mi355q import functional as F
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def t(f, steps=20):
    f()
    _, s = bench.time_steps(f, steps, 2, lambda: None)
    return s / steps


def line(tag, sec, flops=None, bytes_=None):
    msg = f"{tag:70s} {sec * 1e6:9.1f} us"
    if flops:
        msg += f"  {flops / sec / 1e12:8.1f} T(FL)OP/s"
    if bytes_:
        msg += f"  {bytes_ / sec / 1e9:8.1f} GB/s"
    print(msg, flush=True)


# ---- weight-only GEMMs through the plugins: dtype x weight type x m
for dts, dt in (("float16", torch.float16), ("bfloat16", torch.bfloat16)):
    for wt, wname in ((1, "int8"), (2, "int4")):
        for (m, n, k) in ((1, 4096, 4096), (8, 11008, 4096), (16, 4096, 11008), (24, 4096, 4096), (48, 11008, 4096),
                          (128, 11008, 4096), (300, 4096, 4096), (1000, 11008, 4096), (4096, 11008, 4096)):
            w = torch.randint(-128, 128, (k, n if wt == 1 else n // 2), dtype=torch.int8, device=dev)
            sc = torch.rand(n, dtype=dt, device=dev)
            x = torch.randn(m, k, dtype=dt, device=dev)
            s = t(lambda: F.weight_only_quant_matmul(x, w, sc, wt, dts))
            line(f"WeightOnlyQuantMatmul {dts} {wname} m={m} n={n} k={k}", s, 2.0 * m * n * k, w.numel())
# ---- group-wise: fp16, no zero, bias, W4A8
for dts, dt in (("float16", torch.float16),):
    for algo, aname in ((0, "plain"), (1, "bias"), (2, "zero"), (4, "prequant"), (7, "all")):
        for (m, n, k) in ((1, 4096, 4096), (16, 11008, 4096), (64, 11008, 4096), (4096, 4096, 4096)):
            g = 128
            w = torch.randint(-128, 128, (k, n // 2), dtype=torch.int8, device=dev).view(dt).reshape(k, n // 4)
            sc = torch.rand(k // g, n, dtype=dt, device=dev)
            z = torch.randn(k // g, n, dtype=dt, device=dev)
            b = torch.randn(1, n, dtype=dt, device=dev)
            pq = torch.rand(1, k, dtype=dt, device=dev)
            x = torch.randn(m, k, dtype=dt, device=dev)
            s = t(lambda: F.weight_only_groupwise_quant_matmul(x, pq, w, sc, z, b, None, algo, g, dts))
            line(f"Groupwise {dts} algo={aname} m={m} n={n} k={k}", s, 2.0 * m * n * k, k * n // 2)
# ---- SmoothQuant int8 GEMM: scale modes, output types, ragged shapes
for (m, n, k) in ((4096, 13824, 5120), (1000, 5120, 5120), (4096, 5120, 13824), (17, 5120, 5120), (4096, 1280, 5120), (333, 13824, 5120)):
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    for pt, pc in ((1, 1), (0, 0)):
        sa = torch.rand(m if pt else 1, 1, device=dev) * 1e-2
        sb = torch.rand(1, n if pc else 1, device=dev) * 1e-2
        for dts in ("float16", "float32"):
            s = t(lambda: F.smooth_quant_gemm(a, b, sa, sb, bool(pt), bool(pc), dts))
            line(f"SmoothQuantGemm {dts} per_token={pt} per_channel={pc} m={m} n={n} k={k}", s, 2.0 * m * n * k)
# ---- fp8 GEMM shapes
for (m, n, k) in ((4096, 14336, 4096), (1000, 4096, 4096), (64, 14336, 4096), (4096, 6144, 4096)):
    a = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn)
    b = torch.randn(n, k, device=dev).to(torch.float8_e4m3fn)
    one = torch.tensor([0.5], device=dev)
    for dts in ("float16", "bfloat16"):
        s = t(lambda: F.fp8_gemm(a, b, one, one, dts))
        line(f"Fp8Gemm {dts} m={m} n={n} k={k}", s, 2.0 * m * n * k)
# ---- norms / quantisers
for dt in (torch.float16, torch.bfloat16):
    for (rows, h) in ((4096, 5120), (64, 8192), (4096, 4096), (4096, 6656)):
        x = torch.randn(rows, h, device=dev).to(dt)
        wgt = torch.rand(h, device=dev).to(dt)
        sc1 = torch.ones(1, device=dev)
        for dyn in (False, True):
            s = t(lambda: F.smooth_quant_rms_norm(x, h, wgt, None, sc1, 1e-5, dyn))
            line(f"RmsnormQuantization {dt} dyn={dyn} rows={rows} h={h}", s, None, rows * h * 3)
            s = t(lambda: F.smooth_quant_layer_norm(x, h, wgt, None, sc1, 1e-5, True, dyn))
            line(f"LayernormQuantization {dt} dyn={dyn} rows={rows} h={h}", s, None, rows * h * 3)
        s = t(lambda: F.quantize_per_token(x))
        line(f"QuantizePerToken {dt} rows={rows} h={h}", s, None, rows * h * 3)
        s = t(lambda: F.rms_norm(x, wgt))
        line(f"rms_norm {dt} rows={rows} h={h}", s, None, rows * h * 4)
