"""Lab: phased int8 kernel (MI355Q_GEMM8_TILE=3) vs the oracle (small) and vs the 256x256 baseline kernel (large)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
import numpy as np, torch
import oracle
from mi355q import functional as F

dev = torch.device("cuda")
def run(a, b, sa, sb, tile, out="int32"):
    os.environ["MI355Q_GEMM8_TILE"] = tile
    o = F.smooth_quant_gemm(a, b, sa, sb, sa.numel() > 1, sb.numel() > 1, out)
    torch.cuda.synchronize()
    return o
torch.manual_seed(0)
ok = True
for (m, n, k) in [(256, 256, 128), (256, 256, 256), (512, 768, 1024), (256, 512, 640)]:
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8); b = torch.randint(-128, 128, (n, k), dtype=torch.int8)
    one = torch.ones(1, 1, device=dev)
    acc = run(a.to(dev), b.to(dev), one, one, "0").cpu().numpy()
    ref = oracle.gemm_s8s8s32(a.numpy(), b.numpy())
    e = np.array_equal(acc, ref); ok &= e
    print(m, n, k, "int32 exact vs oracle:", e, flush=True)
    st = (torch.randint(1, 10, (m, 1)).float() * 1e-2); sc = (torch.randint(1, 10, (1, n)).float() * 1e-2)
    o16 = run(a.to(dev), b.to(dev), st.to(dev), sc.to(dev), "0", "float16").cpu()
    r16, _ = oracle.smooth_quant_gemm(a.numpy(), b.numpy(), st.numpy(), sc.numpy(), 1, 1, torch.float16)
    e = torch.equal(o16, r16); ok &= e
    print(m, n, k, "fp16 epilogue exact vs oracle:", e, flush=True)
for (m, n, k) in [(4096, 13824, 5120), (4096, 5120, 13824), (2048, 4096, 4096)]:
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev); b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    one = torch.ones(1, 1, device=dev)
    for rep in range(3):  # race screen: repeated runs must agree with the baseline kernel bit for bit
        e = torch.equal(run(a, b, one, one, "auto"), run(a, b, one, one, "3")); ok &= e
    print(m, n, k, "phased == baseline kernel (3 runs):", e, flush=True)
    for tile in ("3", "0", "auto"):
        os.environ["MI355Q_GEMM8_TILE"] = tile
        st = torch.rand(m, 1, device=dev); sc = torch.rand(1, n, device=dev)
        for _ in range(3): F.smooth_quant_gemm(a, b, st, sc, True, True, "float16")
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): F.smooth_quant_gemm(a, b, st, sc, True, True, "float16")
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"  tile {tile}: {t*1e3:.1f} us  {2.0*m*n*k/t/1e9:.0f} TOPS", flush=True)
# ---- fp8: phased (16x16x128 f8f6f4) vs the small-tile kernel (16x16x32-class), Mixtral shapes
for (m, n, k) in [(512, 1024, 4096), (4096, 14336, 4096), (4096, 4096, 14336), (4096, 6144, 4096)]:
    a = (torch.randn(m, k, device=dev)).to(torch.float8_e4m3fn); b = torch.randn(n, k, device=dev).to(torch.float8_e4m3fn)
    sa = torch.tensor([0.5], device=dev); sb = torch.tensor([0.25], device=dev)
    os.environ.pop("MI355Q_FP8_SMALL_TILE", None)
    o1 = F.fp8_gemm(a, b, sa, sb, "float32"); torch.cuda.synchronize()
    os.environ["MI355Q_FP8_SMALL_TILE"] = "1"
    o0 = F.fp8_gemm(a, b, sa, sb, "float32"); torch.cuda.synchronize()
    err = (o1 - o0).abs().max().item(); scale = o0.abs().max().item()
    e = err <= 1e-5 * scale * (k ** 0.5); ok &= e
    print(m, n, k, f"fp8 phased vs small-tile: max err {err:.3e} (max |out| {scale:.1f})", e, flush=True)
    for small in (True, False):
        if small: os.environ["MI355Q_FP8_SMALL_TILE"] = "1"
        else: os.environ.pop("MI355Q_FP8_SMALL_TILE", None)
        for _ in range(3): F.fp8_gemm(a, b, sa, sb, "bfloat16")
        torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): F.fp8_gemm(a, b, sa, sb, "bfloat16")
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"  {'small-tile' if small else 'phased'}: {t*1e3:.1f} us  {2.0*m*n*k/t/1e9:.0f} TFLOPs", flush=True)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
