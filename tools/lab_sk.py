"""Lab: the persistent / stream-K 8-bit GEMM (workspace given) against the one-tile-per-workgroup path (no workspace),
through the C ABI: int8 results must be bit-identical, fp8 within fp32 re-association; then timing of both."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
import torch
from mi355q import _lib as L

dev = torch.device("cuda")
lib = L.lib()
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def int8(a, b, sa, sb, out, ws):
    m, k = a.shape
    n = b.shape[0]
    ot = {torch.float16: L.HALF, torch.int32: L.INT32, torch.bfloat16: L.BF16, torch.float32: L.FLOAT}[out.dtype]
    L.check(lib.mi355q_int8_gemm_ws(a.data_ptr(), b.data_ptr(), 1, 1, sb.data_ptr(), sa.data_ptr(), out.data_ptr(), ot, m, n, k,
                                    ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, st()))


def fp8(a, b, sa, sb, out, ws):
    m, k = a.shape
    n = b.shape[0]
    ot = {torch.float16: L.HALF, torch.bfloat16: L.BF16, torch.float32: L.FLOAT}[out.dtype]
    L.check(lib.mi355q_fp8_gemm_ws(a.data_ptr(), b.data_ptr(), sa.data_ptr(), sb.data_ptr(), out.data_ptr(), ot, m, n, k,
                                   ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0, st()))


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def timeit2(fa, fb, rounds=6, n=5):
    """interleaved A/B (the device's clock drifts by several per cent within a run): median of the rounds"""
    ta, tb = [], []
    for _ in range(rounds):
        ta.append(timeit(fa, n))
        tb.append(timeit(fb, n))
    ta.sort(), tb.sort()
    return ta[len(ta) // 2], tb[len(tb) // 2]


ok = True
torch.manual_seed(0)
shapes = [(256, 512, 1024), (512, 13824, 5120), (1024, 13824, 5120), (2048, 13824, 5120), (4096, 4352, 4096),
          (4096, 13824, 5120), (4096, 5120, 13824), (4096, 27648, 5120), (8192, 8192, 8192)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split("x")) for s in sys.argv[1:]]
for (m, n, k) in shapes:
    need = lib.mi355q_gemm8_workspace_size(m, n, k)
    a = torch.randint(-128, 128, (m, k), dtype=torch.int8, device=dev)
    b = torch.randint(-128, 128, (n, k), dtype=torch.int8, device=dev)
    sa = torch.randint(1, 10, (m, 1), device=dev).float() * 1e-2
    sb = torch.randint(1, 10, (1, n), device=dev).float() * 1e-2
    ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev).fill_(0xAB) if need else None
    line = f"int8 {m}x{n}x{k}: ws {need >> 20} MiB"
    if need:
        for dt in (torch.int32, torch.float16):
            o0 = torch.zeros(m, n, dtype=dt, device=dev)
            o1 = torch.zeros(m, n, dtype=dt, device=dev)
            int8(a, b, sa, sb, o0, None)
            for rep in range(3):
                o1.zero_()
                int8(a, b, sa, sb, o1, ws)
                torch.cuda.synchronize()
                e = torch.equal(o0, o1)
                ok &= e
                if not e:
                    bad = (o0 != o1).nonzero()
                    line += f" MISMATCH {dt} rep {rep}: {bad.shape[0]} elements, first {bad[0].tolist()}"
                    break
        line += " exact"
    o = torch.zeros(m, n, dtype=torch.float16, device=dev)
    if need:
        t0, t1 = timeit2(lambda: int8(a, b, sa, sb, o, None), lambda: int8(a, b, sa, sb, o, ws))
        line += f" | plain {t0:.1f} us {2.0 * m * n * k / t0 / 1e6:.0f} TOPS"
        line += f" | split {t1:.1f} us {2.0 * m * n * k / t1 / 1e6:.0f} TOPS"
    else:
        t0 = timeit(lambda: int8(a, b, sa, sb, o, None))
        line += f" | plain {t0:.1f} us {2.0 * m * n * k / t0 / 1e6:.0f} TOPS"
    print(line, flush=True)
if len(sys.argv) == 1:
    for (m, n, k) in [(512, 14336, 4096), (4096, 14336, 4096), (4096, 6144, 4096), (4096, 4096, 14336)]:
        need = lib.mi355q_gemm8_workspace_size(m, n, k)
        a = torch.randn(m, k, device=dev).to(torch.float8_e4m3fn)
        b = torch.randn(n, k, device=dev).to(torch.float8_e4m3fn)
        sa = torch.tensor([0.5], device=dev)
        sb = torch.tensor([0.25], device=dev)
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device=dev) if need else None
        line = f"fp8 {m}x{n}x{k}: ws {need >> 20} MiB"
        o0 = torch.zeros(m, n, dtype=torch.float32, device=dev)
        fp8(a, b, sa, sb, o0, None)
        if need:
            o1 = torch.zeros(m, n, dtype=torch.float32, device=dev)
            fp8(a, b, sa, sb, o1, ws)
            torch.cuda.synchronize()
            err = (o0 - o1).abs().max().item()
            e = err <= 1e-5 * o0.abs().max().item() * k ** 0.5
            ok &= e
            line += f" max err {err:.2e} {'ok' if e else 'BAD'}"
        o = torch.zeros(m, n, dtype=torch.bfloat16, device=dev)
        if need:
            t0, t1 = timeit2(lambda: fp8(a, b, sa, sb, o, None), lambda: fp8(a, b, sa, sb, o, ws))
            line += f" | plain {t0:.1f} us {2.0 * m * n * k / t0 / 1e6:.0f} TFLOPS"
            line += f" | split {t1:.1f} us {2.0 * m * n * k / t1 / 1e6:.0f} TFLOPS"
        else:
            t0 = timeit(lambda: fp8(a, b, sa, sb, o, None))
            line += f" | plain {t0:.1f} us {2.0 * m * n * k / t0 / 1e6:.0f} TFLOPS"
        print(line, flush=True)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
