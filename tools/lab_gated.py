"""Lab: time the one-pass per-token quantisation and the fused gated activation."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorrt-llm_amd"))
import torch
from mi355q import functional as F
dev = torch.device("cuda", 0)


def timeit(fn, reps=40):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for rows, cols in ((4096, 13824), (4096, 5120), (64, 13824)):
    xs = [torch.randn(rows, cols, dtype=torch.float16, device=dev) for _ in range(6)]
    gu = [torch.randn(rows, 2 * cols, dtype=torch.float16, device=dev) for _ in range(4)]
    sm = torch.rand(cols, dtype=torch.float16, device=dev) + 0.5
    i = [0]
    def ptq():
        F.quantize_per_token(xs[i[0] % 6]); i[0] += 1
    def gated():
        t = gu[i[0] % 4]; i[0] += 1
        F.gated_silu(t[:, :cols], t[:, cols:], sm, dynamic_act_scaling=True)
    def unfused():
        t = gu[i[0] % 4]; i[0] += 1
        y = torch.nn.functional.silu(t[:, :cols]) * t[:, cols:] / sm
        F.quantize_per_token(y)
    t = timeit(ptq); print(f"per-token quant {rows}x{cols}: {t:.1f} us  {rows * cols * 3 / t / 1e3:.0f} GB/s")
    t = timeit(gated); print(f"gated silu + smoother + per-token quant {rows}x{cols}: {t:.1f} us  {rows * cols * 5 / t / 1e3:.0f} GB/s")
    t = timeit(unfused); print(f"  unfused torch silu*gate/smoother + per-token quant: {t:.1f} us")
