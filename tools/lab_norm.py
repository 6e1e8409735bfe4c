"""Lab: time the norm + quant feeder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(), dict(hidden=8192, sets=16), dict(tokens=64, hidden=4096, sets=64), dict(tokens=16384, hidden=4096, sets=8)):
    print(bench.bench_rmsnorm_quant(dev, 96, **kw), flush=True)

# calibration: what a plain copy / cast of the same footprint reaches (mixed read + write traffic)
def _time(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
srcs = [torch.randn(16384, 4096, dtype=torch.float16, device=dev) for _ in range(4)]
dst8 = torch.empty(16384, 4096, dtype=torch.int8, device=dev)
dst16 = torch.empty(16384, 4096, dtype=torch.float16, device=dev)
i = [0]
def cast():
    dst8.copy_(srcs[i[0] % 4]); i[0] += 1
def copy():
    dst16.copy_(srcs[i[0] % 4]); i[0] += 1
t = _time(cast); print("torch fp16->int8 cast 16384x4096: %.1f us  %.0f GB/s" % (t, 16384 * 4096 * 3 / t / 1e3))
t = _time(copy); print("torch fp16 copy 16384x4096: %.1f us  %.0f GB/s" % (t, 16384 * 4096 * 4 / t / 1e3))
