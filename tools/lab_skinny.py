"""Lab: the skinny form of the W4A16 / W8A16 GEMM (16 < m <= 64, woq_skinny_kernel) against the path it replaces
(MI355Q_NO_SKINNY=1: the ping-pong prefill kernel + slab reduce), through the C ABI of the LAB library: results against
each other, then hipGraph timings over rotating weight sets, interleaved.
usage: lab_skinny.py [m,n,k ...]   (env MI355Q_SKINNY_MT / MI355Q_SKINNY_P force the tile)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch
from mi355q import _lib as L

dev = torch.device("cuda")
lib = L.lib()


def case(m, n, k, g=128, int4=True, zero=True, dt=torch.bfloat16):
    sets = max(2, min(48, int(400e6 // (n * k // (2 if int4 else 1)))))
    torch.manual_seed(1)
    w = torch.randint(-128, 128, (sets, k, n // 2 if int4 else n), dtype=torch.int8, device=dev)
    groups = k // g if g else 1
    scale = (torch.rand(sets, groups, n, device=dev) * 0.02 + 0.001).to(dt)
    zeros = (torch.randn(sets, groups, n, device=dev) * 0.05).to(dt) if zero else None
    act = torch.randn(m, k, dtype=dt, device=dev)
    outs = [torch.zeros(m, n, dtype=dt, device=dev) for _ in range(2)]
    wt = L.WEIGHT_INT4 if int4 else L.WEIGHT_INT8
    ws_bytes = lib.mi355q_weight_only_gemm_workspace_size(m, n, k, wt)
    ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)

    def params(s, out):
        p = L.WeightOnlyParams()
        p.qweight, p.scales = w[s].data_ptr(), scale[s].data_ptr()
        p.zeros = zeros[s].data_ptr() if zero else None
        p.in_, p.act_scale, p.bias, p.out = act.data_ptr(), None, None, out.data_ptr()
        p.m, p.n, p.k, p.group_size = m, n, k, g
        p.weight_type, p.act_type = wt, (L.BF16 if dt == torch.bfloat16 else L.HALF)
        p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes
        return p

    def graph(no_skinny, out):
        if no_skinny:
            os.environ["MI355Q_NO_SKINNY"] = "1"
        else:
            os.environ.pop("MI355Q_NO_SKINNY", None)
        ps = [params(s, out) for s in range(sets)]
        side = torch.cuda.Stream()

        def run():
            for p in ps:
                L.check(lib.mi355q_weight_only_gemm(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        with torch.cuda.stream(side):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(gr, stream=side):
                run()
        return gr, ps

    torch.cuda.synchronize()
    g_old, keep0 = graph(True, outs[0])
    g_new, keep1 = graph(False, outs[1])
    g_old.replay()
    g_new.replay()
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        bad = torch.isnan(o.float())
        if bad.any():
            print(f"  out[{i}] has {int(bad.sum())} NaN: rows {bad.any(1).nonzero().flatten().tolist()[:6]} cols {bad.any(0).nonzero().flatten().tolist()[:10]}")
    diff = (outs[0].float() - outs[1].float()).abs().max().item()
    mag = outs[0].float().abs().max().item()

    def t(gr, reps=5):
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            gr.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (reps * sets) * 1e3

    a, b = [], []
    for _ in range(5):
        a.append(t(g_old))
        b.append(t(g_new))
    a.sort(), b.sort()
    wbytes = n * k // 2 if int4 else n * k
    print(f"m={m} n={n} k={k} {'int4' if int4 else 'int8'} g={g}: old {a[2]:.2f} us, skinny {b[2]:.2f} us "
          f"({wbytes / b[2] / 1e6:.2f} TB/s of weights) | max diff {diff:.3e} of {mag:.2f}", flush=True)
    return diff <= 2e-2 * max(mag, 1.0)


shapes = [(64, 3584, 8192), (32, 3584, 8192), (64, 8192, 3584), (64, 1280, 8192), (64, 8192, 1024), (48, 4096, 4096),
          (24, 4096, 4096), (32, 28672, 8192), (64, 28672, 8192), (40, 11008, 4096), (64, 4096, 11008)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in s.split(",")) for s in sys.argv[1:]]
ok = True
for (m, n, k) in shapes:
    ok &= case(m, n, k)
if len(sys.argv) == 1:
    ok &= case(64, 3584, 8192, g=64)
    ok &= case(33, 4096, 4096, g=0, zero=False)
    ok &= case(64, 3584, 8192, int4=False, g=0, zero=False)
    ok &= case(20, 4096, 4096, int4=False, g=128, dt=torch.float16)
print("ALL OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
