"""Lab: phase stamps (s_memrealtime, 100 MHz) of woq_skinny_kernel; LAB build, MI355Q_SKINNY_STAMPS=1.
usage: lab_skinny_stamps.py m n k"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tensorrt-llm_amd"))
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
os.environ["MI355Q_SKINNY_STAMPS"] = "1"
import numpy as np, torch
from mi355q import _lib as L
m, n, k = (int(v) for v in sys.argv[1:4])
dev = torch.device("cuda")
lib = L.lib()
g = 128
sets = 6
w = torch.randint(-128, 128, (sets, k, n // 2), dtype=torch.int8, device=dev)
scale = torch.rand(sets, k // g, n, device=dev).bfloat16()
zero = torch.randn(sets, k // g, n, device=dev).bfloat16()
act = torch.randn(m, k, device=dev).bfloat16()
out = torch.empty(m, n, device=dev).bfloat16()
ws_bytes = lib.mi355q_weight_only_gemm_workspace_size(m, n, k, L.WEIGHT_INT4)
ws = torch.empty(max(ws_bytes, 256), dtype=torch.uint8, device=dev)
for s in range(sets):
    p = L.WeightOnlyParams()
    p.qweight, p.scales, p.zeros = w[s].data_ptr(), scale[s].data_ptr(), zero[s].data_ptr()
    p.in_, p.out = act.data_ptr(), out.data_ptr()
    p.m, p.n, p.k, p.group_size = m, n, k, g
    p.weight_type, p.act_type = L.WEIGHT_INT4, L.BF16
    p.workspace, p.workspace_bytes = ws.data_ptr(), ws_bytes
    L.check(lib.mi355q_weight_only_gemm(ctypes.byref(p), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
torch.cuda.synchronize()
lib.mi355q_lab_skinny_stamps.restype = ctypes.POINTER(ctypes.c_uint64)
ptr = lib.mi355q_lab_skinny_stamps()
a = np.ctypeslib.as_array(ptr, shape=(8192 * 16,)).reshape(8192, 16).astype(np.int64).copy()
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
names = ["start", "loads issued", "tile 0 landed", "barrier", "compute done", "published+drained", "counted", "reduced (last)"]
print("workgroups", len(a))
for i, nm in enumerate(names):
    col = a[:, i]
    ok = col >= a[:, 0]
    if i == 7:
        ok &= col >= a[:, 6]
    if ok.any():
        rel = (col[ok] - t0) / 100.0
        print(f"{nm:18s} n={ok.sum():5d}  min {rel.min():7.2f}  median {np.median(rel):7.2f}  p90 {np.percentile(rel, 90):7.2f}  max {rel.max():7.2f} us")
