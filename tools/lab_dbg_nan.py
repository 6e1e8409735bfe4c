"""Lab: repeat one W4A16 GEMM launch many times and compare every result with the first (a race shows as a difference).
usage: lab_dbg_nan.py m n k [iterations] ; MI355Q_NO_SKINNY=1 for the ping-pong path (LAB library)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tensorrt-llm_amd")
os.environ.setdefault("MI355Q_LIB", ROOT + "/tensorrt-llm_amd/lib/libmi355q_lab.so")
import torch
from mi355q import _lib as L
dev = torch.device("cuda"); lib = L.lib()
m, n, k = (int(v) for v in sys.argv[1:4]); iters = int(sys.argv[4]) if len(sys.argv) > 4 else 200
g = 128
torch.manual_seed(1)
sets = 3
w = torch.randint(-128, 128, (sets, k, n // 2), dtype=torch.int8, device=dev)
scale = (torch.rand(sets, k // g, n, device=dev) * 0.02 + 0.001).bfloat16()
zeros = (torch.randn(sets, k // g, n, device=dev) * 0.05).bfloat16()
act = torch.randn(m, k, device=dev).bfloat16()
wsb = lib.mi355q_weight_only_gemm_workspace_size(m, n, k, L.WEIGHT_INT4)
ws = torch.full((max(wsb, 256),), 0xff, dtype=torch.uint8, device=dev)
outs = [torch.zeros(m, n, device=dev).bfloat16() for _ in range(sets)]
ref = [None] * sets
torch.cuda.synchronize()
bad = 0
for it in range(iters):
    for s in range(sets):
        p = L.WeightOnlyParams()
        p.qweight, p.scales, p.zeros = w[s].data_ptr(), scale[s].data_ptr(), zeros[s].data_ptr()
        p.in_, p.out = act.data_ptr(), outs[s].data_ptr()
        p.m, p.n, p.k, p.group_size = m, n, k, g
        p.weight_type, p.act_type = L.WEIGHT_INT4, L.BF16
        p.workspace, p.workspace_bytes = ws.data_ptr(), wsb
        L.check(lib.mi355q_weight_only_gemm(C.byref(p), C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    torch.cuda.synchronize()
    for s in range(sets):
        if ref[s] is None:
            ref[s] = outs[s].clone()
            assert not torch.isnan(ref[s].float()).any(), "NaN in the first result"
        elif not torch.equal(ref[s].view(torch.int16), outs[s].view(torch.int16)):
            d = (ref[s].view(torch.int16) != outs[s].view(torch.int16))
            bad += 1
            if bad < 5:
                print(f"iteration {it} set {s}: {int(d.sum())} elements differ, rows {d.any(1).nonzero().flatten().tolist()[:6]} cols {d.any(0).nonzero().flatten().tolist()[:8]}", flush=True)
        outs[s].fill_(float("nan"))
print(f"m={m} n={n} k={k}: {iters} iterations x {sets} sets, {bad} differing results")
sys.exit(1 if bad else 0)
