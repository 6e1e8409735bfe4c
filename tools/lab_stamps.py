"""Lab: phase stamps of the GQA decode kernel at cfg4 (needs the LAB build and MI355Q_ATTN_STAMPS=1)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from mi355q import _lib
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
kw = dict(Hq=int(os.environ.get("HQ", 8)), Hkv=int(os.environ.get("HKV", 1)), kv=os.environ.get("KV", "int8"), sets=2)
r = bench.bench_decode_attention(dev, 16, **kw)
torch.cuda.synchronize()
L = _lib.lib()
L.mi355q_lab_decode_stamps.restype = ctypes.POINTER(ctypes.c_uint64)
ptr = L.mi355q_lab_decode_stamps()
n = int(os.environ.get("WGS", 512))
a = np.ctypeslib.as_array(ptr, shape=(4096 * 16,)).reshape(4096, 16)[:n].astype(np.int64).copy()
t0 = a[:, 0].min()
names = ["start", "loads issued", "token staged", "loop begins", "loop done", "lds merge done", "partials stored", "drained+barrier", "counted", "merged (last)"]
print("us_per_call", round(r["us_per_call"], 2))
for i, nm in enumerate(names):
    col = a[:, i]
    ok = (col > 0) & (col >= a[:, 0])
    if ok.any():
        rel = (col[ok] - t0) / 100.0
        print(f"{nm:18s} n={ok.sum():4d}  min {rel.min():7.2f}  median {np.median(rel):7.2f}  max {rel.max():7.2f} us")
d = np.diff(a[:, :9], axis=1) / 100.0
print("median phase durations:", np.round(np.median(d, axis=0), 2))
