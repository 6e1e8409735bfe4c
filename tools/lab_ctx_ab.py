"""Lab: context attention, release library against the LAB library (built with another scheduling strategy), two
processes would see different thermal states, so both are loaded in ONE process through ctypes and interleaved."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mi355q import _lib as L
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = {"release": L.lib(), "lab": C.CDLL(os.path.join(root, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))}
for lb in libs.values():
    lb.mi355q_context_attention.restype = C.c_int
    lb.mi355q_context_attention_workspace_size.restype = C.c_size_t
dev = torch.device("cuda", 0)
real = L.lib
for kw in (dict(B=4, H=32, Hkv=32, S=2048), dict(B=2, H=64, Hkv=8, S=4096), dict(B=1, H=32, Hkv=32, S=8192), dict(B=4, H=32, Hkv=32, S=2048, D=64)):
    res = {k: [] for k in libs}
    for rnd in range(3):
        for name, lb in libs.items():
            L.lib = lambda lb=lb: lb
            r = bench.bench_context_attention(dev, 20, **kw)
            res[name].append(round(r["ms_per_call"] * 1e3, 1))
    L.lib = real
    print(kw, res, flush=True)
