"""Lab: context attention through two builds of the library, interleaved in fresh processes is not possible (one library per
process), so: this process times ONE library (MI355Q_LIB) over the bench's shapes; run it once per build."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(B=4), dict(B=1, S=4096), dict(B=8, S=512), dict(B=2, H=64, Hkv=8, S=4096, kv="fp8"), dict(B=4, S=4096, past=2048)):
    ts = sorted(round(bench.bench_context_attention(dev, 10, **kw)["ms_per_call"] * 1e3, 1) for _ in range(5))
    print(os.environ.get("MI355Q_LIB", "default")[-24:], kw, ts, flush=True)
