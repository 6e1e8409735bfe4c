"""Lab: context attention with q taken straight from the qkv rows (bias + rotation in the attention kernel's prologue)
against the prepare pass writing q to the workspace (MI355Q_CTX_Q_PASS=1), LAB library, interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for kw in (dict(B=4), dict(B=1, S=4096), dict(B=8, S=512), dict(B=4, D=64, S=2048), dict(B=2, H=64, Hkv=8, S=4096),
           dict(B=2, H=64, Hkv=8, S=4096, kv="fp8"), dict(B=4, S=4096, past=2048)):
    res = {"fused": [], "pass": []}
    for rep in range(3):
        for name in ("fused", "pass"):
            if name == "pass":
                os.environ["MI355Q_CTX_Q_PASS"] = "1"
            else:
                os.environ.pop("MI355Q_CTX_Q_PASS", None)
            r = bench.bench_context_attention(dev, 10, **kw)
            res[name].append(round(r["ms_per_call"] * 1e3, 1))
    print(kw, res, flush=True)
