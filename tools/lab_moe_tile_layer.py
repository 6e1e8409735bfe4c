"""Lab: tile height of the MoE layer's expert GEMMs after the grouped tile walk: default choice against forced 256-row
(MI355Q_MOE_MT=16), 128-row (8) and 64-row (4) tiles; whole layer (random routing), LAB library, interleaved."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MI355Q_LIB", os.path.join(ROOT, "tensorrt-llm_amd", "lib", "libmi355q_lab.so"))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for tok in (int(a) for a in (sys.argv[1:] or ["768", "1024", "1536", "2048", "3072"])):
    res = {}
    for rep in range(3):
        for name, env in (("default", {}), ("mt16", {"MI355Q_MOE_MT": "16"}), ("mt8", {"MI355Q_MOE_MT": "8"}), ("mt4", {"MI355Q_MOE_MT": "4"}),
                          ("mt2", {"MI355Q_MOE_MT": "2"})):
            os.environ.pop("MI355Q_MOE_MT", None)
            os.environ.update(env)
            res.setdefault(name, []).append(round(bench.bench_moe_layer(dev, 10, tokens=tok)["us_per_call"]))
    print(tok, res, flush=True)
