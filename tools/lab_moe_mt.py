"""Lab: MoE layer over the token count with the expert GEMMs' tile height forced (MI355Q_MOE_MT: 2 / 4 / 8 m-blocks, 16 =
the 256 x 256 tile where eligible; LAB library).  Real routing: the experts' row counts are ragged."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
toks = [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128, 384, 768, 2048]
for tok in toks:
    row = []
    for mt in ("2", "4", "8", "16"):
        if mt == "16":
            os.environ.pop("MI355Q_MOE_MT", None)
            os.environ.pop("MI355Q_PF_NO256", None)
        else:
            os.environ["MI355Q_MOE_MT"] = mt
            os.environ["MI355Q_PF_NO256"] = "1"
        best = min(bench.bench_moe_layer(dev, 10, tokens=tok)["us_per_call"] for _ in range(2))
        row.append(f"mt{mt} {best:7.1f}")
    print(tok, "tokens (avg rows/expert", tok * 2 // 8, "):", "  ".join(row), flush=True)
