cd $GRAFT_REPO_ROOT
cat > /tmp/g8bench.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
dev = torch.device("cuda", 0)
print("int8", [round(bench.bench_int8_gemm(dev, 20, **kw)["TOPS"]) for kw in (dict(), dict(n=5120, k=13824), dict(m=4096, n=14336, k=4096), dict(m=2048, n=13824, k=5120), dict(m=8192, n=13824, k=5120))], flush=True)
PY
for t in "" 0; do echo "== TILE '$t'"; MI355Q_GEMM8_TILE=$t timeout -k 10 200 python3 /tmp/g8bench.py || exit 1; done
