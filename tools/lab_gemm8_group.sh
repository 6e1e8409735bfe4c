cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py -m gpu -x -q --timeout 120 --timeout-method thread -k "smooth or fp8" 2>&1 | tail -3 || exit 1
cat > /tmp/g8bench.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
dev = torch.device("cuda", 0)
print("int8", [round(bench.bench_int8_gemm(dev, 20, **kw)["TOPS"]) for kw in (dict(), dict(n=5120, k=13824), dict(m=4096, n=4096, k=4096), dict(m=8192, n=8192, k=8192))], flush=True)
print("fp8 ", [round(bench.bench_fp8_gemm(dev, 20, **kw)["TFLOPS"]) for kw in (dict(), dict(n=4096, k=14336), dict(m=4096, n=4096, k=4096), dict(m=8192, n=8192, k=8192))], flush=True)
PY
for gm in 1 2 4 8 16; do echo "== GROUP_M $gm"; MI355Q_GEMM8_GROUP_M=$gm timeout -k 10 200 python3 /tmp/g8bench.py || exit 1; done
