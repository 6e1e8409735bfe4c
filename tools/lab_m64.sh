# Lab: m = 33..64 through the decode (GEMV) kernel instead of the prefill kernel (MI355Q_GEMV_MAX_M), LAB library.
cd $GRAFT_REPO_ROOT
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
for mm in 32 64; do
  export MI355Q_GEMV_MAX_M=$mm
  echo "== GEMV_MAX_M $mm"
  timeout -k 10 200 python3 - <<'PY' || exit 1
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
for (m, n, k) in ((64, 3584, 8192), (64, 8192, 3584), (64, 1280, 8192), (64, 8192, 1024), (48, 4096, 4096), (64, 28672, 8192), (40, 11008, 4096)):
    sets = max(2, min(64, (300 << 20) // (n * k // 2)))
    try:
        r = bench.bench_gemv(dev, 2 * sets, m=m, n=n, k=k, sets=sets)
        print(m, n, k, round(r["us_per_call"], 2), flush=True)
    except Exception as e:
        print(m, n, k, "failed:", str(e)[:120], flush=True)
PY
done
