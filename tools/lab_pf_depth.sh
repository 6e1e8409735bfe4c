# Lab: the 4-slot ring of the prefill kernel's short tiles against the ping-pong (MI355Q_PF_DEPTH2), LAB library.
cd $GRAFT_REPO_ROOT
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
for d in ring pingpong; do
  if [ $d = pingpong ]; then export MI355Q_PF_DEPTH2=1; else unset MI355Q_PF_DEPTH2; fi
  echo "== $d"
  timeout -k 10 200 python3 tools/lab_moe_layer.py 64 2>&1 | grep tokens | cut -c1-200 || exit 1
  timeout -k 10 200 python3 - <<'PY' || exit 1
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
for (m, n, k) in ((64, 3584, 8192), (64, 8192, 3584), (64, 1280, 8192), (64, 8192, 1024), (48, 4096, 4096), (64, 28672, 8192), (40, 11008, 4096)):
    sets = max(2, min(64, (300 << 20) // (n * k // 2)))
    r = bench.bench_gemv(dev, 2 * sets, m=m, n=n, k=k, sets=sets)
    print(m, n, k, None if r is None else round(r["us_per_call"], 2), flush=True)
PY
done
