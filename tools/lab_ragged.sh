# Lab: the ragged cfg2 variant (ctx 1024..2048) with forced split counts, LAB library.
cd $GRAFT_REPO_ROOT
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
for s in 0 2 3 4; do
  if [ $s = 0 ]; then unset MI355Q_ATTN_SPLITS; else export MI355Q_ATTN_SPLITS=$s; fi
  echo "== splits $s"
  timeout -k 10 200 python3 - <<'PY' || exit 1
import sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda", 0)
r = bench.bench_decode_attention(dev, 30, B=64, Hq=32, Hkv=32, kv="bf16", ragged=True, sets=2)
print("ragged", round(r["us_per_call"], 1), round(r["GBps"]), flush=True)
r = bench.bench_decode_attention(dev, 30, B=64, Hq=32, Hkv=32, kv="bf16", ragged=False, sets=2)
print("uniform", round(r["us_per_call"], 1), round(r["GBps"]), flush=True)
PY
done
