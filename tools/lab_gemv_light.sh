# Lab: light decode GEMV (woq_gemv_small.hip) against the persistent kernel, graph-timed (bench.bench_gemv) and traced.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/lab/tests.log 2>&1 || { tail -30 gpurun_out/lab/tests.log; exit 1; }; tail -2 gpurun_out/lab/tests.log
for shape in "1 4096 4096" "4 4096 4096" "16 4096 4096" "1 12288 4096" "1 11008 4096" "1 4096 11008" "1 1280 8192" "1 3584 8192" "1 8192 1024" "1 8192 3584" "8 3584 8192" "1 8192 8192" "1 28672 8192" "16 28672 8192"; do
  set -- $shape
  for mode in 0 1; do
    MI355Q_GEMV_LIGHT=$mode timeout -k 10 120 python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out_$mode.txt 2>gpurun_out/lab/err.txt || { tail -5 gpurun_out/lab/err.txt; exit 1; }
  done
  python3 - "$@" <<'PY'
import json, sys
o = [json.load(open(f"gpurun_out/lab/out_{m}.txt")) for m in (0, 1)]
print(f"m={sys.argv[1]} n={sys.argv[2]} k={sys.argv[3]} MB={o[0]['algorithmic_bytes']/1e6:.1f}  persistent {o[0]['us_per_call']:.2f} us ({o[0]['frac_of_8TBps']*100:.0f}%)  light {o[1]['us_per_call']:.2f} us ({o[1]['frac_of_8TBps']*100:.0f}%)", flush=True)
PY
done
for shape in "1 4096 4096" "16 4096 4096" "1 1280 8192"; do
  set -- $shape
  MI355Q_GEMV_LIGHT=1 timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "traced m=$1 n=$2 k=$3: $(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_gemv | tail -1 | awk '{print $(NF-5), $(NF-2), $(NF-1), $NF}')"
done
