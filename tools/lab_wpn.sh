cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
for mx in 64; do
export MI355Q_GEMV_MAX_M=$mx
echo "== MAX_M $mx"
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py -m gpu -x -q --timeout 120 --timeout-method thread -k "weight_only or groupwise or ingested" > gpurun_out/lab/tests.log 2>&1 || { tail -30 gpurun_out/lab/tests.log; exit 1; }; tail -1 gpurun_out/lab/tests.log
for shape in "64 28672 8192" "48 28672 8192" "64 8192 8192" "64 11008 4096" "64 4096 11008" "64 4096 4096" "40 8192 28672"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "m=$1 n=$2 k=$3 $(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_ | tail -n +2 | awk '{printf "%s ", $(NF-2)}')"
  rm -rf gpurun_out/lab/g
done
done
