cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
for cfg in "256 32" "512 22" "512 31" "512 41" "512 32" "768 22"; do
set -- $cfg
export MI355Q_GEMV_CUS=$1 MI355Q_GEMV_RING=$2
echo "== cus $1 ring $2"
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py -m gpu -x -q --timeout 120 --timeout-method thread -k "weight_only or groupwise or ingested" > gpurun_out/lab/tests.log 2>&1 || { tail -30 gpurun_out/lab/tests.log; exit 1; }; tail -1 gpurun_out/lab/tests.log
for shape in "1 4096 4096" "1 11008 4096" "1 4096 11008" "1 28672 8192" "1 8192 8192" "16 28672 8192" "1 8192 28672"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "m=$1 n=$2 k=$3 rocprof(avg,min,max)=$(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_gemv | tail -1 | awk '{print $(NF-2), $(NF-1), $NF}')"
done
done
