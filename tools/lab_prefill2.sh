cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_gemm.py tests/test_gpu_end_to_end.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/lab/tests.log 2>&1 || { tail -30 gpurun_out/lab/tests.log; exit 1; }; tail -1 gpurun_out/lab/tests.log
for shape in "64 28672 8192" "128 28672 8192" "256 28672 8192" "512 28672 8192" "128 4096 4096" "256 4096 4096" "4096 4096 4096" "4096 11008 4096" "2048 28672 8192"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "m=$1 n=$2 k=$3 $(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_ | tail -n +2 | awk '{print $1, $(NF-2)}' | cut -c9-60 | tr '\n' ';')"
done
