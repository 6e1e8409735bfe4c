cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
timeout -k 10 300 python -u -m pytest tests/test_gpu_gemm.py tests/test_golden.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/lab/tests.log 2>&1 || { tail -30 gpurun_out/lab/tests.log; exit 1; }; tail -2 gpurun_out/lab/tests.log
for shape in "32 4096 4096" "64 4096 4096" "128 4096 4096" "512 4096 4096" "4096 4096 4096" "4096 11008 4096" "4096 4096 11008" "2048 28672 8192"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 --steps 6 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_ | tail -n +2 | awk -v m=$1 -v n=$2 -v k=$3 '{t=$(NF-2); printf "m=%d n=%d k=%d %s avg_us=%s TFLOPs=%.1f\n", m,n,k,$1,t, 2.0*m*n*k/t/1e6}'
done
