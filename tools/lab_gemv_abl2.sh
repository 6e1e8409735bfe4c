cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
for ring in 0 1 2; do
for lab in 0 1 5 2; do
export MI355Q_GEMV_LAB=$lab MI355Q_GEMV_RING=$ring
for shape in "1 28672 8192"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "ring=$ring lab=$lab m=$1 n=$2 k=$3 rocprof(avg,min,max)=$(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_gemv | tail -1 | awk '{print $(NF-2), $(NF-1), $NF}')"
done
done
done
