cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
for shape in "48 28672 8192" "64 28672 8192" "64 8192 8192" "64 11008 4096" "64 4096 4096" "48 4096 4096" "128 28672 8192" "128 4096 4096"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "m=$1 n=$2 k=$3 $(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_ | tail -n +2 | awk '{printf "%s ", $(NF-2)}')"
  rm -rf gpurun_out/lab/g
done
