cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/lab/tests_all.log 2>&1 || { tail -30 gpurun_out/lab/tests_all.log; exit 1; }; tail -1 gpurun_out/lab/tests_all.log
for shape in "1 28672 8192" "24 11008 4096" "16 28672 8192" "8 11008 4096" "4 4096 4096" "1 5120 5120" "1 13824 5120" "1 5120 13824" "1 4096 4096"; do
  set -- $shape
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || { cat gpurun_out/lab/out.txt | tail -5; exit 1; }
  echo "m=$1 n=$2 k=$3 rocprof(avg,min,max)=$(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_ | tail -2 | awk '{print $1, $(NF-2), $(NF-1), $NF}' | tr '\n' ';')"
done
