cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
for dbg in 0 1 2 4 5 6 7; do
for shape in "1 4096 4096" "1 28672 8192"; do
  set -- $shape
  MI355Q_DBG=$dbg timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null
  echo "dbg=$dbg m=$1 n=$2 k=$3 rocprof(avg,min,max)=$(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_gemv | tail -1 | awk '{print $(NF-2), $(NF-1), $NF}')"
done
done
