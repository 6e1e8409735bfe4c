cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
for cfg in "4 2 256" "4 3 256" "2 2 256" "2 3 256"; do
  set -- $cfg
  export MI355Q_GEMV_CH=$1 MI355Q_GEMV_DEPTH=$2 MI355Q_GEMV_SLOTS=$3
  for shape in "1 4096 4096" "16 4096 4096" "1 11008 4096" "1 4096 11008" "1 8192 8192" "1 28672 8192" "16 28672 8192"; do
    set -- $shape
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
    echo "cfg=[$cfg] m=$1 n=$2 k=$3 rocprof(avg,min,max)=$(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_gemv | tail -1 | awk '{print $(NF-2), $(NF-1), $NF}')"
  done
done
