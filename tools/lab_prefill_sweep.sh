cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
set -o pipefail
for shape in "128 28672 8192" "256 28672 8192" "512 28672 8192" "128 8192 8192" "256 8192 8192" "128 11008 4096" "256 11008 4096" "512 4096 4096" "1024 4096 4096"; do
for nb in 2 4; do
for ks in 1 2 3 4 6; do
  set -- $shape
  MI355Q_PF_NB=$nb MI355Q_PF_KS=$ks timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/g -- python3 tools/microbench.py gemv --m $1 --n $2 --k $3 > gpurun_out/lab/out.txt 2>/dev/null || exit 1
  echo "m=$1 n=$2 k=$3 nb=$nb ks=$ks $(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/g -name '*kernel_trace.csv') | head -1) woq_ | tail -n +2 | awk '{printf "%s ", $(NF-2)}')"
  rm -rf gpurun_out/lab/g
done
done
done
