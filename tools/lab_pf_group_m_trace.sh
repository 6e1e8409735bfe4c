# Lab: per-kernel times of the MoE layer at ${TOK:-4096} tokens under MI355Q_PF_GROUP_M = 0 / 4 (LAB library, kernel trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
for g in 0 4; do
  export MI355Q_PF_GROUP_M=$g
  rm -rf gpurun_out/lab/gm
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/gm -- python3 tools/lab_moe_layer.py ${TOK:-4096} > gpurun_out/lab/gm.txt 2>/dev/null || exit 1
  echo "G=$g"; python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/gm -name '*kernel_trace.csv') | head -1) woq_ | cut -c1-60,100-200
done
