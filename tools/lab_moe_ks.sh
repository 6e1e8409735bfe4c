# Lab: the two expert GEMMs of the decode MoE layer (64 tokens) under forced split-K (MI355Q_MOE_KS), LAB library, kernel trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
for ks in 0 1 2 4 8; do
  if [ $ks -gt 0 ]; then export MI355Q_MOE_KS=$ks; fi
  rm -rf gpurun_out/lab/mk
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/mk -- python3 tools/lab_moe_layer.py ${TOK:-64} > gpurun_out/lab/mk.txt 2>/dev/null || exit 1
  echo "ks=$ks $(python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/mk -name '*kernel_trace.csv') | head -1) woq_ | tail -n +2 | awk '{print $(NF-6), $(NF-2)}' | tr '\n' ';')  layer: $(grep -o "us_per_call.: [0-9.]*" gpurun_out/lab/mk.txt | head -1)"
done
