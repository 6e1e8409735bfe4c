cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
for c in cfg2 cfg2r cfg4 cfg4b cfg5 gqa16 b1 b8; do
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/a_$c -- python3 tools/lab_attn.py $c > gpurun_out/lab/attn_$c.txt 2>/dev/null || { echo "FAILED $c"; exit 1; }
  grep CASE gpurun_out/lab/attn_$c.txt
  python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/a_$c -name '*kernel_trace.csv') | head -1) decode_attn | tail -n +2 | cut -c1-80,112-200
done
