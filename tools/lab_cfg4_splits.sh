cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
for s in ${SPLITS:-0 2 4 8}; do
echo "== splits $s"
MI355Q_ATTN_SPLITS=$s timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lab/c4 -- python3 tools/lab_cfg4.py > gpurun_out/lab/c4.out 2>/dev/null
python3 tools/summarize_rocprof.py $(ls -t $(find gpurun_out/lab/c4 -name '*kernel_trace.csv') | head -1) decode_attn | awk '{print $1, $2, $3, $(NF-5), $(NF-4), $(NF-3), $(NF-2)}' | cut -c1-150
rm -rf gpurun_out/lab/c4
done
