# tests + timing + kernel trace for the context attention (run on the GPU box)
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_context_attention.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/ctx_test.log 2>&1 || { tail -30 gpurun_out/ctx_test.log; exit 1; }
tail -2 gpurun_out/ctx_test.log
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/ctxprof
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ctxprof -- python3 $GRAFT_REPO_ROOT/tools/lab_ctx.py > $GRAFT_REPO_ROOT/gpurun_out/ctx_lab.log 2>&1 || { tail $GRAFT_REPO_ROOT/gpurun_out/ctx_lab.log; exit 1; }
grep batch $GRAFT_REPO_ROOT/gpurun_out/ctx_lab.log | cut -c1-200
python3 $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $(find $GRAFT_REPO_ROOT/gpurun_out/ctxprof -name "*kernel_trace.csv" | head -1) | grep "ctx_\|kernel " | cut -c1-60,110-190
