# Round-end style GPU run: tests, bench, kernel trace, PMC traffic.  Run through gpurun from the repo root.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/round
set -o pipefail
O=gpurun_out/round
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
# the headline ALONE (no secondary lines, no TP leg, no CPU baseline): the decode kernel's row is then cfg2 only and its
# average reproduces the line's roofline.achieved without de-blending (measurement hygiene)
rm -rf $O/trace_headline $O/trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_headline -- python3 bench.py --steps 50 --no-secondary --no-tp --no-cpu-baseline > $O/bench_traced_headline.json 2> $O/trace_headline.err || { tail -20 $O/trace_headline.err; exit 1; }
python3 tools/summarize_rocprof.py $(ls -t $(find $O/trace_headline -name '*kernel_trace.csv') | head -1) > $O/kernel_trace_summary_headline.txt || exit 1
# every line of the bench (secondary kernels included)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 20 --no-cpu-baseline > $O/bench_traced.json 2> $O/trace.err || { tail -20 $O/trace.err; exit 1; }
python3 tools/summarize_rocprof.py $(ls -t $(find $O/trace -name '*kernel_trace.csv') | head -1) > $O/kernel_trace_summary.txt || exit 1
rm -rf $O/pmc_fetch $O/pmc_write
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 5 --warmup 2 --no-secondary --no-tp --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err || { tail -20 $O/pmc_fetch.err; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 5 --warmup 2 --no-secondary --no-tp --no-cpu-baseline > /dev/null 2> $O/pmc_write.err || { tail -20 $O/pmc_write.err; exit 1; }
python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write decode_attn_kernel $O/hbm_traffic.json
cat $O/bench.json
