# Per-kernel counters of the whole bench (every secondary line): three rocprofv3 passes, each a run of its own.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc_all
O=gpurun_out/pmc_all
rm -rf $O/a $O/b $O/c
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/a -- python3 bench.py --steps 5 --warmup 2 --no-tp --no-cpu-baseline > $O/a.json 2> $O/a.err || { tail -20 $O/a.err; exit 1; }
echo pass A done
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/b -- python3 bench.py --steps 5 --warmup 2 --no-tp --no-cpu-baseline > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
echo pass B done
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c -- python3 bench.py --steps 5 --warmup 2 --no-tp --no-cpu-baseline > $O/c.json 2> $O/c.err || { tail -20 $O/c.err; exit 1; }
echo pass C done
python3 tools/pmc_all.py $O/a $O/b $O/c > $O/table.txt && cut -c1-190 $O/table.txt | head -70
