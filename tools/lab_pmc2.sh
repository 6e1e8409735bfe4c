cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/${LAB_LIB:-libmi355q.so}
export MI355Q_GEMV_CH=4 MI355Q_GEMV_DEPTH=2
run() { # name, counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python3 tools/microbench.py gemv --m 1 --n 28672 --k 8192 > gpurun_out/pmc/$name.out 2>&1
  f=$(find gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "woq_gemv" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:28s} avg/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
}
echo "== $MI355Q_LIB"
run a1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run a2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA
run a3 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_ACTIVE_INST_FLAT SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_SMEM
run g GRBM_GUI_ACTIVE GRBM_COUNT
