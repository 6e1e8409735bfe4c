cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc
run() { # name, counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc/$name -- python3 tools/lab_ctx_one.py > gpurun_out/pmc/$name.out 2>&1
  f=$(find gpurun_out/pmc/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ctx_attn" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"  {k:28s} avg/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
}
run d1 GRBM_GUI_ACTIVE GRBM_COUNT
run d2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_CYCLES
