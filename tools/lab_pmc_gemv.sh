cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcv
M=${M:-1}; N=${N:-28672}; K=${K:-8192}
run() { # name, counters...
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcv/$name -- python3 tools/microbench.py gemv --m $M --n $N --k $K > gpurun_out/pmcv/$name.out 2>&1
  f=$(find gpurun_out/pmcv/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
dur = []
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "woq_gemv" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r:
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print(f"  {k:28s} avg/launch = {sum(v)/len(v):.5g}  (n={len(v)})")
if dur:
    print(f"  duration under pmc: avg {sum(dur)/len(dur):.2f} us")
PY
}
for lab in ${LABS:-0 1 2}; do
export MI355Q_GEMV_LAB=$lab
echo "== lab $lab"
run c$lab SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU
run g$lab GRBM_GUI_ACTIVE
done
