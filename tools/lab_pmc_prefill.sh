cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcp
run() { # name, m n k, counters...
  name=$1; M=$2; N=$3; K=$4; shift 4
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcp/$name -- python3 tools/microbench.py gemv --m $M --n $N --k $K > gpurun_out/pmcp/$name.out 2>&1
  f=$(find gpurun_out/pmcp/$name -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
dur = []
for r in csv.DictReader(open(sys.argv[1])):
    if "woq_prefill" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    print(f"  {k:28s} avg/launch = {sum(v)/len(v):.5g}  (n={len(v)})")
print(f"  duration under pmc: avg {sum(dur)/max(1,len(dur)):.2f} us")
PY
}
for shape in "128 28672 8192" "4096 4096 4096"; do
set -- $shape
echo "== $shape"
run a$1 $1 $2 $3 SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
run b$1 $1 $2 $3 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES
done
