# PMC for the W4A16 prefill GEMM, 4096 x 4096 x 11008: 128 x 256 tile on 4 waves against 256 x 256 on 8 waves (LAB library)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmcp
export MI355Q_LIB=$GRAFT_REPO_ROOT/tensorrt-llm_amd/lib/libmi355q_lab.so
run() { # name, counters...
  name=$1; shift 1
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmcp/$name -- python3 tools/microbench.py gemv --m 4096 --n 4096 --k 11008 > gpurun_out/pmcp/$name.out 2>&1
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
echo "== 256 x 256, 8 waves"
run p256a SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES && \
run p256b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES && \
export MI355Q_PF_NO256=1 && echo "== 128 x 256, 4 waves" && \
run p128a SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES && \
run p128b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAVES
