# Round-2 int8 GEMM evidence: per variant, kernel duration (kernel trace) + GRBM_GUI_ACTIVE + MFMA busy cycles in ONE
# run each (kernel-trace + pmc only), so that effective clock = GRBM_GUI_ACTIVE / 8 / duration and
# MFMA-busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8) come from the same dispatches.
# Needs the LAB library (variants are lab switches).  Usage: bash tools/lab_pmc_gemm8_r2.sh M N K
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/pmc2
M=${1:-8192}; N=${2:-8192}; K=${3:-8192}
run() { # name [VAR=value ...]: the variables are exported in a subshell around the profiler, never behind its "--"
  name=$1; shift
  dir=$(echo "$name" | tr -c "a-zA-Z0-9_" "_")
  (
  for kv in "$@"; do export "$kv"; done
  rm -rf gpurun_out/pmc2/$dir
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc2/$dir -- python3 tools/microbench.py gemm8 --m $M --n $N --k $K --steps 12 > gpurun_out/pmc2/$dir.out 2>&1 || return 1
  python3 - gpurun_out/pmc2/$dir "$name" $M $N $K <<'PY'
import csv, sys, glob, collections
d, name, M, N, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    if "gemm8" in r["Kernel_Name"]:
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
acc = collections.defaultdict(dict)
for r in csv.DictReader(open(cc)):
    if "gemm8" in r["Kernel_Name"]:
        acc[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(set(dur) & set(acc), key=int)[2:]          # skip warm-up
# a problem may be two launches (main + remainder): sum per step is not needed for the ratios; report the longest kernel
longest = max(dur[i] for i in ids)
sel = [i for i in ids if dur[i] > 0.6 * longest]
t = sum(dur[i] for i in sel) / len(sel)
g = sum(acc[i]["GRBM_GUI_ACTIVE"] for i in sel) / len(sel) / 8
mf = sum(acc[i]["SQ_VALU_MFMA_BUSY_CYCLES"] for i in sel) / len(sel)
print(f"{name:22s} {M}x{N}x{K}: kernel {t*1e6:7.1f} us under PMC, effective clock {g/t/1e9:5.2f} GHz, "
      f"MFMA pipe busy {mf/(1024*g)*100:5.1f} % of SIMD cycles, {len(sel)} dispatches")
PY
  )
}
run "phased (shipped)" &&
run "w128" MI355Q_GEMM8_W128=1 &&
run "w128 no-DMA" MI355Q_GEMM8_W128=1 MI355Q_GEMM8_ABL=1 &&
run "w128 no-DMA no-LDS" MI355Q_GEMM8_W128=1 MI355Q_GEMM8_ABL=3
