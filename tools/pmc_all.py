"""Per-kernel counter table of a whole bench run, from three rocprofv3 passes:
  pass A: --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT
  pass B: --pmc FETCH_SIZE      pass C: --pmc WRITE_SIZE
Keyed by (kernel, grid).  Effective clock = GRBM_GUI_ACTIVE / 8 / duration; MFMA-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES /
(1024 SIMDs x GRBM_GUI_ACTIVE / 8); HBM bytes = 2 x FETCH_SIZE KiB (gfx950: a wide coalesced read stream reports half,
MI355X_MICROARCH.md) + WRITE_SIZE KiB.  Durations are those of the PMC run itself (kernels serialised by the profiler).
usage: pmc_all.py dirA dirB dirC > table.txt"""
import csv, glob, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::|mi355q::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:96]


def counters(d):
    out = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mi355q" not in r["Kernel_Name"]:
                continue
            key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "?")))
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def durations(d):
    out = defaultdict(list)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "mi355q" not in r["Kernel_Name"]:
                continue
            if "Grid_Size_X" in r:  # the counter files give the product of the three dimensions
                grid = str(int(r["Grid_Size_X"]) * int(r.get("Grid_Size_Y", 1)) * int(r.get("Grid_Size_Z", 1)))
            else:
                grid = r.get("Grid_Size", "?")
            key = (short(r["Kernel_Name"]), grid)
            out[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    return out


a, b, c = (counters(x) for x in sys.argv[1:4])
dur = durations(sys.argv[1])
mean = lambda v: sum(v) / len(v) if v else None
print(f"{'kernel':<98} {'grid':>9} {'calls':>5} {'us':>9} {'GHz':>5} {'MFMA%':>6} {'LDSconf%':>8} {'readMB':>9} {'writeMB':>9} {'TB/s':>6}")
for key in sorted(dur, key=lambda k: -sum(dur[k])):
    t = mean(dur[key])
    g = mean(a[key].get("GRBM_GUI_ACTIVE", []))
    mf = mean(a[key].get("SQ_VALU_MFMA_BUSY_CYCLES", []))
    sb = mean(a[key].get("SQ_BUSY_CYCLES", []))
    lc = mean(a[key].get("SQ_LDS_BANK_CONFLICT", []))
    fs = mean(b.get(key, {}).get("FETCH_SIZE", []))
    ws = mean(c.get(key, {}).get("WRITE_SIZE", []))
    ghz = f"{g / 8 / t / 1e9:5.2f}" if g and t else "    -"
    mfp = f"{mf / (1024 * g / 8) * 100:6.1f}" if mf is not None and g else "     -"
    lcp = f"{lc / (1024 * g / 8) * 100:8.2f}" if lc is not None and g else "       -"
    rd = 2 * fs * 1024 if fs is not None else None
    wr = ws * 1024 if ws is not None else None
    tb = f"{(rd + wr) / t / 1e12:6.2f}" if rd is not None and wr is not None else "     -"
    print(f"{key[0]:<98} {key[1]:>9} {len(dur[key]):>5} {t * 1e6:>9.2f} {ghz} {mfp} {lcp} "
          f"{(rd / 1e6 if rd is not None else float('nan')):>9.1f} {(wr / 1e6 if wr is not None else float('nan')):>9.1f} {tb}")
