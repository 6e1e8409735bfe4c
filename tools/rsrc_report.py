"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output: kernel, VGPRs, AGPRs, spills, scratch, occupancy, LDS."""
import re, sys, subprocess
cur = None
rows = []
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"remark: [^ ]+ +(?:Function )?Name: (\S+)", line) or re.search(r"Name: (\S+)", line)
    if m and "Name:" in line:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"(VGPRs|AGPRs|SGPRs Spill|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" [")[0]] = int(m.group(2))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    if flt and flt not in name:
        continue
    name = re.sub(r"mi355q::\(anonymous namespace\)::", "", name)[:90]
    print(f'{name:90s} v{r.get("VGPRs", -1):4d} a{r.get("AGPRs", -1):4d} vspill{r.get("VGPRs Spill", -1):4d} sspill{r.get("SGPRs Spill", -1):4d} scratch{r.get("ScratchSize", -1):5d} occ{r.get("Occupancy", -1):2d} lds{r.get("LDS Size", -1):6d}')
