"""List every kernel of csrc/*.hip that uses scratch memory (register spills), from hipcc's resource remarks.
A spill inside a hot loop is silent in every parity test and can cost several x (the W8A16 128 x 256 prefill tile ran at
120 TFLOP/s for that reason); run this after touching a kernel's register budget.  CPU only: python3 tools/scratch_scan.py
[-DMI355Q_LAB]"""
import glob, os, re, subprocess, sys, tempfile

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tensorrt-llm_amd", "csrc")
extra = sys.argv[1:]
bad = 0
for src in sorted(glob.glob(os.path.join(root, "*.hip"))):
    # the Makefile's per-file flags (SLP off for the weight-only GEMMs)
    per_file = ["-fno-slp-vectorize"] if os.path.basename(src) in ("woq_gemm.hip", "woq_gemv_small.hip", "woq_prefill.hip") else []
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off",
                            "-Rpass-analysis=kernel-resource-usage", *per_file, *extra, "-c", src, "-o", os.path.join(td, "x.o")],
                           capture_output=True, text=True, cwd=root)
    seen = set()
    for blk in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
        name = blk.split("\n")[0].strip().split()[0]
        if name in seen:
            continue
        seen.add(name)
        sc = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", blk).group(1))
        if sc:
            vg = int(re.search(r"VGPRs: (\d+)", blk).group(1))
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
            dn = re.sub(r"mi355q::|\(anonymous namespace\)::", "", dn)
            print(f"{os.path.basename(src):22s} scratch {sc:5d} B/lane  vgpr {vg:3d}  {dn[:120]}")
            bad += 1
print(f"{bad} kernel(s) with scratch")
