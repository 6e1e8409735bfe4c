#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into a short per-(kernel, grid) table (names trimmed)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path, only=None):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            n = short(r["Kernel_Name"])
            if only and only not in n:
                continue
            key = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")),
                   r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?"))
            rows[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':<112} {'grid':>9} {'wg':>5} {'vgpr':>5} {'lds':>7} {'calls':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}")
    for key, d in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        n, g, w, v, l = key
        print(f"{n:<112} {g:>9} {w:>5} {v:>5} {l:>7} {len(d):>6} {sum(d)/len(d)/1e3:>9.2f} {min(d)/1e3:>9.2f} {max(d)/1e3:>9.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
