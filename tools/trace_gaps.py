#!/usr/bin/env python3
"""Per-launch duration and the gap to the previous kernel, from a rocprofv3 --kernel-trace csv.
  python tools/trace_gaps.py <dir-or-csv> [kernel-name-substring]"""
import csv, glob, os, sys
p = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "nsg_spec_step"
files = [p] if os.path.isfile(p) else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
prev_end = None
out = []
for s, e, name in rows:
    if sub in name:
        out.append((s, e, (s - prev_end) if prev_end is not None else 0, name))
    prev_end = e
print(f"{len(out)} launches of *{sub}*")
# split into runs separated by gaps > 1 ms
runs, cur = [], []
for s, e, gap, name in out:
    if cur and gap > 1_000_000:
        runs.append(cur); cur = []
    cur.append((s, e, gap))
if cur:
    runs.append(cur)
for r in runs:
    d = [(e - s) / 1e3 for s, e, _ in r]
    g = [gap / 1e3 for _, _, gap in r[1:]]
    span = (r[-1][1] - r[0][0]) / 1e3
    print(f"run of {len(r):5d}: span {span:9.1f} us  mean dur {sum(d)/len(d):6.2f}  min {min(d):6.2f} max {max(d):6.2f}  "
          f"mean gap {sum(g)/max(len(g),1):6.2f} max gap {max(g) if g else 0:7.2f}")
    if len(r) <= 40:
        print("   dur:", " ".join(f"{x:.1f}" for x in d))
        print("   gap:", " ".join(f"{x:.1f}" for x in g))
