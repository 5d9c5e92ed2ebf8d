"""The largest idle gaps of a rocprofv3 kernel trace and the kernels on either side: gaps.py <kernel_trace.csv> [top]
(steady part = the last 60 % of the launches).  Where a step's GPU time goes missing."""
import csv
import sys
from collections import Counter

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:70]))
rows.sort()
rows = rows[int(len(rows) * 0.4):]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
gaps = []
end = rows[0][1]
last = rows[0][2]
for s, e, n in rows[1:]:
    if s > end:
        gaps.append((s - end, last, n))
    if e > end:
        end, last = e, n
tot = sum(g[0] for g in gaps)
span = rows[-1][1] - rows[0][0]
print(f"idle {tot / 1e6:.2f} ms of {span / 1e6:.2f} ms ({tot / span:.3f}); {len(gaps)} gaps")
by = Counter()
cnt = Counter()
for g, a, b in gaps:
    by[(a, b)] += g
    cnt[(a, b)] += 1
for (a, b), g in by.most_common(top):
    print(f"{g / 1e3:10.1f} us in {cnt[(a, b)]:5d} gaps (mean {g / cnt[(a, b)] / 1e3:7.1f} us)  after [{a}]  before [{b}]")
