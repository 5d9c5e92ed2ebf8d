"""GPU-busy fraction out of a rocprofv3 kernel trace: the union of the kernels' [start, end] intervals over the span from the
first to the last kernel of the steady part (the last 60 % of the trace), the number of launches per millisecond, and the
time inside this library's kernels.  Usage: gpu_busy.py <kernel_trace.csv>"""
import csv
import sys

rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[int(len(rows) * 0.4):]
span = rows[-1][1] - rows[0][0]
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
ours = sum(e - s for s, e, n in rows if "glb::" in n or "anonymous namespace" in n)
gaps = sorted((rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1))
print(f"steady part: {len(rows)} launches over {span / 1e6:.2f} ms; GPU busy {busy / span:.3f}; {len(rows) / (span / 1e6):.0f} launches/ms; "
      f"this library's kernels {ours / span:.3f} of the span; median gap between launches {gaps[len(gaps) // 2] / 1e3:.2f} us")
