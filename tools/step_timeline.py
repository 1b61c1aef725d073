"""One step of bench.py as a launch-by-launch timeline from a rocprofv3 kernel trace:
     python tools/step_timeline.py <kernel_trace.csv> [step index from the end, default 3] [marker kernel, default act_backward_kernel]
A step = the launches between two consecutive first-of-step kernels on the main stream (the stream with the most busy time).
Prints start offset, duration, gap to the previous launch on the same stream, grid size and kernel name; then per-stream sums."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
marker = sys.argv[3] if len(sys.argv) > 3 else "act_backward_kernel"  # a kernel launched once per step
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
busy = defaultdict(int)
for r in rows:
    busy[r["Queue_Id"]] += r["e"] - r["s"]
main_q = max(busy, key=busy.get)
main = [r for r in rows if r["Queue_Id"] == main_q]
# the step starts with the projection product: the first NT product after the last TN product / reduce of the previous step
marks = [i for i, r in enumerate(main) if marker in r["Kernel_Name"]]  # once per step (default: top of the backward pass)
if len(marks) < back + 2:
    sys.exit("not enough steps in the trace")
a, b = marks[-back - 1], marks[-back]
t0, t1 = main[a]["s"], main[b]["s"]
print(f"step window {1e-3 * (t1 - t0):.1f} us ({marker} to {marker}), main queue {main_q}")
prev_end = {}
tot = defaultdict(float)
for r in rows:
    if not (t0 <= r["s"] < t1):
        continue
    q = r["Queue_Id"]
    gap = (r["s"] - prev_end[q]) * 1e-3 if q in prev_end else 0.0
    prev_end[q] = r["e"]
    dur = (r["e"] - r["s"]) * 1e-3
    tot[q] += dur
    tag = "M" if q == main_q else "s"
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    print(f"{tag} +{1e-3 * (r['s'] - t0):8.1f}  dur {dur:7.1f}  gap {gap:6.1f}  grid {grid:>9s}/{wg:<5s} {r['Kernel_Name'][:90]}")
for q, v in tot.items():
    print(("main" if q == main_q else "side"), q, f"kernel time {v:.1f} us")
