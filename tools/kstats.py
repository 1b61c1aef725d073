"""Print the top kernels of a rocprofv3 kernel_stats.csv: python tools/kstats.py <csv> [n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms", tot / 1e6)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:n]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:9.1f} tot%={100 * float(r['TotalDurationNs']) / tot:5.1f}")
