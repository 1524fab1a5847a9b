"""rocprofv3 --stats kernel_stats.csv -> the fixed-width table kept under profiles/ (calls, total ns, average ns, %)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1], newline="")))
print(f"{'kernel':78s} {'calls':>7s} {'total_ns':>12s} {'avg_ns':>10s} {'pct':>6s}")
for r in rows:
    name = r["Name"].replace("(anonymous namespace)::", "")
    if name.startswith("void "):
        name = name[5:]
    name = name.split("(")[0]
    print(f"{name[:78]:78s} {int(r['Calls']):7d} {int(float(r['TotalDurationNs'])):12d} {int(float(r['AverageNs'])):10d} {float(r['Percentage']):6.2f}")
