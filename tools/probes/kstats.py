"""Print kernel durations from a rocprofv3 --kernel-trace --stats output dir: kstats.py <dir> [substr]"""
import csv, glob, re, sys
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
rows = []
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            nm = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            nm = re.sub(r"\((float|dfhip|WG|SG|Tensor|BnFwd|BnBwd|unsigned|int|long|GDesc|Geo).*$", "", nm)
            rows.append((float(r["TotalDurationNs"]) if "TotalDurationNs" in r else float(r["AverageNs"]) * int(r["Calls"]), nm, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
for tot, nm, calls, avg in sorted(rows, reverse=True):
    print(f"{nm[:70]:70s} {calls:6d} {avg:9.2f} us {tot / 1e3:10.1f}")
