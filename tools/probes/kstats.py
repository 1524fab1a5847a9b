"""Print avg kernel durations from a rocprofv3 --kernel-trace --stats output dir: kstats.py <dir> [substr]"""
import csv, glob, sys
d, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Name"]:
            print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:9.2f} us")
