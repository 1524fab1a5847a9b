"""Per kernel name: mean of every counter in a rocprofv3 --pmc counter_collection CSV (any counters).
usage: python tools/pmc_any.py <csv> [<csv> ...] [--filter substring]"""
import csv
import sys
from collections import defaultdict


def main():
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for p in args:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                if flt and flt not in k:
                    continue
                a = agg[k[:110]][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    for k, cs in agg.items():
        print(k)
        for c, (n, tot) in sorted(cs.items()):
            print(f"    {c:36s} calls {n:5d}  mean {tot / n:16.1f}")


if __name__ == "__main__":
    main()
