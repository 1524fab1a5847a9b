"""Grid / workgroup / LDS / register figures of every kernel of the benchmarked step, from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 3 --warmup 2 --no-graph-timing ...
    python3 tools/grid_table.py DIR/**/t_kernel_trace.csv
One line per kernel name: launches, workgroups, threads per workgroup, LDS bytes, VGPRs / AGPRs, mean duration."""
import csv
import sys
from collections import OrderedDict


def main():
    rows = OrderedDict()
    for path in sys.argv[1:]:
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
                grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                key = (name, grid // wg, wg)
                e = rows.setdefault(key, [0, 0.0, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""),
                                          r.get("Accum_VGPR_Count", ""), r.get("SGPR_Count", "")])
                e[0] += 1
                e[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-78s %7s %6s %5s %7s %5s %5s %8s" % ("kernel", "calls", "wgs", "thr", "lds", "vgpr", "agpr", "us"))
    for (name, wgs, wg), e in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%-78s %7d %6d %5d %7s %5s %5s %8.2f" % (name[:78], e[0], wgs, wg, e[2], e[3], e[4], e[1] / e[0]))


if __name__ == "__main__":
    main()
