"""Paste a measurement set's per-kernel table into DESIGN.md between the table markers:
    python3 tools/update_design_table.py r04z      (reads profiles/<tag>_kernel_table.md)"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
table = open(os.path.join(root, "profiles", tag + "_kernel_table.md")).read().strip()
path = os.path.join(root, "DESIGN.md")
s = open(path).read()
a, b = s.index("<!-- table:begin -->"), s.index("<!-- table:end -->")
s = s[:a] + "<!-- table:begin -->\n" + table + "\n" + s[b:]
open(path, "w").write(s)
print("pasted", len(table.splitlines()), "lines from", tag)
