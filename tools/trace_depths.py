"""Summarises a rocprofv3 kernel_trace.csv: per-dispatch durations of the last frame, in launch order.
usage: python tools/trace_depths.py <kernel_trace.csv> [launches-per-frame]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
t0 = None
for r in rows[-n:]:
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void rtk::", "").replace("rtk::", ""))
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    t0 = t0 or s
    print("%-28s start %9.3f ms  dur %8.3f ms  grid %s" % (name[:28], (s - t0) / 1e6, (e - s) / 1e6, r.get("Grid_Size", r.get("Grid_Size_X", ""))))
