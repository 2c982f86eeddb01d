"""Prints the per-dispatch durations of the last frame from a rocprofv3 --kernel-trace CSV directory."""
import csv, glob, os, sys
root = sys.argv[1]
rows = []
for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
        rows += list(csv.DictReader(f))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last k_accumulate marks the end of a frame; print back to the previous one
idx = [i for i, n in enumerate(names) if "k_accumulate" in n]
lo, hi = (idx[-2] + 1, idx[-1] + 1) if len(idx) > 1 else (0, len(rows))
t0 = int(rows[lo]["Start_Timestamp"])
for r in rows[lo:hi]:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("rtk::", "").split("(")[0]
    print("%9.1f us  +%8.1f  %s  grid=%s vgpr=%s lds=%s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (int(r["Start_Timestamp"]) - t0) / 1e3, n,
          r.get("Grid_Size_X", "?"), r.get("VGPR_Count", "?"), r.get("LDS_Block_Size", "?")))
print("frame span %.1f us" % ((int(rows[hi - 1]["End_Timestamp"]) - t0) / 1e3))
