"""Folds a rocprofv3 --kernel-trace --stats CSV (per kernel template) into bench.py's kernel families (extend / shadow / shade /
accumulate), so that the CSV's average launch duration can be set beside bench.py's live `avg_launch_ms` directly.
With the per-launch trace (kernel_trace.csv) and a frame count T it also averages each family over the LAST T frames' launches
only -- the frames bench.py --family-times measured with HIP event pairs -- so that both clocks describe the same launches.
usage: python tools/family_stats.py <kernel_stats.csv> [frames [kernel_trace.csv T]]  -> JSON on stdout"""
import csv
import json
import re
import sys

MODES = {0: "CLOSEST", 1: "SHADOW", 2: "ANYPROBE", 3: "PRIMARY", 4: "PRIMARY_GROUP", 5: "SHADOW_PT"}


def family(name):
    m = re.search(r"k_trace(_packet)?<(\d+),\s*(true|false)>", name)
    if m:
        mode = MODES.get(int(m.group(2)))
        if mode == "ANYPROBE":
            return None
        return "shadow" if mode.startswith("SHADOW") else "extend"
    for k in ("shade", "accumulate", "generate"):
        if "k_" + k in name:
            return k
    return None


def main():
    fam = {}
    for row in csv.DictReader(open(sys.argv[1])):
        f = family(row["Name"])
        if not f:
            continue
        e = fam.setdefault(f, {"launches": 0, "total_ms": 0.0, "kernels": []})
        e["launches"] += int(row["Calls"])
        e["total_ms"] += float(row["TotalDurationNs"]) / 1e6
        e["kernels"].append(re.sub(r"\(.*", "", row["Name"].replace("void rtk::", "").replace("rtk::", "")))
    frames = float(sys.argv[2]) if len(sys.argv) > 2 else None
    for e in fam.values():
        e["avg_launch_ms"] = round(e["total_ms"] / max(1, e["launches"]), 5)
        e["total_ms"] = round(e["total_ms"], 3)
        if frames:
            e["ms_per_frame"] = round(e["total_ms"] / frames, 3)
            e["launches_per_frame"] = round(e["launches"] / frames, 2)
    if len(sys.argv) > 4 and frames:
        last = int(sys.argv[4])
        per = {}
        for row in csv.DictReader(open(sys.argv[3])):
            f = family(row["Kernel_Name"])
            if f:
                per.setdefault(f, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"])))
        for f, spans in per.items():
            spans.sort()
            n = int(round(fam[f]["launches"] / frames)) * last
            tail = spans[-n:]
            fam[f]["last_frames"] = {"frames": last, "launches": len(tail),
                                     "avg_launch_ms": round(sum(e - b for b, e in tail) / max(1, len(tail)) / 1e6, 5)}
    print(json.dumps(fam, indent=1, sort_keys=True))


main()
