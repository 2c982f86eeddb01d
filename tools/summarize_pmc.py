"""Summarises rocprofv3 --pmc CSV output (one directory per pass) per kernel: sums every counter over the dispatches of a
kernel and reports per-dispatch averages plus a few derived figures.  gfx950 corrections of MI355X_MICROARCH.md ("HBM"):
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE tallies 128-B requests at 64 B, so read bytes = FETCH_SIZE * 1024 * 2
(stated per kernel both raw and corrected); WRITE_SIZE reads the bytes exactly.

usage: summarize_pmc.py <pass-root> [--traffic <workload> <traffic.json>]
The optional --traffic step merges per-launch HBM bytes (2*FETCH + WRITE), keyed like bench.py's kernel_ms_per_step
(extend / shadow / shade / accumulate / compact / generate), into the JSON file bench.py reads for roofline.traffic."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

MODES = {0: "CLOSEST", 1: "SHADOW", 2: "ANYPROBE", 3: "PRIMARY", 4: "PRIMARY_GROUP"}


def short_name(full):
    s = full.replace("(anonymous namespace)::", "").replace("rtk::", "")
    s = re.sub(r"^void\s+", "", s)
    m = re.match(r"k_trace<(\d+),\s*(true|false)>", s)
    if m:
        return "k_trace<%s%s>" % (MODES.get(int(m.group(1)), m.group(1)), ",COUNT" if m.group(2) == "true" else "")
    m = re.match(r"(k_\w+)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or "")) if m else s.split("(")[0]


def bench_key(name):
    if name.startswith("k_trace<SHADOW"):
        return "shadow"
    if name.startswith("k_trace<ANYPROBE"):
        return None
    if name.startswith("k_trace<"):
        return "extend"
    for k in ("shade", "accumulate", "compact", "generate"):
        if name.startswith("k_" + k):
            return k
    return None


def main():
    root = sys.argv[1]
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(lambda: defaultdict(set))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = short_name(row["Kernel_Name"])
                c = row["Counter_Name"]
                per[k][c] += float(row["Counter_Value"])
                calls[k][c].add((path, row["Dispatch_Id"]))
    out = {}
    traffic_sum = defaultdict(lambda: [0.0, 0.0, 0, 0])  # key -> [fetch KiB, write KiB, fetch dispatches, write dispatches]
    valu_sum = defaultdict(lambda: [0.0, 0, 0.0, 0])     # key -> [SQ_INSTS_VALU, dispatches, GRBM_GUI_ACTIVE, dispatches]
    for k, cs in per.items():
        if not k.startswith("k_"):
            continue  # torch / runtime kernels
        n = {c: max(1, len(calls[k][c])) for c in cs}
        avg = {c: cs[c] / n[c] for c in cs}
        d = {"dispatches": max(n.values()), "per_dispatch": {c: round(v, 1) for c, v in sorted(avg.items())}}
        g = avg.get
        if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
            # lanes active per VALU instruction (of 64): thread-cycles / (instruction-cycles * 64)
            d["valu_lane_utilisation"] = round(g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0), 4)
        if g("SQ_ACTIVE_INST_VALU") and g("SQ_BUSY_CYCLES"):
            d["valu_busy_share"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / g("SQ_BUSY_CYCLES"), 4)  # quad-cycles -> cycles, summed over SQs
        if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
            d["mean_waves_resident_per_sq_cycle"] = round(g("SQ_WAVE_CYCLES") * 4.0 / g("SQ_BUSY_CYCLES"), 3)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
            d["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("FETCH_SIZE") is not None:
            d["fetch_bytes_raw"] = g("FETCH_SIZE") * 1024.0
            d["fetch_bytes_x2_gfx950"] = g("FETCH_SIZE") * 2048.0
        if g("WRITE_SIZE") is not None:
            d["write_bytes"] = g("WRITE_SIZE") * 1024.0
        out[k] = d
        bk = bench_key(k)
        if bk and "COUNT" not in k:
            t = traffic_sum[bk]
            if "FETCH_SIZE" in cs:
                t[0] += cs["FETCH_SIZE"]
                t[2] += n["FETCH_SIZE"]
            if "WRITE_SIZE" in cs:
                t[1] += cs["WRITE_SIZE"]
                t[3] += n["WRITE_SIZE"]
            v = valu_sum[bk]
            if "SQ_INSTS_VALU" in cs:
                v[0] += cs["SQ_INSTS_VALU"]
                v[1] += n["SQ_INSTS_VALU"]
            if "GRBM_GUI_ACTIVE" in cs:
                v[2] += cs["GRBM_GUI_ACTIVE"]
                v[3] += n["GRBM_GUI_ACTIVE"]
    traffic = {k: round(2048.0 * t[0] / t[2] + 1024.0 * t[1] / t[3], 1) for k, t in traffic_sum.items() if t[2] and t[3]}
    out["_hbm_bytes_per_launch"] = traffic
    # VALU issue utilisation: wave-instructions x 4 cycles (a wave64 op occupies its 16-lane SIMD for 4 cycles; quarter-rate
    # ops for longer, so this is a lower bound) over the SIMD-cycles of the launch.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.
    NUM_SIMD, NUM_XCD = 256 * 4, 8
    valu = {k: round((v[0] / v[1]) * 4.0 / (NUM_SIMD * (v[2] / v[3]) / NUM_XCD), 4) for k, v in valu_sum.items() if v[1] and v[3] and v[2] > 0}
    out["_valu_issue_utilisation"] = valu
    print(json.dumps(out, indent=1, sort_keys=True))
    if len(sys.argv) >= 5 and sys.argv[2] == "--traffic":
        workload, path = sys.argv[3], sys.argv[4]
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc["_note"] = ("HBM bytes per launch from rocprofv3 PMC passes (tools/profile_pmc.sh): 2*FETCH_SIZE*1024 + WRITE_SIZE*1024, "
                        "averaged over the launches of each kernel family in bench.py runs of the named workload")
        doc[workload] = traffic
        doc.setdefault("_valu_issue_utilisation", {})[workload] = valu
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)


main()
