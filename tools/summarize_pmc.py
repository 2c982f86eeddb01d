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

MODES = {0: "CLOSEST", 1: "SHADOW", 2: "ANYPROBE", 3: "PRIMARY", 4: "PRIMARY_GROUP", 5: "SHADOW_PT"}


def short_name(full):
    s = full.replace("(anonymous namespace)::", "").replace("rtk::", "")
    s = re.sub(r"^void\s+", "", s)
    m = re.match(r"(k_trace|k_trace_packet)<(\d+),\s*(true|false)>", s)
    if m:
        return "%s<%s%s>" % (m.group(1), MODES.get(int(m.group(2)), m.group(2)), ",COUNT" if m.group(3) == "true" else "")
    m = re.match(r"(k_\w+)(<[^>]*>)?", s)
    return (m.group(1) + (m.group(2) or "")) if m else s.split("(")[0]


def bench_key(name):
    if name.startswith("k_trace<SHADOW") or name.startswith("k_trace_packet<SHADOW"):
        return "shadow"
    if name.startswith("k_trace_packet<"):
        return "extend"
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
    fam = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))  # family -> counter -> [sum, dispatches]
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
        if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
            d["mean_waves_resident_per_sq_cycle"] = round(g("SQ_WAVE_CYCLES") * 4.0 / g("SQ_BUSY_CYCLES"), 3)
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and (g("TCC_HIT_sum") + g("TCC_MISS_sum")) > 0:
            d["l2_hit_rate"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 4)
        if g("FETCH_SIZE") is not None:
            d["fetch_bytes_raw"] = g("FETCH_SIZE") * 1024.0
            d["fetch_bytes_x2_gfx950"] = g("FETCH_SIZE") * 2048.0
        if g("WRITE_SIZE") is not None:
            d["write_bytes"] = g("WRITE_SIZE") * 1024.0
        if g("SQ_INSTS_VALU") and g("GRBM_GUI_ACTIVE"):
            # VALU issue utilisation: a wave64 VALU instruction occupies its SIMD for VALU_CYCLES_PER_INST cycles (measured:
            # profiles/r02_valu_rate.txt, 2 with several waves per SIMD; quarter-rate and IEEE divide / sqrt sequences take
            # longer, so this is a lower bound); GRBM_GUI_ACTIVE is summed over the 8 XCDs
            d["valu_issue_utilisation"] = round(g("SQ_INSTS_VALU") * VALU_CYCLES_PER_INST / (NUM_SIMD * g("GRBM_GUI_ACTIVE") / NUM_XCD), 4)
        if g("SQ_ACTIVE_INST_VALU") and g("GRBM_GUI_ACTIVE"):
            # the hardware's own figure: SQ_ACTIVE_INST_VALU counts quad-cycles (4 clocks) a SIMD spends executing VALU
            # instructions.  Only fp32 add / mul / fma take 2 clocks (tools/ubench/valu_asm.hip: 1.13 ns against 1.81 ns for
            # conversions, integer, compare / select and three-operand instructions), so for instruction mixes like the
            # traversal kernels' -- mostly the latter -- this is the fraction of time the vector ALU port is taken
            d["valu_busy"] = round(g("SQ_ACTIVE_INST_VALU") * 4.0 / (NUM_SIMD * g("GRBM_GUI_ACTIVE") / NUM_XCD), 4)
        out[k] = d
        bk = bench_key(k)
        if bk and "COUNT" not in k:
            for c in cs:
                fam[bk][c][0] += cs[c]
                fam[bk][c][1] += n[c]
    families = {}
    for bk, cs in fam.items():
        a = {c: v[0] / v[1] for c, v in cs.items() if v[1]}
        e = {}
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            e["hbm_bytes_per_launch"] = round(2048.0 * a["FETCH_SIZE"] + 1024.0 * a["WRITE_SIZE"], 1)
        if "SQ_INSTS_VALU" in a:
            e["valu_insts_per_launch"] = round(a["SQ_INSTS_VALU"], 1)
        if "SQ_INSTS_SALU" in a:
            e["salu_insts_per_launch"] = round(a["SQ_INSTS_SALU"], 1)
        if a.get("TCC_HIT_sum") is not None and a.get("TCC_MISS_sum") is not None and a["TCC_HIT_sum"] + a["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(a["TCC_HIT_sum"] / (a["TCC_HIT_sum"] + a["TCC_MISS_sum"]), 4)
        if a.get("SQ_THREAD_CYCLES_VALU") and a.get("SQ_ACTIVE_INST_VALU"):
            e["valu_lane_utilisation"] = round(a["SQ_THREAD_CYCLES_VALU"] / (a["SQ_ACTIVE_INST_VALU"] * 64.0), 4)
        if a.get("SQ_INSTS_VALU") and a.get("GRBM_GUI_ACTIVE"):
            e["valu_issue_utilisation_under_pmc"] = round(a["SQ_INSTS_VALU"] * VALU_CYCLES_PER_INST / (NUM_SIMD * a["GRBM_GUI_ACTIVE"] / NUM_XCD), 4)
        if a.get("SQ_ACTIVE_INST_VALU") and a.get("GRBM_GUI_ACTIVE"):
            e["valu_busy_under_pmc"] = round(a["SQ_ACTIVE_INST_VALU"] * 4.0 / (NUM_SIMD * a["GRBM_GUI_ACTIVE"] / NUM_XCD), 4)
        if a.get("SQ_WAVE_CYCLES"):
            for c, name in (("SQ_WAIT_ANY", "wave_time_waiting_on_counters"), ("SQ_WAIT_INST_ANY", "wave_time_waiting_to_issue"),
                            ("SQ_ACTIVE_INST_ANY", "wave_time_issuing")):
                if a.get(c) is not None:
                    e[name] = round(a[c] / a["SQ_WAVE_CYCLES"], 4)
        e["launches_profiled"] = int(max(v[1] for v in cs.values()))
        families[bk] = e
    out["_per_kernel_family"] = families
    print(json.dumps(out, indent=1, sort_keys=True))
    if len(sys.argv) >= 5 and sys.argv[2] == "--traffic":
        workload, path = sys.argv[3], sys.argv[4]
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc = {k: v for k, v in doc.items() if isinstance(v, dict) and all(isinstance(x, dict) for x in v.values())}  # drop the round-1 layout
        doc["_note"] = {"text": {"what": "per kernel family of bench.py (extend / shadow / shade / accumulate), averaged over the launches of rocprofv3 --pmc "
                                 "passes of `bench.py --workload <name>` (tools/profile_pmc.sh): HBM bytes per launch = 2*FETCH_SIZE*1024 + "
                                 "WRITE_SIZE*1024 (gfx950 correction of MI355X_MICROARCH.md), VALU wave-instructions per launch, L2 hit rate, lane "
                                 "utilisation; bench.py divides by ITS OWN live launch durations"}}
        families = dict(families)
        families["_stamp"] = stamp()
        doc[workload] = families
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)


def stamp():
    """what the counters belong to: the fingerprint of the kernel sources (bench.py compares it with the sources it runs on)
    and, where a git checkout is at hand, the commit"""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    st = {"kernel_sources_sha16": bench.kernel_sources_sha16(), "collected": time.strftime("%Y-%m-%d %H:%M:%S")}
    commit = os.environ.get("RT_COMMIT")
    if not commit:
        try:
            commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip() or None
        except Exception:
            commit = None
    st["commit"] = commit
    return st


NUM_SIMD, NUM_XCD = 256 * 4, 8
VALU_CYCLES_PER_INST = 2.0
main()
