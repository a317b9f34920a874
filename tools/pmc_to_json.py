#!/usr/bin/env python3
"""gpurun_out/pmc_fetch_<tag>.csv + pmc_write_<tag>.csv + prof_<tag>_kernel_stats.csv  ->  profiles/<out>.json

HBM traffic per kernel from rocprofv3 PMC counters, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) prescribes: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF of the bytes of a wide
coalesced read -> x2.  WRITE_SIZE was calibrated on k_raygen (a pure streaming write of a known 85 B/path): factor 1.0.
Optional: the joined SQ counter table of the same run (tools/pmc_join.py) adds, per kernel, the wave-level VALU instructions
retired per cycle and SIMD and the share of enabled lanes (SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU)).
Usage: python tools/pmc_to_json.py <tag> <out.json> [calls-csv [sq-csv]]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    out = {}
    for line in open(path):
        name, _, val = line.strip().rpartition(",")  # kernel names may contain commas (template arguments)
        if name and name != "kernel":
            out[name.replace("void ", "").split("<")[0]] = float(val)
    return out


def main():
    tag, out = sys.argv[1], sys.argv[2]
    fetch = load(f"gpurun_out/pmc_fetch_{tag}.csv")
    write = load(f"gpurun_out/pmc_write_{tag}.csv")
    calls = {}
    if len(sys.argv) > 3:
        for row in csv.DictReader(open(sys.argv[3])):
            calls[row["Name"].replace("void ", "").split("<")[0].split("(")[0]] = (int(row["Calls"]), float(row["TotalDurationNs"]))
    res = {}
    for k in sorted(set(fetch) | set(write)):
        if "rayn" not in k:
            continue
        fb, wb = fetch.get(k, 0.0) * 1024 * 2.0, write.get(k, 0.0) * 1024
        e = {"fetch_bytes": fb, "write_bytes": wb, "hbm_bytes": fb + wb}
        if k in calls:
            n, ns = calls[k]
            e.update({"launches": n, "hbm_bytes_per_launch": (fb + wb) / n, "ms_total_unprofiled": ns / 1e6,
                      "hbm_GBps": (fb + wb) / ns})
        res[k.split("::")[-1]] = e
    if len(sys.argv) > 4 and os.path.exists(sys.argv[4]):
        lines = [ln.strip() for ln in open(sys.argv[4]) if ln.strip()]
        cols = lines[0].split(",")[1:]
        for ln in lines[1:]:
            parts = ln.rsplit(",", len(cols))
            name = parts[0].replace("void ", "").split("<")[0].split("::")[-1]
            v = dict(zip(cols, (float(x) for x in parts[1:])))
            if name in res and v.get("GRBM_GUI_ACTIVE") and v.get("SQ_ACTIVE_INST_VALU"):
                res[name]["valu_inst_per_cycle_simd"] = v["SQ_INSTS_VALU"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)  # 8 XCDs, 1024 SIMDs
                res[name]["lanes_enabled"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
    from bench import kernel_source_hash
    json.dump({"workload": tag, "source_hash": kernel_source_hash(), "corrections": {"unit": "KiB", "FETCH_SIZE": "x2 (gfx950)", "WRITE_SIZE": "x1 (calibrated on k_raygen)"},
               "kernels": res}, open(out, "w"), indent=1)
    for k, e in res.items():
        print(f"{k:24s} fetch {e['fetch_bytes']/1e9:8.2f} GB  write {e['write_bytes']/1e9:8.2f} GB  " + (f"{e['hbm_GBps']:7.0f} GB/s" if "hbm_GBps" in e else ""))


if __name__ == "__main__":
    main()
