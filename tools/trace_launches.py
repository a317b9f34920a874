#!/usr/bin/env python3
"""Per-launch durations of the march kernels from a rocprofv3 --kernel-trace CSV (last frame of the run).
usage: trace_launches.py <kernel_trace.csv> <launches_per_frame>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
for kname in ("k_extend1", "k_shadow1", "k_shade_setup"):
    d = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows if kname in r["Kernel_Name"]]
    d.sort()
    last = [round(x[1], 3) for x in d[-n:]]
    print(kname, "launches", len(d), "last frame ms:", last, "sum", round(sum(last), 2))
