#!/usr/bin/env python3
"""Join per-kernel counter tables written by tools/gpu_pmc.sh (one pass each) on the kernel name; adds, for every kernel,
valu_per_cycle_simd = SQ_INSTS_VALU / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) when both are present.
usage: python tools/pmc_join.py a.csv b.csv ... > joined.csv"""
import sys


def load(path):
    rows, names = {}, []
    for i, line in enumerate(open(path)):
        line = line.strip()
        if not line:
            continue
        if i == 0:
            names = line.split(",")[1:]
            continue
        parts = line.rsplit(",", len(names))  # kernel names may contain commas
        rows[parts[0]] = dict(zip(names, (float(x) for x in parts[1:])))
    return names, rows


def main():
    cols, table = [], {}
    for path in sys.argv[1:]:
        names, rows = load(path)
        cols += [n for n in names if n not in cols]
        for k, v in rows.items():
            table.setdefault(k, {}).update(v)
    extra = "SQ_INSTS_VALU" in cols and "GRBM_GUI_ACTIVE" in cols
    print("kernel," + ",".join(cols) + (",valu_per_cycle_simd" if extra else ""))
    for k in sorted(table):
        v = table[k]
        line = k + "," + ",".join(f"{v.get(c, 0):.0f}" for c in cols)
        if extra:
            cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0
            line += f",{(v.get('SQ_INSTS_VALU', 0) / cyc if cyc else 0):.4f}"
        print(line)


if __name__ == "__main__":
    main()
