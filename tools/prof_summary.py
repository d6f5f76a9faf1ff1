#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small text files for profiles/.

usage: prof_summary.py --stats <dir with *_kernel_stats.csv> --pmc NAME=<dir with *_counter_collection.csv> ... -o out.md
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; per MI355X_MICROARCH.md (HBM section) gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced streams, i.e. HALF the bytes -- the
corrected column doubles it; the calibration factor for this repo's 8-B-per-lane access pattern is
measured with tools/pmc_calib.py on kernels of known byte count.
"""
import argparse
import collections
import csv
import glob
import os
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--only", default="k_", help="keep kernels whose short name starts with this")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--title", default="rocprofv3 summary")
    a = ap.parse_args()
    lines = ["# " + a.title, ""]
    if a.stats:
        f = glob.glob(os.path.join(a.stats, "**", "*_kernel_stats.csv"), recursive=True)[0]
        lines += ["## kernel-trace --stats (%s)" % os.path.relpath(f), "",
                  "| kernel | calls | avg ms | min ms | max ms | total ms | % |", "|---|---|---|---|---|---|---|"]
        for r in csv.DictReader(open(f)):
            s = short(r["Name"])
            if not s.startswith(a.only):
                continue
            lines.append("| %s | %s | %.4f | %.4f | %.4f | %.3f | %s |" % (
                s, r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6,
                float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
        lines.append("")
    for spec in a.pmc:
        cname, d = spec.split("=", 1)
        f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != cname:
                continue
            s = short(r["Kernel_Name"])
            if not s.startswith(a.only):
                continue
            key = (s, r["Grid_Size"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
        lines += ["## --pmc %s (%s)" % (cname, os.path.relpath(f)), "",
                  "| kernel | grid | launches | mean %s (KB) | mean bytes (raw) | x2 (gfx950 FETCH correction) |" % cname,
                  "|---|---|---|---|---|---|"]
        for (s, g), v in agg.items():
            m = sum(v) / len(v)
            lines.append("| %s | %s | %d | %.1f | %.4e | %s |" % (s, g, len(v), m, m * 1024,
                                                                  ("%.4e" % (2 * m * 1024)) if cname == "FETCH_SIZE" else "-"))
        lines.append("")
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
