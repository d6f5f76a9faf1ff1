#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into small text files for profiles/.

usage: prof_summary.py --stats <dir with *kernel_stats.csv> --pmc NAME=<dir with *counter_collection.csv> ... -o out.md
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
    name = re.sub(r"^dxk::", "", name)   # the register-chain / fused kernel templates live in namespace dxk
    m = re.match(r"([A-Za-z_0-9:]+(<[^(]*>)?)", name)
    return (m.group(1) if m else name)[:70]


def planes_of(s):
    """number of map planes a launch of this kernel works on, from its template arguments (None: not a per-plane kernel)"""
    m = re.match(r"(k_[a-z_]+)<([^>]*)>", s)
    if not m:
        return None
    args = [x.strip() for x in m.group(2).split(",")]
    try:
        if m.group(1) in ("k_plane_set", "k_index_mh_pair"):
            return int(args[0])
        if m.group(1) in ("k_amp_index", "k_index_mh_reg"):
            return int(args[1])
    except (ValueError, IndexError):
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stats")
    ap.add_argument("--pmc", action="append", default=[])
    ap.add_argument("--only", default="k_", help="keep kernels whose short name starts with this")
    ap.add_argument("--sq", help="dir with the SQ_* counter pass (VALU issue occupancy table)")
    ap.add_argument("--valu-json", help="write {kernel family: VALU issue busy, VALU lane-ops per launch} from the SQ pass")
    ap.add_argument("--traffic-json", help="write {kernel family: HBM bytes per launch} from the FETCH_SIZE/WRITE_SIZE passes")
    ap.add_argument("--fetch-factor", type=float, default=2.0, help="FETCH_SIZE correction (tools/pmc_calib.py: 2.000 here)")
    ap.add_argument("--config", default="C3")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--title", default="rocprofv3 summary")
    a = ap.parse_args()
    lines = ["# " + a.title, ""]
    if a.stats:
        f = glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True)[0]
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
        f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
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
    if a.sq:
        f = glob.glob(os.path.join(a.sq, "**", "*counter_collection.csv"), recursive=True)[0]
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            if not s.startswith(a.only):
                continue
            key = (s, r["Grid_Size"])
            e = agg.setdefault(key, {"n": collections.Counter(), "v": collections.Counter(), "meta": r})
            e["n"][r["Counter_Name"]] += 1
            e["v"][r["Counter_Name"]] += float(r["Counter_Value"])
        lines += ["## SQ counters (%s)" % os.path.relpath(f), "",
                  "resident waves/SIMD from the dispatch's VGPR (arch+accum, 512 per lane) and LDS (160 KiB/CU) footprint; "
                  "VALU issue busy = resident waves/SIMD x the share of a wave's cycles with a VALU instruction issuing.", "",
                  "| kernel | grid | vgpr (arch+acc) | LDS B/block | VALU instr / wave | inst active % | VALU active % | wait_any % | wait_inst % | waves/SIMD | VALU issue busy |",
                  "|---|---|---|---|---|---|---|---|---|---|---|"]
        for (s, g), e in agg.items():
            m = {k: e["v"][k] / e["n"][k] for k in e["v"]}
            r = e["meta"]
            vg = 2 * (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))  # the CSV counts register PAIRS for wave64 dispatches
            lds, wg = int(r["LDS_Block_Size"]), int(r["Workgroup_Size"])
            w_v = min(8, 512 // max(8 * ((vg + 7) // 8), 8))
            w_l = ((160 * 1024) // lds) * (wg // 64) // 4 if lds else 8
            w = max(1, min(w_v, w_l))
            wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
            valu = 100.0 * m.get("SQ_ACTIVE_INST_VALU", 0.0) / wc
            lines.append("| %s | %s | %d | %d | %.0f | %.1f | %.1f | %.1f | %.1f | %d | %.0f %% |" % (
                s, g, vg, lds, m.get("SQ_INSTS_VALU", 0.0) / max(m.get("SQ_WAVES", 1.0), 1.0),
                100.0 * m.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, valu, 100.0 * m.get("SQ_WAIT_ANY", 0.0) / wc,
                100.0 * m.get("SQ_WAIT_INST_ANY", 0.0) / wc, w, min(100.0, w * valu)))
        lines.append("")
    if a.valu_json and a.sq:
        import json
        f = glob.glob(os.path.join(a.sq, "**", "*counter_collection.csv"), recursive=True)[0]
        disp = collections.OrderedDict()          # one entry per dispatch
        for r in csv.DictReader(open(f)):
            s = short(r["Kernel_Name"])
            k = ("k_amp_index" if s.startswith(("k_amp_index", "k_plane_set")) else "k_amp_direct" if s.startswith("k_amp_") else
                     "k_index_mh" if s.startswith("k_index_mh") else None)
            if not k:
                continue
            d = disp.setdefault((k, r["Dispatch_Id"]), {"meta": r})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        # per kernel NAME: mean busy and mean lane-ops per dispatch (SQ pass); calls and mean duration (kernel-trace pass).
        # A family's figures weight its kernels by the time they take in the traced run (a family mixes kernels of
        # different length: single sweeps and two-index sweeps), and its lane-op RATE comes from the trace's durations.
        byname = collections.OrderedDict()
        for (k, _), d in disp.items():
            r = d["meta"]
            vg = 2 * (int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
            w = max(1, min(8, 512 // max(8 * ((vg + 7) // 8), 8)))
            busy = min(1.0, w * d.get("SQ_ACTIVE_INST_VALU", 0.0) / max(d.get("SQ_WAVE_CYCLES", 1.0), 1.0))
            e = byname.setdefault((k, short(r["Kernel_Name"])), {"busy": [], "lane_ops": []})
            e["busy"].append(busy)
            e["lane_ops"].append(64.0 * d.get("SQ_INSTS_VALU", 0.0))
        trace = {}
        if a.stats:
            tf = glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True)[0]
            for r in csv.DictReader(open(tf)):
                trace[short(r["Name"])] = (float(r["Calls"]), float(r["AverageNs"]) * 1e-9)
        fam = collections.OrderedDict()
        for (k, name), e in byname.items():
            calls, dur = trace.get(name, (float(len(e["busy"])), 0.0))
            wt = calls * dur if dur > 0 else float(len(e["busy"]))
            f_ = fam.setdefault(k, {"busy": [], "lane_ops": [], "wt": [], "ops_total": 0.0, "time_total": 0.0, "n": 0})
            f_["busy"].append(sum(e["busy"]) / len(e["busy"])); f_["lane_ops"].append(sum(e["lane_ops"]) / len(e["lane_ops"]))
            f_["wt"].append(wt); f_["n"] += len(e["busy"])
            if dur > 0:
                f_["ops_total"] += calls * sum(e["lane_ops"]) / len(e["lane_ops"]); f_["time_total"] += calls * dur
        out = {"source": "%s (rocprofv3 --pmc SQ_* pass; busy = resident waves/SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES per "
                         "dispatch, lane-ops = 64 x SQ_INSTS_VALU; per family the kernels are weighted by their time in the "
                         "kernel-trace pass, whose durations also give the lane-op rate)" % os.path.join("profiles", os.path.basename(a.out)),
               "config": a.config, "kernels": {}}
        for k, e in fam.items():
            W = sum(e["wt"])
            out["kernels"][k] = {"valu_issue_busy": sum(b * w for b, w in zip(e["busy"], e["wt"])) / W,
                                 "valu_lane_ops_per_launch": sum(o * w for o, w in zip(e["lane_ops"], e["wt"])) / W,
                                 "valu_lane_ops_per_s": (e["ops_total"] / e["time_total"]) if e["time_total"] > 0 else None,
                                 "dispatches": e["n"]}
        # ... and every kernel by the name rocprof prints: the T and Q+U instances of a family are different code objects
        out["instances"] = {}
        for (k, name), e in byname.items():
            calls, dur = trace.get(name, (float(len(e["busy"])), 0.0))
            out["instances"][name] = {"family": k, "planes": planes_of(name), "calls_in_trace": calls, "avg_ms_in_trace": dur * 1e3,
                                      "valu_issue_busy": sum(e["busy"]) / len(e["busy"]),
                                      "valu_lane_ops_per_launch": sum(e["lane_ops"]) / len(e["lane_ops"])}
        json.dump(out, open(a.valu_json, "w"), indent=1)
    if a.traffic_json:
        import json
        fam = collections.OrderedDict()
        for spec in a.pmc:
            cname, d = spec.split("=", 1)
            if cname not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cname:
                    continue
                s = short(r["Kernel_Name"])
                k = ("k_amp_index" if s.startswith(("k_amp_index", "k_plane_set")) else "k_amp_direct" if s.startswith("k_amp_") else
                     "k_index_mh" if s.startswith("k_index_mh") else None)
                if k:
                    fam.setdefault(k, {}).setdefault(cname, {}).setdefault(s, []).append(float(r["Counter_Value"]) * 1024.0)
        # a family's bytes per launch: its kernels' means weighted by their call counts in the kernel-trace pass (the
        # run the bench line describes; the short PMC passes see a different mix of first-iteration kernels)
        calls = {}
        if a.stats:
            tf = glob.glob(os.path.join(a.stats, "**", "*kernel_stats.csv"), recursive=True)[0]
            for r in csv.DictReader(open(tf)):
                calls[short(r["Name"])] = float(r["Calls"])

        def wmean(per_name):
            if not per_name:
                return 0.0
            num = sum(calls.get(n, 1.0) * sum(v) / len(v) for n, v in per_name.items())
            return num / sum(calls.get(n, 1.0) for n in per_name)
        out = {"source": "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH corrected by the measured "
                         "k_cg_vec calibration factor %.3f; kernels of a family weighted by their calls in the kernel-trace pass)"
                         % (os.path.join("profiles", os.path.basename(a.out)), a.fetch_factor),
               "config": a.config, "fetch_correction": a.fetch_factor, "kernels": {}}
        out["instances"] = {}
        for k, v in fam.items():
            fr = wmean(v.get("FETCH_SIZE", {}))
            wr = wmean(v.get("WRITE_SIZE", {}))
            out["kernels"][k] = {"fetch_raw_bytes": fr, "fetch_corrected_bytes": fr * a.fetch_factor, "write_bytes": wr,
                                 "hbm_bytes_per_launch": fr * a.fetch_factor + wr}
            for name in set(v.get("FETCH_SIZE", {})) | set(v.get("WRITE_SIZE", {})):   # per kernel name (never averaged over T / Q+U)
                f1 = v.get("FETCH_SIZE", {}).get(name, [0.0]); w1 = v.get("WRITE_SIZE", {}).get(name, [0.0])
                fb, wb = sum(f1) / len(f1), sum(w1) / len(w1)
                out["instances"][name] = {"family": k, "planes": planes_of(name), "fetch_corrected_bytes": fb * a.fetch_factor,
                                          "write_bytes": wb, "hbm_bytes_per_launch": fb * a.fetch_factor + wb}
        json.dump(out, open(a.traffic_json, "w"), indent=1)
    open(a.out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
