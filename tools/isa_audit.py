#!/usr/bin/env python3
"""ISA audit of the hot kernels: per kernel the register / spill footprint and, per loop, what the loop body issues.

usage: tools/isa_audit.py [--unit dangx_planeset.hip] [--defs "-DX ..."] [--match k_plane_set] [-o profiles/rNN_isa_audit.md]

Compiles one translation unit of dang_amd/csrc with --save-temps (gfx950), walks the device assembly and reports for every
kernel whose mangled name contains --match:
  * VGPRs / AGPRs / SGPRs, spilled VGPRs and SGPRs, scratch bytes, occupancy (the .amdhsa / remark values);
  * every loop (a backward branch): instructions, VALU instructions, issue cycles by the measured classes of
    profiles/r02_isa_rate.txt (64-bit and three-operand 32-bit VALU ops 4 cycles, simple 32-bit ops 2, v_rcp/v_rsq/v_sqrt_f64 16),
    v_readlane/v_writelane (= SGPR spill traffic), scratch loads/stores, LDS and global/flat memory operations, v_mov_b64 copies.
The proposal loop of a Metropolis chain is the loop that contains the Philox products (v_mad_u64_u32).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TWO_CYCLE = ("v_xor_b32", "v_add_u32", "v_sub_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32",
             "v_add_co_u32", "v_addc_co_u32", "v_subrev_u32", "v_not_b32", "v_add_f32", "v_mul_f32", "v_fma_f32",
             "v_ashrrev_i32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_xad_u32",
             "v_xor3_b32", "v_or3_b32", "v_and_or_b32", "v_lshl_or_b32", "v_add3_u32", "v_lshl_add_u32", "v_add_lshl_u32")
SIXTEEN = ("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")


def cycles_of(op):
    base = op.split("_e32")[0].split("_e64")[0]
    if base in SIXTEEN:
        return 16
    # three-operand 32-bit forms measured at 4 cycles (profiles/r02_isa_rate.txt): v_and_or, v_lshl_add, v_bfe, v_perm, v_mad_*
    if base in ("v_xor_b32", "v_add_u32", "v_sub_u32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32",
                "v_add_co_u32", "v_addc_co_u32", "v_subrev_u32", "v_not_b32", "v_add_f32", "v_mul_f32", "v_fma_f32",
                "v_ashrrev_i32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32"):
        return 2
    return 4


def compile_unit(unit, defs, workdir):
    src = os.path.join(ROOT, "dang_amd", "csrc", unit)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "dang_amd", "lib", "obj"), "--save-temps", "-c", "-o", "unit.o", src] + defs
    r = subprocess.run(cmd, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout)
        raise SystemExit("compile failed")
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return os.path.join(workdir, f)
    raise SystemExit("no device assembly found")


def demangle(names):
    r = subprocess.run(["c++filt"] + names, stdout=subprocess.PIPE, text=True)
    out = r.stdout.strip().split("\n")
    return [re.sub(r"\(.*", "", o).replace("void ", "").replace("dxk::", "") for o in out]


def audit(asm, match):
    lines = open(asm).read().split("\n")
    kernels = []
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", lines[i])
        if m and match in m.group(1):
            name = m.group(1)
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            body = lines[i:j]
            meta = {}
            for l in lines[j:j + 600]:
                for key in ("NumVgprs", "NumAgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize", "TotalNumVgprs"):
                    mm = re.match(r"^;\s*%s:\s*(\d+)" % key, l.strip())
                    if mm and key not in meta:
                        meta[key] = int(mm.group(1))
                mm = re.search(r"\.(sgpr|vgpr)_spill_count:\s*(\d+)", l)
                if mm and (mm.group(1) + "_spill") not in meta:
                    meta[mm.group(1) + "_spill"] = int(mm.group(2))
                if l.startswith("_Z") and l.rstrip().endswith(":"):
                    break
            kernels.append((name, body, meta))
            i = j
        else:
            i += 1
    # spill counts live in the metadata block at the end of the file (per kernel .name / .sgpr_spill_count)
    text = "\n".join(lines)
    for name, body, meta in kernels:
        mm = re.search(r"\.name:\s+%s\n(?:.*\n){0,40}?\s+\.sgpr_spill_count:\s*(\d+)" % re.escape(name), text)
        blk = re.search(r"- \.agpr_count:.*?\.name:\s+%s\b.*?\.wavefront_size" % re.escape(name), text, re.S)
        if blk:
            seg = blk.group(0)
            seg = seg[seg.rfind("- .agpr_count"):]
            for key in ("sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "vgpr_count", "sgpr_count", "agpr_count"):
                m2 = re.search(r"\.%s:\s*(\d+)" % key, seg)
                if m2:
                    meta[key] = int(m2.group(1))
    out = []
    for name, body, meta in kernels:
        labels = {}
        for k, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = k
        loops = []
        for k, l in enumerate(body):
            m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
            if m:
                t = m.group(1) or m.group(2)
                if t in labels and labels[t] < k:
                    ins = [x.strip().split()[0] for x in body[labels[t]:k + 1]
                           if x.startswith("\t") and not x.strip().startswith((".", ";"))]
                    loops.append((labels[t], k, ins))
        # one row per loop header: the widest back edge (a `continue` is a second, shorter back edge to the same header)
        byhead = {}
        for (b, e, ins) in loops:
            if b not in byhead or e > byhead[b][1]:
                byhead[b] = (b, e, ins)
        loops = sorted(byhead.values())
        # whole-kernel totals too
        allins = [x.strip().split()[0] for x in body if x.startswith("\t") and not x.strip().startswith((".", ";"))]
        out.append((name, meta, loops, allins))
    return out


def stats(ins):
    valu = [x for x in ins if x.startswith("v_") and not x.startswith(("v_readlane", "v_writelane", "v_readfirstlane"))]
    return dict(n=len(ins), valu=len(valu), cyc=sum(cycles_of(x) for x in valu),
                lane=sum(1 for x in ins if x.startswith(("v_readlane", "v_writelane"))),
                scratch=sum(1 for x in ins if x.startswith("scratch_")),
                lds=sum(1 for x in ins if x.startswith("ds_")),
                mem=sum(1 for x in ins if x.startswith(("global_", "flat_", "buffer_"))),
                mov64=sum(1 for x in ins if x.startswith("v_mov_b64")),
                philox=sum(1 for x in ins if x.startswith("v_mad_u64_u32")),
                sload=sum(1 for x in ins if x.startswith("s_load")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--unit", default="dangx_planeset.hip")
    ap.add_argument("--defs", default="")
    ap.add_argument("--match", default="k_plane_set")
    ap.add_argument("--asm", help="use an existing device .s instead of compiling")
    ap.add_argument("--min-loop", type=int, default=40, help="hide loops with fewer instructions")
    ap.add_argument("-o", "--out")
    ap.add_argument("--json", help="merge {kernel name as rocprof prints it: spills, proposal-loop mix} into this file (bench.py reads it)")
    ap.add_argument("--title", default=None)
    a = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        asm = a.asm or compile_unit(a.unit, a.defs.split(), tmp)
        res = audit(asm, a.match)
    names = demangle([r[0] for r in res]) if res else []
    L = ["# " + (a.title or "ISA audit: %s %s (kernels matching '%s')" % (a.unit, a.defs, a.match)), "",
         "Issue cycles: 64-bit / three-operand VALU 4, simple 32-bit 2, v_rcp/v_rsq/v_sqrt_f64 16 (profiles/r02_isa_rate.txt). "
         "`lane` = v_readlane + v_writelane (SGPR spill traffic), `scratch` = scratch_load/store (VGPR spill traffic). "
         "A loop with Philox products (`philox` > 0) is a Metropolis proposal loop (lane pairs: two proposals per trip).", ""]
    for (name, meta, loops, allins), dn in zip(res, names):
        L.append("## %s" % dn)
        L.append("")
        L.append("VGPR %s, AGPR %s, SGPR %s, spilled VGPR %s, spilled SGPR %s, scratch %s B" % (
            meta.get("vgpr_count", meta.get("NumVgprs")), meta.get("agpr_count", meta.get("NumAgprs")),
            meta.get("sgpr_count", meta.get("NumSgprs")), meta.get("vgpr_spill_count", "?"), meta.get("sgpr_spill_count", "?"),
            meta.get("private_segment_fixed_size", meta.get("ScratchSize"))))
        s = stats(allins)
        L.append("")
        L.append("whole kernel (static): %d instructions, %d VALU, lane ops %d, scratch ops %d, LDS ops %d, memory ops %d" % (
            s["n"], s["valu"], s["lane"], s["scratch"], s["lds"], s["mem"]))
        L.append("")
        L.append("| loop (asm lines) | instr | VALU | VALU cycles | philox | lane | scratch | LDS | mem | s_load | v_mov_b64 |")
        L.append("|---|---|---|---|---|---|---|---|---|---|---|")
        for (b, e, ins) in loops:
            if len(ins) < a.min_loop:
                continue
            s = stats(ins)
            L.append("| %d-%d | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (
                b, e, s["n"], s["valu"], s["cyc"], s["philox"], s["lane"], s["scratch"], s["lds"], s["mem"], s["sload"], s["mov64"]))
        L.append("")
    if a.json:
        import json
        J = {}
        if os.path.exists(a.json):
            J = json.load(open(a.json))
        for (name, meta, loops, allins), dn in zip(res, names):
            # proposal loops = loops with Philox products that no other such loop contains
            pl = [(b, e, ins) for (b, e, ins) in loops if stats(ins)["philox"] >= 8]
            outer = [x for x in pl if not any(y is not x and y[0] <= x[0] and x[1] <= y[1] for y in pl)]
            st = [stats(ins) for (_, _, ins) in outer]
            nv, nc = sum(x["valu"] for x in st), sum(x["cyc"] for x in st)
            J[dn] = {"unit": a.unit, "vgpr": meta.get("vgpr_count"), "sgpr": meta.get("sgpr_count"), "vgpr_spill": meta.get("vgpr_spill_count"),
                     "sgpr_spill": meta.get("sgpr_spill_count"), "scratch_bytes": meta.get("private_segment_fixed_size"),
                     "proposal_loops": len(outer), "proposal_valu_per_trip": [x["valu"] for x in st],
                     "proposal_cycles_per_instr": (nc / nv) if nv else None,
                     "proposal_lane_ops": sum(x["lane"] for x in st), "proposal_scratch_ops": sum(x["scratch"] for x in st)}
        json.dump(J, open(a.json, "w"), indent=1, sort_keys=True)
    txt = "\n".join(L) + "\n"
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt)
    print(txt)


if __name__ == "__main__":
    main()
