#!/usr/bin/env python3
"""Relate the CPU oracle ("port") to the reference's OWN timings (measured in the survey session with the
flang-built reference on this container's 8 Xeon cores): time the oracle, in reference-algorithm mode (global CG with
per-iteration SED evaluation + per-pixel Metropolis), on the two configurations measured there, with the same thread count.

    python tools/ref_ratio.py            -> profiles/r03_ref_ratio.json

Run in the BUILD container (the host BASELINE.md's numbers were taken on); bench.py copies the two ratios into
cpu_baseline.ref_ratio so that the GPU box's oracle timing can be related to the true reference."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import dang_amd as da  # noqa: E402
from dang_amd import _build, synth  # noqa: E402
from dang_amd import _lib as L  # noqa: E402

# BASELINE.md section 2 (8 threads): seconds per Gibbs iteration of the reference, [low, high]
REFERENCE = {
    "nside64_3b_2c_Q": {"amp_s": [0.56, 0.77], "index_s": [0.81, 0.83]},
    "nside256_5b_2c_QU": {"amp_s": [44.0, 48.0], "index_s": [15.7, 15.7]},
}


def case(nside, nbands, pol):
    """power-law synch + MBB dust, delta bands log-spaced, no mask, beta_s and beta_d sampled (dust T fixed), chisq + gaussian."""
    nmaps = 3
    dpar, ddata, bands, comps, meta = synth.make_sky(None, nside=nside, nbands=nbands, comps=["synch", "dust"], nmaps=nmaps,
                                                     mask_frac=(2.0, 3.0), start="prior")
    keep_flag = L.FLAG_QU if pol == "QU" else L.FLAG_Q
    for c in comps:
        if c.label.endswith("_P"):
            c.pol_flag = [[keep_flag]] * c.nindices
            if c.type == "mbb":
                c.sample_index = [True, False]
        else:
            c.sample_index = [False] * c.nindices
            c.sample_amplitude = False
    for g in dpar.cg_groups:
        g.sample = g.cg_group == 2
        if g.cg_group == 2:
            g.pol_flag = [keep_flag]
    return dpar, ddata, bands, comps, meta


def time_oracle(name, nside, nbands, pol, threads=8):
    import oracle_ffi as O
    dpar, ddata, bands, comps, meta = case(nside, nbands, pol)
    orc = O.Oracle(bands, comps, ddata, nthreads=threads)
    mapn = {1: 1, 2: 2, 4: 3, 8: -1}
    res = {"amp_s": [], "index_s": [], "cg_iters": []}
    for it in (1, 2, 3):   # warm start from the second iteration on, as in the survey's "iterations 2+"
        t0 = time.time()
        for g in dpar.cg_groups:
            if g.sample:
                for f in g.pol_flag:
                    n = orc.amp_sample_cg(g.cg_group, f, "sample", dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f), i_max=100, converge=1e-8)
                    res["cg_iters"].append(n)
        orc.chisq(1, meta["nmaps"], ddata.nump)
        t1 = time.time()
        for l, c in enumerate(comps):
            for j in range(c.nindices):
                if c.sample_index[j]:
                    for f in c.pol_flag[j]:
                        orc.sample_index_mh(l, j, mapn[f], 10, "sample", dpar.seed, da.stream_id(it, 1, l, j, f))
        orc.chisq(1, meta["nmaps"], ddata.nump)
        t2 = time.time()
        if it > 1:
            res["amp_s"].append(t1 - t0); res["index_s"].append(t2 - t1)
    ref = REFERENCE[name]
    mid = lambda v: 0.5 * (v[0] + v[1])
    amp, idx = min(res["amp_s"]), min(res["index_s"])
    out = {"config": name, "threads": threads, "oracle_amp_s": amp, "oracle_index_s": idx, "cg_iters": res["cg_iters"],
           "reference_amp_s": ref["amp_s"], "reference_index_s": ref["index_s"],
           "ratio_amp": amp / mid(ref["amp_s"]), "ratio_index": idx / mid(ref["index_s"]),
           "ratio_iteration": (amp + idx) / (mid(ref["amp_s"]) + mid(ref["index_s"]))}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    _build.build_oracle()
    rows = [time_oracle("nside64_3b_2c_Q", 64, 3, "Q"), time_oracle("nside256_5b_2c_QU", 256, 5, "QU")]
    out = {"what": "oracle (port) seconds per Gibbs iteration / the reference's own seconds (BASELINE.md section 2), same host, 8 threads",
           "host": os.uname().nodename, "cores": len(os.sched_getaffinity(0)), "rows": rows}
    path = os.path.join(ROOT, "profiles", "r03_ref_ratio.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)
