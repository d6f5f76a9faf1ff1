#!/usr/bin/env python3
"""bench.py -- Gibbs iterations/sec of the MI355X path on synthetic HEALPix-shaped maps.

One "step" = one full Gibbs iteration of the hot path: `sample_cg_groups` (amplitude solve for
every CG group / poltype flag, then update_sky_model + compute_chisq) followed by
`sample_spectral_parameters` (per-pixel Metropolis sweep of every sampled spectral index, then
update_sky_model + compute_chisq) -- src/dang.f90:101-106 without FITS/ASCII output.

Workload at N=1: BASELINE config "C3" = Nside 1024, 10 bands, 4 components (cmb, synch, dust,
free-free), IQU, fp64.  For N>1 the SAME sky is pixel-sharded over the ranks (strong scaling,
BASELINE config 4); the only collective is the scalar chi^2 all-reduce (RCCL).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")  # oracle leg: no spinning OpenMP workers

import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# VALU issue peak: one vector instruction per SIMD every 4 cycles (16 lanes x 4 passes per wave64 instruction, fp64 FMA
# included), 256 CUs x 4 SIMDs, at the 2.4 GHz peak engine clock = 39.3e12 lane-ops/s; the clock the part holds under a
# pure fp64-FMA load gives 30.9e12 (measured: profiles/r01_ubench_valu.txt)
VALU_PEAK_LANE_OPS = 256 * 4 * 16 * 2.4e9
VALU_SUSTAINED_FMA = 30.9e12
# rocprofv3 --pmc summaries of the SAME command (tools/profile_round.sh): FETCH_SIZE / WRITE_SIZE and the SQ_* pass
PROFILE_TAG = {"C3": ["r04_z", "r03_z", "r02_z"], "C5": ["r04_c5z", "r03_c5z", "r02_c5z"]}   # newest committed profile of the configuration first


def _profile_json(config, kind):
    for tag in PROFILE_TAG.get(config, []):
        path = os.path.join(ROOT, "profiles", "%s_%s.json" % (tag, kind))
        if os.path.exists(path):
            return path
    return None


def measured_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed PMC profile of the same command (FETCH_SIZE and
    WRITE_SIZE collected in separate passes, FETCH corrected by the factor calibrated on kernels of known byte
    count: profiles/r01_z_final.md).  None when no profile of this configuration is committed."""
    try:
        t = json.load(open(_profile_json(config, "traffic")))
        if t.get("config") != config:
            return None
        return float(t["kernels"][kernel]["hbm_bytes_per_launch"])
    except Exception:
        return None


def measured_valu(config, kernel, avg_launch_s):
    """VALU-side roofline of `kernel` from the committed SQ-counter profile of the same command: share of the issue
    cycles of a SIMD in which a vector instruction issues (`busy`), and vector lane-operations per second = the
    profile's lane-ops per launch (64 x the instruction count of the SQ pass: a property of the code, not of the box) / THIS
    run's average launch duration, against the issue peak.  None without a committed profile of the configuration."""
    try:
        t = json.load(open(_profile_json(config, "valu")))
        if t.get("config") != config:
            return None
        k = t["kernels"][kernel]
        rate = float(k["valu_lane_ops_per_launch"]) / avg_launch_s
        return {"busy_in_profile": float(k["valu_issue_busy"]), "lane_ops_per_launch": float(k["valu_lane_ops_per_launch"]),
                "lane_ops_per_s": rate, "peak": VALU_PEAK_LANE_OPS,
                "frac": rate / VALU_PEAK_LANE_OPS, "sustained_fp64_fma": VALU_SUSTAINED_FMA,
                "source": os.path.relpath(_profile_json(config, "valu"), ROOT)}
    except Exception:
        return None


def kernel_instances(config, prof_pl, meta, comps, nmaps):
    """roofline.kernels: one entry per kernel INSTANCE of the plane-set family, by the name rocprof prints -- the T and the Q+U
    launch are different code objects with different costs.  avg_ms: THIS run's HIP events (library buckets by plane count);
    lane-ops / busy / HBM bytes per launch: the committed SQ and FETCH/WRITE passes of the same command; algorithmic bytes:
    SURVEY 8d's per-iteration bytes of the instance's planes.  valu_frac counts every issued vector instruction as one 4-cycle
    slot (issue-slot UTILISATION, not efficiency: the compiler's instructions count as useful; the static mix of the proposal
    loops averages `cycles_per_instr` cycles -- 32-bit ops issue in 2, fp64 reciprocals in 16: profiles/r04_isa_audit.json)."""
    try:
        vj = json.load(open(_profile_json(config, "valu")))
        tj = json.load(open(_profile_json(config, "traffic")))
        if vj.get("config") != config or "instances" not in vj:
            return None
    except Exception:
        return None
    try:
        audit = json.load(open(os.path.join(ROOT, "profiles", "r04_isa_audit.json")))
    except Exception:
        audit = {}
    nphys = len(meta["phys"])
    nb = meta["nbands"]
    nidx = sum(c.nindices for c in comps[:nphys])
    nidx_s = sum(1 for c in comps[:nphys] for j in range(c.nindices) if c.sample_index[j])
    per_unit = 8.0 * ((2 * nb + nidx + 1 + nphys) + (2 * nb + nphys + nidx + 1 + nidx_s))
    out = {}
    # the instance that IS the timed launch of each plane count: the one with the most time in the traced run (the first
    # iteration's solve-only launches belong to the same family)
    best = {}
    for name, v in vj["instances"].items():
        if v.get("family") == "k_amp_index" and v.get("planes"):
            w = v.get("calls_in_trace", 0.0) * v.get("avg_ms_in_trace", 0.0)
            if v["planes"] not in best or w > best[v["planes"]][0]:
                best[v["planes"]] = (w, name)
    for name, v in vj["instances"].items():
        if v.get("family") != "k_amp_index" or not v.get("planes") or best[v["planes"]][1] != name:
            continue
        live = prof_pl.get(("k_amp_index", v["planes"]))
        if not live:
            continue
        t = live["avg_ms"] * 1e-3
        alg = per_unit * meta["npix"] * v["planes"]
        hbm = (tj.get("instances", {}).get(name) or {}).get("hbm_bytes_per_launch")
        a = audit.get(name, {})
        out[name] = {"planes": v["planes"], "avg_ms": round(live["avg_ms"], 4), "launches": live["launches"],
                     "valu_frac": v["valu_lane_ops_per_launch"] / t / VALU_PEAK_LANE_OPS, "busy": v["valu_issue_busy"],
                     "lane_ops_per_launch": v["valu_lane_ops_per_launch"], "cycles_per_instr": a.get("proposal_cycles_per_instr"),
                     "spilled_vgprs": a.get("vgpr_spill"), "spilled_sgprs": a.get("sgpr_spill"),
                     "lane_ops_in_proposal_loops_from_sgpr_spills": a.get("proposal_lane_ops"),
                     "hbm_frac": alg / t / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes": alg, "traffic": hbm,
                     "traffic_ratio": (hbm / alg) if hbm else None}
    return out or None


def algorithmic_bytes(meta, comps, kernel, units):
    """Algorithmic HBM bytes of one launch (DESIGN.md 'Algorithmic bytes'), 8 B per double.

    amplitude solve, per (pixel, plane) unit: read d and sigma for every band (2 nb), the group's
      spectral indices (nidx_g) and the mask (1); write the group's nc amplitudes.
    index sweep, per unit: read d and sigma (2 nb), every amplitude that enters the model on that
      plane (nc_g), the group's indices (nidx_g) and the mask (1); write 1 index value.
    fused solve + first sweep (k_amp_index), per unit: the solve's traffic plus the one index value written.
    """
    nb = meta["nbands"]
    nphys = len(meta["phys"])
    nidx = sum(c.nindices for c in comps[:nphys])
    if kernel == "k_amp_direct":
        per_unit = 2 * nb + nidx + 1 + nphys
    elif kernel == "k_index_mh":
        per_unit = 2 * nb + nphys + nidx + 1 + 1
    elif kernel == "k_amp_index":  # solve + first sweep in one launch: the maps once, amplitudes and one index written
        per_unit = 2 * nb + nidx + 1 + nphys + 1
    else:  # k_sky_chisq: read d, sigma, amplitudes, indices, mask
        per_unit = 2 * nb + nphys + nidx + 1
    return 8.0 * per_unit * units


def log(msg):
    sys.stderr.write("[bench %.1fs] %s\n" % (time.time() - _T0, msg))
    sys.stderr.flush()


_T0 = time.time()


def cpu_baseline(config_name, nsample, nside_sample=128):
    """Time the CPU oracle ("port" of the reference algorithm: global CG with per-iteration SED
    re-evaluation + per-pixel Metropolis) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    import dang_amd as da
    from dang_amd import synth

    # the GPU box gives one GPU's share of the host: at most 16 cores (never oversubscribe OpenMP)
    cores = max(1, min(16, len(os.sched_getaffinity(0))))
    dpar, ddata, bands, comps, meta = synth.make_sky(config_name, nside=nside_sample, nsample=nsample)
    full_npix = 12 * synth.CONFIGS[config_name]["nside"] ** 2
    orc = O.Oracle(bands, comps, ddata, nthreads=cores)
    mapn = {1: 1, 2: 2, 4: 3, 8: -1}
    t0 = time.time()
    cg_iters = []
    for g in dpar.cg_groups:
        for f in g.pol_flag:
            cg_iters.append(orc.amp_sample_cg(g.cg_group, f, "sample", dpar.seed, da.stream_id(1, 0, g.cg_group, 0, f),
                                              i_max=g.i_max, converge=g.converge))
    orc.chisq(1, meta["nmaps"], ddata.nump)
    for l, c in enumerate(comps):
        for j in range(c.nindices):
            if c.sample_index[j]:
                for f in c.pol_flag[j]:
                    orc.sample_index_mh(l, j, mapn[f], dpar.nsample, "sample", dpar.seed, da.stream_id(2, 1, l, j, f))
    orc.chisq(1, meta["nmaps"], ddata.nump)
    dt = time.time() - t0
    scale = full_npix / meta["npix_global"]
    # how the port relates to the reference itself: oracle seconds / the reference's own seconds per Gibbs iteration on the two
    # configurations BASELINE.md section 2 holds reference timings for, both measured on the build container's 8 cores
    # (tools/ref_ratio.py -> profiles/r03_ref_ratio.json).  < 1: the port is FASTER than the reference, i.e. the reference
    # would post a lower it/s than `value` on the same cores.
    try:
        rr = json.load(open(os.path.join(ROOT, "profiles", "r03_ref_ratio.json")))
        ref_ratio = {r["config"]: round(r["ratio_iteration"], 4) for r in rr["rows"]}
        ref_ratio["source"] = "profiles/r03_ref_ratio.json (tools/ref_ratio.py, build container, 8 threads)"
    except Exception:
        ref_ratio = None
    return {"value": 1.0 / (dt * scale), "unit": "it/s", "cores": cores, "kind": "port", "ref_ratio": ref_ratio,
            "sample": "1 Gibbs iteration of config %s at Nside=%d (%d px; CG iterations %s at i_max=100, converge=1e-8) "
                      "took %.2f s on %d OpenMP threads; scaled by pixel count x%d to Nside=%d"
                      % (config_name, nside_sample, meta["npix_global"], cg_iters, dt, cores, int(scale),
                         synth.CONFIGS[config_name]["nside"])}


def fortran_seam(config_name, nsample, steps=10):
    """Gibbs it/s through the REFERENCE-SIDE Fortran wrapper (fortran/reference_side/dang_gpu_mod.f90, run by dang_gpu_drive):
    the two-call seam north_star names (`call sample_cg_groups_gpu` ; `call sample_spectral_parameters_gpu`, statistics after
    every group as the reference prints them) beside `call gibbs_iteration_gpu` (solves fused with the first sweeps).  The
    driver reads an Nside-8 problem of the same model and repeats its maps along the pixel axis up to the configuration's
    pixel count (random streams are keyed by the global pixel); child processes, after this process's own timed region."""
    import tempfile
    from dang_amd import fdrive, synth
    full_npix = 12 * synth.CONFIGS[config_name]["nside"] ** 2
    dpar, ddata, bands, comps, meta = synth.make_sky(config_name, nside=8, nsample=nsample)
    tile = full_npix // meta["npix_global"]
    out = {"what": "it/s through the Fortran wrapper on %s tiled to %d pixels (%d timed iterations each)" % (config_name, full_npix, steps)}
    with tempfile.TemporaryDirectory() as tmp:
        fin = os.path.join(tmp, "in.bin")
        fdrive.write_problem(fin, dpar, ddata, comps, meta, niter=steps + 3)
        for mode in ("twocall", "fused"):
            txt = fdrive.run(fin, os.path.join(tmp, "out_%s.bin" % mode), nctx=1, mode=mode, tile=tile)
            secs = [float(l.split("=")[1].split()[0]) for l in txt.splitlines() if l.startswith("drive seconds per iteration")]
            out["two_call_seam" if mode == "twocall" else "fused_gibbs_iteration_gpu"] = {"it_per_s": 1.0 / secs[0], "ms_per_step": 1e3 * secs[0]}
    return out


def launch_ranks(n, backend):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per GPU, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, as torch.distributed.run would), wait for them, return the worst
    exit code.  Fewer than N visible GPUs is an error (with --backend nccl; gloo rehearsals may share a GPU)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if backend == "nccl" and ndev < n:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) visible; refusing to measure fewer ranks than asked\n"
                         % (n, ndev))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:   # one rank failed: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def template_model(config_name, nsample, steps=6, warmup=2):
    """The same workload with a Q/U template fitted at three bands in the Q+U group (the model class of real dang
    runs): whole Gibbs iterations through da.gibbs_iteration -- T group one plane-set launch, Q+U group pass 1 of the Schur solve +
    one launch that back-substitutes and sweeps.  A secondary figure beside the headline, never `value`."""
    import dang_amd as da
    from dang_amd import synth
    dev = torch.device("cuda", 0)
    dpar, ddata, bands, comps, meta = synth.make_sky(config_name, device=dev, nsample=nsample, as_numpy=False)
    nb = meta["nbands"]
    fit = tuple(range(nb - 3, nb))
    tl = synth.add_qu_template(ddata, comps, meta, fit_bands=fit)
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
    try:
        for it in range(1, warmup + 1):
            da.gibbs_iteration(dpar, ddata, it, want_counts=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(warmup + 1, warmup + 1 + steps):
            da.gibbs_iteration(dpar, ddata, it, want_counts=False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        (resid, _), nref = eng.schur_info()
        ta = eng.get_template_amplitudes(tl)[1, list(fit)]
        return {"workload": "%s + a Q/U template fitted at bands %s of the Q+U group (injected 2.0, -1.5, 0.7)" % (config_name, list(fit)),
                "it_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "steps": steps, "chisq": float(ddata.chisq),
                "template_amplitudes": [float(x) for x in ta], "schur_residual_bound": float(resid), "refinements": int(nref)}
    finally:
        eng.close()


def index_mode_figures(config_name, nsample, steps=4, warmup=2):
    """Two more secondary figures: whole Gibbs iterations of the workload with every sampled index in FULL-SKY mode (index_mode = 1,
    SURVEY 8f rank 2: the chains run on the sufficient statistics of one pass per sweep) and with every index sampled per pixel at
    Nside / 8 (stage, degrade, one chain per coarse pixel).  Never `value`."""
    import dang_amd as da
    from dang_amd import synth
    dev = torch.device("cuda", 0)
    out = {}
    for mode in ("fullsky", "coarse"):
        dpar, ddata, bands, comps, meta = synth.make_sky(config_name, device=dev, nsample=nsample, as_numpy=False)
        nside = int(round((meta["npix_global"] / 12.0) ** 0.5))
        for c in comps:
            if mode == "fullsky":
                c.index_mode = [1] * c.nindices
                c.step_size = [0.05 * g[1] for g in c.gauss_prior]
            else:
                c.sample_nside = [max(nside // 8, 1)] * c.nindices
        eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
        try:
            for it in range(1, warmup + 1):
                da.gibbs_iteration(dpar, ddata, it, want_counts=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for it in range(warmup + 1, warmup + 1 + steps):
                da.gibbs_iteration(dpar, ddata, it, want_counts=False)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            out[mode] = {"workload": "%s, every sampled index %s" % (config_name, "in full-sky mode" if mode == "fullsky" else "per pixel at Nside %d of %d" % (max(nside // 8, 1), nside)),
                         "it_per_s": 1.0 / dt, "ms_per_step": 1e3 * dt, "steps": steps, "chisq": float(ddata.chisq)}
        finally:
            eng.close()
            del ddata.sig_map, ddata.rms_map
            torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--nside", type=int, default=None)
    ap.add_argument("--nsample", type=int, default=10)
    ap.add_argument("--nbands", type=int, default=None, help="diagnostic (not a BASELINE config): the configuration's model on another "
                    "number of bands -- a shape without a built-in kernel instantiation is specialised at run time (hiprtc)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-template-model", action="store_true", help="skip the secondary figure: the workload with a fitted Q/U template")
    ap.add_argument("--no-fortran-seam", action="store_true", help="skip the it/s of the reference-side Fortran wrapper (two-call "
                    "seam and gibbs_iteration_gpu), which bench.py otherwise reports beside the headline value at N=1")
    ap.add_argument("--no-fuse", action="store_true", help="diagnostic: amplitude solve and first index sweep of a plane set as "
                    "two calls (two launches) instead of dangx_amp_index_sample")
    ap.add_argument("--bandpass", type=int, default=0, help="diagnostic (not a BASELINE config): integrate every second "
                    "band over an N-sample +-10%% top-hat bandpass instead of a delta (SURVEY 8f rank 3)")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2], help="2: the temperature chain (group 1, T sweeps) and the "
                    "polarisation chain (group 2, Q+U sweeps) of one Gibbs iteration are independent: run them on two HIP "
                    "streams (two contexts over the same resident maps), which fills the launch tails of small shards "
                    "(+2.5%% at the 8-rank shard size, -2%% at the 2- and 4-rank sizes).  0 = auto: 2 from 8 ranks on")
    ap.add_argument("--shard-of", type=int, default=0, help="diagnostic: with --gpus 1, run ONE rank's shard of an N-rank run "
                    "(rank 0 of N, no process group): the per-rank time at that shard size on one GPU")
    ap.add_argument("--equal-shards", action="store_true", help="shard by pixel count instead of by unmasked pixel count (the "
                    "ranks that hold the masked band then idle: A/B of the balanced boundaries)")
    ap.add_argument("--gather", action="store_true", help="after the timed region: gather the pixel-sharded synchrotron amplitude and "
                    "index maps to rank 0 (dist.gather_maps with the run's shard boundaries -- north_star's 'gather for map output') "
                    "and report their checksum, which does not depend on the number of ranks")
    ap.add_argument("--calibrated", action="store_true", help="diagnostic (not a BASELINE config): band gains /= 1 and offsets /= 0 on the "
                    "T plane (the state after sample_calibrators and a fitted monopole): the plane-set launches must stay")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only for rehearsing "
                                                      "the N>1 path with several ranks on ONE GPU)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N rank processes.  It has made
        # no GPU call (device_count() does not initialise the runtime) and never execs: children are spawned.
        raise SystemExit(launch_ranks(args.gpus, args.backend))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d): refusing to report a line for a different rank count"
                         % (world, args.gpus))
    ndev = torch.cuda.device_count()
    if ndev == 0:
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if local_rank >= ndev and args.backend == "nccl":
        raise SystemExit("LOCAL_RANK %d but only %d GPU(s) visible" % (local_rank, ndev))
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            td.init_process_group(args.backend, rank=rank, world_size=world)

    import dang_amd as da
    from dang_amd import synth

    log("building synthetic sky %s on %s" % (args.config, dev))
    shard_of = args.shard_of if (args.shard_of > 1 and world == 1) else 0
    dpar, ddata, bands, comps, meta = synth.make_sky(args.config, nside=args.nside, nbands=args.nbands, device=dev, rank=rank,
                                                     nranks=shard_of if shard_of else world, nsample=args.nsample, as_numpy=False,
                                                     balance=not args.equal_shards,
                                                     gain=[1.0 + 0.01 * ((j % 3) - 1) for j in range(args.nbands or synth.CONFIGS[args.config]["nbands"])] if args.calibrated else None,
                                                     offset=[0.5 * ((j % 4) - 1.5) for j in range(args.nbands or synth.CONFIGS[args.config]["nbands"])] if args.calibrated else None)
    if args.bandpass > 0:
        import numpy as np
        for b in bands[1::2]:
            b.id = "tophat"
            b.nu0 = b.nu_c * 1e9 * np.linspace(0.9, 1.1, args.bandpass)
            b.tau0 = np.full(args.bandpass, 1.0 / args.bandpass)
    stream = torch.cuda.current_stream().cuda_stream
    eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=dev_index,
                        stream=stream if stream else None)
    nmaps, nb = meta["nmaps"], meta["nbands"]
    chisq_buf = torch.zeros(2, dtype=torch.float64, device=dev)
    mapn = {1: 1, 2: 2, 4: 3, 8: -1}
    # auto: two streams from 8 ranks on (measured per-rank time at the N-rank shard size, one GPU, --shard-of N:
    # N=2 8.09 / 8.29 ms with 1 / 2 streams, N=4 4.57 / 4.64, N=8 2.36 / 2.30)
    two = (args.streams == 2 or (args.streams == 0 and max(world, shard_of) >= 8)) and nmaps == 3
    if two:
        side = torch.cuda.Stream(device=dev)
        engP = da.Engine(bands, comps, ddata, npix_global=meta["npix_global"], pix0=meta["pix0"], device=dev_index,
                         stream=side.cuda_stream)
        chisq_P = torch.zeros(2, dtype=torch.float64, device=dev)
    eng_of = (lambda f: engP if (two and f != 1) else eng)

    # Which sweeps go with which solve: ONE rule for every Python host (da.plan_plane_sets = dangx_plan_fusion behind the ABI plus
    # the foreign-flag guard): the first sampled (component, index) of a (CG group, flag) directly follows that group's solve on
    # the same planes, and EVERY sweep on the group's planes follows it through the same entry point (dangx_plane_set_sample:
    # one launch where the model allows it; for C3 it IS the fused solve + first sweep and the paired dust sweeps)
    plane_sets = {} if args.no_fuse else da.plan_plane_sets(dpar, eng)
    first_sweep = {k: v[0] for k, v in plane_sets.items()}
    fused_sweeps = set((l, j, f) for (grp, f), (l, j) in first_sweep.items())
    in_plane_set = set((l, j, f) for (grp, f), lst in plane_sets.items() for l, j in lst)
    # what one iteration launches on the index side, for the byte accounting: (planes, index values written) per launch
    index_launches = []
    for l, c in enumerate(comps):
        j = 0
        while j < c.nindices:
            if not c.sample_index[j]:
                j += 1
                continue
            pair = (not args.no_fuse and j + 1 < c.nindices and c.sample_index[j + 1] and c.pol_flag[j] == c.pol_flag[j + 1])
            for f in c.pol_flag[j]:
                planes = 2 if f == 8 else 1
                if (l, j, f) in fused_sweeps:
                    if pair:
                        index_launches.append((planes, 1))
                elif pair:
                    index_launches.append((planes, 2))
                else:
                    index_launches.append((planes, 1))
            j += 2 if pair else 1

    def gibbs_iteration(it):
        # sample_cg_groups (src/dang_cg_mod.f90:142-177)
        for g in dpar.cg_groups:
            for f in g.pol_flag:
                if (g.cg_group, f) in plane_sets:
                    eng_of(f).plane_set_sample(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f),
                                               [(l, j, da.stream_id(it, 1, l, j, f)) for l, j in plane_sets[(g.cg_group, f)]],
                                               dpar.nsample, dpar.seed, solver="direct", fluct_mode=dpar.fluct_mode, want_counts=False)
                else:
                    eng_of(f).amp_sample(g.cg_group, f, dpar.ml_mode, dpar.seed, da.stream_id(it, 0, g.cg_group, 0, f),
                                         solver="direct", fluct_mode=dpar.fluct_mode, want_counts=False)
        # the chi^2 after the amplitude phase is captured by the first index sweep on each plane (fused);
        # sample_spectral_parameters (src/dang_sample_mod.f90:21-86)
        # consecutive sampled indices of one component on the same planes (dust beta, dust T) go through
        # dangx_index_sample_pair: one launch and one staging of the maps where the chain covers both, the two sweeps otherwise
        for l, c in enumerate(comps):
            j = 0
            while j < c.nindices:
                if not c.sample_index[j]:
                    j += 1
                    continue
                pair = (not args.no_fuse and j + 1 < c.nindices and c.sample_index[j + 1] and c.pol_flag[j] == c.pol_flag[j + 1])
                for f in c.pol_flag[j]:
                    if (l, j, f) in in_plane_set:   # went with its group's solve
                        continue
                    if pair:
                        eng_of(f).index_sample_pair(l, j, mapn[f], dpar.nsample, dpar.ml_mode, dpar.seed,
                                                    da.stream_id(it, 1, l, j, f), da.stream_id(it, 1, l, j + 1, f), want_counts=False)
                    else:
                        eng_of(f).index_sample(l, j, mapn[f], dpar.nsample, dpar.ml_mode, dpar.seed,
                                               da.stream_id(it, 1, l, j, f), want_counts=False)
                j += 2 if pair else 1
        # update_sky_model + compute_chisq after each phase (src/dang_cg_mod.f90:172-173,
        # src/dang_sample_mod.f90:81-84): both values come out of the sweeps, no extra pass over the maps
        if two:  # T planes from the main context, Q/U planes from the side context; joined on the main stream
            okT = eng.chisq_cached_dev(0, 1, 1, chisq_buf[0:1]) and eng.chisq_cached_dev(1, 1, 1, chisq_buf[1:2])
            okP = engP.chisq_cached_dev(0, 2, 3, chisq_P[0:1]) and engP.chisq_cached_dev(1, 2, 3, chisq_P[1:2])
            if not (okT and okP):
                raise SystemExit("--streams 2 needs a sampled index on every plane")
            torch.cuda.current_stream().wait_stream(side)
            chisq_buf.add_(chisq_P)
            side.wait_stream(torch.cuda.current_stream())
        elif not (eng.chisq_cached_dev(0, 1, nmaps, chisq_buf[0:1]) and eng.chisq_cached_dev(1, 1, nmaps, chisq_buf[1:2])):
            eng.sky_model_chisq_dev(1, nmaps, chisq_buf[1:2])  # a plane without a sampled index: explicit pass
        if world > 1:
            td.all_reduce(chisq_buf)  # global chi^2 (RCCL over xGMI); 16-byte message

    def fence():
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    log("engine ready: npix(local)=%d nbands=%d ncomp=%d" % (meta["npix"], nb, meta["ncomp"]))
    it = 1
    for _ in range(args.warmup):
        gibbs_iteration(it)
        it += 1
    eng.profile(True)
    if two:
        engP.profile(True)
    fence()
    log("warmup done, timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        gibbs_iteration(it)
        it += 1
    fence()
    elapsed = time.perf_counter() - t0
    log("timed region: %.3f s" % elapsed)
    prof = eng.profile_get()
    prof_pl = eng.profile_get(by_planes=True)
    eng.profile(False)
    if two:
        for k, v in engP.profile_get(by_planes=True).items():
            prof_pl.setdefault(k, v)      # Q+U launches live on the side context
        for k, v in engP.profile_get().items():
            if k in prof:
                t = prof[k]
                t["total_ms"] += v["total_ms"]; t["launches"] += v["launches"]; t["avg_ms"] = t["total_ms"] / t["launches"]
            else:
                prof[k] = v
        engP.profile(False)
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        td.all_reduce(ones)   # every rank contributes 1 through the same process group the chi^2 uses
        ranks_seen = int(round(float(ones.item())))
    chisq = (chisq_buf / nb / ddata.nump).tolist()
    gathered = None
    if args.gather:
        # map output (north_star: "a gather for map output"): the shards of two state maps to rank 0, reassembled with the SAME
        # boundaries the sky was split by; the checksum (sum of the doubles' bit patterns mod 2^63) is what a one-rank run gives
        from dang_amd import dist
        lsyn = [l for l, c in enumerate(comps) if c.nindices == 1][0]
        tg = time.perf_counter()
        full = [dist.gather_maps(t, meta["npix_global"], dst=0, bounds=meta["bounds"]) for t in (comps[lsyn].amplitude, comps[lsyn].indices[0])]
        fence()
        if rank == 0:
            ck = 0
            for t in full:
                assert t.shape[-1] == meta["npix_global"]
                bits = t.contiguous().view(torch.int64)
                ck = (ck + int((bits & 0x7FFFFFFF).sum().item()) + 3 * int(((bits >> 31) & 0x7FFFFFFF).sum().item())) % (1 << 62)
            gathered = {"maps": "amplitude and index 0 of component %d (%s)" % (lsyn, comps[lsyn].label), "npix": meta["npix_global"],
                        "bytes": sum(t.numel() * 8 for t in full), "ms": 1e3 * (time.perf_counter() - tg), "checksum": ck,
                        "bounds": "work-balanced" if meta["bounds"] is not None else "equal ranges"}

    if rank == 0:
        # dominant kernel (largest total time on this rank) and its HBM roofline fraction
        dom = max(prof, key=lambda k: prof[k]["total_ms"])
        in_fused = fused_sweeps if "k_amp_index" in prof else set()
        units_per_step = {"k_amp_direct": meta["npix"] * nmaps, "k_amp_index": meta["npix"] * nmaps,
                          "k_index_mh": sum(meta["npix"] * (2 if f == 8 else 1) for l, c in enumerate(comps) for j in range(c.nindices)
                                            if c.sample_index[j] for f in c.pol_flag[j] if (l, j, f) not in in_fused),
                          "k_sky_chisq": 2 * meta["npix"] * nmaps}
        launches_per_step = max(prof[dom]["launches"] // args.steps, 1)
        bytes_per_launch = algorithmic_bytes(meta, comps, dom, units_per_step.get(dom, meta["npix"] * nmaps)) / launches_per_step
        if dom == "k_index_mh" and "k_amp_index" in prof:
            # launches that sweep two consecutive indices of a component stage the maps once and write two index values:
            # per launch 8 (2nb + nc + nidx + 1 + written) bytes per (pixel, plane) unit, averaged over this step's launches
            nphys_, nidx_ = len(meta["phys"]), sum(c.nindices for c in comps[:len(meta["phys"])])
            bytes_per_launch = sum(8.0 * (2 * nb + nphys_ + nidx_ + 1 + w) * meta["npix"] * pl for pl, w in index_launches) / max(len(index_launches), 1)
        if dom == "k_amp_index" and "k_index_mh" not in prof:
            # plane-set launches (dangx_plane_set_sample): a launch IS a plane set's share of the iteration -- the solve and every
            # sweep on those planes; SURVEY 8d's per-iteration bytes of this rank's shard over the launches of a step
            nphys_, nidx_ = len(meta["phys"]), sum(c.nindices for c in comps[:len(meta["phys"])])
            nidx_s_ = sum(1 for c in comps[:nphys_] for j in range(c.nindices) if c.sample_index[j])
            bytes_per_launch = 8.0 * meta["npix"] * nmaps * ((2 * nb + nidx_ + 1 + nphys_) + (2 * nb + nphys_ + nidx_ + 1 + nidx_s_)) / launches_per_step
        achieved = bytes_per_launch / (prof[dom]["avg_ms"] * 1e-3) / 1e9
        standard = (world == 1 and args.nside is None and args.nbands is None and args.nsample == 10 and not args.bandpass
                    and not shard_of and not args.no_fuse and not args.calibrated)   # what the committed profiles are of
        # B_iter = 8 N_sp [(2nb + nidx + 1 + nc) + (2nb + nc + nidx + 1 + nidx_s)] over the WHOLE sky
        nphys = len(meta["phys"])
        nidx = sum(c.nindices for c in comps[:nphys])
        nidx_s = sum(1 for c in comps[:nphys] for j in range(c.nindices) if c.sample_index[j])
        iter_bytes = 8.0 * meta["npix_global"] * nmaps * ((2 * nb + nidx + 1 + nphys) + (2 * nb + nphys + nidx + 1 + nidx_s))
        out = {
            "metric": "gibbs_iterations_per_sec", "value": args.steps / elapsed, "unit": "it/s",
            "n_gpus": world, "ranks_seen": ranks_seen, "backend": (args.backend if world > 1 else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: Nside=%d, %d bands, %d components (%s), %s, NUMSAMPLE=%d, per-pixel indices, "
                                   "direct block solve, reference fluctuation term%s; pixel-sharded over %d rank(s)%s, %d stream(s) per rank%s"
                                   % (args.config, meta["nside"], nb, len(meta["phys"]), "+".join(meta["phys"]),
                                      "IQU" if nmaps == 3 else "I", args.nsample,
                                      (", each group's solve and every sweep on its planes in ONE launch (dangx_plane_set_sample)" if "k_index_mh" not in prof else
                                       ", each group's solve and the first sweep on its planes in one launch, consecutive indices of a component in one launch")
                                      if "k_amp_index" in prof else "",
                                      world, (" (contiguous RING ranges of equal " + ("pixel" if args.equal_shards else "unmasked-pixel") + " count)")
                                      if max(world, shard_of) > 1 else "", 2 if two else 1,
                                      ("; DIAGNOSTIC: every second band integrated over a %d-sample bandpass" % args.bandpass
                                       if args.bandpass else "") +
                                      ("; DIAGNOSTIC: %d bands instead of the configuration's; kernels specialised at run time: %s"
                                       % (args.nbands, ", ".join(eng.rtc_kernels()) or "none") if args.nbands else "") +
                                      ("; DIAGNOSTIC: ONE rank's shard of a %d-rank run" % shard_of if shard_of else "") +
                                      ("; DIAGNOSTIC: band gains /= 1 and offsets /= 0 on the T plane" if args.calibrated else "")),
                       "npix": meta["npix_global"], "chisq_after_amp": chisq[0], "chisq_after_index": chisq[1]},
            # The dominant kernel is bound by fp64 VECTOR ISSUE, not by HBM: achieved / peak / frac are that roofline -- vector
            # lane-operations per second (the instruction count of the committed SQ-counter profile x 64, per launch, over
            # THIS run's launch duration) against one vector instruction per SIMD every 4 cycles.  hbm_* = the HBM side
            # (algorithmic bytes per launch / launch duration; traffic = counter-measured bytes); iter_frac = SURVEY 8d's
            # per-ITERATION bytes against the time of a step.
            "roofline": dict(
                {"bound": "fp64-valu", "kernel": dom, "avg_launch_ms": prof[dom]["avg_ms"]},
                **(lambda v: {"achieved": (v["lane_ops_per_s"] / 1e12 if v else None), "peak": VALU_PEAK_LANE_OPS / 1e12,
                              "unit": "T lane-op/s", "frac": (v["frac"] if v else None), "valu": v})(
                    measured_valu(args.config, dom, prof[dom]["avg_ms"] * 1e-3) if standard else None),
                hbm_achieved=achieved, hbm_peak=HBM_PEAK_GBS, hbm_unit="GB/s", hbm_frac=achieved / HBM_PEAK_GBS,
                bytes_per_launch=bytes_per_launch, traffic=measured_traffic(args.config, dom) if standard else None,
                iter_bytes=iter_bytes, iter_frac=iter_bytes / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS / world,
                note="fp64 VALU issue bound; HBM traffic <= algorithmic bytes, so hbm_frac is what the arithmetic leaves. "
                     "frac is null when no SQ profile of this exact command is committed (non-standard flags)"),
            "kernels": {k: {"avg_ms": round(v["avg_ms"], 4), "launches": v["launches"],
                            "ms_per_step": round(v["total_ms"] / args.steps, 4)} for k, v in prof.items()},
        }
        inst = kernel_instances(args.config, prof_pl, meta, comps, nmaps) if standard else None
        if inst:
            # per instance, and the headline keys from the LONGER one (never an average over the T and Q+U launches)
            longest = max(inst, key=lambda n: inst[n]["avg_ms"])
            r = out["roofline"]
            r["kernels"] = inst
            r["kernel"] = longest
            r["avg_launch_ms"] = inst[longest]["avg_ms"]
            r["achieved"] = inst[longest]["lane_ops_per_launch"] / (inst[longest]["avg_ms"] * 1e-3) / 1e12
            r["frac"] = inst[longest]["valu_frac"]
            r["hbm_frac"] = inst[longest]["hbm_frac"]
            r["hbm_achieved"] = inst[longest]["algorithmic_bytes"] / (inst[longest]["avg_ms"] * 1e-3) / 1e9
            r["bytes_per_launch"] = inst[longest]["algorithmic_bytes"]
            r["traffic"] = inst[longest]["traffic"]
            r["note"] = ("fp64 VALU issue bound; frac = issue-slot UTILISATION of the longer of the two plane-set launches (every issued "
                         "vector instruction = one 4-cycle slot; kernels[*].cycles_per_instr gives the static mix); hbm_* = SURVEY 8d's "
                         "algorithmic bytes of that launch / its duration; traffic = counter-measured HBM bytes per launch")
        if gathered is not None:
            out["gather"] = gathered
        if world == 1 and not args.no_cpu_baseline:
            try:
                log("timing the CPU oracle (cpu_baseline)")
                out["cpu_baseline"] = cpu_baseline(args.config, args.nsample)
                log("cpu_baseline done")
            except Exception as e:  # the oracle is optional infrastructure; never let it break the bench line
                out["cpu_baseline"] = {"value": None, "unit": "it/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        if world == 1 and standard and not args.no_fortran_seam:
            try:
                log("timing the Fortran wrapper (fortran_seam)")
                eng.close()
                if two:
                    engP.close()
                del ddata.sig_map, ddata.rms_map
                torch.cuda.empty_cache()
                out["fortran_seam"] = fortran_seam(args.config, args.nsample)
                log("fortran_seam done")
            except Exception as e:  # flang absent / driver failed: say so in the line, never break it
                out["fortran_seam"] = {"error": repr(e)[:500]}
        if world == 1 and standard and not args.no_template_model:
            try:
                log("timing the template model (template_model)")
                if args.no_fortran_seam:   # (the seam branch above has released the headline's maps already)
                    eng.close()
                    if two:
                        engP.close()
                    del ddata.sig_map, ddata.rms_map
                    torch.cuda.empty_cache()
                out["template_model"] = template_model(args.config, args.nsample)
                log("template_model done")
            except Exception as e:
                out["template_model"] = {"error": repr(e)[:500]}
            try:
                log("timing the full-sky and coarse-Nside index modes (index_modes)")
                out["index_modes"] = index_mode_figures(args.config, args.nsample)
                log("index_modes done")
            except Exception as e:
                out["index_modes"] = {"error": repr(e)[:500]}
        print(json.dumps(out))
    if world > 1:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
