"""The nccl (= RCCL) branches of dang_amd/dist.py on the one GPU a test box has: a process group of ONE rank with
DANGX_DIST_SINGLE=1, so that every sky-wide sum is really handed to RCCL (device tensors, the dangx_set_allreduce callback of the
Schur solve and of the device CG, the chi^2 all-reduce, broadcast, gather).  The numbers must be those of a run without a process
group.  What this does NOT show is xGMI traffic between ranks: no multi-GPU node was available to rounds 1-3."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np, torch
import torch.distributed as td
import dang_amd as da
from dang_amd import dist, _lib as L
from test_oracle_templates_cpu import add_globals
from util import make_case

use_pg = sys.argv[1] == "nccl"
if use_pg:
    torch.cuda.set_device(0)
    td.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.active()

out = {}
def tweak(dpar, ddata, bands, comps):
    add_globals(dpar, ddata, bands, comps, ("template",), 2, skip_band0=True)
dpar, ddata, bands, comps, meta = make_case("C2", nside=4, start="truth", tweak=tweak)
eng = da.initialize(bands, comps, ddata, npix_global=meta["npix_global"], device=0)
assert (eng._allreduce_cb is not None) == use_pg
for it in (1, 2):
    da.sample_cg_groups(dpar, ddata, it=it)                      # the template group's Schur rows pass through the callback
    da.sample_spectral_parameters(dpar, ddata, it=it)            # chi^2: allreduce_sum_float
out["chisq"] = ddata.chisq
out["amp"] = [float(np.abs(eng.get_amplitude(l)).sum()) for l in range(len(comps))]
out["tamp"] = [eng.get_template_amplitudes(l).tolist() for l, c in enumerate(comps) if c.type == "template"]
iters, _ = eng.amp_sample(1, L.FLAG_T, "sample", dpar.seed, da.stream_id(3, 0, 1, 0, 1), solver="cg", i_max=100, converge=1e-8)
out["cg_iters"] = int(iters)                                      # the device CG's dot products pass through the callback
out["amp_cg"] = float(np.abs(eng.get_amplitude(0)).sum())
out["bcast"] = dist.bcast_from_rank0([1.5, -2.25])
t = torch.arange(6, dtype=torch.float64, device="cuda").reshape(2, 3)
out["gather"] = dist.gather_maps(t, 3).cpu().tolist()
out["sum_"] = dist.allreduce_sum_(torch.full((4,), 0.125, dtype=torch.float64, device="cuda")).cpu().tolist()
if use_pg:
    td.barrier(); td.destroy_process_group()
print("RESULT " + json.dumps(out))
"""


def _child(mode):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = CHILD % {"root": root, "tests": os.path.join(root, "tests")}
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29571", DANGX_DIST_SINGLE="1",
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, "-c", code, mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert r.returncode == 0 and lines, r.stdout[-3000:]
    import json
    return json.loads(lines[-1][7:])


def test_single_rank_rccl_group_gives_the_numbers_of_a_run_without_one(built):
    a, b = _child("nccl"), _child("none")
    assert a == b, (a, b)          # a sum over one rank is the value itself: bit for bit
    assert a["cg_iters"] > 1 and a["gather"] == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
