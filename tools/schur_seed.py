#!/usr/bin/env python3
"""One seed of tests/test_gpu_fuzz.py::test_random_template_group_direct_solve with the numbers printed (python tools/schur_seed.py 243)."""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
from dang_amd import _lib as L  # noqa: E402
from util import make_case, pair  # noqa: E402
from test_gpu_fuzz import _packed  # noqa: E402
from test_oracle_templates_cpu import add_globals  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(5000 + seed)
nb = int(rng.choice([4, 5, 6, 8]))
pool = ["cmb", "synch", "dust", "ff"]
comps_l = ["synch"] + list(rng.permutation([p for p in pool if p != "synch"])[: int(rng.integers(0, min(3, nb - 3) + 1))])
pol = bool(rng.integers(0, 2))
which = ("template",) if pol else tuple(rng.permutation(["monopole", "hi_fit"])[: int(rng.integers(1, 3))])
group, flag = (2, L.FLAG_QU) if pol else (1, L.FLAG_T)
fit = sorted(rng.choice(nb, size=int(rng.integers(1, 3)), replace=False).tolist())
ml_mode = str(rng.choice(["sample", "optimize"]))
case = make_case(None, nside=int(rng.choice([2, 4])), nbands=nb, comps=comps_l, nmaps=3,
                 tweak=lambda dpar, ddata, bands, comps: add_globals(dpar, ddata, bands, comps, which, group, fit_bands=fit), start="truth")
dpar, ddata, bands, comps, meta = case
eng, orc = pair(case)
b = orc.compute_rhs(group, flag)
if ml_mode == "sample":
    b = b + orc.compute_sample_vector(group, flag, orc.draw_eta(flag, 8, 9))
it, bad = eng.amp_sample(group, flag, ml_mode, 8, 9, solver="direct")
x = _packed(eng, comps, group, flag, nb)
Ax = orc.compute_Ax(group, flag, x)
R = sum(c.nfit for c in comps if c.cg_group == group and c.type in ("template", "monopole", "hi_fit"))
worst = (np.abs(Ax - b)[-R:] / np.maximum(np.abs(b)[-R:], 1e-300)).max()
print(dict(nb=nb, comps=comps_l, which=which, fit=fit, ml=ml_mode, R=R), "schur_info", eng.schur_info(), "worst global row |Ax-b|/|b| =", worst,
      "global amplitudes", x[-R:])
