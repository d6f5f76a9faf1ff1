"""bench.py --gpus N without a launcher starts its own N ranks, and never reports a line for fewer ranks than asked."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return env


def test_more_gpus_than_visible_is_an_error_not_a_one_gpu_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "64", "--steps", "1", "--warmup", "0"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert "--gpus 64" in r.stderr and "visible" in r.stderr
    assert r.stdout.strip() == ""            # no JSON line at all


def test_world_size_must_match_gpus():
    env = dict(_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE (1) != --gpus (8)" in r.stderr
    assert r.stdout.strip() == ""


@pytest.mark.gpu
def test_self_launched_ranks_report_their_count(built):
    """Two self-spawned ranks (gloo process group, both on cuda:0 -- a rehearsal of the N-rank path on a one-GPU box;
    on a node with N GPUs the same launcher runs N RCCL ranks): the line says n_gpus = 2 and the all-reduced rank
    count agrees; chi^2 equals the one-rank run's (the sky is the same, sharded)."""
    def run(*extra):
        r = subprocess.run([sys.executable, BENCH, "--config", "C2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + list(extra),
                           env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])
    two = run("--gpus", "2", "--backend", "gloo")
    one = run("--gpus", "1")
    assert two["n_gpus"] == 2 and two["ranks_seen"] == 2 and two["backend"] == "gloo"
    assert one["n_gpus"] == 1 and one["ranks_seen"] == 1
    for k in ("chisq_after_amp", "chisq_after_index"):
        assert abs(two["config"][k] - one["config"][k]) <= 1e-12 * abs(one["config"][k])


@pytest.mark.gpu
def test_five_rank_rehearsal_of_the_sharded_path_on_one_gpu(built):
    """The N-rank data path with what one GPU allows (no 8-GPU node has been available to any round; a box admits six
    processes on its card, and the test runner may be one of them): bench.py launches 5 gloo ranks on cuda:0 over the C3 model at Nside 256, each on its RING shard (boundaries by
    unmasked pixel count) with the two-stream form the 8-rank runs take (T chain and Q+U chain on separate HIP streams).  All ranks are seen by the
    all-reduce, and chi^2 equals the one-rank value -- the sky does not depend on how it is sharded.  RCCL itself stays
    unexercised."""
    def run(*extra):
        r = subprocess.run([sys.executable, BENCH, "--config", "C3", "--nside", "256", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                            "--no-fortran-seam"] + list(extra), env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])
    five = run("--gpus", "5", "--backend", "gloo", "--streams", "2", "--gather")
    one = run("--gpus", "1", "--gather")
    assert five["n_gpus"] == 5 and five["ranks_seen"] == 5 and five["backend"] == "gloo"
    # north_star's "gather for map output" with the run's (work-balanced, unequal) shard boundaries: the maps reassembled on rank 0
    # are the one-rank maps bit for bit
    assert five["gather"]["bounds"] == "work-balanced" and five["gather"]["npix"] == one["gather"]["npix"] == 12 * 256 * 256
    assert five["gather"]["checksum"] == one["gather"]["checksum"], (five["gather"], one["gather"])
    assert "5 rank(s) (contiguous RING ranges of equal unmasked-pixel count), 2 stream(s) per rank" in five["config"]["workload"]
    for k in ("chisq_after_amp", "chisq_after_index"):
        assert abs(five["config"][k] - one["config"][k]) <= 1e-12 * abs(one["config"][k]), k
    # the old split by pixel count: the same sky
    eq = run("--gpus", "2", "--backend", "gloo", "--equal-shards")
    assert "equal pixel count" in eq["config"]["workload"]
    for k in ("chisq_after_amp", "chisq_after_index"):
        assert abs(eq["config"][k] - one["config"][k]) <= 1e-12 * abs(one["config"][k]), k


@pytest.mark.gpu
def test_bench_line_carries_the_template_model_figure(built):
    """The default line (C3 at full size) with its secondary figure: the same workload with a Q/U template fitted in the Q+U group,
    whole iterations through gibbs_iteration -- the injected amplitudes come back, the Schur solve needed no refinement, and the
    headline keys are what they were."""
    r = subprocess.run([sys.executable, BENCH, "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-fortran-seam"], env=_env(),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["metric"] and d["n_gpus"] == 1 and d["roofline"]["frac"] > 0 and "C3" in d["config"]["workload"]
    t = d["template_model"]
    assert "error" not in t, t
    assert t["it_per_s"] > 0 and t["refinements"] == 0 and t["schur_residual_bound"] <= 1e-12
    for got, want in zip(t["template_amplitudes"], (2.0, -1.5, 0.7)):
        assert abs(got - want) < 0.01, t
    m = d["index_modes"]
    assert "error" not in m and m["fullsky"]["it_per_s"] > 0 and m["coarse"]["it_per_s"] > 0, m
    assert np.isfinite(m["fullsky"]["chisq"]) and np.isfinite(m["coarse"]["chisq"])
