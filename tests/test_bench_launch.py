"""bench.py's own multi-rank launch path on CPU: `python bench.py --gpus 2` must spawn two ranks itself (no torchrun wrapper),
form a world of 2 (gloo here, RCCL on the GPU box), shard the sweep, gather it with ONE all_gather and print one JSON line whose
`n_gpus` is the world size it observed.  The compute runs through the CPU kernel-logic emulator at a toy size
(TRX_BENCH_EMU=1: plumbing only, never a measurement)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*flags, gpus=1):
    env = dict(os.environ, TRX_BENCH_EMU="1", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0", "--order", "1", "--grid", "12",
           "--no-cpu-baseline"] + list(flags)
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_spawns_its_own_ranks_weak_and_strong():
    one = _bench("--batch", "3", gpus=1)
    two = _bench("--batch", "3", gpus=2)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["backend"] == "gloo"
    assert one["scaling"] == "weak" and two["scaling"] == "weak"
    assert one["gathered_points"] == 3 and two["gathered_points"] == 6          # weak: 3 points per rank
    assert np.allclose(one["txx00_sample"], two["txx00_sample"], rtol=1e-6)     # rank 0 solves the same first point
    s = two["strong_scaling"]                                                     # the same 3-point sweep split 2 + 1
    assert s["gathered_points"] == 3 and s["value"] > 0 and s["numerical_failures"] == 0          # (the strong leg's own record; not the JSON line's `value`)
    assert "strong_scaling" not in one
    for r in (one, two):
        assert r["metric"].startswith("RCWA layer-solves/sec") and r["unit"] == "layer-solves/s" and r["higher_is_better"] is True
        # an emulator run must not carry a `value`: the plumbing figures live under their own key
        assert r["value"] is None and r["ms_per_step"] is None and r["roofline"] is None and "EMULATOR" in r["data"]
        assert r["emulator_plumbing"]["units_per_s"] > 0 and r["emulator_plumbing"]["ms_per_step"] > 0


def test_bench_config4_strong_sharded():
    r = _bench("--config", "4", "--points", "5", gpus=2)                          # ragged shards: 3 + 2
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["gathered_points"] == 5
    assert r["config"]["points_total"] == 5 and r["config"]["points_per_gpu"] == 3 and "configs[3]" in r["config"]["workload"]
    r1 = _bench("--config", "4", "--points", "5", gpus=1)
    assert np.allclose(r1["txx00_sample"], r["txx00_sample"], rtol=1e-6)
    assert r1["value"] is None and r1["emulator_plumbing"]["units_per_s"] > 0
    rc = _bench("--config", "4", "--points", "5", "--cyclic", gpus=2)            # cyclic shards: points 0, 2, 4 | 1, 3; same sweep order after the gather
    assert rc["config"]["sharding"].startswith("cyclic") and rc["gathered_points"] == 5 and np.allclose(rc["txx00_sample"], r["txx00_sample"], rtol=1e-6)


def test_bench_config3_stack_and_config5_adjoint():
    """The 4-layer stack (configs[2]: 4 layer-solves per sweep point, ragged chunks) and the forward + adjoint step (configs[4]:
    replicas at N > 1) run through the same launcher and JSON contract."""
    r = _bench("--config", "3", "--batch", "3", "--chunk", "2")
    assert "configs[2]" in r["config"]["workload"] and r["config"]["layer_solves_per_point"] == 4 and r["gathered_points"] == 3
    assert r["scaling"] == "weak" and r["value"] is None
    assert abs(r["emulator_plumbing"]["units_per_s"] * r["emulator_plumbing"]["ms_per_step"] / 1e3 - 12) < 1e-6                  # 3 points x 4 layers per step
    one = _bench("--config", "5")
    two = _bench("--config", "5", gpus=2)
    assert "configs[4]" in one["config"]["workload"] and one["dtype"] == "c128" and one["gathered_points"] == 1 and two["gathered_points"] == 2
    assert one["fom"] > 0 and one["grad_norm"] > 0 and np.isclose(one["fom"], two["fom"], rtol=1e-9)
    assert "forward + adjoint" in one["metric"]
