"""The bench line's contract: `cpu_baseline` and `roofline` are JSON OBJECTS — the driver's parser keeps
nothing else —, the headline workload is SURVEY §8(d)'s literal config-2 bundle, and the per-solve kernel time rides next to the
per-launch average."""
import json
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    sys.path.insert(0, ROOT)
    import bench

    return bench


def test_cpu_baseline_is_one_object_with_the_contract_keys():
    """bench.cpu_baseline on a small sample of config 1 (CPU only: the oracle and the host build of the lane code)."""
    import bmo_amd as bmo
    from scenes import c1_scene, disc_bundle

    bench = _bench()
    system, _ = c1_scene()
    bundle = disc_bundle(512, center=[0, -0.05, 0], direction=[0, 1, 0], diameter=0.01, lam=1.064e-6)
    case = types.SimpleNamespace(bmo=bmo, scene=bmo.CompiledScene(system, bundle.lambdas), bundle=bundle)
    cb = bench.cpu_baseline(case, 64, 100)
    assert isinstance(cb, dict)
    for rec in (cb, cb["all_cores"], cb["lane_code_all_cores"]):
        assert isinstance(rec, dict)
        assert rec["value"] > 0 and rec["unit"] == "intersections/s" and rec["kind"] == "port"
        assert isinstance(rec["cores"], int) and rec["cores"] >= 1 and isinstance(rec["sample"], str)
        assert rec["nproc"] >= 1 and rec["cpu"]
    assert cb["cores"] == 1  # the headline baseline: the reference's serial trace loop (System.jl:463-468)
    json.dumps(cb)


def test_default_workload_is_the_survey_bundle():
    bench = _bench()
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'name = args.workload or ("c5" if multi else "c2s")' in src
    assert bench.CONFIG_KEY["c2"] == "c2_cone" and bench.DEFAULT_RAYS["c2s"] == 1 << 20


@pytest.mark.gpu
def test_bench_line_objects_on_the_gpu():
    """One short run of bench.py itself (small bundle, small CPU sample): the printed line parses and carries the objects."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--rays", "16384", "--cpu-sample", "128", "--no-extras"]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
    out = json.loads(line)
    assert out["metric"] == "ray-surface intersections/s" and out["n_gpus"] == 1 and out["steps"] == 2
    assert out["config"]["name"] == "c2s" and "SURVEY 8(d)" in out["config"]["workload"]
    rl, cb = out["roofline"], out["cpu_baseline"]
    assert isinstance(rl, dict) and isinstance(cb, dict)
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-12
    assert rl["kernel_ms_per_solve"] > 0 and rl["kernel_ms_per_solve"] >= rl["avg_launch_ms"]
    assert rl["valu_issue"] is None  # the committed counter pass is of the default size only: no VALU figure for another bundle size
    assert cb["value"] > 0 and cb["cores"] == 1 and cb["kind"] == "port" and "all_cores" in cb and "lane_code_all_cores" in cb
    assert abs(out["vs_baseline"] - out["value"] / cb["value"]) <= 1e-9 * out["vs_baseline"]


def test_valu_issue_figure_reads_the_committed_counter_pass():
    """roofline.valu_issue: vector instructions per solve from the committed SQ pass over the live kernel time, against 256 CUs x 4 SIMDs x 2.4 GHz / 4."""
    bench = _bench()
    tj = json.load(open(os.path.join(ROOT, "profiles", "r04_traffic.json")))["c2s"]
    v = bench.valu_issue("c2s", 1 << 20, 100, 3.0)
    assert v["peak"] == 256 * 4 * 2.4e9 / 4 and v["wave_instructions_per_solve"] == tj["valu_wave_instructions"]
    assert abs(v["achieved"] - tj["valu_wave_instructions"] / 3.0e-3) <= 1e-6 * v["achieved"] and abs(v["frac"] - v["achieved"] / v["peak"]) < 1e-12
    assert 0.2 < v["fp64_arithmetic_share"] < 0.8
    assert bench.valu_issue("c2s", 1 << 16, 100, 3.0) is None and bench.valu_issue("c2s", 1 << 20, 50, 3.0) is None
