"""bench.py without a GPU: the CPU leg alone (`--cpu-baseline-only`: the oracle = port, and the unmodified reference when
VC_REFERENCE_ROOT names its tree - build container only) prints one JSON object with the fields the GPU line's `cpu_baseline`
carries, and the in-process A/B's statistics behave on hand-made timings."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env):
    env = dict(os.environ, **extra_env)
    env.pop("VC_REFERENCE_ROOT", None) if "VC_REFERENCE_ROOT" not in extra_env else None
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--preset", "tiny", "--lx", "6",
                        "--prompt-frames", "21", "--top-k", "1", "--cpu-steps", "12", "--cpu-threads", "2"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_cpu_leg_runs_alone_and_reports_the_port():
    j = _run({})
    cb = j["cpu_baseline"]
    assert j["config"] == "tiny/lx6/t21/k1"
    assert cb["kind"] == "port" and cb["unit"] == "codec-tokens/s" and cb["value"] > 0 and cb["cores"] == 2
    assert len(cb["ms_per_step"]) == 3 and "reference" not in cb           # no reference tree named: the port alone


def test_cpu_leg_times_the_unmodified_reference_when_its_tree_is_named(tmp_path):
    from oracle import ref_loader
    if not ref_loader.available():
        import pytest
        pytest.skip("reference tree not present (GPU box)")
    before = open(os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json")).read()
    try:
        j = _run({"VC_REFERENCE_ROOT": ref_loader.REFERENCE_ROOT})
        ref = j["cpu_baseline"]["reference"]
        assert ref["kind"] == "reference" and ref["value"] > 0 and ref["port_speed_over_reference"] > 0
    finally:       # the run merges its ratio into the committed file: this tiny configuration does not belong there
        open(os.path.join(ROOT, "profiles", "cpu_reference_vs_port.json"), "w").write(before)
