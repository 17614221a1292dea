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


def test_option_state_text_round_trips_through_the_bench_helpers():
    """The engine's option state (`key=v,v|...`) as the bench line's JSON object, and the value of one option as vc_set_option
    takes it (what an in-process A/B restores afterwards)."""
    sys.path.insert(0, ROOT)
    import bench
    t = "g=8|nt=63,2|fr=16,1,1,1|ta=2,768|r1=1,1,2|q16=1,1,1,1|sh=1"
    o = bench.options_object(t)
    assert o["text"] == t and o["graph_steps"] == 8 and o["tile_attn"] == {"kernel": 2, "min_rows": 768}
    assert o["one_row"] == {"fr_one": 1, "attn_fast": 1, "qkv_p8": 2} and o["nt"] == {"weights_mask": 63, "attn_kv": 2}
    assert o["many_rows"] == {"qkv16": 1, "wide_heads": 1, "wide_gemm": 1, "wd_stage": 1} and o["shrink"] == 1 and o["finished_rows"] == {"max_rows": 16, "paired": 1, "bf16_partials": 1, "centred_copy": 1}
    assert bench.option_value(t, "tile_attn") == "2,768" and bench.option_value(t, "wide_gemm") == "1" and bench.option_value(t, "shrink") == "1"
    assert bench.option_value(t, "fr_one") == "1" and bench.option_value(t, "nt") == "63,2" and bench.option_value(t, "no_such") is None
    assert len(bench.OPTION_STATE) + 1 == 16          # + prefill_rows, which shapes no captured decode step: the engine's sixteen options (two of them - att_p16, hq - measurement arms of round 6; the K/V hint is the second value of "nt")


def test_in_situ_figure_is_quoted_only_for_the_configuration_it_was_traced_on(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    cfg = {"preset": "giga830M", "dtype": "bf16", "batch": 1, "lx": 80, "prompt_frames": 150, "mode": "tts"}
    (prof / "in_situ.json").write_text(json.dumps({"source": "x", "config": cfg, "lib_stamp": "0123456789abcdef", "box": "b", "kernels": {
        "ffn2": {"name": "k", "calls": 10, "avg_us": 7.0, "algorithmic_bytes": 33579008}}}))
    (prof / "pmc_traffic.json").write_text(json.dumps({"source": "y", "config": cfg, "lib_stamp": "0123456789abcdef", "box": "b", "kernels": {
        "ffn2": {"fetch_bytes_per_launch": 33700000}}}))
    (tmp_path / "voicecraft_amd").mkdir()
    (tmp_path / "voicecraft_amd" / ".build_stamp").write_text("0123456789abcdef" + "f" * 48)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    a = argparse.Namespace(**cfg)
    got = bench.in_situ("ffn2", a)
    assert got["avg_us"] == 7.0 and abs(got["frac"] - 33579008 / 7e-6 / 8e12) < 1e-3 and got["box"] == "b"
    assert bench.pmc_traffic("ffn2", a) == {"bytes": 33700000, "source": "y", "box": "b", "lib_stamp": "0123456789abcdef"}
    a.batch = 8
    assert bench.in_situ("ffn2", a) is None
    # ... and only for the BUILD it was traced on: another digest of the library's sources -> nothing is quoted
    a.batch = 1
    (tmp_path / "voicecraft_amd" / ".build_stamp").write_text("f" * 64)
    assert bench.in_situ("ffn2", a) is None and bench.pmc_traffic("ffn2", a) is None


def test_plain_multi_gpu_entry_builds_the_launcher_command_and_refuses_a_box_without_the_gpus():
    """`python bench.py --gpus N` with no WORLD_SIZE starts its own N ranks (torch.distributed.run, rendezvous on 127.0.0.1, the
    caller's own arguments); here, with no GPU at all, it must exit non-zero with a clear message and print no JSON line."""
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.spawn_command(4, ["--gpus", "4", "--steps", "2"], 29517)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29517"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"]
    env = dict(os.environ, VC_RANKS_SHARE_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    import torch
    n = torch.cuda.device_count() + 2
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and f"--gpus {n} but only" in r.stdout, r.stdout[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
