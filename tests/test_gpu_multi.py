"""-m gpu: bench.py's N > 1 path - one process per rank under torch.distributed.run, the barriers, the MAX / SUM reductions
and the ONE all_gather of the token blocks - and the gathered blocks must equal what ONE engine produces for the same
utterances (utterance u -> rank u mod N).  Two forms: two ranks SHARING the one GPU of the test box over
gloo (runs everywhere: the code path the driver's 8-GPU run takes, minus RCCL itself), and two GPUs over RCCL (skipped on
a 1-GPU box)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _two_rank_bench(tmp_path, backend, share, plain_entry=False):
    import json
    common = ["--steps", "2", "--warmup", "1", "--preset", "tiny128", "--lx", "8", "--prompt-frames", "20", "--batch", "2",
              "--no-cpu-baseline", "--no-codec", "--dist-backend", backend]
    dump2 = str(tmp_path / "two.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", VC_RANKS_SHARE_DEVICE="1" if share else "0")
    env.pop("WORLD_SIZE", None)
    launcher = [] if plain_entry else ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                       "--master-port", str(_free_port())]
    # plain_entry: `python bench.py --gpus 2` with no launcher around it - bench.py starts its own two ranks
    r = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump", dump2, *common],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                # rank 0 prints exactly one JSON line
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "weak" and j["config"]["utterances_per_step"] == 4
    col = j["collective"]
    assert col["world"] == 2 and col["backend"] == backend and col["gather_ms"] > 0 and col["ranks_share_one_device"] == share
    # 2 steps x 2 ranks x 2 utterances x K codebooks x (10 Lx - 20) frames: the length cap ends every utterance (muted terminator)
    assert j["value"] > 0 and abs(j["value"] * j["ms_per_step"] * 2 / 1e3 - 2 * 2 * 2 * 4 * 60) < 20.0, j     # (both figures are rounded)
    return dump2


def _check_against_one_engine(dump2):
    two = np.load(dump2)
    assert len(two.files) == 4                                   # 2 ranks x 2 utterances
    # the same four utterances on one engine: global utterance g = u * world + rank has prompt seed 1 + g and
    # is decoded in slot u of its rank with the step's seed
    from voicecraft_amd import synth
    from voicecraft_amd.engine import VoiceCraftEngine
    a = synth.make_args("tiny128")
    sd = synth.make_state_dict(a, seed=0, perturb=False, mute_eos=True, fast=True)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=2, max_positions=1024)
    for rank in range(2):
        prompts = [synth.random_prompt(a, 8, 20, seed=1 + (u * 2 + rank)) for u in range(2)]
        outs = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=40, top_p=1.0, temperature=1.0,
                                       stop_repetition=3, silence_tokens=[1388, 1898, 131], _seed=1001)      # the dump holds the last timed step: seed 1000 + 1
        for u in range(2):
            assert np.array_equal(two[f"u{rank + u * 2}"], outs[u][1][0].cpu().numpy()), (rank, u)


def test_two_ranks_on_one_gpu_run_the_whole_n_gpu_code_path(tmp_path):
    """... through the PLAIN entry `python bench.py --gpus 2`: bench.py launches its two ranks itself."""
    _check_against_one_engine(_two_rank_bench(tmp_path, "gloo", share=True, plain_entry=True))


def test_plain_entry_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus N` on a box with fewer than N devices must fail loudly, not measure one GPU and print n_gpus 1."""
    env = dict(os.environ, VC_RANKS_SHARE_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0", "--preset", "tiny128"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and f"--gpus {n} but only" in r.stdout, r.stdout[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_gpu_bench_gathers_what_one_engine_generates(tmp_path):
    _check_against_one_engine(_two_rank_bench(tmp_path, "nccl", share=False))
