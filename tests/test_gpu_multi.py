"""-m gpu, needs >= 2 GPUs (skipped on the 1-GPU box): bench.py's N > 1 path for real - one process per GPU under
torch.distributed.run, RCCL all_gather of the token blocks - and the gathered blocks must equal what ONE engine
produces for the same utterances (utterance u -> rank u mod N)."""
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


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs")
def test_two_gpu_bench_gathers_what_one_engine_generates(tmp_path):
    common = ["--steps", "1", "--warmup", "0", "--preset", "tiny128", "--lx", "8", "--prompt-frames", "20", "--batch", "2",
              "--no-cpu-baseline", "--no-codec"]
    dump2, dump1 = str(tmp_path / "two.npz"), str(tmp_path / "one.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dump", dump2, *common],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    assert '"n_gpus": 2' in line
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
                                       stop_repetition=3, silence_tokens=[1388, 1898, 131], _seed=1000)
        for u in range(2):
            assert np.array_equal(two[f"u{rank + u * 2}"], outs[u][1][0].cpu().numpy()), (rank, u)
