"""N>1 path on CPU: two gloo processes shard utterances, 'decode' them (a deterministic stub, then
the oracle on a tiny model, stand in for the GPU engine) and exchange the results with the single
all_gather of voicecraft_amd.dist.  Padding removal and utterance order are checked on every rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from voicecraft_amd import dist as vdist

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("use_oracle,n_total", [(0, 5), (1, 5), (0, 1), (0, 2)])
def test_two_rank_shard_and_gather(use_oracle, n_total, tmp_path):
    """n_total = 5: ragged lengths and an empty slot on rank 1; n_total = 1: rank 1 has NO utterance at all
    (n_total < world) and still takes part in the collective with an all-empty block."""
    world = 2
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), str(n_total), str(use_oracle),
                                       str(tmp_path)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for p in procs:
        out, _ = p.communicate(timeout=240)
        assert p.returncode == 0, out
    sys.path.insert(0, HERE)
    import _dist_worker as w
    if use_oracle:
        from oracle.voicecraft_oracle import VoiceCraftOracle
        from voicecraft_amd import synth
        a = synth.make_args("tiny")
        orc = VoiceCraftOracle(a, synth.make_state_dict(a, seed=3))
        want = [w.decode_oracle(orc, a, u).numpy() for u in range(n_total)]
    else:
        want = [w.decode_stub(u).numpy() for u in range(n_total)]
    for r in range(world):
        got = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert len(got.files) == n_total
        for u in range(n_total):
            assert np.array_equal(got[f"u{u}"], want[u]), (r, u)


def test_sharding_is_a_partition():
    for world in (1, 2, 4, 8):
        seen = sorted(u for r in range(world) for u in vdist.shard_utterances(64, r, world))
        assert seen == list(range(64))
        assert max(len(vdist.shard_utterances(64, r, world)) for r in range(world)) == 64 // world
