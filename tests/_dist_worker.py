"""One rank of the CPU multi-process test (launched by tests/test_dist_cpu.py, torchrun-style env)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from voicecraft_amd import dist as vdist  # noqa: E402
from voicecraft_amd import synth  # noqa: E402


def decode_stub(u: int, K: int = 4):
    """Deterministic ragged 'generation' for utterance u."""
    rs = np.random.RandomState(1000 + u)
    return torch.from_numpy(rs.randint(0, 2048, size=(K, 5 + 3 * u)).astype(np.int64))


def decode_oracle(orc, a, u: int):
    x, xl, y = synth.random_prompt(a, 3 + u, 8, seed=1 + u)
    return orc.inference_tts(x, xl, y, top_k=1)[1][0]


def main():
    n_total, use_oracle, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    mine = vdist.shard_utterances(n_total, rank, world)
    if use_oracle:
        from oracle.voicecraft_oracle import VoiceCraftOracle
        a = synth.make_args("tiny")
        orc = VoiceCraftOracle(a, synth.make_state_dict(a, seed=3))
        gens = [decode_oracle(orc, a, u) for u in mine]
    else:
        gens = [decode_stub(u) for u in mine]
    per_rank = vdist.gather_token_blocks(gens, 64, n_slots=-(-n_total // world))
    merged = vdist.merge_in_utterance_order(per_rank)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **{f"u{u}": m.numpy() for u, m in enumerate(merged)})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
