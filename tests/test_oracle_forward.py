"""CPU: the oracle's restatement of the training objective (VoiceCraft.forward, models/voicecraft.py:472-559) against the
reference-made fixtures tests/golden/fwd_*.npz.  The reference ran with its random mask intervals replaced
by the fixture's; loss is compared to fp32 rounding, hit counts and token counts exactly."""
import ast
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, build_forward_case
from oracle.gen_golden import FORWARD_CASES
from oracle.voicecraft_oracle import VoiceCraftOracle


def weights_of(spec, K):
    return [float(w) for w in ast.literal_eval(spec["codebook_weight"])] if spec["codebook_weight"] else [1.0] * K


@pytest.mark.parametrize("name", sorted(FORWARD_CASES))
def test_oracle_forward_matches_reference(name):
    spec, args, sd, batch = build_forward_case(name)
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    for k in ("x", "x_lens", "y", "y_lens"):
        assert np.array_equal(g[k], batch[k].numpy()), k
    orc = VoiceCraftOracle(args, sd)
    out = orc.forward(batch, spec["spans"], codebook_weight=weights_of(spec, args.n_codebooks))
    assert int(out["effective_ntoken"]) == int(g["effective_ntoken"])
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-6 * abs(float(g["loss"]))
    got = np.array([float(t) for t in out["top10acc_by_codebook"]])
    assert np.allclose(got, g["top10acc_by_codebook"], atol=1e-3)          # n * mean(hits): integers up to fp32 rounding
    assert abs(float(out["top10acc"]) - float(g["top10acc"])) <= 1e-3


def test_padding_does_not_change_a_sample():
    """The reference pads the batch and masks padded keys; a sample alone must give its own share of the batch loss."""
    spec, args, sd, batch = build_forward_case("fwd_b3_ragged")
    orc = VoiceCraftOracle(args, sd)
    whole = orc.forward(batch, spec["spans"])
    total, ntok = 0.0, 0
    for i in range(batch["x"].shape[0]):
        one = {"x": batch["x"][i: i + 1], "x_lens": batch["x_lens"][i: i + 1], "y": batch["y"][i: i + 1], "y_lens": batch["y_lens"][i: i + 1]}
        o = orc.forward(one, [spec["spans"][i]])
        total += float(o["loss"]); ntok += int(o["effective_ntoken"])
    assert ntok == int(whole["effective_ntoken"])
    assert abs(total - float(whole["loss"])) <= 1e-4 * abs(float(whole["loss"]))


def test_layout_of_a_training_sequence():
    spec, args, sd, _ = build_forward_case("fwd_b1_1span")
    orc = VoiceCraftOracle(args, sd)
    desc = orc.train_layout(48, [(12, 25)], [0])
    assert desc == [("piece", 0, 12, -1), ("mask", 0), ("piece", 25, 48, args.eos), ("mask", 0), ("piece", 12, 25, args.eog)]


def test_interval_sampler_reproduces_the_reference_under_the_same_seed():
    """voicecraft_amd.engine.draw_mask_intervals restates prepare_mask_intervals (models/voicecraft.py:198-237) call for
    call on `random` / torch's generator: the fixture holds what the live reference returned for the same seeds."""
    import random
    from oracle.gen_golden import INTERVAL_CASES, interval_args
    from voicecraft_amd.engine import draw_mask_intervals
    g = np.load(os.path.join(GOLDEN, "mask_intervals.npz"))
    for ci, case in enumerate(INTERVAL_CASES):
        random.seed(case[6]); torch.manual_seed(case[6])
        mi, nmi = draw_mask_intervals(interval_args(case), case[5])
        got_m = np.array([[i, s0, e0] for i, v in enumerate(mi) for (s0, e0) in v], dtype=np.int64)
        got_n = np.array([[i, s0, e0] for i, v in enumerate(nmi) for (s0, e0) in v], dtype=np.int64)
        assert np.array_equal(got_m, g[f"mask_{ci}"]), (ci, got_m, g[f"mask_{ci}"])
        assert np.array_equal(got_n, g[f"nonmask_{ci}"]), ci
