"""-m gpu: the engine's teacher-forced training objective (vc_eval_forward) against the reference-made
fixtures tests/golden/fwd_*.npz and the oracle.

fp32: summed cross-entropy per call within 2e-4 relative of the reference's loss, top-10 hit counts and the number of
targets exact (a hit flips only on an exact tie with the 10th logit).  bf16: loss within 2e-2 relative, hits within 3 %."""
import ast
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, build_forward_case
from oracle.gen_golden import FORWARD_CASES
from oracle.voicecraft_oracle import VoiceCraftOracle

pytestmark = pytest.mark.gpu


def engine_for(args, sd, dtype, max_seqs=4):
    from voicecraft_amd.engine import VoiceCraftEngine
    return VoiceCraftEngine(args, sd, device="cuda:0", dtype=dtype, max_seqs=max_seqs, max_positions=512)


@pytest.mark.parametrize("name", sorted(FORWARD_CASES))
def test_forward_fp32_matches_the_reference_fixture(name):
    spec, args, sd, batch = build_forward_case(name)
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    eng = engine_for(args, sd, "fp32")
    out = eng.forward({k: v.cuda() for k, v in batch.items()}, spec["spans"])
    assert int(out["effective_ntoken"]) == int(g["effective_ntoken"])
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])), (float(out["loss"]), float(g["loss"]))
    got = np.array([float(t) for t in out["top10acc_by_codebook"]])
    assert np.array_equal(np.rint(got), np.rint(g["top10acc_by_codebook"])), (got, g["top10acc_by_codebook"])


def test_forward_per_target_terms_match_the_oracle():
    """Every single cross-entropy term (not only the sum): the engine's per-row terms, keyed by their target, against the
    oracle's per-target logits."""
    name = "fwd_b3_ragged"
    spec, args, sd, batch = build_forward_case(name)
    orc = VoiceCraftOracle(args, sd)
    want = orc.forward(batch, spec["spans"])
    lg, tg = want["_per_token_logits"], want["_targets"]                     # [K,N,V], [K,N]
    want_nll = torch.stack([torch.nn.functional.cross_entropy(lg[k], tg[k], reduction="none") for k in range(lg.shape[0])])
    eng = engine_for(args, sd, "fp32")
    out = eng.forward({k: v.cuda() for k, v in batch.items()}, spec["spans"], _per_row=True)
    nll, tgt = out["_nll_rows"].cpu(), out["_tgt_rows"].cpu()
    K = nll.shape[1]
    for k in range(K):
        got = np.sort(nll[:, k][tgt[:, k] != -1].numpy())
        ref = np.sort(want_nll[k].numpy())
        assert got.shape == ref.shape
        assert np.abs(got - ref).max() <= 1e-3 * max(1.0, float(np.abs(ref).max())), float(np.abs(got - ref).max())


def test_forward_bf16_and_sample_by_sample():
    name = "fwd_hd128"
    spec, args, sd, batch = build_forward_case(name)
    g = np.load(os.path.join(GOLDEN, f"{name}.npz"))
    eng = engine_for(args, sd, "bf16")
    cu = {k: v.cuda() for k, v in batch.items()}
    out = eng.forward(cu, spec["spans"])
    assert int(out["effective_ntoken"]) == int(g["effective_ntoken"])
    assert abs(float(out["loss"]) - float(g["loss"])) <= 2e-2 * abs(float(g["loss"]))
    hits = float(out["top10acc"])
    assert abs(hits - float(g["top10acc"])) <= 0.03 * float(g["effective_ntoken"]) / args.n_codebooks
    # one utterance at a time gives the same total (no padding inside the engine: rows are per sequence)
    total = 0.0
    for i in range(cu["x"].shape[0]):
        one = {"x": cu["x"][i: i + 1], "x_lens": cu["x_lens"][i: i + 1], "y": cu["y"][i: i + 1], "y_lens": cu["y_lens"][i: i + 1]}
        total += float(eng.forward(one, [spec["spans"][i]])["loss"])
    assert abs(total - float(out["loss"])) <= 1e-3 * abs(float(out["loss"]))


def test_forward_takes_the_spans_from_a_caller_supplied_sampler():
    """`model(batch)` of the reference samples its spans inside (prepare_mask_intervals); the engine takes that sampler as a hook
    (`mask_sampler(y_lens)`), and without spans or sampler it says what it needs."""
    name = "fwd_b3_ragged"
    spec, args, sd, batch = build_forward_case(name)
    eng = engine_for(args, sd, "fp32")
    cu = {k: v.cuda() for k, v in batch.items()}
    want = eng.forward(cu, spec["spans"])
    seen = []

    def sampler(y_lens):
        seen.append([int(v) for v in y_lens])
        return [torch.tensor(iv) for iv in spec["spans"]]          # (the reference returns tensors / nested lists: both are accepted)
    got = eng.forward(cu, mask_sampler=sampler)
    assert seen == [[int(v) for v in batch["y_lens"]]]
    assert float(got["loss"]) == float(want["loss"]) and int(got["effective_ntoken"]) == int(want["effective_ntoken"])
    with pytest.raises(TypeError):
        eng.forward(cu)
