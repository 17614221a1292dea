"""Shared helpers for the parity tests (CPU side)."""
from __future__ import annotations

import os

import numpy as np
import torch

from oracle.gen_golden import MODEL_CASES, case_state_dict  # specs only; importing this module does not touch /root/reference
from oracle.voicecraft_oracle import VoiceCraftOracle
from voicecraft_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, f"model_{name}.npz"))


def build_case(name: str):
    """(spec, args, state_dict, x, x_lens, y) of a golden model case, regenerated from seeds."""
    spec = MODEL_CASES[name]
    args = synth.make_args(spec["preset"], **spec["arg_kw"])
    sd = case_state_dict(spec, args)
    Lx, T, pseed = spec["prompt"]
    x, x_lens, y = synth.random_prompt(args, Lx, T, seed=pseed)
    return spec, args, sd, x, x_lens, y


def run_oracle_case(name: str, trace=None):
    spec, args, sd, x, x_lens, y = build_case(name)
    orc = VoiceCraftOracle(args, sd)
    kn = dict(spec["knobs"])
    if "tseed" in spec:
        torch.manual_seed(spec["tseed"])
    if spec["mode"] == "tts":
        res, gen = orc.inference_tts(x, x_lens, y, trace=trace, **kn)
        return res, gen
    if spec["mode"] == "tts_batch":
        res, gen = orc.inference_tts_batch(x, x_lens, y, trace=trace, **kn)
        return res, gen
    mi = torch.tensor([spec["spans"]], dtype=torch.int64)
    return orc.inference(x, x_lens, y, mi, trace=trace, **kn), None


def build_forward_case(name: str):
    """(spec, args, state_dict, batch dict) of a training-objective golden case (oracle/gen_golden.py FORWARD_CASES)."""
    from oracle.gen_golden import FORWARD_CASES, forward_inputs
    spec = FORWARD_CASES[name]
    args = synth.make_args(spec["preset"], **spec["arg_kw"])
    args.codebook_weight = spec["codebook_weight"]
    sd = case_state_dict(spec, args)
    return spec, args, sd, forward_inputs(spec, args)
