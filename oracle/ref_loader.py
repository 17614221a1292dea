"""Imports the REAL reference (jasonppy/VoiceCraft) from /root/reference.  TEST INFRASTRUCTURE ONLY.

Only usable in the build container: /root/reference does not exist on the GPU box, so nothing in
the `-m gpu` tests, smoke() or bench.py may import this module.  It is used by gen_golden.py to
produce tests/golden/*.npz and by the container-only cross-check test.

`torchmetrics` is absent from the image and only feeds a training metric
(models/voicecraft.py:10, :187-195), so a stand-in that restates the one metric the reference uses is put on
sys.modules first.
"""
from __future__ import annotations

import os
import sys
import types

import torch

REFERENCE_ROOT = os.environ.get("VC_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "models", "voicecraft.py"))


def _stub_torchmetrics() -> None:
    if "torchmetrics" in sys.modules:
        return
    tm = types.ModuleType("torchmetrics")
    cl = types.ModuleType("torchmetrics.classification")

    class MulticlassAccuracy(torch.nn.Module):
        """Stand-in for torchmetrics 0.11.1 (README.md:113) `MulticlassAccuracy(num_classes, top_k, average="micro",
        multidim_average="global", ignore_index=None)` as the reference constructs it (models/voicecraft.py:187-195):
        the fraction of samples whose target is among the top_k logits (`select_topk` = `Tensor.topk`)."""

        def __init__(self, num_classes=None, top_k=1, average="micro", multidim_average="global", ignore_index=None, **k):
            super().__init__()
            assert average == "micro" and multidim_average == "global" and ignore_index is None
            self.top_k = int(top_k)

        def forward(self, preds, target):
            hit = (preds.topk(self.top_k, dim=1).indices == target[:, None]).any(dim=1)
            return hit.float().mean()

    cl.MulticlassAccuracy = MulticlassAccuracy
    tm.classification = cl
    sys.modules["torchmetrics"] = tm
    sys.modules["torchmetrics.classification"] = cl


def import_reference():
    """Returns the reference's `models.voicecraft` and `models.codebooks_patterns` modules."""
    if not available():
        raise RuntimeError(f"reference tree not found at {REFERENCE_ROOT}")
    _stub_torchmetrics()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from models import codebooks_patterns, voicecraft  # type: ignore

    return voicecraft, codebooks_patterns


def build_reference_model(args, state_dict):
    """VoiceCraft(args).eval() of the reference with `state_dict` loaded."""
    voicecraft, _ = import_reference()
    import copy

    model = voicecraft.VoiceCraft(copy.copy(args)).eval()
    missing, unexpected = model.load_state_dict(state_dict, strict=False)
    bad = [k for k in missing if not (k in ("eog", "eos") or k.startswith("accuracy_metrics"))]
    assert not bad and not unexpected, (bad, unexpected)
    return model
