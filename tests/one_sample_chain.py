"""Test / bench helper (NOT part of the product package): the three engine stages a TTS request goes through, timed one by one.

INTEGRATION.md's claim is that the reference's own driver (`inference_tts_scale.inference_one_sample`) runs unchanged on
`VoiceCraftEngine` + `AudioTokenizer`; the driver itself also phonemizes text and reads an audio file, which are out of
scope here.  This helper drives exactly the part in between - voice prompt -> codes, codes + phonemes -> generated codes,
codes -> waveforms - so that `tests/test_gpu_pipeline.py` can check every stage against the oracles and `bench.py` can
report the end-to-end time of one request (`one_sample`).
"""
from __future__ import annotations

import ast
import time
from dataclasses import dataclass, field

import torch


@dataclass
class ChainResult:
    prompt_codes: torch.Tensor          # int64 [1,T,K] as the model takes them
    all_codes: torch.Tensor             # int64 [1,K,T+Tg]   prompt followed by the generated frames
    new_codes: torch.Tensor             # int64 [1,K,Tg]     generated frames only
    wave_all: torch.Tensor              # fp32 waveform of all_codes
    wave_new: torch.Tensor              # fp32 waveform of new_codes
    seconds: dict = field(default_factory=dict)      # stage -> wall seconds (device-synchronised)


class OneSampleChain:
    """engine: VoiceCraftEngine (or anything with the reference's model interface); codec: AudioTokenizer."""

    def __init__(self, engine, codec, n_codebooks: int, device):
        self.engine, self.codec, self.K, self.device = engine, codec, int(n_codebooks), device

    def _tick(self):
        torch.cuda.synchronize(self.device)
        return time.perf_counter()

    def encode_prompt(self, waveform, n_prompt_samples: int | None = None) -> torch.Tensor:
        w = torch.as_tensor(waveform, dtype=torch.float32).reshape(1, 1, -1)
        if n_prompt_samples is not None and n_prompt_samples > 0:
            w = w[..., : int(n_prompt_samples)]
        codes = self.codec.encode(w.to(self.device))[0][0]            # [1,K,T]
        codes = codes.transpose(2, 1)                                  # the model wants time-major [1,T,K]
        assert codes.ndim == 3 and codes.shape[0] == 1 and codes.shape[2] == self.K, codes.shape
        return codes

    def generate(self, phonemes, prompt_codes, sampling: dict):
        x = torch.as_tensor(phonemes, dtype=torch.int64).reshape(1, -1).to(self.device)
        x_len = torch.tensor([x.shape[1]], dtype=torch.int64, device=self.device)
        silence = sampling["silence_tokens"]
        if isinstance(silence, str):                                   # configs carry it as text
            silence = ast.literal_eval(silence)
        knobs = {k: sampling[k] for k in ("top_k", "top_p", "temperature", "stop_repetition", "kvcache")}
        best_of = int(sampling.get("sample_batch_size", 1))
        y = prompt_codes[..., : self.K].to(self.device)
        if best_of > 1:
            return self.engine.inference_tts_batch(x, x_len, y, batch_size=best_of, silence_tokens=silence, **knobs)
        return self.engine.inference_tts(x, x_len, y, silence_tokens=silence, **knobs)

    def decode_both(self, all_codes, new_codes):
        return self.codec.decode([(all_codes, None)]), self.codec.decode([(new_codes, None)])

    def run(self, phonemes, waveform, sampling: dict, n_prompt_samples: int | None = None) -> ChainResult:
        t0 = self._tick()
        prompt = self.encode_prompt(waveform, n_prompt_samples)
        t1 = self._tick()
        both, new = self.generate(phonemes, prompt, sampling)
        t2 = self._tick()
        wave_all, wave_new = self.decode_both(both, new)
        t3 = self._tick()
        return ChainResult(prompt, both, new, wave_all, wave_new,
                           {"encode": t1 - t0, "model": t2 - t1, "decode": t3 - t2, "total": t3 - t0})
