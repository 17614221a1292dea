"""The caller of the hot path, as one function: `inference_one_sample` of the reference's TTS driver
(inference_tts_scale.py:42-105) with the two front-ends that are out of scope taken as arguments - the phonemizer
(`text_tokens` are the integer phoneme ids it would produce, :43-50) and the audio file loader (`wav` is the fp32 16 kHz
waveform `tokenize_audio` would read, data/tokenizer.py:136-150).  Everything between them is the engine:

    AudioTokenizer.encode(wav[: prompt_end_frame])      -> codes [1,K,T]         (:53-54)
    model.inference_tts / inference_tts_batch            -> concat [1,K,T+Tg], gen [1,K,Tg]   (:60-86)
    AudioTokenizer.decode(concat), .decode(gen)          -> two waveforms        (:95-100)

`model` is a VoiceCraftEngine (or anything with the reference's model interface), `audio_tokenizer` an AudioTokenizer.
"""
from __future__ import annotations

import ast
import logging
import time

import torch


@torch.no_grad()
def inference_one_sample(model, model_args, text_tokens, audio_tokenizer, wav, device, decode_config, prompt_end_frame,
                         timings: dict | None = None):
    """text_tokens: int64 [Lx] or [1,Lx]; wav: fp32 [N], [1,N] or [1,1,N] at audio_tokenizer.sample_rate;
    prompt_end_frame: samples of `wav` that form the voice prompt (-1 = all), as the reference's `num_frames`.
    decode_config: the reference's dict (top_k, top_p, temperature, stop_repetition, kvcache, codec_sr,
    silence_tokens, sample_batch_size).  Returns (concat_sample, gen_sample), fp32 [1,1,320*frames] each.
    `timings` (optional dict) receives the wall time of the three stages in seconds (synchronised)."""
    x = torch.as_tensor(text_tokens, dtype=torch.int64).reshape(1, -1)
    x_lens = torch.tensor([x.shape[-1]], dtype=torch.int64)
    w = torch.as_tensor(wav, dtype=torch.float32).reshape(1, 1, -1)
    if prompt_end_frame is not None and prompt_end_frame > 0:
        w = w[..., : int(prompt_end_frame)]

    def clock():
        if timings is not None:
            torch.cuda.synchronize(device)
        return time.perf_counter()

    t0 = clock()
    encoded_frames = audio_tokenizer.encode(w.to(device))
    original_audio = encoded_frames[0][0].transpose(2, 1)                      # [1,T,K]
    K = int(model_args.n_codebooks)
    assert original_audio.ndim == 3 and original_audio.shape[0] == 1 and original_audio.shape[2] == K, original_audio.shape
    logging.info(f"original audio length: {original_audio.shape[1]} codec frames, "
                 f"which is {original_audio.shape[1] / decode_config['codec_sr']:.2f} sec.")
    t1 = clock()
    sil = decode_config["silence_tokens"]
    sil = ast.literal_eval(sil) if isinstance(sil, str) else sil                # the reference eval()s the string form
    kw = dict(top_k=decode_config["top_k"], top_p=decode_config["top_p"], temperature=decode_config["temperature"],
              stop_repetition=decode_config["stop_repetition"], kvcache=decode_config["kvcache"], silence_tokens=sil)
    n = int(decode_config.get("sample_batch_size", 1))
    if n <= 1:
        concat_frames, gen_frames = model.inference_tts(x.to(device), x_lens.to(device), original_audio[..., :K].to(device), **kw)
    else:
        concat_frames, gen_frames = model.inference_tts_batch(x.to(device), x_lens.to(device), original_audio[..., :K].to(device),
                                                              batch_size=n, **kw)
    t2 = clock()
    logging.info(f"generated encoded_frames.shape: {gen_frames.shape}, which is {gen_frames.shape[-1] / decode_config['codec_sr']} sec.")
    concat_sample = audio_tokenizer.decode([(concat_frames, None)])
    gen_sample = audio_tokenizer.decode([(gen_frames, None)])
    t3 = clock()
    if timings is not None:
        timings.update(encode_s=t1 - t0, model_s=t2 - t1, decode_s=t3 - t2, total_s=t3 - t0,
                       prompt_frames=int(original_audio.shape[1]), gen_frames=int(gen_frames.shape[-1]))
    return concat_sample, gen_sample
