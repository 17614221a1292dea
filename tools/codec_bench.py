"""EnCodec encode/decode timing on the GPU box: HIP engine (HIP-event ms) vs the transformers CPU restatement."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicecraft_amd import synth
from voicecraft_amd.codec import AudioTokenizer
from oracle import encodec_oracle as eo
sd = synth.make_codec_state_dict(0)
tok = AudioTokenizer(sd, device="cuda:0", max_seconds=17.0)
for secs in (3, 16):
    n = 16000 * secs
    wav = torch.randn(1, 1, n) * 0.1
    wc = wav.cuda()
    for _ in range(2):
        codes = tok.encode(wc)[0][0]
    enc_ms = tok.last_ms()
    for _ in range(2):
        back = tok.decode([(codes, None)])
    dec_ms = tok.last_ms()
    T = codes.shape[2]
    print(f"[codec] {secs}s audio ({T} frames): HIP encode {enc_ms:.2f} ms (RTF {enc_ms/1e3/secs:.5f}), decode {dec_ms:.2f} ms (RTF {dec_ms/1e3/secs:.5f})", flush=True)
    if secs == 3:
        m = eo.build(sd)
        torch.set_num_threads(16)
        t = time.perf_counter(); co, _ = eo.encode(m, wav); t1 = time.perf_counter() - t
        t = time.perf_counter(); eo.decode(m, co); t2 = time.perf_counter() - t
        print(f"[codec] {secs}s audio: CPU restatement (16 threads) encode {t1*1e3:.0f} ms, decode {t2*1e3:.0f} ms; codes equal: {float((co == codes[0].cpu()).float().mean()):.4f}", flush=True)
