"""Times the persistent LSTM (lstm_persist_k, the default) against the launch-per-step two-layer wavefront
(lstm_wave_k, VC_LSTM_WAVE=1) and checks that codes and waveform are identical.
usage: timeout 120 python tools/lstm_probe.py [batch]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from voicecraft_amd import synth
from voicecraft_amd.codec import AudioTokenizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sd = synth.make_codec_state_dict(0)
tok = AudioTokenizer(sd, device="cuda:0", max_seconds=17.0, max_batch=B)
torch.manual_seed(0)
res = {}
for secs in (1, 16):
    wav = (torch.randn(B, 1, 16000 * secs) * 0.1).cuda()
    for mode in ("wave", "persist"):
        if mode == "wave":
            os.environ["VC_LSTM_WAVE"] = "1"
        else:
            os.environ.pop("VC_LSTM_WAVE", None)
        for _ in range(2):
            codes = tok.encode(wav)[0][0]
        enc_ms = tok.last_ms()
        lstm_ms, _ = tok.last_lstm_ms()
        for _ in range(2):
            back = tok.decode([(codes, None)])
        dec_ms = tok.last_ms()
        res[(secs, mode)] = (codes.cpu(), back.cpu())
        print(f"[lstm] {secs:2d} s x {B}: {mode:8s} encode {enc_ms:7.2f} ms (LSTM part {lstm_ms:6.2f} ms), decode {dec_ms:7.2f} ms", flush=True)
    same_codes = bool((res[(secs, "wave")][0] == res[(secs, "persist")][0]).all())
    err = float((res[(secs, "wave")][1] - res[(secs, "persist")][1]).abs().max())
    print(f"[lstm] {secs:2d} s x {B}: persist: codes identical {same_codes}, waveform max |diff| {err:.3g}", flush=True)
os.environ.pop("VC_LSTM_WAVE", None)
