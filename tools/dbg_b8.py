import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from oracle.voicecraft_oracle import VoiceCraftOracle
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine
from test_gpu_model import rel_l2
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
a = synth.make_args("tiny_h16")
sd = synth.make_state_dict(a, seed=4)
prompts = [synth.random_prompt(a, 4 + (u % 5), 9 + 3 * (u % 7), seed=300 + u) for u in range(B)]
orc = VoiceCraftOracle(a, sd)
traces = []
for (xx, xl, yy) in prompts:
    tr = []
    orc.inference_tts(xx, xl, yy, top_k=1, stop_repetition=3, trace=tr)
    traces.append(tr)
n = max(len(t) for t in traces)
forced = np.zeros((n, B, 4), dtype=np.int64)
for b, tr in enumerate(traces):
    forced[: len(tr), b] = torch.stack([t["tokens"] for t in tr]).numpy()
for dtype in ("bf16", "fp32"):
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype=dtype, max_seqs=B, max_positions=256)
    outs, lg = eng.inference_tts_multi([p[0][0] for p in prompts], [p[2][0] for p in prompts], top_k=1, stop_repetition=3, _forced=forced, _logit_steps=n)
    lg = lg.cpu().numpy()
    for b, tr in enumerate(traces):
        want = torch.stack([t["logits"][0] for t in tr]).numpy()
        r = rel_l2(lg[: len(tr), b], want)
        print(dtype, 'seq', b, 'steps', len(tr), 'max rel', float(r.max()), 'at', int(r.argmax()), 'first5', np.round(r[:5], 4).tolist())
