"""Validates the stream engine (VC_STREAM=1: the batch-1 decode step as one persistent launch, vc_stream.hip) against the
launch path and the CPU oracle on a small model, and bisects a mismatch op by op (VC_STREAM_DBG dumps the inputs of
every op of the last executed step; the same quantities are recomputed here on the CPU from the state dict).
usage: timeout 300 python tools/stream_probe.py [preset=tiny128] [layers=2] [Lx=7] [T=20]"""
import os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.voicecraft_oracle import VoiceCraftOracle, delayed_shift
from voicecraft_amd import synth
from voicecraft_amd.engine import VoiceCraftEngine

preset = sys.argv[1] if len(sys.argv) > 1 else "tiny128"
L = int(sys.argv[2]) if len(sys.argv) > 2 else 2
Lx = int(sys.argv[3]) if len(sys.argv) > 3 else 7
T = int(sys.argv[4]) if len(sys.argv) > 4 else 20
a = synth.make_args(preset)
a.num_decoder_layers = L
sd = synth.make_state_dict(a, seed=3, fast=(a.d_model >= 1024))
x, xl, y = synth.random_prompt(a, Lx, T, seed=5)
orc = VoiceCraftOracle(a, sd)
torch.set_num_threads(16)
if a.d_model >= 1024:          # full size: a forced random trajectory, evaluated in one pass
    n = 24
    K = a.n_codebooks
    forced = np.random.RandomState(3).randint(0, 2048, size=(n, K)).astype(np.int64)
    for j in range(K):
        forced[n - K + j, :j] = a.empty_token
        forced[n - K + j, j] = a.eos
    want = orc.tts_logits_for_trajectory(x, y, forced, steps=list(range(n))).numpy()
else:
    trace = []
    orc.inference_tts(x, xl, y, top_k=1, stop_repetition=3, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    n = len(trace)


def rel(got, ref):
    live = np.abs(ref) < 1e3
    return np.sqrt((((got - ref) * live) ** 2).reshape(len(ref), -1).sum(1)) / np.sqrt(((ref * live) ** 2).reshape(len(ref), -1).sum(1))


def run(stream):
    if stream:
        os.environ["VC_STREAM"] = "1"; os.environ["VC_STREAM_DBG"] = "1"
    else:
        os.environ.pop("VC_STREAM", None); os.environ.pop("VC_STREAM_DBG", None)
    eng = VoiceCraftEngine(a, sd, device="cuda:0", dtype="bf16", max_seqs=1, max_positions=1024)
    res, gen, lg = eng.inference_tts(x.cuda(), xl.cuda(), y.cuda(), top_k=1, stop_repetition=3, _forced=forced, _logit_steps=n)
    return eng, lg.cpu().numpy(), eng.last_timing_ms()


e0, lg0, tm0 = run(False)
print(f"[probe] {preset} L={L} d={a.d_model} H={a.nhead}: launches  rel-L2 vs oracle max {rel(lg0, want).max():.4f}  {tm0}", flush=True)
try:
    e1, lg1, tm1 = run(True)
except Exception as ex:
    print("[probe] stream run FAILED:", ex, flush=True)
    sys.exit(1)
r1 = rel(lg1, want)
print(f"[probe] stream    rel-L2 vs oracle max {r1.max():.4f} (per step: {' '.join(f'{v:.3f}' for v in r1[:12])})  {tm1}", flush=True)
print(f"[probe] stream vs launches rel-L2 max {rel(lg1, lg0).max():.5f}; persistent launches counted: {e1.launch_counts()['persist']}", flush=True)

# ---- op-by-op bisect of the LAST executed decode step (step n-1: input = tokens of step n-2 at audio position T+n-1)
d, H = a.d_model, a.nhead
hd = d // H
dbg = e1.debug_read("stream_dbg", (L, 5, 4 * d))
yk = y.transpose(2, 1)
cols = torch.from_numpy(np.ascontiguousarray(delayed_shift(yk.numpy(), a.empty_token)[0][:, :T + 1]))
allc = torch.cat([cols, torch.from_numpy(forced[: n - 1]).t().contiguous()], dim=1)
x_in = orc._pos(F.embedding(x, sd["text_embedding.word_embeddings.weight"]), "text")
y_in = orc._pos(orc._embed_cols(allc.unsqueeze(-1)), "audio")
h = torch.cat([x_in, y_in], dim=1)[0]                    # [S,d]
S = h.shape[0]
names = ["h_in", "attn_out", "h_after_attn", "ffn_act", "h_out"]
for l in range(L):
    p = f"decoder.layers.{l}."
    xn = F.layer_norm(h, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    qkv = F.linear(xn, sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"])
    q, k, v = qkv.split(d, dim=1)
    qh, kh, vh = (t.view(S, H, hd).transpose(0, 1) for t in (q, k, v))
    att = F.scaled_dot_product_attention(qh, kh, vh, is_causal=True).transpose(0, 1).reshape(S, d)
    h2 = h + F.linear(att, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
    act = F.relu(F.linear(F.layer_norm(h2, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5), sd[p + "linear1.weight"], sd[p + "linear1.bias"]))
    h3 = h2 + F.linear(act, sd[p + "linear2.weight"], sd[p + "linear2.bias"])
    refs = [h[-1], att[-1], h2[-1], act[-1], h3[-1]]
    out = []
    for i, (nm, r) in enumerate(zip(names, refs)):
        g = dbg[l, i, : r.numel()]
        out.append(f"{nm} {float((g - r).norm() / (r.norm() + 1e-9)):.4f}")
    print(f"[probe] layer {l}: rel-L2 of the kernel's view vs fp32 CPU: " + " | ".join(out), flush=True)
    h = h3

# ---- phase stamps of the last executed step (100 MHz wall clock -> us), workgroups 0 (a head leader), 1 and G-1
G = a.d_model // 8
raw = e1.debug_read("stream_ts", (3 * L * 16 + G * 4,), torch.int64).numpy().astype(np.float64) / 100.0
ts = raw[: 3 * L * 16].reshape(3, L, 16)
per_cu = raw[3 * L * 16:].reshape(G, 4)                  # layer L/2, every workgroup: ffn1 done, act gathered, ffn2 done, h gathered
if per_cu[0, 0] > 0:
    t0 = per_cu[:, 0].min()
    q = lambda v: " ".join(f"{np.percentile(v - t0, p):.2f}" for p in (0, 10, 50, 90, 100))
    print(f"[probe] layer {L // 2}, all {G} workgroups, us after the first ffn1-done (min p10 p50 p90 max): ffn1-done {q(per_cu[:, 0])} | act-gathered {q(per_cu[:, 1])} | "
          f"ffn2-done {q(per_cu[:, 2])} | h-gathered {q(per_cu[:, 3])}", flush=True)
lab = ["ln1", "qkv", "q-edge", "attn", "merge", "o-edge", "oproj", "h2-edge", "ln2", "ffn1", "act-edge", "ffn2", "h-edge"]
for w in range(3):
    t = ts[w]
    if t[0, 0] == 0:
        continue
    dur = np.diff(t[:, :14], axis=1)                      # [L,13]
    mean = dur[1:-1].mean(axis=0) if L > 2 else dur.mean(axis=0)
    print(f"[probe] wg {['0', '1', 'G-1'][w]}: step {(t[-1, 13] - t[0, 0]):.1f} us; per layer (mean of the inner layers) "
          + " ".join(f"{n} {v:.2f}" for n, v in zip(lab, mean)) + f" | layer {mean.sum():.2f} us", flush=True)
