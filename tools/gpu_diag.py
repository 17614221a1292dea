"""Stage-by-stage numerical diagnosis of the HIP engine against the oracle (run on the GPU box).
Prints one line per stage so a single gpurun call localises a wrong kernel.  Not a test."""
from __future__ import annotations

import os
import sys
import time
import traceback

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from _util import MODEL_CASES, build_case, load_golden, run_oracle_case  # noqa: E402
from oracle.voicecraft_oracle import VoiceCraftOracle, prompt_columns_tts  # noqa: E402


def stage(name, fn):
    t = time.time()
    try:
        msg = fn()
        print(f"[diag] {name}: {msg}  ({time.time() - t:.2f}s)", flush=True)
    except Exception:
        print(f"[diag] {name}: EXCEPTION\n{traceback.format_exc()}", flush=True)


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def diag_case(name, dtype):
    from voicecraft_amd.engine import VoiceCraftEngine
    spec, args, sd, x, x_lens, y = build_case(name)
    trace = []
    res_o, _ = run_oracle_case(name, trace=trace)
    want = torch.stack([t["logits"][0] for t in trace]).numpy()
    forced = torch.stack([t["tokens"] for t in trace]).numpy()
    eng = VoiceCraftEngine(args, sd, device="cuda:0", dtype=dtype, max_seqs=4, max_positions=512, use_graph=False)
    orc = VoiceCraftOracle(args, sd)
    d, K = args.d_model, args.n_codebooks
    Lx, T = x.shape[1], y.shape[1]

    def run_forced():
        kn = dict(spec["knobs"])
        if spec["mode"] == "tts":
            out = eng.inference_tts(x.cuda(), x_lens.cuda(), y.cuda(), **kn, _forced=forced, _logit_steps=len(trace), _seed=1)
            res, lg = out[0], out[2]
        else:
            mi = torch.tensor([spec["spans"]], dtype=torch.int64)
            res, lg = eng.inference(x.cuda(), x_lens.cuda(), y.cuda(), mi, **kn, _forced=forced, _logit_steps=len(trace), _seed=1)
        got = lg.cpu().numpy()
        err = np.abs(got - want).max(axis=(1, 2))
        live = np.abs(want) < 1e3
        rel = (np.sqrt((((got - want) * live) ** 2).reshape(len(trace), -1).sum(1)) /
               np.sqrt(((want * live) ** 2).reshape(len(trace), -1).sum(1)))
        same = np.array_equal(res.cpu().numpy(), res_o.numpy())
        return (f"steps={len(trace)} engine_steps={eng.last_steps} res_equal={same} max|d| step0={err[0]:.3g} "
                f"step1={err[1]:.3g} worst={err.max():.3g}@{int(err.argmax())} rel_l2 max={rel.max():.3g} "
                f"nan={int(np.isnan(got).sum())}")

    stage(f"{name}/{dtype} teacher-forced logits", run_forced)

    if spec["mode"] == "tts":
        def emb_check():
            rows = Lx + T + 1
            emb = eng.debug_read("emb", (rows, d))
            xin = orc._pos(F.embedding(x, orc.sd["text_embedding.word_embeddings.weight"]), "text")[0]
            cols = torch.from_numpy(np.ascontiguousarray(prompt_columns_tts(y[0].numpy(), args.empty_token)))
            yin = orc._pos(orc._embed_cols(cols.unsqueeze(-1)), "audio")[0]
            ref = torch.cat([xin, yin], dim=0)
            return f"prompt embedding rows={rows} max|d|={float((emb - ref).abs().max()):.3g}"
        stage(f"{name}/{dtype} prompt embedding", emb_check)

        def kv_check():
            rows = Lx + T + 1
            H, hd, S = args.nhead, args.d_model // args.nhead, 512
            tdt = torch.float32 if dtype == "fp32" else torch.bfloat16
            kc = eng.debug_read("kcache0", (4, H, S, hd), dtype=tdt).float()[0, :, :rows]
            vc = eng.debug_read("vcache0", (4, H, S, hd), dtype=tdt).float()[0, :, :rows]
            xin = orc._pos(F.embedding(x, orc.sd["text_embedding.word_embeddings.weight"]), "text")
            cols = torch.from_numpy(np.ascontiguousarray(prompt_columns_tts(y[0].numpy(), args.empty_token)))
            yin = orc._pos(orc._embed_cols(cols.unsqueeze(-1)), "audio")
            xy = torch.cat([xin, yin], dim=1)
            p = "decoder.layers.0."
            h = F.layer_norm(xy, (d,), orc.sd[p + "norm1.weight"], orc.sd[p + "norm1.bias"], 1e-5)
            proj = F.linear(h, orc.sd[p + "self_attn.in_proj_weight"], orc.sd[p + "self_attn.in_proj_bias"])[0]
            k = proj[:, d:2 * d].view(rows, H, hd).transpose(0, 1)
            v = proj[:, 2 * d:].view(rows, H, hd).transpose(0, 1)
            return (f"layer0 K max|d|={float((kc - k).abs().max()):.3g} V max|d|={float((vc - v).abs().max()):.3g} "
                    f"(|K|max={float(k.abs().max()):.3g})")
        stage(f"{name}/{dtype} layer-0 KV cache after the run (prefill rows)", kv_check)

    def free_run():
        kn = dict(spec["knobs"])
        if spec["mode"] == "tts":
            res = eng.inference_tts(x.cuda(), x_lens.cuda(), y.cuda(), **kn, _seed=1)[0]
        else:
            mi = torch.tensor([spec["spans"]], dtype=torch.int64)
            res = eng.inference(x.cuda(), x_lens.cuda(), y.cuda(), mi, **kn, _seed=1)
        g = load_golden(name)["res"]
        r = res.cpu().numpy()
        if r.shape != g.shape:
            return f"free-running shape {r.shape} vs golden {g.shape}"
        agree = float((r == g).mean())
        return f"free-running tokens equal golden: {np.array_equal(r, g)} (agreement {agree:.3f}) timing={eng.last_timing_ms()}"
    stage(f"{name}/{dtype} free-running", free_run)
    for g in (True,):
        def graph_run():
            eng.use_graph = True
            kn = dict(spec["knobs"])
            if spec["mode"] == "tts":
                res = eng.inference_tts(x.cuda(), x_lens.cuda(), y.cuda(), **kn, _seed=1)[0]
            else:
                mi = torch.tensor([spec["spans"]], dtype=torch.int64)
                res = eng.inference(x.cuda(), x_lens.cuda(), y.cuda(), mi, **kn, _seed=1)
            eng.use_graph = False
            gg = load_golden(name)["res"]
            r = res.cpu().numpy()
            return f"hipGraph tokens equal golden: {r.shape == gg.shape and np.array_equal(r, gg)} timing={eng.last_timing_ms()}"
        stage(f"{name}/{dtype} hipGraph", graph_run)
    del eng


def main():
    print("[diag] torch", torch.__version__, "cuda", torch.cuda.is_available(), torch.cuda.get_device_name(0) if torch.cuda.is_available() else "-", flush=True)
    names = sys.argv[1:] or ["tts_greedy", "tts_greedy_hd128", "edit_2span"]
    for name in names:
        for dtype in ("fp32", "bf16"):
            diag_case(name, dtype)


if __name__ == "__main__":
    main()
