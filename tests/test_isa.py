"""CPU: properties of the compiled gfx950 code that the measurements of round 3 depend on (hipcc cross-compiles here; the
device-only ISA of all translation units takes ~35 s, once per run).

* no hot kernel carries a chain of stores that each wait for the previous one (`s_waitcnt vmcnt(0)` between two stores):
  the block-GEMM epilogues lost 10-25 us per launch to exactly that before `tile_epilogue`;
* the decode and prefill kernels use no scratch memory;
* the instructions a kernel is built around are really there (LDS-DMA and bf16 MFMA in the 256 x 256 GEMM, the LDS
  transpose-read in the bf16 tile attention, the fp32 MFMA in exact mode).
"""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_store_scan as isa  # noqa: E402

pytestmark = pytest.mark.skipif(not (os.path.exists(isa.HIPCC) or shutil.which(isa.HIPCC)), reason="hipcc not available")


@pytest.fixture(scope="module")
def kern(tmp_path_factory):
    out = {}
    for stem, text in isa.compile_isa(str(tmp_path_factory.mktemp("isa"))).items():
        for name, (body, scratch, vgpr) in isa.kernels(text).items():
            out[isa.demangle(name)] = (body, scratch, vgpr)
    assert len(out) > 100, len(out)
    return out


def pick(kern, prefix):
    sel = {n: v for n, v in kern.items() if n.replace("void ", "").startswith(prefix)}
    assert sel, prefix
    return sel


# a cold path that keeps a short chain: the state-machine tail of best-of-N (one thread, three stores, after the keep decision)
CHAIN_ALLOWED = ("advance_only_k(",)


def test_no_serialised_store_chains_in_hot_kernels(kern):
    bad = {}
    for name, (body, _, _) in kern.items():
        n, total = isa.store_chains(body)
        if n >= 2 and not any(a in name for a in CHAIN_ALLOWED):
            bad[name] = (n, total)
    assert not bad, bad
    # the kernels the rule was found on, explicitly: every store of their epilogues is issued back to back
    for prefix in ("rows_gemm_big_k<", "rows_gemm_blk_k<bf16_t", "sample_fused_k("):
        for name, (body, _, _) in pick(kern, prefix).items():
            assert isa.store_chains(body)[0] <= 1, name


def test_kernels_use_no_scratch(kern):
    spilled = {n: s for n, (_, s, _) in kern.items() if s > 0}
    assert not spilled, spilled


def test_kernels_are_built_around_the_intended_instructions(kern):
    import re
    for name, (body, _, vgpr) in pick(kern, "rows_gemm_big_k<").items():
        assert body.count("global_load_lds_dwordx4") >= 4 and body.count("v_mfma_f32_16x16x32_bf16") >= 32, name
        assert "s_waitcnt vmcnt(8)" in body and vgpr <= 256, name        # counted wait of the 4-stage ring; two waves per SIMD
    for name, (body, _, vgpr) in pick(kern, "tile_attn_k<bf16_t, 128").items():
        assert body.count("ds_read_b64_tr_b16") == 16 and body.count("v_mfma_f32_16x16x32_bf16") == 16, name
        assert body.count("v_exp_f32") >= 8 and vgpr <= 256, name
    for name, (body, _, vgpr) in pick(kern, "tile_attn64_k(").items():     # second prefill attention: P stays in registers, one barrier per key tile
        assert body.count("v_mfma_f32_16x16x32_bf16") == 32 and body.count("ds_read_b64_tr_b16") == 32 and body.count("ds_read_b128") == 16, name
        assert body.count("v_cvt_pk_bf16_f32") >= 8 and vgpr <= 256, (name, vgpr)
    for name, (body, _, _) in pick(kern, "tile_attn_k<float, 128").items():
        assert "ds_read_b64_tr_b16" not in body and body.count("v_mfma_f32_16x16x4_f32") >= 64, name
    for name, (body, _, _) in pick(kern, "rows_attn_k<").items():      # <WT, NT, FAST>
        # the four words every address depends on come in ONE scalar batch
        assert len(re.findall(r"s_load_dword s\d+, s\[\d+:\d+\], 0x0\n\ts_load_dword s\d+, s\[\d+:\d+\], 0x0\n\ts_load_dword", body)) >= 1, name
    decode = pick(kern, "rows_gemm_k<bf16_t, 16, 0, 2, 1, true, false, 2>")      # FFN-up of a one-row step (h + the out-projection's two slabs)
    for name, (body, _, _) in decode.items():
        assert body.count("v_mfma_f32_16x16x32_bf16") >= 2 and " nt" in body, name     # non-temporal weight stream
    # finished-row form (2..16-row decode steps): the FFN down-projection streams a wave's whole share (32 fragments) in ONE burst
    # of non-temporal loads, 512 threads at <= 256 registers; the consumers with two tiles per workgroup are 512-thread kernels
    for name, (body, _, vgpr) in pick(kern, "rows_gemm_fr_k<bf16_t, 32, 1, 1, false>").items():
        assert body.count("global_load_dwordx4") >= 32 + 16 and body.count(" nt") >= 32 and body.count("v_mfma_f32_16x16x32_bf16") == 32, name
        assert vgpr <= 256, (name, vgpr)
    for name, (body, _, vgpr) in pick(kern, "rows_gemm_fr2_k<bf16_t, 16>").items():      # 9..16 rows: both halves' fragments up front
        assert body.count(" nt") >= 32 and body.count("v_mfma_f32_16x16x32_bf16") == 32 and vgpr <= 256, (name, vgpr)
    # one-row finished-row producer (round 5): the FFN down-projection at d = 2048 - 16 fragment PAIRS per wave (every lane's 16 bytes
    # used: two k-tiles per MFMA), all non-temporal, one MFMA per KB
    for name, (body, _, vgpr) in pick(kern, "row_gemm_fr1_k<bf16_t, 16, true, 8, 1, 5>").items():
        assert len(re.findall(r"global_load_dwordx4[^\n]* nt", body)) == 16 and body.count("v_mfma_f32_16x16x32_bf16") == 16, name
        assert vgpr <= 128, (name, vgpr)
    # ... and the QKV projection in the same paired form (LayerNorm prologue, QKV epilogue): 8 pairs per wave of four at d = 2048
    for name, (body, _, vgpr) in pick(kern, "row_gemm_fr1_k<bf16_t, 8, true, 4, 0, 0>").items():
        assert len(re.findall(r"global_load_dwordx4[^\n]* nt", body)) == 8 and body.count("v_mfma_f32_16x16x32_bf16") == 8, name
        assert vgpr <= 128, (name, vgpr)
    # ... and the FFN down-projection of 2..8-row steps: 16 full-KB fragments per wave instead of 32 half-filled ones
    for name, (body, _, vgpr) in pick(kern, "rows_gemm_frp_k<bf16_t, 16, 8>").items():
        assert len(re.findall(r"global_load_dwordx4[^\n]* nt", body)) == 16 and body.count("v_mfma_f32_16x16x32_bf16") == 16, name
        assert vgpr <= 256, (name, vgpr)
    # ... and the QKV projection of 2..8 finished rows in the same paired form (round 6): three 8-channel tiles per workgroup, 4 pairs per
    # wave and tile at d = 2048 - 12 full-KB weight requests (non-temporal) behind the 4 requests of the wave's row, 12 MFMAs, no scratch
    qp = pick(kern, "rows_gemm_qp_k<bf16_t, 4, 8>")
    assert qp
    for name, (body, _, vgpr) in qp.items():
        assert len(re.findall(r"global_load_dwordx4[^\n]* nt", body)) == 12 and body.count("v_mfma_f32_16x16x32_bf16") == 12, name
        assert "scratch_" not in body and vgpr <= 128, (name, vgpr)
    # the GEMM path reads `*a.n_active` with a SCALAR load (round 5: an inline-asm prefetch role in the same kernel had turned it into a
    # vector load; the roles left the tree in round 6)
    for prefix in ("rows_gemm_k<bf16_t, 16, 0, 2, 1, true, false, 2>", "rows_gemm_k<bf16_t, 8, 2, 1, 1, true, false, 4>"):
        for name, (body, _, _) in pick(kern, prefix).items():
            assert ";;#ASMSTART" not in body and re.search(r"s_load_dword s\d+, s\[\d+:\d+\], 0x0\n", body), name
    # the trimmed LayerNorm prologues: every slab the prologue does not request is two 16-byte loads per thread and row less
    n_plain = {}
    for np_ in ("0", "2", "4"):
        for name, (body, _, _) in pick(kern, f"rows_gemm_k<bf16_t, 16, 0, 0, 1, true, false, {np_}>").items():
            n_plain[np_] = len(re.findall(r"global_load_dwordx4 ", body)) - len(re.findall(r"global_load_dwordx4[^\n]* nt", body))
    assert n_plain["0"] + 4 <= n_plain["2"] and n_plain["2"] + 4 <= n_plain["4"], n_plain
    for prefix in ("rows_gemm_k<bf16_t, 16, 3, 0, 2, true, false, 4>", "rows_gemm_k<bf16_t, 16, 3, 2, 2, true, false, 4>"):
        for name, (body, _, vgpr) in pick(kern, prefix).items():
            assert vgpr <= 128 and " nt" in body, (name, vgpr)             # 8 waves per workgroup, two workgroups per CU


def test_wide_decode_kernel_keeps_every_row_tile_in_flight(kern):
    """rows_gemm_wd_k / rows_gemm_wds_k (round 6, 17..64-row steps): nothing is waited for before everything is requested - at d = 2048 /
    64 rows a wave of the direct form asks for its 32 X fragments (plain loads, L2) and then its 16 weight fragments (non-temporal, HBM) in
    one batch; the staged form asks for two chunks of whole-line X requests and the 16 weight fragments, parks X in its own LDS stage
    (ds_write_b128) and reads the fragments back; both run 64 MFMAs into 8 accumulators and meet the other waves at ONE barrier; 512
    threads at <= 256 registers, no scratch."""
    import re
    direct = {n: v for n, v in kern.items() if "rows_gemm_wd_k<" in n}
    staged = {n: v for n, v in kern.items() if "rows_gemm_wds_k<" in n}
    assert len(direct) >= 80 and len(staged) >= 60, (len(direct), len(staged))
    for sel, pat in ((direct, r"rows_gemm_wd_k<(\w+), (\d+), (\d+), (\d+)>"), (staged, r"rows_gemm_wds_k<(\w+), (\d+), (\d+), (\d+)>")):
        for name, (body, scratch, vgpr) in sel.items():
            m = re.search(pat, name)
            assert m, name
            wt, kpw, rt = m.group(1), int(m.group(2)), int(m.group(3))
            assert scratch == 0 and vgpr <= 256, (name, scratch, vgpr)
            assert body.count("s_barrier") == 1, name
            nt_loads = len(re.findall(r"global_load_dwordx4[^\n]* nt", body))
            mf = body.count("v_mfma_f32_16x16x32_bf16") if wt == "bf16_t" else body.count("v_mfma_f32_16x16x4_f32") // 4
            if sel is staged or kpw <= (8 if wt == "bf16_t" else 4):         # straight-line code: every fragment counted once
                assert nt_loads == 2 * kpw and mf == 2 * rt * kpw, (name, nt_loads, mf)
    main = {n: v for n, v in direct.items() if "rows_gemm_wd_k<bf16_t, 8, 4, 2>" in n}
    assert main
    for name, (body, _, vgpr) in main.items():      # FFN-up at d = 2048, 33..64 rows, direct form
        first_wait = body.index("s_waitcnt vmcnt")
        assert len(re.findall(r"global_load_dwordx4", body[:first_wait])) >= 48, name         # 32 X + 16 W (+ the bias) before the first wait
    main = {n: v for n, v in staged.items() if "rows_gemm_wds_k<bf16_t, 8, 4, 2>" in n}
    assert main
    for name, (body, _, vgpr) in main.items():      # ... and the staged form: 2 chunks x 8 whole-line requests + 16 W ahead of the first wait
        first_wait = body.index("s_waitcnt vmcnt")
        assert len(re.findall(r"global_load_dwordx4", body[:first_wait])) >= 32, name
        assert body.count("global_load_dwordx4") == 4 * 8 + 16 + 1 and body.count("ds_write_b128") >= 32 + 8 and vgpr <= 200, (name, vgpr)


def test_the_non_temporal_hint_survives_in_every_decode_gemm(kern):
    """Through round 3 the hint was a RUNTIME flag: the compiler merged the kernel's two load arms and dropped it - no
    non-temporal load at all in the compiled QKV, out-projection and heads-2 forms, 15 of 16 in the FFN forms (found in the ISA
    in round 4).  It is a template parameter now: every `..., true>` form carries at least its KTW weight loads with the hint
    (twice that once a refill exists), every `..., false>` form none."""
    import re
    seen = 0
    for name, (body, _, _) in pick(kern, "rows_gemm_k<").items():
        m = re.search(r"rows_gemm_k<(\w+), (\d+), (\d+), (\d+), (\d+), (true|false), (true|false), (\d+)>", name)
        assert m, name
        ktw, nt = int(m.group(2)), m.group(6) == "true"
        n = len(re.findall(r"global_load_dwordx4[^\n]* nt", body))
        assert (n >= ktw) if nt else (n == 0), (name, n)
        seen += 1
    assert seen >= 100, seen
