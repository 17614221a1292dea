"""Host-side pass planning without a GPU (vc_debug_plan): for EVERY model width the engine accepts (d a multiple of 256 up to
2048; head_dim 32 / 64 / 128), both compute dtypes and every row count of a decode pass, the form the engine would pick must be
one the kernel launchers accept - the finished-row producers have shape constraints (rows x K of X in one workgroup's LDS in one or
two pieces, a wave's fragments in registers, rows x splits <= 16 partials per thread) that only a few widths exercise on hardware."""
import ctypes as C

import pytest

from voicecraft_amd import _lib
from voicecraft_amd._lib import ModelCfg

WIDTHS = [(d, h) for d in range(256, 2049, 256) for h in (d // 32, d // 64, d // 128) if h >= 1 and d % h == 0 and d // h in (32, 64, 128)]


def plan(d, h, dtype, rows):
    cfg = ModelCfg(d_model=d, nhead=h, num_layers=2, n_codebooks=4, audio_vocab_size=2048, n_special=4, text_rows=101, head_hidden=1024,
                   empty_token=2048, eog=2049, audio_pad_token=2050, eos=2051, reduced_eog=1, encodec_sr=50, max_n_spans=3, max_seqs=64,
                   max_positions=1024)
    out = (C.c_int32 * 16)()
    assert _lib.load().vc_debug_plan(C.byref(cfg), dtype, rows, out) == 0
    return list(out)


@pytest.mark.parametrize("dtype", [_lib.VC_DTYPE_BF16, _lib.VC_DTYPE_F32])
def test_every_planned_pass_is_launchable(dtype):
    seen_forms = set()
    for d, h in WIDTHS:
        for rows in range(1, 65):
            frmax, form, nsplit, mt, oform, dform, heads_lnw, even, fr1, qkvp8, frp, wd, wd_ko, wd_kf, wd_kpw, wd_kpw_h = plan(d, h, dtype, rows)
            assert even == 1, (d, h)
            assert 0 <= frmax <= 16 and 1 <= nsplit <= 8
            if rows == 1:
                assert form == 0
                # round 5's one-row forms: every width has them in bf16; the exact mode wherever a wave's fragments fit its registers
                assert fr1 in (0, 1) and qkvp8 in (0, 1) and (qkvp8 <= fr1)
                if dtype == _lib.VC_DTYPE_BF16:
                    assert fr1 == 1 and qkvp8 == 1, (d, h)
            else:
                # 2..8 finished rows: 2 = the paired QKV consumer of round 6 (rows_gemm_qp_k; vc_debug_plan asks the launcher's own predicate), 0 = 12-channel tiles
                assert fr1 == -1 and (qkvp8 in (0, 2) if (form == 1 and rows <= 8) else qkvp8 == -1)
                if qkvp8 == 2:
                    assert (d * (2 if dtype == _lib.VC_DTYPE_BF16 else 4)) % 1024 == 0 and rows <= (6 if d >= 2048 else 8)
            if rows > 16:
                assert form == 2 and nsplit == 1
                # round 6: the wide-decode kernel has a form for the power-of-two widths (a wave owns 1, 2, 4, 8 or 16 k-tiles) in both
                # dtypes; other widths fall back to the weight-stationary kernel of rounds 2-5
                assert wd in (0, 1)
                if d in (256, 512, 1024, 2048):
                    assert wd == 1 and wd_ko in (1, 2, 4) and wd_kf in (1, 2, 4) and wd_kpw in (1, 2, 4, 8, 16) and wd_kpw_h in (4, 8), (d, h, dtype, plan(d, h, dtype, rows))
                    assert (d // 16 + 1) // 2 * wd_ko <= 256 or wd_ko == 1
            else:
                assert wd == -1
            if form == 1 and rows <= 8:
                assert frp in (0, 1)
                if dtype == _lib.VC_DTYPE_BF16 and d in (256, 512, 1024, 2048):      # (a wave's share of K must be a power of two of fragment pairs)
                    assert frp == 1, (d, rows)
            if form == 1:
                assert 2 <= rows <= frmax
                assert oform == 1 and dform in (1, 2), (d, h, dtype, rows, oform, dform)
                assert mt in ((4,) if rows > 8 else (0, 3)), (d, rows, mt)
                assert nsplit * rows <= 16 or nsplit == 1
                assert (nsplit == 1) == (rows > 8)
                assert heads_lnw == (1 if rows <= 8 else 0)
                seen_forms.add((dform, mt))
            elif rows <= 16:
                assert rows == 1 or rows > frmax
    assert (1, 0) in seen_forms and (1, 3) in seen_forms and (2, 4) in seen_forms, seen_forms


def test_the_benchmarked_shapes_take_the_forms_the_profiles_describe():
    bf, f32 = _lib.VC_DTYPE_BF16, _lib.VC_DTYPE_F32
    assert plan(2048, 16, bf, 8)[:6] == [16, 1, 2, 3, 1, 1]          # config 5's share: split 2, two tiles per consumer workgroup, one piece
    assert plan(2048, 16, bf, 16)[:6] == [16, 1, 1, 4, 1, 2]         # 16 rows: unsplit attention, two rows per wave, FFN-down in two halves
    assert plan(2048, 16, bf, 4)[:6] == [16, 1, 4, 0, 1, 1]
    assert plan(2048, 16, bf, 2)[2] == 8 and plan(2048, 16, bf, 1)[:3] == [16, 0, 8]
    assert plan(2048, 16, f32, 4)[:2] == [4, 1] and plan(2048, 16, f32, 5)[1] == 0      # exact mode at d = 2048: X of 4 rows fills the LDS
    assert plan(1024, 16, bf, 16)[:6] == [16, 1, 1, 4, 1, 1]         # giga330M: 16 rows x 8 KB still one piece
    assert plan(2048, 16, bf, 32)[1] == 2
    assert plan(2048, 16, bf, 1)[8:10] == [1, 1] and plan(2048, 16, bf, 8)[10] == 1              # round 5: finished row + paired QKV at one row, paired FFN-down at 8
    # round 6: the paired QKV consumer of 2..8 finished rows - up to 6 rows at d = 2048 (8 rows measured slower), up to 8 below; not where the
    # centred copy's rows are not whole 1 KB requests (d = 256 in bf16, d = 768)
    assert [plan(2048, 16, bf, r)[9] for r in (2, 6, 7, 8, 9)] == [2, 2, 0, 0, -1]
    assert plan(1024, 16, bf, 8)[9] == 2 and plan(512, 16, f32, 3)[9] == 2 and plan(256, 4, bf, 4)[9] == 0 and plan(768, 12, bf, 4)[9] == 0
    assert plan(2048, 16, bf, 64)[11:16] == [1, 4, 4, 8, 4]        # round 6: wide decode on rows_gemm_wd_k, 4 K slices for both producers (256 workgroups)
    assert plan(1024, 16, bf, 32)[11:16] == [1, 4, 4, 4, 4]
