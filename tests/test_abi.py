"""The C-ABI boundary without a GPU: the shared library loads, exports every symbol that
include/*.h declares with the declared arity, and refuses loudly to run when there is no device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    out = {}
    inc = os.path.join(ROOT, "include")
    for fn in sorted(os.listdir(inc)):
        src = open(os.path.join(inc, fn)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(?:int|void|const char\s*\*)\s*(vc_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
            args = m.group(2).strip()
            n = 0 if args in ("", "void") else args.count(",") + 1
            out[m.group(1)] = n
    return out


@pytest.fixture(scope="module")
def lib():
    from voicecraft_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from voicecraft_amd import _lib, codec
    decl = _declared()
    assert len(decl) >= 15
    protos = dict(_lib.PROTOTYPES)
    protos.update(codec.PROTOTYPES)
    for name, nargs in decl.items():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported by libvcengine.so"
        assert name in protos, f"{name} has no ctypes prototype"
        assert len(protos[name][1]) == nargs, f"{name}: header has {nargs} parameters, binding has {len(protos[name][1])}"
    for name in protos:
        assert name in decl, f"{name} bound in Python but not declared in include/"


def test_struct_layout_matches_header():
    from voicecraft_amd._lib import ModelCfg, SampleCfg
    assert C.sizeof(ModelCfg) == 17 * 4
    # int32 top_k, float top_p, float temperature, int32 stop_repetition, int32 n_silence, int32[8], (pad) u64 seed,
    # 3x int32 (use_graph, poll_every, forced_mode) + tail padding to the 8-byte alignment of the struct
    assert SampleCfg.seed.offset % 8 == 0 and C.sizeof(SampleCfg) == SampleCfg.seed.offset + 8 + 16
    assert SampleCfg.forced_mode.offset == SampleCfg.seed.offset + 16


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from voicecraft_amd import synth
    from voicecraft_amd._lib import ModelCfg
    a = synth.make_args("tiny")
    cfg = ModelCfg(d_model=a.d_model, nhead=a.nhead, num_layers=a.num_decoder_layers, n_codebooks=4,
                   audio_vocab_size=2048, n_special=4, text_rows=101, head_hidden=1024, empty_token=2048, eog=2049,
                   audio_pad_token=2050, eos=2051, reduced_eog=1, encodec_sr=50, max_n_spans=3, max_seqs=2, max_positions=256)
    h = C.c_void_p()
    rc = lib.vc_create(C.byref(cfg), 0, C.byref(h))
    assert rc == -3 and not h.value                      # VC_EHIP, no engine
    assert b"hipSetDevice" in lib.vc_last_error(None)
    from voicecraft_amd.engine import VoiceCraftEngine
    with pytest.raises(RuntimeError):
        VoiceCraftEngine(a, {}, device="cpu")


def test_bad_config_is_rejected(lib):
    from voicecraft_amd._lib import ModelCfg
    cfg = ModelCfg(d_model=300, nhead=4, num_layers=1, n_codebooks=4, audio_vocab_size=2048, n_special=4, text_rows=101,
                   head_hidden=1024, empty_token=2048, eog=2049, audio_pad_token=2050, eos=2051, reduced_eog=1,
                   encodec_sr=50, max_n_spans=3, max_seqs=1, max_positions=256)
    h = C.c_void_p()
    assert lib.vc_create(C.byref(cfg), 0, C.byref(h)) == -1
    assert b"d_model" in lib.vc_last_error(None)


def test_pattern_entry_points_validate_arguments(lib):
    assert lib.vc_pattern_shift(None, 1, 4, 3, 0, None, None) == -1
    assert lib.vc_pattern_unshift(None, 2, 4, None, None) == -1      # N < K
    assert lib.vc_pattern_unshift(None, 4, 4, None, None) == 0       # N == K: nothing to write
