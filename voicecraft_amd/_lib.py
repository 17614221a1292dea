"""ctypes binding of libvcengine.so (include/vc_engine.h).  No fallback: if the HIP library is
missing or fails to load, importing code gets a RuntimeError telling how to build it."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "libvcengine.so"

VC_DTYPE_F32, VC_DTYPE_BF16, VC_DTYPE_I64 = 0, 1, 2
VC_MAX_SILENCE = 8


class ModelCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "d_model", "nhead", "num_layers", "n_codebooks", "audio_vocab_size", "n_special", "text_rows",
        "head_hidden", "empty_token", "eog", "audio_pad_token", "eos", "reduced_eog", "encodec_sr",
        "max_n_spans", "max_seqs", "max_positions")]


class SampleCfg(C.Structure):
    _fields_ = [
        ("top_k", C.c_int32), ("top_p", C.c_float), ("temperature", C.c_float),
        ("stop_repetition", C.c_int32), ("n_silence", C.c_int32),
        ("silence_tokens", C.c_int32 * VC_MAX_SILENCE), ("seed", C.c_uint64),
        ("use_graph", C.c_int32), ("poll_every", C.c_int32), ("forced_mode", C.c_int32),
    ]


# name -> (restype, argtypes); the single source of truth the symbol test checks against the header
PROTOTYPES = {
    "vc_create": (C.c_int, [C.POINTER(ModelCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "vc_destroy": (None, [C.c_void_p]),
    "vc_last_error": (C.c_char_p, [C.c_void_p]),
    "vc_version": (C.c_char_p, []),
    "vc_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p]),
    "vc_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "vc_finalize_weights": (C.c_int, [C.c_void_p, C.c_int]),
    "vc_tts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(SampleCfg), C.c_int,
                         C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p, C.c_int,
                         C.POINTER(C.c_int), C.c_void_p]),
    "vc_tts_multi": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32),
                               C.POINTER(SampleCfg), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int),
                               C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "vc_eval_forward": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p, C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                  C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "vc_eval_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int32), C.c_int64]),
    "vc_debug_sample": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(SampleCfg), C.c_int, C.c_void_p, C.c_void_p]),
    "vc_edit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_int,
                          C.POINTER(C.c_int32), C.POINTER(SampleCfg), C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                          C.POINTER(C.c_int), C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_void_p]),
    "vc_pattern_shift": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vc_pattern_revert": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "vc_pattern_unshift": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "vc_debug_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32)]),
    "vc_box_probe": (C.c_int, [C.c_longlong, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "vc_debug_read": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "vc_last_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "vc_bench_kernel": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_float),
                                  C.POINTER(C.c_double), C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libvcengine.so once; raises RuntimeError (never falls back) when it is unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("VC_ENGINE_LIB", LIB_PATH))
    if not path.exists():
        raise RuntimeError(
            f"{path} not found: the HIP engine is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `python voicecraft_amd/build.py`). There is no CPU fallback.")
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # pragma: no cover
        raise RuntimeError(f"failed to load {path}: {e}. There is no CPU fallback.") from e
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift, let it surface
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class EngineError(RuntimeError):
    pass


def check(rc: int, handle=None, what: str = "") -> None:
    if rc == 0:
        return
    msg = load().vc_last_error(handle)
    text = msg.decode() if msg else ""
    if rc == -1:
        # the reference signals bad inputs with AssertionError (SURVEY.md §8b "Errors")
        raise AssertionError(f"{what}: {text}")
    raise EngineError(f"{what} failed (code {rc}): {text}")
