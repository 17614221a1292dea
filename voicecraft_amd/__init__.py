"""voicecraft_amd — MI355X-native engine for VoiceCraft's token-infilling decode path.

Public surface (mirrors the reference's model interface):
    VoiceCraftEngine.inference_tts / inference_tts_batch / inference
    pattern_shift / pattern_revert / pattern_unshift   (delayed-codebook pattern, bit-exact)
    AudioTokenizer                                     (EnCodec encode/decode)
Everything computes in libvcengine.so (HIP, gfx950); importing this package does not need a GPU,
constructing an engine does.
"""
from .synth import PRESETS, make_args, make_state_dict, random_prompt  # noqa: F401


def __getattr__(name):
    if name in ("VoiceCraftEngine", "pattern_shift", "pattern_revert", "pattern_unshift"):
        from . import engine
        return getattr(engine, name)
    if name == "AudioTokenizer":
        from . import codec
        return codec.AudioTokenizer
    raise AttributeError(name)
