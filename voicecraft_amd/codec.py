"""EnCodec tokenizer boundary (data/tokenizer.py:101-133 of the reference) — HIP implementation pending.

PROTOTYPES lists the ctypes bindings of include/vc_codec.h (none yet)."""
PROTOTYPES: dict = {}


class AudioTokenizer:  # pragma: no cover - placeholder until the conv/LSTM/RVQ kernels land
    def __init__(self, *a, **k):
        raise NotImplementedError("EnCodec HIP kernels are not built yet; there is no CPU fallback")
