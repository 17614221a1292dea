"""Delayed-codebook pattern — CPU oracle (numpy / pure Python).  TEST INFRASTRUCTURE ONLY.

Restates models/codebooks_patterns.py of the reference:
  * `delayed_layout`            <- DelayedPatternProvider.get_pattern            (:336-352)
  * `build_sequence_from_layout`<- Pattern._build_pattern_sequence_scatter_indexes + build_pattern_sequence (:117-176)
  * `revert_sequence_from_layout`<- _build_reverted_sequence_scatter_indexes + revert_pattern_sequence (:177-245)
and gives the closed forms the HIP kernels implement (`delayed_shift`, `delayed_revert`,
`unshift_span`), which the tests prove equal to the layout-driven versions.
"""
from __future__ import annotations

import numpy as np


def delayed_layout(T: int, K: int) -> list[list[tuple[int, int]]]:
    """Layout for delays = [0..K-1]: step 0 is empty, step s>=1 lists the (t, q) it holds."""
    layout: list[list[tuple[int, int]]] = [[]]
    for t in range(T + K - 1):
        step = []
        for q in range(K):
            tq = t - q
            if tq >= 0:
                step.append((tq, q))
        layout.append(step)
    return layout


def build_sequence_from_layout(z: np.ndarray, special: int) -> np.ndarray:
    """z [B,K,T] -> [B,K,T+K] by walking the layout (keep_only_valid_steps=False)."""
    B, K, T = z.shape
    layout = delayed_layout(T, K)
    idx = np.full((K, len(layout)), K * T, dtype=np.int64)      # K*T = slot of the special token
    for s, coords in enumerate(layout):
        for (t, q) in coords:
            if t < T:
                idx[q, s] = t + q * T
    flat = np.concatenate([z.reshape(B, -1), np.full((B, 1), special, dtype=z.dtype)], axis=1)
    return flat[:, idx.reshape(-1)].reshape(B, K, len(layout))


def revert_sequence_from_layout(s: np.ndarray, T: int, special: int) -> np.ndarray:
    """s [B,K,S] (S <= T+K) -> [B,K,T]."""
    B, K, S = s.shape
    layout = delayed_layout(T, K)
    assert S <= len(layout)
    idx = np.full((K, T), K * S, dtype=np.int64)
    for step, coords in enumerate(layout):
        if step < S:
            for (t, q) in coords:
                if t < T:
                    idx[q, t] = step + q * S
    flat = np.concatenate([s.reshape(B, -1), np.full((B, 1), special, dtype=s.dtype)], axis=1)
    return flat[:, idx.reshape(-1)].reshape(B, K, T)


# ---------------------------------------------------------------- closed forms (what the kernels do)
def delayed_shift(z: np.ndarray, special: int) -> np.ndarray:
    """out[b,q,s] = z[b,q,s-1-q] if 0 <= s-1-q < T else special;  [B,K,T] -> [B,K,T+K]."""
    B, K, T = z.shape
    out = np.full((B, K, T + K), special, dtype=z.dtype)
    for q in range(K):
        out[:, q, 1 + q: 1 + q + T] = z[:, q, :]
    return out


def delayed_revert(s: np.ndarray, T: int, special: int) -> np.ndarray:
    """out[b,q,t] = s[b,q,t+1+q] if t+1+q < S else special;  [B,K,S] -> [B,K,T]."""
    B, K, S = s.shape
    out = np.full((B, K, T), special, dtype=s.dtype)
    for q in range(K):
        n = max(0, min(T, S - 1 - q))
        out[:, q, :n] = s[:, q, 1 + q: 1 + q + n]
    return out


def unshift_span(span: np.ndarray) -> np.ndarray:
    """span [N,K] (one row per decode step) -> [K,N-K]; row j = span[j : N-(K-j), j]
    (models/voicecraft.py:1125-1139)."""
    N, K = span.shape
    return np.stack([span[j: N - (K - j), j] for j in range(K)], axis=0)
