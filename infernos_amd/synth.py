"""Synthetic call audio used by benchmarks and tests: a gated harmonic
tone plus noise, 1 s of leading/trailing silence, mu-law framed into 20 ms packets."""
import numpy as np


def synth_utterance(seed: int, seconds: float = 10.0, sr: int = 8000) -> np.ndarray:
    rng = np.random.default_rng(seed)
    n = int(seconds * sr)
    t = np.arange(n) / sr
    f0 = rng.uniform(100, 300)
    env = 0.5 - 0.5 * np.cos(2 * np.pi * 4.0 * t)
    env[(t < 1.0) | (t > seconds - 1.0)] = 0.0
    x = 0.3 * env * sum(a * np.sin(2 * np.pi * k * f0 * t) for k, a in ((1, 1.0), (2, 0.5), (3, 0.25)))
    x = x + 0.01 * rng.standard_normal(n)
    return x.astype(np.float32)
