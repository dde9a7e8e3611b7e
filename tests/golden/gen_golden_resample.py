"""Fixture F18: sample-rate conversion as librosa.resample(..., res_type="polyphase") performs it, i.e. scipy.signal.resample_poly
(scipy is importable in the build container; librosa and its default "soxr_hq" backend are not).  Inputs are seeded noise +
a few sinusoids; outputs are scipy's, float32.    python tests/golden/gen_golden_resample.py"""
import math
import os

import numpy as np
import scipy
import scipy.signal

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [(44100, 22050, 4099), (48000, 22050, 6001), (16000, 22050, 2500), (8000, 22050, 1203), (32000, 22050, 3333), (22050, 22050, 64)]


def signal(n, sr, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    return (0.3 * rng.standard_normal(n) + np.sin(2 * np.pi * 440.0 * t) + 0.5 * np.sin(2 * np.pi * 1999.0 * t + 0.3)).astype(np.float32)


def main():
    out = {"scipy_version": np.array(scipy.__version__)}
    for k, (o, t, n) in enumerate(CASES):
        x = signal(n, o, 100 + k)
        g = math.gcd(o, t)
        up, down = t // g, o // g
        y = x if up == down else scipy.signal.resample_poly(x.astype(np.float64), up, down)
        out[f"c{k}_meta"] = np.array([o, t, n])
        out[f"c{k}_x"] = x
        out[f"c{k}_y"] = np.asarray(y, dtype=np.float32)
    np.savez(os.path.join(HERE, "f18_resample.npz"), **out)
    print("wrote f18_resample.npz", {k: v.shape for k, v in out.items() if k.endswith("_y")})


if __name__ == "__main__":
    main()
