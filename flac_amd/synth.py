"""Deterministic synthetic PCM used by the benchmark and the parity tests.

Three detuned tones plus AR(2)-coloured noise per channel; channel 1 is correlated with
channel 0 so that mid/side coding has something to find, further channels are independent.
numpy + scipy only; the same seed always gives the same samples.
"""
import numpy as np
from scipy.signal import lfilter


def synth_pcm(seconds=None, sample_rate=44100, channels=2, bits_per_sample=16, seed=1234,
              wide_samples=None):
    """Return interleaved int32 PCM of shape (wide_samples, channels)."""
    if wide_samples is None:
        wide_samples = int(round(seconds * sample_rate))
    n = np.arange(wide_samples, dtype=np.float64)
    t = n / float(sample_rate)
    rng = np.random.default_rng(seed)

    def tone(phi):
        return (0.35 * np.sin(2 * np.pi * 220.0 * t + phi)
                + 0.20 * np.sin(2 * np.pi * 1337.7 * t)
                + 0.10 * np.sin(2 * np.pi * 5003.1 * t * (1 + 1e-4 * np.sin(2 * np.pi * 0.3 * t))))

    def colour():
        w = rng.normal(0.0, 0.02, wide_samples)
        return lfilter([1.0], [1.0, -1.6, 0.8], w) * 0.2

    chans = []
    ch0 = tone(0.0) + colour()
    chans.append(ch0)
    if channels > 1:
        chans.append(0.8 * ch0 + 0.2 * (tone(1.0) + colour()))
    for k in range(2, channels):
        chans.append(tone(float(k)) + colour())
    x = np.stack(chans, axis=1)
    scale = 20000.0 * (2.0 ** (bits_per_sample - 16))
    lim = 2 ** (bits_per_sample - 1)
    pcm = np.clip(np.trunc(x * scale), -lim, lim - 1).astype(np.int32)
    return np.ascontiguousarray(pcm)


def noise_pcm(wide_samples, channels, bits_per_sample, seed, full_scale=True):
    """Uniform white noise, the worst case for the predictors (mostly verbatim subframes)."""
    rng = np.random.default_rng(seed)
    lim = 2 ** (bits_per_sample - 1)
    if not full_scale:
        lim //= 64
    return rng.integers(-lim, lim, size=(wide_samples, channels), dtype=np.int64).astype(np.int32)
