"""
oracle/features_ref.py -- NumPy restatement of the reference's featuriser.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/speech/loader.py:156-166 (log_specgram: scipy.signal.spectrogram with a periodic Hann window
of nperseg = window_size ms, noverlap = step_size ms, detrend=False, default one-sided 'density' PSD scaling; then
log(spec.T + eps)) and loader.py:65-69 (z-normalisation) with explicit framing / rFFT / scaling in float64.
Pinned against the live reference function by oracle/gen_golden.py -> tests/golden/specgram.npz.
"""
import numpy as np


def log_specgram(audio, sample_rate, window_size=20, step_size=10, eps=1e-10):
    audio = np.asarray(audio, dtype=np.float64)
    nperseg = int(window_size * sample_rate / 1e3)
    noverlap = int(step_size * sample_rate / 1e3)
    hop = nperseg - noverlap
    frames = 1 + (len(audio) - nperseg) // hop
    idx = np.arange(nperseg)[None, :] + hop * np.arange(frames)[:, None]
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(nperseg) / nperseg)  # periodic Hann
    spec = np.abs(np.fft.rfft(audio[idx] * win, axis=1)) ** 2
    spec /= sample_rate * (win ** 2).sum()
    if nperseg % 2 == 0:
        spec[:, 1:-1] *= 2.0
    else:
        spec[:, 1:] *= 2.0
    return np.log(spec + eps)


def normalise(feats, mean, std):
    return (feats - mean) / std
