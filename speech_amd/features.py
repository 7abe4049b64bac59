"""speech_amd.features -- on-device log spectrogram, the GPU counterpart of speech.loader.log_specgram
(/root/reference/speech/loader.py:156-166) and of Preprocessor.preprocess's normalisation (loader.py:65-69).

    feats = log_specgram(audio_int16, sample_rate)                       # (frames, nperseg // 2 + 1) on the GPU
    feats = log_specgram(audio_int16, sample_rate, mean=m, std=s)        # z-normalised, ready for CTC.forward_impl

All compute is libspeech_amd.so (sa_log_specgram: int16 -> fp32, overlapping-frame view x windowed-DFT MFMA GEMM,
power / scipy 'density' scaling / log / normalise epilogue)."""
import numpy as np
import torch

from . import _lib

_DFT = {}


def _dft_table(nperseg, device):
    key = (nperseg, str(device))
    tab = _DFT.get(key)
    if tab is None:
        nbins = nperseg // 2 + 1
        tab = torch.empty(nperseg, 2 * nbins, dtype=torch.float32, device=device)
        _lib.check(_lib.lib().sa_specgram_build_dft(_lib.ptr(tab), nperseg, _lib.cur_stream()), "sa_specgram_build_dft")
        _DFT[key] = tab
    return tab


def log_specgram(audio, sample_rate, window_size=20, step_size=10, eps=1e-10, mean=None, std=None, device=None):
    """audio: int16 samples (numpy array or tensor, host or device).  Same arguments as the reference's log_specgram;
    returns a float32 (frames, bins) CUDA tensor."""
    if not torch.is_tensor(audio):
        audio = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.int16))
    if audio.dtype != torch.int16 or audio.dim() != 1:
        raise _lib.SpeechAmdError("audio must be a 1-D int16 signal")
    if not audio.is_cuda:
        audio = audio.to(device or "cuda")
    dev = audio.device
    nperseg = int(window_size * sample_rate / 1e3)
    noverlap = int(step_size * sample_rate / 1e3)
    hop = nperseg - noverlap
    L = _lib.lib()
    n = audio.numel()
    frames = L.sa_specgram_frames(n, nperseg, hop)
    if frames <= 0:
        raise _lib.SpeechAmdError("signal shorter than one window")
    nbins = nperseg // 2 + 1
    tab = _dft_table(nperseg, dev)

    def stat(v):
        if v is None:
            return None
        t = torch.as_tensor(np.asarray(v, dtype=np.float32)) if not torch.is_tensor(v) else v.float()
        return t.to(dev).contiguous()

    m, s = stat(mean), stat(std)
    out = torch.empty(frames, nbins, dtype=torch.float32, device=dev)
    ws = _lib.WORKSPACE.get(L.sa_log_specgram_workspace_bytes(n, nperseg, hop), dev, "specgram")
    _lib.check(L.sa_log_specgram(_lib.ptr(audio), n, int(sample_rate), nperseg, hop, _lib.ptr(tab), _lib.ptr(m),
                                 _lib.ptr(s), eps, _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.cur_stream()),
               "sa_log_specgram")
    return out
