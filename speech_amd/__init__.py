"""speech_amd -- MI355X (gfx950) native CTC hot path of awni/speech behind the reference's own Python API.

Layout: csrc/ (HIP kernels + the C ABI of include/speech_amd.h), _lib.py (ctypes loader, no fallback),
ops.py (tensor wrappers), ctc.py (functions.ctc.CTCLoss), decoder.py (ctc_decoder), encoder.py + models.py
(speech.models.Model / CTC), loader.py, io.py (speech.loader / save / load / compute_cer), dist.py (RCCL data parallel).

The top-level `speech/` and `functions/` packages re-export these under the reference's import paths."""
__version__ = "0.1.0"
