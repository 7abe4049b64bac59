"""speech_amd -- MI355X (gfx950) native CTC hot path of awni/speech behind the reference's own Python API.

Layout: csrc/ (HIP kernels + the C ABI of include/speech_amd.h), _lib.py (ctypes loader, no fallback),
ctc.py (functions.ctc.CTCLoss), and the host-side mirrors of speech.models / speech.loader."""
__version__ = "0.1.0"
