"""speech.utils.score -> speech_amd.io.compute_cer (/root/reference/speech/utils/score.py)."""
from speech_amd.io import compute_cer  # noqa: F401
