"""speech.utils.io -> speech_amd.io (/root/reference/speech/utils/io.py: get_names, save, load)."""
from speech_amd.io import MODEL, PREPROC, get_names, save, load  # noqa: F401
