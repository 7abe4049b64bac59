"""speech.models.ctc_decoder -> speech_amd.decoder (/root/reference/speech/models/ctc_decoder.py: decode)."""
from speech_amd.decoder import *  # noqa: F401,F403
from speech_amd.decoder import decode  # noqa: F401
