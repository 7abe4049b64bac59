"""speech.models.ctc_model -> speech_amd.models.CTC (/root/reference/speech/models/ctc_model.py)."""
from speech_amd.models import CTC  # noqa: F401
from . import model  # noqa: F401
