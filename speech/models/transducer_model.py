"""speech.models.transducer_model -> speech_amd.models.Transducer (/root/reference/speech/models/transducer_model.py)."""
from speech_amd.models import Transducer  # noqa: F401
from . import model  # noqa: F401
