"""speech.models.model -> speech_amd.models (the reference's module path: /root/reference/speech/models/model.py;
part of its whole-module checkpoint format, speech/utils/io.py:15-19)."""
from speech_amd.models import Model, LinearND, zero_pad_concat  # noqa: F401
