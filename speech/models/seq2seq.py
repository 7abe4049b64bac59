"""speech.models.seq2seq -> speech_amd.models (/root/reference/speech/models/seq2seq.py)."""
from speech_amd.models import Seq2Seq, NNAttention, end_pad_concat  # noqa: F401
from . import model  # noqa: F401
