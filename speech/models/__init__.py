"""speech.models -> speech_amd.models (Model, CTC, Transducer, Seq2Seq)."""
from speech_amd.models import Model, CTC, Transducer, Seq2Seq, LinearND, zero_pad_concat  # noqa: F401
from speech_amd import decoder as ctc_decoder  # noqa: F401
from speech_amd.io import save, load, compute_cer  # noqa: F401
