"""speech.models -> speech_amd.models (Model, CTC, Transducer, Seq2Seq) with the reference's submodule paths."""
from speech_amd.models import Model, CTC, Transducer, Seq2Seq, LinearND, zero_pad_concat  # noqa: F401
from . import model, ctc_model, ctc_decoder, seq2seq, transducer_model  # noqa: F401
