"""transducer.functions.transducer.TransducerLoss -> speech_amd.transducer.TransducerLoss (sa_transducer_loss)."""
from speech_amd.transducer import TransducerLoss  # noqa: F401
