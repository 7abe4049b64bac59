"""speech.loader -> speech_amd.loader (same names as /root/reference/speech/loader.py)."""
from speech_amd.loader import *  # noqa: F401,F403
from speech_amd.loader import Preprocessor, AudioDataset, BatchRandomSampler, make_loader, log_specgram  # noqa: F401
