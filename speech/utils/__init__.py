from speech_amd.io import save, load, compute_cer  # noqa: F401
from . import io, score, wave  # noqa: F401
