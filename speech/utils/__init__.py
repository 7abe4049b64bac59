from speech_amd import io  # noqa: F401
from speech_amd.io import save, load, compute_cer  # noqa: F401
