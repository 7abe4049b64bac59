"""speech.utils.wave -> speech_amd.loader (/root/reference/speech/utils/wave.py: array_from_wave, wav_duration)."""
from speech_amd.loader import array_from_wave, wav_duration  # noqa: F401
