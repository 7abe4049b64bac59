"""Import-path shim: `import speech` resolves to the MI355X implementation (speech_amd) with the reference's names
(/root/reference/speech/__init__.py: save, load, compute_cer)."""
from speech_amd.io import save, load, compute_cer  # noqa: F401
from . import loader, models  # noqa: F401
