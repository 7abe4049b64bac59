"""`transducer` import shim: the reference imports `transducer.decoders` and `transducer.functions.transducer`
(/root/reference/speech/models/transducer_model.py:10-11) from the un-vendored github.com/awni/transducer checkout on
PYTHONPATH (setup.sh); this package provides the same names on the HIP library (speech_amd.transducer)."""
from . import decoders, functions  # noqa: F401
