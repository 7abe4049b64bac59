"""transducer.decoders.decode_static -> speech_amd.transducer.decode_static (sa_transducer_decode_static)."""
from speech_amd.transducer import decode_static  # noqa: F401
