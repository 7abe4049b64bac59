from . import transducer  # noqa: F401
