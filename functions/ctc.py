"""functions.ctc -> speech_amd.ctc: the module path the reference imports the warp-ctc binding from
(/root/reference/speech/models/ctc_model.py:9)."""
from speech_amd.ctc import CTCLoss, CTCLabels, ctc_loss_raw  # noqa: F401
