"""speech_amd.decoder -- device-side CTC decoding behind the reference's decoder API.

  beam_decode   <- /root/reference/speech/models/ctc_decoder.py:38-113  decode(probs, beam_size, blank), batched:
                   CTC.infer (ctc_model.py:55-60) loops `decode(p, beam_size=1, blank=self.blank)[0]` over the batch
                   on the host; here the whole batch is one kernel launch and only the label lists come back.
  decode        <- the same single-utterance signature the reference exports (returns (labels tuple, nll)).
  greedy_decode <- np.argmax + CTC.max_decode (ctc_model.py:62-70).
All compute is libspeech_amd.so (sa_ctc_beam_decode / sa_ctc_greedy_decode); inputs must be GPU tensors."""
import numpy as np
import torch

from . import _lib


def _prep(x, lengths):
    _lib.require_cuda(x, "decoder input")
    if x.dtype != torch.float32 or x.dim() != 3:
        raise _lib.SpeechAmdError("decoder input must be a float32 (B, T, S) tensor")
    x = x.detach()
    if x.stride(2) != 1:
        x = x.contiguous()
    B, T, S = x.shape
    if lengths is None:
        lens = torch.full((B,), T, dtype=torch.int32, device=x.device)  # the reference decodes the full padded T'
    else:
        lens = torch.as_tensor(lengths, dtype=torch.int32).to(x.device)
        if lens.numel() != B or int(lens.max()) > T or int(lens.min()) < 0:
            raise _lib.SpeechAmdError("bad lengths")
    return x, lens, B, T, S


def _to_lists(labels, lens):
    """(B, T) int32 labels + (B,) lengths on the GPU -> list of tuples: ONE device-to-host copy (the lengths ride in an
    extra column), then NumPy slicing (no per-label Python loop)."""
    both = torch.cat([lens.view(-1, 1), labels], dim=1).cpu().numpy()
    return [tuple(both[b, 1:1 + both[b, 0]].tolist()) for b in range(both.shape[0])]


def beam_decode(x, beam_size=10, blank=0, input_is_logits=False, lengths=None):
    """x: (B, T, S) probabilities (post-softmax, as the reference's decode takes) or raw logits.
    Returns (list of label tuples, nll tensor (B,) on the GPU)."""
    x, lens, B, T, S = _prep(x, lengths)
    if not 0 <= blank < S:
        raise _lib.SpeechAmdError("blank out of range")
    L = _lib.lib()
    out_labels = torch.empty(B, T, dtype=torch.int32, device=x.device)
    out_lens = torch.empty(B, dtype=torch.int32, device=x.device)
    nll = torch.empty(B, dtype=torch.float32, device=x.device)
    ws = _lib.WORKSPACE.get(L.sa_ctc_beam_workspace_bytes(T, S, B, beam_size), x.device, "beam")
    _lib.check(L.sa_ctc_beam_decode(_lib.ptr(x), x.stride(1), x.stride(0), _lib.ptr(lens), S, B, T, beam_size, blank,
                                    int(input_is_logits), _lib.ptr(out_labels), _lib.ptr(out_lens), _lib.ptr(nll),
                                    _lib.ptr(ws), ws.numel(), _lib.cur_stream()), "sa_ctc_beam_decode")
    return _to_lists(out_labels, out_lens), nll


def decode(probs, beam_size=10, blank=0):
    """Single-utterance face with the reference's exact signature: probs (T, S) -> (labels tuple, nll float)."""
    p = probs if torch.is_tensor(probs) else torch.from_numpy(np.ascontiguousarray(probs, dtype=np.float32))
    if not p.is_cuda:
        p = p.cuda()
    labels, nll = beam_decode(p.float().unsqueeze(0), beam_size=beam_size, blank=blank)
    return labels[0], float(nll[0])


def greedy_decode(x, blank, lengths=None):
    """argmax over classes per frame, then drop repeats and blanks.  Returns a list of label tuples."""
    x, lens, B, T, S = _prep(x, lengths)
    out_labels = torch.empty(B, T, dtype=torch.int32, device=x.device)
    out_lens = torch.empty(B, dtype=torch.int32, device=x.device)
    _lib.check(_lib.lib().sa_ctc_greedy_decode(_lib.ptr(x), x.stride(1), x.stride(0), _lib.ptr(lens), S, B, T, blank,
                                               _lib.ptr(out_labels), _lib.ptr(out_lens), _lib.cur_stream()),
               "sa_ctc_greedy_decode")
    return _to_lists(out_labels, out_lens)
