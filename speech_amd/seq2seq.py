"""speech_amd.seq2seq -- the attention decoder of /root/reference/speech/models/seq2seq.py on the HIP ops.

DecoderFunction is ONE autograd node for Seq2Seq.decode (:77-112): the teacher-forced (or scheduled-sampled) loop over
output tokens -- embedding lookup (+ previous context), nn.GRUCell, NNAttention, fc -- with an explicit backward
through the tokens (BPTT) that writes every parameter gradient straight into its slot.  The loop lives in the library
(sa_s2s_decoder_fwd / _bwd): per token an index pick, an embedding gather (+ context), ONE launch for both GRU
projections (MFMA "skinny" product), the gate kernel and two attention kernels; everything that does not depend on the
loop (the fc over all tokens, every weight gradient, the embedding gradient) is one batched product after it.
`step` is the same arithmetic for one token without autograd (Seq2Seq.decode_step, :114-138).
All compute is the HIP library; a CPU tensor raises.
"""
import ctypes
import math
import random

import torch

from . import _lib, ops


def _L():
    return _lib.lib()


def embedding_rows(table, idx, out):
    idx = idx.contiguous()  # a column slice y[:, t] is strided; the kernel reads a packed int64 vector
    _lib.check(_L().sa_embedding_fwd(_lib.ptr(table), _lib.ptr(idx), _lib.ptr(out), idx.numel(), table.shape[1],
                                     _lib.cur_stream()), "sa_embedding_fwd")
    return out


def argmax_rows(x):
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    _lib.check(_L().sa_argmax_rows(_lib.ptr(x), _lib.ptr(out), x.shape[0], x.shape[1], _lib.cur_stream()),
               "sa_argmax_rows")
    return out


def gru_cell(ix, hprev, w_ih, w_hh, b_ih, b_hh, stash=None):
    """nn.GRUCell forward (seq2seq.py:97): two projections + the gate kernel."""
    B, H = hprev.shape
    gi = ops.gemm(ix, w_ih, trans_b=True, bias=b_ih)
    gh = ops.gemm(hprev, w_hh, trans_b=True, bias=b_hh)
    hx = torch.empty(B, H, dtype=torch.float32, device=ix.device)
    _lib.check(_L().sa_grucell_gates_fwd(_lib.ptr(gi), _lib.ptr(gh), _lib.ptr(hprev), _lib.ptr(hx), _lib.ptr(stash), B,
                                         H, _lib.cur_stream()), "sa_grucell_gates_fwd")
    return hx


def _att_ws(B, T, H, KS, device):
    n = _L().sa_attention_workspace_bytes(B, T, H, KS)
    if n == 0:
        raise _lib.SpeechAmdError("attention: unsupported shape (odd location kernel <= 15 taps)")
    return _lib.WORKSPACE.get(n, device, "attention")


def attention(eh, ox, ax_prev, conv_w, conv_b, nn_w, nn_b, log_t):
    """NNAttention.forward (seq2seq.py:341-360): returns (sx (B, H), ax (B, T))."""
    B, T, H = eh.shape
    KS = conv_w.shape[-1]
    ax = torch.empty(B, T, dtype=torch.float32, device=eh.device)
    sx = torch.empty(B, H, dtype=torch.float32, device=eh.device)
    scale = math.log(T) if log_t else 1.0
    ws = _att_ws(B, T, H, KS, eh.device)
    _lib.check(_L().sa_attention_fwd(_lib.ptr(eh), _lib.ptr(ox), _lib.ptr(ax_prev), _lib.ptr(conv_w), _lib.ptr(conv_b),
                                     _lib.ptr(nn_w), _lib.ptr(nn_b), scale, _lib.ptr(ax), _lib.ptr(sx), B, T, H, KS,
                                     _lib.ptr(ws), ws.numel(), _lib.cur_stream()), "sa_attention_fwd")
    return sx, ax


def step(eh, idx, state, P, log_t):
    """One decoder token without autograd.  idx (B,) int64; state None or (hx, ax, sx).  P: parameter dict.
    Returns (logits (B, V-1), (hx, ax, sx))."""
    B, T, H = eh.shape
    E, K, KS = P["emb"].shape[1], P["fc_w"].shape[0], P["conv_w"].shape[-1]
    L = _L()
    nbytes = L.sa_s2s_decoder_workspace_bytes(B, T, 1, H, E, KS, K)
    if nbytes:
        # the whole token in the library: 7 launches, both GRU projections and the fc on the small-M MFMA kernel
        f = dict(dtype=torch.float32, device=eh.device)
        hx, ax, sx = torch.empty(B, H, **f), torch.empty(B, T, **f), torch.empty(B, H, **f)
        out = torch.empty(B, K, **f)
        ws = _lib.WORKSPACE.get(nbytes, eh.device, "s2s_decoder")
        hp, ap, sp = state if state is not None else (None, None, None)
        plist = [P[n].contiguous() for n in _NAMES]
        _lib.check(L.sa_s2s_decoder_step(_lib.ptr(eh.contiguous()), _lib.ptr(idx.contiguous()), _lib.ptr(hp),
                                         _lib.ptr(ap), _lib.ptr(sp), _ptr_array(plist), B, T, H, E, KS, K,
                                         math.log(T) if log_t else 1.0, _lib.ptr(hx), _lib.ptr(ax), _lib.ptr(sx),
                                         _lib.ptr(out), _lib.ptr(ws), ws.numel(), _lib.cur_stream()),
                   "sa_s2s_decoder_step")
        return out, (hx, ax, sx)
    ix = torch.empty(B, E, dtype=torch.float32, device=eh.device)
    embedding_rows(P["emb"], idx, ix)
    if state is None:
        hprev, ax_prev = torch.zeros(B, H, dtype=torch.float32, device=eh.device), None
    else:
        hprev, ax_prev, sx_prev = state
        ix = ops.add_rows(ix, sx_prev)
    hx = gru_cell(ix, hprev, P["w_ih"], P["w_hh"], P["b_ih"], P["b_hh"])
    sx, ax = attention(eh, hx, ax_prev, P["conv_w"], P["conv_b"], P["nn_w"], P["nn_b"], log_t)
    out = ops.gemm(ops.add_rows(hx, sx), P["fc_w"], trans_b=True, bias=P["fc_b"])
    return out, (hx, ax, sx)


def greedy_decode(eh, y0, P, log_t, end_tok, max_len, check_every=8):
    """Seq2Seq.infer_decode's token matrix (seq2seq.py:140-158) for a batch, the whole loop in the library
    (sa_s2s_greedy_decode): eh (B, T, H), y0 (B,) start tokens.  Returns an int64 CPU tensor (B, steps + 1)."""
    B, T, H = eh.shape
    E, K, KS = P["emb"].shape[1], P["fc_w"].shape[0], P["conv_w"].shape[-1]
    L = _L()
    nbytes = L.sa_s2s_greedy_workspace_bytes(B, T, H, E, KS, K, max_len)
    if nbytes == 0:
        raise _lib.SpeechAmdError("Seq2Seq greedy decode: unsupported shape")
    ws = _lib.WORKSPACE.get(nbytes, eh.device, "s2s_greedy")
    rec = torch.zeros(B * (max_len + 1) + 1, dtype=torch.int64, device=eh.device)  # tokens | steps (low int32)
    tokens = rec[:B * (max_len + 1)].view(B, max_len + 1)
    tokens[:, 0] = y0.reshape(-1).to(eh.device)
    plist = [P[n].contiguous() for n in _NAMES]
    _lib.check(L.sa_s2s_greedy_decode(_lib.ptr(eh.contiguous()), _ptr_array(plist), B, T, H, E, KS, K,
                                      math.log(T) if log_t else 1.0, int(end_tok), int(max_len), int(check_every),
                                      tokens.data_ptr(), rec.data_ptr() + 8 * B * (max_len + 1), _lib.ptr(ws), ws.numel(),
                                      _lib.cur_stream()), "sa_s2s_greedy_decode")
    host = rec.cpu()
    steps = int(host[-1:].numpy().view("int32")[0])
    return host[:B * (max_len + 1)].view(B, max_len + 1)[:, :steps + 1]


def beam_search(eh, P, log_t, start_tok, end_tok, beam_size, max_len, check_every=8):
    """Seq2Seq.beam_search (seq2seq.py:180-227) for ONE utterance on the device (sa_s2s_beam_search): eh (T, H) encoder
    states.  Returns (hypothesis tuple incl. the start token, score, (search steps, completed hypotheses)) after ONE
    device -> host copy at the end."""
    T, H = eh.shape
    E, K, KS = P["emb"].shape[1], P["fc_w"].shape[0], P["conv_w"].shape[-1]
    L = _L()
    nbytes = L.sa_s2s_beam_workspace_bytes(T, H, E, KS, K, beam_size, max_len)
    if nbytes == 0:
        raise _lib.SpeechAmdError("Seq2Seq beam search: unsupported shape (embedding_dim == rnn dim, dims % 4 == 0, odd "
                                  "location kernel <= 15 taps, beam_size <= 64)")
    ws = _lib.WORKSPACE.get(nbytes, eh.device, "s2s_beam")
    # one result record: [hyp (max_len + 1) int64 | score double | len, steps, n_complete, pad int32]
    rec = torch.empty(max_len + 1 + 1 + 2, dtype=torch.int64, device=eh.device)
    base = rec.data_ptr()
    p_score = base + 8 * (max_len + 1)
    p_len, p_info = p_score + 8, p_score + 12
    plist = [P[n].contiguous() for n in _NAMES]
    _lib.check(L.sa_s2s_beam_search(_lib.ptr(eh.contiguous()), _ptr_array(plist), T, H, E, KS, K,
                                    math.log(T) if log_t else 1.0, int(start_tok), int(end_tok), int(beam_size),
                                    int(max_len), int(check_every), base, p_len, p_score, p_info, _lib.ptr(ws),
                                    ws.numel(), _lib.cur_stream()), "sa_s2s_beam_search")
    host = rec.cpu()
    tail = host[max_len + 1:].numpy()
    score = float(tail[:1].view("float64")[0])
    n, steps, ncomp = (int(v) for v in tail[1:].view("int32")[:3])
    return tuple(int(t) for t in host[:n].tolist()), score, (steps, ncomp)


_NAMES = ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "conv_w", "conv_b", "nn_w", "nn_b", "fc_w", "fc_b")


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class DecoderFunction(torch.autograd.Function):
    """(logits (B, U-1, V-1), aligns (B, U-1, T)) = decode(eh, y)  -- seq2seq.py:77-112.
    params (in _NAMES order): embedding.weight, dec_rnn.{weight_ih,weight_hh,bias_ih,bias_hh}, attend.conv.{weight,bias},
    attend.nn.1.fc.{weight,bias}, fc.fc.{weight,bias}.  The token loop and its backward are ONE library call each
    (sa_s2s_decoder_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, eh, y, log_t, sample_prob, *params):
        _lib.require_cuda(eh, "encoder states (move the model to the GPU: model.cuda())")
        eh = eh.detach().contiguous()
        y = y.to(device=eh.device, dtype=torch.int64).contiguous()
        B, T, H = eh.shape
        U = y.shape[1]
        U1 = U - 1
        V, E = params[0].shape
        K = params[9].shape[0]
        KS = params[5].shape[-1]
        dev = eh.device
        plist = [p.detach().contiguous() for p in params]
        # scheduled sampling (:91-96): one draw of Python's RNG per token after the first, in the reference's order
        flags = None
        if sample_prob:
            flags = (ctypes.c_ubyte * U1)(*([0] + [1 if random.random() < sample_prob else 0 for _ in range(U1 - 1)]))
        f = dict(dtype=torch.float32, device=dev)
        out = torch.empty(U1, B, K, **f)
        IDX = torch.empty(U1, B, dtype=torch.int64, device=dev)
        IX, ST, HX = torch.empty(U1, B, E, **f), torch.empty(U1, B, 4 * H, **f), torch.empty(U1, B, H, **f)
        AX, OIN = torch.empty(U1, B, T, **f), torch.empty(U1, B, H, **f)
        scale = math.log(T) if log_t else 1.0
        L = _L()
        nbytes = L.sa_s2s_decoder_workspace_bytes(B, T, U1, H, E, KS, K)
        if nbytes == 0:
            raise _lib.SpeechAmdError("Seq2Seq decoder: unsupported shape (embedding_dim == rnn dim, dims % 4 == 0, "
                                      "odd location kernel <= 15 taps)")
        ws = _lib.WORKSPACE.get(nbytes, dev, "s2s_decoder")
        _lib.check(L.sa_s2s_decoder_fwd(_lib.ptr(eh), _lib.ptr(y), flags, _ptr_array(plist), B, T, U, H, E, KS, K, scale,
                                        _lib.ptr(out), _lib.ptr(IDX), _lib.ptr(IX), _lib.ptr(ST), _lib.ptr(HX),
                                        _lib.ptr(AX), _lib.ptr(OIN), _lib.ptr(ws), ws.numel(), _lib.cur_stream()),
                   "sa_s2s_decoder_fwd")
        if any(ctx.needs_input_grad):
            ctx.saved = (eh, plist, IDX, IX, ST, HX, AX, OIN, scale, (B, T, U, H, E, KS, K, V))
            ctx.slots = [getattr(p, "_grad_slot", None) for p in params]
            ctx.param_refs = params
            ctx.shapes = [tuple(p.shape) for p in params]
        aligns = AX.transpose(0, 1)
        ctx.mark_non_differentiable(aligns)
        return out.transpose(0, 1), aligns

    @staticmethod
    def backward(ctx, d_out, _d_aligns):
        eh, plist, IDX, IX, ST, HX, AX, OIN, scale, (B, T, U, H, E, KS, K, V) = ctx.saved
        dev = eh.device
        dO = d_out.transpose(0, 1).contiguous()
        d_eh = torch.empty(B, T, H, dtype=torch.float32, device=dev)
        # every gradient goes straight into its slot of the flat buffer when the parameter has one (and is handed over by
        # reference: ops.slot_hand_over -- autograd cloned all 11 into fresh .grad tensors every step before)
        carry = ops.slot_carry(ctx.param_refs, ctx.slots)
        grads = [s if s is not None else torch.empty(shape, dtype=torch.float32, device=dev)
                 for s, shape in zip(ctx.slots, ctx.shapes)]
        L = _L()
        ws = _lib.WORKSPACE.get(L.sa_s2s_decoder_workspace_bytes(B, T, U - 1, H, E, KS, K), dev, "s2s_decoder")
        _lib.check(L.sa_s2s_decoder_bwd(_lib.ptr(eh), _ptr_array(plist), _lib.ptr(dO), _lib.ptr(IDX), _lib.ptr(IX),
                                        _lib.ptr(ST), _lib.ptr(HX), _lib.ptr(AX), _lib.ptr(OIN), B, T, U, H, E, KS, K, V,
                                        scale, _lib.ptr(d_eh), _ptr_array(grads), _lib.ptr(ws), ws.numel(),
                                        _lib.cur_stream()), "sa_s2s_decoder_bwd")
        return (d_eh, None, None, None) + tuple(ops.slot_hand_over(ctx.param_refs, ctx.slots, grads, carry))


class XentFunction(torch.autograd.Function):
    """sum over rows of cross_entropy(logits, targets) * scale  (seq2seq.py:59-63 with scale = 1 / batch_size);
    the gradient is produced in the forward pass."""

    @staticmethod
    def forward(ctx, logits, targets, scale):
        _lib.require_cuda(logits, "logits")
        x = logits.detach().contiguous()
        N, K = x.shape
        tg = targets.to(device=x.device, dtype=torch.int64).contiguous()
        rows = torch.empty(N, 1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        _lib.check(_L().sa_softmax_xent(_lib.ptr(x), _lib.ptr(tg), scale, _lib.ptr(rows), _lib.ptr(dx), N, K,
                                        _lib.cur_stream()), "sa_softmax_xent")
        ctx.dx = dx
        return ops.colsum(rows) * scale

    @staticmethod
    def backward(ctx, go):
        return (ctx.dx * go.reshape(()) if ctx.dx is not None else None), None, None
