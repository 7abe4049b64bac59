"""speech_amd.seq2seq -- the attention decoder of /root/reference/speech/models/seq2seq.py on the HIP ops.

DecoderFunction is ONE autograd node for Seq2Seq.decode (:77-112): the teacher-forced (or scheduled-sampled) loop over
output tokens -- embedding lookup (+ previous context), nn.GRUCell, NNAttention, fc -- with an explicit backward
through the tokens (BPTT) that writes every parameter gradient straight into its slot.  Per token the forward issues
an embedding gather, two small projections (sa_gemm_f32), the gate kernel, the attention kernel and an add; everything
that does not depend on the loop (the fc over all tokens, every weight gradient, the embedding gradient) is one batched
call after the loop.  `step` is the same arithmetic for one token without autograd (Seq2Seq.decode_step, :114-138).
All compute is the HIP library; a CPU tensor raises.
"""
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
    B, H = eh.shape[0], eh.shape[2]
    ix = torch.empty(B, P["emb"].shape[1], dtype=torch.float32, device=eh.device)
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


_NAMES = ("emb", "w_ih", "w_hh", "b_ih", "b_hh", "conv_w", "conv_b", "nn_w", "nn_b", "fc_w", "fc_b")


class DecoderFunction(torch.autograd.Function):
    """(logits (B, U-1, V-1), aligns (B, U-1, T)) = decode(eh, y)  -- seq2seq.py:77-112.
    params (in _NAMES order): embedding.weight, dec_rnn.{weight_ih,weight_hh,bias_ih,bias_hh}, attend.conv.{weight,bias},
    attend.nn.1.fc.{weight,bias}, fc.fc.{weight,bias}."""

    @staticmethod
    def forward(ctx, eh, y, log_t, sample_prob, *params):
        _lib.require_cuda(eh, "encoder states (move the model to the GPU: model.cuda())")
        P = dict(zip(_NAMES, [p.detach() for p in params]))
        eh = eh.detach().contiguous()
        y = y.to(device=eh.device, dtype=torch.int64)
        B, T, H = eh.shape
        U1 = y.shape[1] - 1
        E = P["emb"].shape[1]
        K = P["fc_w"].shape[0]
        dev = eh.device
        conv_w = P["conv_w"].reshape(H, -1).contiguous()
        nn_w = P["nn_w"].reshape(H).contiguous()
        f = dict(dtype=torch.float32, device=dev)
        IDX = torch.empty(U1, B, dtype=torch.int64, device=dev)
        IX, HPREV = torch.empty(U1, B, E, **f), torch.empty(U1, B, H, **f)
        ST, HX = torch.empty(U1, B, 4 * H, **f), torch.empty(U1, B, H, **f)
        AX, OIN = torch.empty(U1, B, T, **f), torch.empty(U1, B, H, **f)
        scale = math.log(T) if log_t else 1.0
        L = _L()
        hprev = torch.zeros(B, H, **f)
        sx = torch.empty(B, H, **f)
        ws = _att_ws(B, T, H, conv_w.shape[1], dev)
        for t in range(U1):
            idx = y[:, t]
            # scheduled sampling (:91-96): feed back the argmax of the previous token's logits
            if t > 0 and sample_prob and random.random() < sample_prob:
                idx = argmax_rows(ops.gemm(OIN[t - 1], P["fc_w"], trans_b=True, bias=P["fc_b"]))
            IDX[t].copy_(idx)
            embedding_rows(P["emb"], IDX[t], IX[t])
            if t > 0:
                ops.add_rows(IX[t], sx, out=IX[t])
            HPREV[t].copy_(hprev)
            gi = ops.gemm(IX[t], P["w_ih"], trans_b=True, bias=P["b_ih"])
            gh = ops.gemm(hprev, P["w_hh"], trans_b=True, bias=P["b_hh"])
            _lib.check(L.sa_grucell_gates_fwd(_lib.ptr(gi), _lib.ptr(gh), _lib.ptr(hprev), _lib.ptr(HX[t]),
                                              _lib.ptr(ST[t]), B, H, _lib.cur_stream()), "sa_grucell_gates_fwd")
            _lib.check(L.sa_attention_fwd(_lib.ptr(eh), _lib.ptr(HX[t]), _lib.ptr(AX[t - 1]) if t > 0 else None,
                                          _lib.ptr(conv_w), _lib.ptr(P["conv_b"]), _lib.ptr(nn_w), _lib.ptr(P["nn_b"]),
                                          scale, _lib.ptr(AX[t]), _lib.ptr(sx), B, T, H, conv_w.shape[1],
                                          _lib.ptr(ws), ws.numel(), _lib.cur_stream()), "sa_attention_fwd")
            ops.add_rows(HX[t], sx, out=OIN[t])
            hprev = HX[t]
        out = ops.gemm(OIN.view(U1 * B, H), P["fc_w"], trans_b=True, bias=P["fc_b"]).view(U1, B, K)
        if any(ctx.needs_input_grad):
            ctx.saved = (eh, P, conv_w, nn_w, IDX, IX, HPREV, ST, HX, AX, OIN, scale)
            ctx.slots = [getattr(p, "_grad_slot", None) for p in params]
        aligns = AX.transpose(0, 1)
        ctx.mark_non_differentiable(aligns)
        return out.transpose(0, 1), aligns

    @staticmethod
    def backward(ctx, d_out, _d_aligns):
        eh, P, conv_w, nn_w, IDX, IX, HPREV, ST, HX, AX, OIN, scale = ctx.saved
        S = dict(zip(_NAMES, ctx.slots))
        B, T, H = eh.shape
        U1, E = IX.shape[0], IX.shape[2]
        K = P["fc_w"].shape[0]
        KS = conv_w.shape[1]
        dev = eh.device
        f = dict(dtype=torch.float32, device=dev)
        L = _L()
        dO = d_out.transpose(0, 1).contiguous().view(U1 * B, K)
        g = {}
        g["fc_w"] = ops.gemm(dO, OIN.view(U1 * B, H), trans_a=True, out=S["fc_w"])
        g["fc_b"] = ops.colsum(dO, out=S["fc_b"])
        dOIN = ops.gemm(dO, P["fc_w"]).view(U1, B, H)
        d_eh = torch.zeros(B, T, H, **f)
        g_cw, g_cb = torch.zeros(B, H * KS, **f), torch.zeros(B, H, **f)
        g_nw, g_nb = torch.zeros(B, H, **f), torch.zeros(B, 1, **f)
        DGI, DGH = torch.empty(U1, B, 3 * H, **f), torch.empty(U1, B, 3 * H, **f)
        DIX = torch.empty(U1, B, E, **f)
        d_ox, d_hx, d_hprev = torch.empty(B, H, **f), torch.empty(B, H, **f), torch.zeros(B, H, **f)
        d_sx = torch.empty(B, H, **f)
        d_ax = [torch.empty(B, T, **f), torch.empty(B, T, **f)]
        have_next = False
        ws = _att_ws(B, T, H, KS, dev)
        for t in range(U1 - 1, -1, -1):
            # the context of token t feeds the fc (dOIN[t]) and the next token's GRU input (DIX[t + 1])
            if have_next:
                ops.add_rows(dOIN[t], DIX[t + 1], out=d_sx)
            else:
                d_sx.copy_(dOIN[t])
            d_ax_next = d_ax[(t + 1) & 1] if have_next else None
            _lib.check(L.sa_attention_bwd(_lib.ptr(eh), _lib.ptr(HX[t]), _lib.ptr(AX[t - 1]) if t > 0 else None,
                                          _lib.ptr(conv_w), _lib.ptr(P["conv_b"]), _lib.ptr(nn_w), _lib.ptr(P["nn_b"]),
                                          scale, _lib.ptr(AX[t]), _lib.ptr(d_sx), _lib.ptr(d_ax_next), _lib.ptr(d_eh),
                                          _lib.ptr(d_ox), _lib.ptr(d_ax[t & 1]) if t > 0 else None, _lib.ptr(g_cw),
                                          _lib.ptr(g_cb), _lib.ptr(g_nw), _lib.ptr(g_nb), B, T, H, KS,
                                          _lib.ptr(ws), ws.numel(), _lib.cur_stream()), "sa_attention_bwd")
            # the decoder state of token t feeds the fc, the attention and the next token's GRU
            ops.add_rows(dOIN[t], d_ox, out=d_hx)
            ops.add_rows(d_hx, d_hprev, out=d_hx)
            _lib.check(L.sa_grucell_gates_bwd(_lib.ptr(d_hx), _lib.ptr(ST[t]), _lib.ptr(HPREV[t]), _lib.ptr(DGI[t]),
                                              _lib.ptr(DGH[t]), _lib.ptr(d_hprev), B, H, _lib.cur_stream()),
                       "sa_grucell_gates_bwd")
            ops.gemm(DGH[t], P["w_hh"], out=d_hprev, beta=1.0)
            ops.gemm(DGI[t], P["w_ih"], out=DIX[t])
            have_next = True
        g["w_ih"] = ops.gemm(DGI.view(U1 * B, 3 * H), IX.view(U1 * B, E), trans_a=True, out=S["w_ih"])
        g["w_hh"] = ops.gemm(DGH.view(U1 * B, 3 * H), HPREV.view(U1 * B, H), trans_a=True, out=S["w_hh"])
        g["b_ih"] = ops.colsum(DGI.view(U1 * B, 3 * H), out=S["b_ih"])
        g["b_hh"] = ops.colsum(DGH.view(U1 * B, 3 * H), out=S["b_hh"])
        V = P["emb"].shape[0]
        g["emb"] = S["emb"] if S["emb"] is not None else torch.empty(V, E, **f)
        _lib.check(L.sa_embedding_bwd(_lib.ptr(DIX), _lib.ptr(IDX), _lib.ptr(g["emb"]), U1 * B, E, V,
                                      _lib.cur_stream()), "sa_embedding_bwd")

        def reduce_b(part, name, shape):  # per-utterance partials -> the parameter's gradient (slot if it has one)
            out = S[name].view(-1) if S[name] is not None else None
            return ops.colsum(part, out=out).view(shape)

        g["conv_w"] = reduce_b(g_cw, "conv_w", (H, 1, KS))
        g["conv_b"] = reduce_b(g_cb, "conv_b", (H,))
        g["nn_w"] = reduce_b(g_nw, "nn_w", (1, H))
        g["nn_b"] = reduce_b(g_nb, "nn_b", (1,))
        return (d_eh, None, None, None) + tuple(g[n] for n in _NAMES)


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
