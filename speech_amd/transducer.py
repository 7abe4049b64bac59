"""speech_amd.transducer -- the replacement for the `transducer` package the reference imports at
/root/reference/speech/models/transducer_model.py:10-11 and calls at :50-51 and :98:

    loss_fn = transducer.TransducerLoss()         # no-arg constructor, built per call
    loss = loss_fn(out, y, x_lens, y_lens)        # out (B, T', U+1, V+1) LOG-softmax lattice on the GPU, requires grad
                                                  # y flat int32, x_lens / y_lens int32 (B,) -- CPU tensors
    td.decode_static(lp, beam_size, blank=V)[0]   # static beam search over one utterance's lattice

The package is un-vendored and un-pinned (Makefile:11), so conventions the reference tree does not show are constructor
keywords with documented defaults:
  blank_label=None  -> the LAST class, matching Transducer.blank = vocab_size (transducer_model.py:32)
  size_average=True -> the summed cost is divided by the batch size
Like the CTC loss, the gradient (with respect to the lattice entries) is produced in the forward pass.
This module also holds the autograd pieces of Transducer.decode (:54-78) over the HIP ops: embedding, prediction GRU,
broadcast joint, log_softmax.  All compute is the HIP library; a CPU tensor raises.
"""
import numpy as np
import torch

from . import _lib, ops
from .ctc import _host_i32


class TransducerLabels:
    """Labels / lengths of one batch, validated on the host once and staged on the device (cf. ctc.CTCLabels)."""

    def __init__(self, labels, act_lens, label_lens, device):
        lab_h, alen_h, llen_h = _host_i32(labels), _host_i32(act_lens), _host_i32(label_lens)
        self.B = alen_h.shape[0]
        if llen_h.shape[0] != self.B or lab_h.shape[0] != int(llen_h.sum()):
            raise _lib.SpeechAmdError("TransducerLoss: label / length tensors do not match the batch")
        if alen_h.min() < 1 or llen_h.min() < 0:
            raise _lib.SpeechAmdError("TransducerLoss: bad lengths")
        self.lab_h, self.alen_h, self.llen_h = lab_h, alen_h, llen_h
        ints = _lib.ints_to_device(np.concatenate([alen_h, llen_h, lab_h, np.zeros(1, np.int32)]).astype(np.int32), device)
        self.d_alen, self.d_llen, self.d_lab = ints[:self.B], ints[self.B:2 * self.B], ints[2 * self.B:]

    def check(self, B, T, U1, K, blank):
        if self.B != B or int(self.alen_h.max()) > T or int(self.llen_h.max()) + 1 > U1:
            raise _lib.SpeechAmdError("TransducerLoss: label / length tensors do not fit the lattice")
        if self.lab_h.size and (self.lab_h.min() < 0 or self.lab_h.max() >= K or (self.lab_h == blank).any()):
            raise _lib.SpeechAmdError("TransducerLoss: labels must be in [0, %d) and differ from blank=%d" % (K, blank))


def transducer_loss_raw(log_probs, labels, act_lens=None, label_lens=None, blank=None, want_grad=True):
    """Un-reduced face of the HIP kernels: (costs (B,), grads like log_probs or None), both on the GPU."""
    _lib.require_cuda(log_probs, "log_probs")
    if log_probs.dtype != torch.float32 or log_probs.dim() != 4:
        raise _lib.SpeechAmdError("log_probs must be a float32 (B, T, U+1, V) tensor")
    lp = log_probs.detach().contiguous()
    B, T, U1, K = lp.shape
    if blank is None:
        blank = K - 1
    lab = labels if isinstance(labels, TransducerLabels) else TransducerLabels(labels, act_lens, label_lens, lp.device)
    lab.check(B, T, U1, K, blank)
    L = _lib.lib()
    costs = torch.empty(B, dtype=torch.float32, device=lp.device)
    grads = torch.empty_like(lp) if want_grad else None
    nbytes = L.sa_transducer_workspace_bytes(T, U1, B)
    if nbytes == 0:
        raise _lib.SpeechAmdError("TransducerLoss: unsupported lattice size (U+1 <= 512)")
    ws = _lib.WORKSPACE.get(nbytes, lp.device, "transducer")
    _lib.check(L.sa_transducer_loss(_lib.ptr(lp), _lib.ptr(grads), _lib.ptr(lab.d_lab), _lib.ptr(lab.d_llen),
                                    _lib.ptr(lab.d_alen), K, B, T, U1, blank, _lib.ptr(costs), _lib.ptr(ws),
                                    ws.numel(), _lib.cur_stream()), "sa_transducer_loss")
    return costs, grads


class _TransducerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, log_probs, labels, act_lens, label_lens, blank, size_average, denom):
        costs, grads = transducer_loss_raw(log_probs, labels, act_lens, label_lens, blank,
                                           want_grad=ctx.needs_input_grad[0])
        ctx.grads = grads
        ctx.scale = 1.0 / (denom if denom else costs.shape[0]) if size_average else 1.0
        return (costs.sum() * ctx.scale).reshape(1)

    @staticmethod
    def backward(ctx, grad_out):
        g = ctx.grads
        if g is None:
            return (None,) * 7
        return (g * (grad_out.reshape(()) * ctx.scale),) + (None,) * 6


class TransducerLoss(torch.nn.Module):
    """Drop-in for transducer.functions.transducer.TransducerLoss (no-arg constructor; keyword defaults above)."""

    def __init__(self, blank_label=None, size_average=True, denom=None):
        super().__init__()
        self.blank_label = blank_label
        self.size_average = size_average
        self.denom = denom

    def forward(self, log_probs, labels, lengths, label_lengths):
        return _TransducerFunction.apply(log_probs, labels, lengths, label_lengths, self.blank_label,
                                         self.size_average, self.denom)


# ---- autograd pieces of Transducer.decode over the HIP ops ----------------------------------------------------------
class EmbeddingFunction(torch.autograd.Function):
    """nn.Embedding lookup (transducer_model.py:59): idx int64 (any shape) -> (..., E)."""

    @staticmethod
    def forward(ctx, idx, table):
        _lib.require_cuda(table, "embedding table")
        flat = idx.reshape(-1).to(device=table.device, dtype=torch.int64).contiguous()
        n, (V, E) = flat.numel(), table.shape
        out = torch.empty(n, E, dtype=torch.float32, device=table.device)
        _lib.check(_lib.lib().sa_embedding_fwd(_lib.ptr(table.detach().contiguous()), _lib.ptr(flat), _lib.ptr(out),
                                               n, E, _lib.cur_stream()), "sa_embedding_fwd")
        ctx.flat, ctx.shape_t = flat, (V, E)
        ctx.slot = getattr(table, "_grad_slot", None)
        ctx.table_ref = table
        return out.view(tuple(idx.shape) + (E,))

    @staticmethod
    def backward(ctx, dout):
        V, E = ctx.shape_t
        d = dout.contiguous().view(-1, E)
        carry = ops.slot_carry((ctx.table_ref,), (ctx.slot,))
        dt = ctx.slot if ctx.slot is not None else torch.empty(V, E, dtype=torch.float32, device=d.device)
        _lib.check(_lib.lib().sa_embedding_bwd(_lib.ptr(d), _lib.ptr(ctx.flat), _lib.ptr(dt), d.shape[0], E, V,
                                               _lib.cur_stream()), "sa_embedding_bwd")
        return None, ops.slot_hand_over((ctx.table_ref,), (ctx.slot,), (dt,), carry)[0]


class DropoutFunction(torch.autograd.Function):
    """x * Bernoulli mask / (1 - p) with the library's counter-based mask (csrc/dropout.h): element i of the contiguous
    tensor <-> index i of mask stream `stream_id` under key `seed`; backward re-applies the same mask.  Used between the
    layers of a multi-layer prediction network (nn.GRU(dropout=p), transducer_model.py:23-26)."""

    @staticmethod
    def forward(ctx, x, p, seed, stream_id):
        _lib.require_cuda(x, "x")
        ctx.args = (float(p), int(seed), int(stream_id))
        return ops.dropout_apply(x.contiguous(), *ctx.args)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout_apply(dy.contiguous(), *ctx.args), None, None, None


class GRUStackFunction(torch.autograd.Function):
    """Unidirectional nn.GRU(batch_first=True) stack (the prediction network, transducer_model.py:23-26,69) on the
    recurrence kernels: x (B, U, E) -> (B, U, H).  params per layer: w_ih, w_hh, b_ih, b_hh."""

    @staticmethod
    def forward(ctx, x, H, *params):
        L = len(params) // 4
        w_ih, w_hh = [params[4 * l] for l in range(L)], [params[4 * l + 1] for l in range(L)]
        b_ih, b_hh = [params[4 * l + 2] for l in range(L)], [params[4 * l + 3] for l in range(L)]
        xt = x.transpose(0, 1).contiguous()  # time-major (U, B, E)
        need = any(ctx.needs_input_grad)
        h_out, stash = ops.gru_stack_fwd(xt, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=need)
        if need:
            ctx.saved = (xt, h_out, stash, w_ih, w_hh, L, H)
            ctx.slots = [getattr(p, "_grad_slot", None) for p in params]
            ctx.param_refs = params
        return h_out[-1].transpose(0, 1)

    @staticmethod
    def backward(ctx, dy):
        xt, h_out, stash, w_ih, w_hh, L, H = ctx.saved
        U, B, E = xt.shape
        dtop = dy.transpose(0, 1).contiguous()
        carry = ops.slot_carry(ctx.param_refs, ctx.slots)
        dai, dah, dx = ops.gru_stack_bwd(dtop, stash, w_ih, w_hh, L, 1, H, E, want_dx=True)
        grads = []
        for l in range(L):
            lay_in = (xt if l == 0 else h_out[l - 1]).view(U * B, -1)
            dai2, dah2 = dai[l].view(U * B, 3 * H), dah[l].view(U * B, 3 * H)
            s = ctx.slots[4 * l:4 * l + 4]
            grads += [ops.gemm(dai2, lay_in, trans_a=True, out=s[0]),
                      ops.gemm(dah2, stash[l].view(U * B, 5 * H)[:, 4 * H:], trans_a=True, out=s[1]),
                      ops.colsum(dai2, out=s[2]), ops.colsum(dah2, out=s[3])]
        return (dx.transpose(0, 1), None) + tuple(ops.slot_hand_over(ctx.param_refs, ctx.slots, grads, carry))


class JointFunction(torch.autograd.Function):
    """relu(xa[:, :, None, :] + ya[:, None, :, :])  (transducer_model.py:72-74): (B,T,H), (B,U1,H) -> (B,T,U1,H)."""

    @staticmethod
    def forward(ctx, xa, ya):
        _lib.require_cuda(xa, "xa")
        xa, ya = xa.detach().contiguous(), ya.detach().contiguous()
        B, T, H = xa.shape
        U1 = ya.shape[1]
        z = torch.empty(B, T, U1, H, dtype=torch.float32, device=xa.device)
        _lib.check(_lib.lib().sa_joint_relu_fwd(_lib.ptr(xa), _lib.ptr(ya), _lib.ptr(z), B, T, U1, H,
                                                _lib.cur_stream()), "sa_joint_relu_fwd")
        ctx.save_for_backward(xa, ya)
        return z

    @staticmethod
    def backward(ctx, dz):
        xa, ya = ctx.saved_tensors
        B, T, H = xa.shape
        U1 = ya.shape[1]
        dz = dz.contiguous()
        dxa, dya = torch.empty_like(xa), torch.empty_like(ya)
        _lib.check(_lib.lib().sa_joint_relu_bwd(_lib.ptr(dz), _lib.ptr(xa), _lib.ptr(ya), _lib.ptr(dxa), _lib.ptr(dya),
                                                B, T, U1, H, _lib.cur_stream()), "sa_joint_relu_bwd")
        return dxa, dya


def joint_fused_supported(B, T, U1, H, K):
    """True when the one-operator joint (sa_joint_fused_*) takes this shape."""
    return _lib.lib().sa_joint_fused_workspace_bytes(B, T, U1, H, K) != 0


class FusedJointFunction(torch.autograd.Function):
    """log_softmax(fc2(relu(xa[:, :, None, :] + ya[:, None, :, :])))  (transducer_model.py:72-76) as one operator:
    (B,T,H), (B,U1,H), fc2.weight (K,H), fc2.bias (K) -> (B,T,U1,K) log-probabilities.  The (B,T,U1,H) joint tensor is
    never written; the backward rebuilds it inside the matrix products and writes the fc2 gradients into their slots."""

    @staticmethod
    def forward(ctx, xa, ya, w2, b2):
        _lib.require_cuda(xa, "xa")
        xa, ya = xa.detach().contiguous(), ya.detach().contiguous()
        w, b = w2.detach().contiguous(), b2.detach().contiguous()
        B, T, H = xa.shape
        U1, K = ya.shape[1], w.shape[0]
        logp = torch.empty(B, T, U1, K, dtype=torch.float32, device=xa.device)
        _lib.check(_lib.lib().sa_joint_fused_fwd(_lib.ptr(xa), _lib.ptr(ya), _lib.ptr(w), _lib.ptr(b), _lib.ptr(logp),
                                                 B, T, U1, H, K, _lib.cur_stream()), "sa_joint_fused_fwd")
        ctx.save_for_backward(xa, ya, w, logp)
        ctx.slots = (getattr(w2, "_grad_slot", None), getattr(b2, "_grad_slot", None))
        ctx.param_refs = (w2, b2)
        return logp

    @staticmethod
    def backward(ctx, glp):
        xa, ya, w, logp = ctx.saved_tensors
        B, T, H = xa.shape
        U1, K = ya.shape[1], w.shape[0]
        glp = glp.contiguous()
        L = _lib.lib()
        ws = _lib.WORKSPACE.get(L.sa_joint_fused_workspace_bytes(B, T, U1, H, K), xa.device, "joint_fused")
        dxa, dya = torch.empty_like(xa), torch.empty_like(ya)
        carry = ops.slot_carry(ctx.param_refs, ctx.slots)
        dw = ctx.slots[0] if ctx.slots[0] is not None else torch.empty_like(w)
        db = ctx.slots[1] if ctx.slots[1] is not None else torch.empty(K, dtype=torch.float32, device=xa.device)
        _lib.check(L.sa_joint_fused_bwd(_lib.ptr(glp), _lib.ptr(logp), _lib.ptr(xa), _lib.ptr(ya), _lib.ptr(w),
                                        _lib.ptr(dxa), _lib.ptr(dya), _lib.ptr(dw), _lib.ptr(db), B, T, U1, H, K,
                                        _lib.ptr(ws), ws.numel(), _lib.cur_stream()), "sa_joint_fused_bwd")
        return (dxa, dya) + tuple(ops.slot_hand_over(ctx.param_refs, ctx.slots, (dw, db), carry))


class LogSoftmaxFunction(torch.autograd.Function):
    """log_softmax over the last axis (transducer_model.py:76)."""

    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x, "x")
        x = x.detach().contiguous()
        K = x.shape[-1]
        y = torch.empty_like(x)
        _lib.check(_lib.lib().sa_log_softmax_fwd(_lib.ptr(x), _lib.ptr(y), x.numel() // K, K, _lib.cur_stream()),
                   "sa_log_softmax_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        K = y.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        _lib.check(_lib.lib().sa_log_softmax_bwd(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dx), y.numel() // K, K,
                                                 _lib.cur_stream()), "sa_log_softmax_bwd")
        return dx


# ---- static beam search (transducer.decoders.decode_static) ----------------------------------------------------------
def decode_static_batch(log_probs, u1=None, t=None, beam_size=1, blank=None):
    """log_probs (B, T, U1, K) lattice on the GPU; utterance b searches rows [0, u1[b]) and frames [0, t[b])
    (defaults: all).  Returns (list of label tuples, scores tensor (B,) float64 on the GPU)."""
    _lib.require_cuda(log_probs, "log_probs")
    lp = log_probs.detach().float().contiguous()
    B, T, U1, K = lp.shape
    if blank is None:
        blank = K - 1
    u1_h = np.full(B, U1, np.int32) if u1 is None else np.asarray(u1, np.int32)
    t_h = np.full(B, T, np.int32) if t is None else np.minimum(np.asarray(t, np.int32), T)
    if u1_h.min() < 1 or u1_h.max() > U1 or t_h.min() < 1:
        raise _lib.SpeechAmdError("decode_static: rows / frames out of range")
    ints = torch.from_numpy(np.concatenate([t_h, u1_h])).to(lp.device)
    L = _lib.lib()
    nbytes = L.sa_transducer_decode_workspace_bytes(T, U1, B, beam_size)
    if nbytes == 0 or K > 64:
        raise _lib.SpeechAmdError("decode_static: beam_size <= 16 and at most 64 classes are supported")
    ws = _lib.WORKSPACE.get(nbytes, lp.device, "transducer_decode")
    labels = torch.zeros(B, U1, dtype=torch.int32, device=lp.device)
    lens = torch.zeros(B, dtype=torch.int32, device=lp.device)
    scores = torch.zeros(B, dtype=torch.float64, device=lp.device)
    _lib.check(L.sa_transducer_decode_static(_lib.ptr(lp), _lib.ptr(ints[:B]), _lib.ptr(ints[B:]), K, B, T, U1,
                                             beam_size, blank, _lib.ptr(labels), _lib.ptr(lens), _lib.ptr(scores),
                                             _lib.ptr(ws), ws.numel(), _lib.cur_stream()),
               "sa_transducer_decode_static")
    lab_h, len_h = labels.cpu().numpy(), lens.cpu().numpy()
    return [tuple(int(v) for v in lab_h[b, :len_h[b]]) for b in range(B)], scores


def decode_static(log_probs, beam_size=1, blank=0):
    """Single-utterance face with the package's signature: log_probs (T, U, V) -> (labels tuple, score float)."""
    p = log_probs if torch.is_tensor(log_probs) else torch.from_numpy(np.ascontiguousarray(log_probs, np.float32))
    if not p.is_cuda:
        p = p.cuda()
    hyps, scores = decode_static_batch(p.float().unsqueeze(0), beam_size=beam_size, blank=blank)
    return hyps[0], float(scores[0])
