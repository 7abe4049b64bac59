"""speech_amd.ops -- thin torch-tensor wrappers over the C ABI (include/speech_amd.h).

torch supplies device memory and the current stream; every FLOP is done by libspeech_amd.so.  Nothing here has a
CPU path: CPU tensors raise SpeechAmdError."""
import math

import torch

from . import _lib
from ._lib import WORKSPACE, check, cur_stream, ptr


def _f32(t, name):
    _lib.require_cuda(t, name)
    if t.dtype != torch.float32:
        raise _lib.SpeechAmdError("%s must be float32" % name)
    return t


def gemm(a, b, trans_a=False, trans_b=False, bias=None, out=None, alpha=1.0, beta=0.0):
    """out = alpha * op(a) @ op(b) (+ bias) (+ beta * out); a, b 2-D row-major views (last stride 1)."""
    _f32(a, "a"), _f32(b, "b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    K, M = (a.shape if trans_a else (a.shape[1], a.shape[0]))
    Kb, N = ((b.shape[1], b.shape[0]) if trans_b else b.shape)
    assert K == Kb, (a.shape, b.shape, trans_a, trans_b)
    if out is None:
        assert beta == 0.0
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1
    L = _lib.lib()
    nbytes = L.sa_gemm_workspace_bytes(M, N, K)
    ws = WORKSPACE.get(nbytes, a.device, "gemm") if nbytes else None
    check(L.sa_gemm_f32(int(trans_a), int(trans_b), M, N, K, alpha, ptr(a), a.stride(0), ptr(b), b.stride(0), beta,
                        ptr(out), out.stride(0), ptr(bias), ptr(ws), ws.numel() if ws is not None else 0,
                        cur_stream()), "sa_gemm_f32")
    return out


def conv_out_size(n, k, s):
    return int(math.ceil((n - k + 1) / s))


def conv2d_relu_fwd(x, w, bias, s, feature_layout):
    """x (B,C,T,F) contiguous -> relu(conv(x)).  feature_layout: return (B, T', O*F') channel-major features
    (model.py:66-71) instead of NCHW."""
    _f32(x, "x"), _f32(w, "w"), _f32(bias, "bias")
    assert x.is_contiguous() and w.is_contiguous()
    B, C, T, F = x.shape
    O, C2, kh, kw = w.shape
    assert C == C2
    To, Fo = conv_out_size(T, kh, s), conv_out_size(F, kw, s)
    if To <= 0 or Fo <= 0:
        raise _lib.SpeechAmdError("convolution output is empty")
    if feature_layout:
        y = torch.empty(B, To, O * Fo, dtype=torch.float32, device=x.device)
        ys = (To * O * Fo, Fo, O * Fo)
    else:
        y = torch.empty(B, O, To, Fo, dtype=torch.float32, device=x.device)
        ys = (O * To * Fo, To * Fo, Fo)
    L = _lib.lib()
    nbytes = L.sa_conv2d_fwd_workspace_bytes(B, C, T, F, O, kh, kw, s)
    ws = WORKSPACE.get(nbytes, x.device, "conv")
    check(L.sa_conv2d_relu_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, C, T, F, O, kh, kw, s, ys[0], ys[1], ys[2],
                               ptr(ws), ws.numel(), cur_stream()), "sa_conv2d_relu_fwd")
    return y, ys


def conv2d_relu_bwd(x, w, y, dy, ys, s, need_dx):
    """Gradients of relu(conv(x)): returns (dx or None, dw, dbias).  y / dy share the strides `ys`."""
    B, C, T, F = x.shape
    O, _, kh, kw = w.shape
    dw = torch.empty_like(w)
    db = torch.empty(O, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if need_dx else None
    L = _lib.lib()
    nbytes = L.sa_conv2d_bwd_workspace_bytes(B, C, T, F, O, kh, kw, s)
    ws = WORKSPACE.get(nbytes, x.device, "conv")
    check(L.sa_conv2d_relu_bwd(ptr(x), ptr(w), ptr(y), ptr(dy), ptr(dx), ptr(dw), ptr(db), B, C, T, F, O, kh, kw, s,
                               ys[0], ys[1], ys[2], ptr(ws), ws.numel(), cur_stream()), "sa_conv2d_relu_bwd")
    return dx, dw, db


def gru_fwd(ai, w_hh, b_hh, h_out, stash, reverse):
    """ai (B,T,3H); h_out a (B,T,H) view (last stride 1) written in place; stash (B,T,5H) or None."""
    B, T, H3 = ai.shape
    H = H3 // 3
    assert ai.is_contiguous() and w_hh.is_contiguous() and h_out.shape == (B, T, H) and h_out.stride(2) == 1
    check(_lib.lib().sa_gru_fwd(ptr(ai), ptr(w_hh), ptr(b_hh), ptr(h_out), h_out.stride(0), h_out.stride(1),
                                ptr(stash), B, T, H, int(reverse), cur_stream()), "sa_gru_fwd")


def gru_bwd(dh_out, h_out, stash, w_hh, dai, dah, reverse):
    """dh_out: (B,T,H) view of the gradient wrt h_out; fills dai, dah (B,T,3H)."""
    B, T, H = dh_out.shape
    assert dh_out.stride(2) == 1 and dai.is_contiguous() and dah.is_contiguous() and stash.is_contiguous()
    L = _lib.lib()
    nbytes = L.sa_gru_bwd_workspace_bytes(B, T, H)
    ws = WORKSPACE.get(nbytes, dh_out.device, "gru")
    check(L.sa_gru_bwd(ptr(dh_out), dh_out.stride(0), dh_out.stride(1), ptr(h_out), ptr(stash), ptr(w_hh), ptr(dai),
                       ptr(dah), B, T, H, int(reverse), ptr(ws), ws.numel(), cur_stream()), "sa_gru_bwd")


def colsum(a, out=None, accumulate=False):
    """Column sums of a 2-D view (bias gradients)."""
    assert a.dim() == 2 and a.stride(1) == 1
    M, N = a.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    check(_lib.lib().sa_colsum_f32(ptr(a), a.stride(0), M, N, ptr(out), int(accumulate), cur_stream()),
          "sa_colsum_f32")
    return out


def add_rows(a, b, out=None):
    """out = a + b for 2-D views with unit last stride (bidirectional sum, model.py:75-77)."""
    assert a.shape == b.shape and a.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    check(_lib.lib().sa_add_rows_f32(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), a.shape[0],
                                     a.shape[1], cur_stream()), "sa_add_rows_f32")
    return out


def clip_sgd_step(params, grads, momentum_buf, lr, momentum, max_norm, grad_scale=1.0, norm_out=None):
    """Fused train.py:32,35 on flat buffers.  Returns the device scalar holding the pre-clip gradient norm."""
    _f32(params, "params"), _f32(grads, "grads")
    n = params.numel()
    assert grads.numel() == n and params.is_contiguous() and grads.is_contiguous()
    if norm_out is None:
        norm_out = torch.empty(1, dtype=torch.float32, device=params.device)
    L = _lib.lib()
    ws = WORKSPACE.get(L.sa_sgd_workspace_bytes(n), params.device, "sgd")
    check(L.sa_clip_sgd_step(ptr(params), ptr(grads), ptr(momentum_buf), n, lr, momentum, max_norm, grad_scale,
                             ptr(norm_out), ptr(ws), ws.numel(), cur_stream()), "sa_clip_sgd_step")
    return norm_out
