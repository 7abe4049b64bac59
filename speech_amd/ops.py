"""speech_amd.ops -- thin torch-tensor wrappers over the C ABI (include/speech_amd.h).

torch supplies device memory and the current stream; every FLOP is done by libspeech_amd.so.  Nothing here has a
CPU path: CPU tensors raise SpeechAmdError."""
import ctypes
import math

import os

import torch

from . import _lib
from ._lib import WORKSPACE, check, cur_stream, ptr


class Profile:
    """Optional HIP-event timing of op calls on the current stream (bench.py's live roofline measurement).
    Usage: ops.PROFILE = ops.Profile(); ...run...; torch.cuda.synchronize(); ops.PROFILE.summary()"""

    def __init__(self):
        self.spans = []  # (name, start_event, end_event, launches, work)

    def summary(self):
        out = {}
        for name, e0, e1, n, work in self.spans:
            d = out.setdefault(name, {"ms": 0.0, "calls": 0, "launches": 0, "work": 0.0})
            d["ms"] += e0.elapsed_time(e1)
            d["calls"] += 1
            d["launches"] += n
            d["work"] += work
        return out


PROFILE = None


class _span:
    def __init__(self, name, launches=1, work=0.0):
        self.name, self.launches, self.work = name, launches, work

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record()
            PROFILE.spans.append((self.name, self.e0, self.e1, self.launches, self.work))
        return False


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
    with _span("gemm", 1, 2.0 * M * N * K):
        check(L.sa_gemm_f32(int(trans_a), int(trans_b), M, N, K, alpha, ptr(a), a.stride(0), ptr(b), b.stride(0),
                            beta, ptr(out), out.stride(0), ptr(bias), ptr(ws), ws.numel() if ws is not None else 0,
                            cur_stream()), "sa_gemm_f32")
    return out


def gemm_tn_colsum(a, b, out=None, colsum_out=None):
    """(a^T @ b, column sums of a) for a (K, M), b (K, N): the weight and the bias gradient of a linear layer from the
    gradient of its output, in one library call (sa_gemm_tn_colsum_f32).  Both outputs are overwritten."""
    _f32(a, "a"), _f32(b, "b")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1 and a.shape[0] == b.shape[0]
    K, M = a.shape
    N = b.shape[1]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
    if colsum_out is None:
        colsum_out = torch.empty(M, dtype=torch.float32, device=a.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and colsum_out.numel() == M and colsum_out.is_contiguous()
    L = _lib.lib()
    nbytes = L.sa_gemm_workspace_bytes(M, N, K)
    ws = WORKSPACE.get(nbytes, a.device, "gemm") if nbytes else None
    with _span("gemm", 2, 2.0 * M * N * K):
        check(L.sa_gemm_tn_colsum_f32(M, N, K, ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0),
                                      ptr(colsum_out), ptr(ws), ws.numel() if ws is not None else 0, cur_stream()),
              "sa_gemm_tn_colsum_f32")
    return out, colsum_out


def conv_out_size(n, k, s):
    return int(math.ceil((n - k + 1) / s))


# Mask streams of one forward pass (csrc/dropout.h): conv layer i -> DROP_STREAM_CONV + i, GRU layer l -> DROP_STREAM_GRU + l
DROP_STREAM_CONV, DROP_STREAM_GRU, DROP_STREAM_PRED = 0, 64, 128


def new_dropout_seed():
    """A fresh 64-bit Philox key for one forward pass, drawn from torch's CPU generator (so torch.manual_seed makes a
    run reproducible); no device work, no sync.  Data-parallel ranks seed that generator identically (train.py seeds every
    rank from the config) and index their masks by LOCAL element position, so the rank is mixed into the key: utterance b
    of rank 0 and utterance b of rank 1 get independent masks.  A rank that skips a forward pass (empty shard) must still
    call this once per step so that the ranks' generators stay in lock-step (train.py does)."""
    draw = int(torch.randint(0, 2 ** 62, (1,)).item())
    rank = int(os.environ.get("RANK", "0")) if "WORLD_SIZE" in os.environ else 0
    return (draw ^ ((rank * 0x9E3779B97F4A7C15) & (2 ** 62 - 1))) if rank else draw


def dropout_mask(n, p, seed, stream_id, device, idx0=0):
    """The factors (0 or 1/(1-p)) the kernels apply to elements idx0 .. idx0+n-1 of mask stream `stream_id`."""
    out = torch.empty(n, dtype=torch.float32, device=device)
    check(_lib.lib().sa_dropout_mask_f32(ptr(out), n, idx0, p, seed, stream_id, cur_stream()), "sa_dropout_mask_f32")
    return out


def dropout_apply(x, p, seed, stream_id, out=None):
    """out = x * mask (element i of the contiguous tensor <-> mask index i)."""
    _f32(x, "x")
    assert x.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(_lib.lib().sa_dropout_apply_f32(ptr(x), ptr(out), x.numel(), 0, p, seed, stream_id, cur_stream()),
          "sa_dropout_apply_f32")
    return out


def conv2d_relu_fwd(x, w, bias, s, layout="nchw", keep_cols=False, drop=None):
    """x (B,C,T,F) contiguous -> relu(conv(x)) in one of three output layouts:
      "nchw": (B, O, T', F')              (input of a following conv)
      "btf" : (B, T', O*F') channel-major features, batch-major   (model.py:66-71)
      "tbf" : (T', B, O*F') the same features time-major          (what the GRU stack consumes)
    Returns (y, (ys_b, ys_c, ys_t)) with y[b*ys_b + c*ys_c + t*ys_t + f]; with keep_cols also the im2col matrix
    (a tensor the caller passes to conv2d_relu_bwd(cols=...) so that backward does not rebuild it).
    drop = (p, seed, stream_id): nn.Dropout(p) behind the ReLU (model.py:25-27), applied in the kernel's epilogue; y is
    then the DROPPED output and conv2d_relu_bwd must be given the same p."""
    _f32(x, "x"), _f32(w, "w"), _f32(bias, "bias")
    assert x.is_contiguous() and w.is_contiguous()
    B, C, T, F = x.shape
    O, C2, kh, kw = w.shape
    assert C == C2
    To, Fo = conv_out_size(T, kh, s), conv_out_size(F, kw, s)
    if To <= 0 or Fo <= 0:
        raise _lib.SpeechAmdError("convolution output is empty")
    if layout == "btf":
        y = torch.empty(B, To, O * Fo, dtype=torch.float32, device=x.device)
        ys = (To * O * Fo, Fo, O * Fo)
    elif layout == "tbf":
        y = torch.empty(To, B, O * Fo, dtype=torch.float32, device=x.device)
        ys = (O * Fo, Fo, B * O * Fo)
    else:
        y = torch.empty(B, O, To, Fo, dtype=torch.float32, device=x.device)
        ys = (O * To * Fo, To * Fo, Fo)
    L = _lib.lib()
    nbytes = L.sa_conv2d_fwd_workspace_bytes(B, C, T, F, O, kh, kw, s)
    ws = WORKSPACE.get(nbytes, x.device, "conv")
    if keep_cols and L.sa_conv2d_is_direct(C, F, O, kh, kw, s):
        keep_cols = False  # the direct conv kernels build no im2col matrix: nothing to keep
        direct = True
    else:
        direct = False
    cols = torch.empty(B * To * Fo, C * kh * kw, dtype=torch.float32, device=x.device) if keep_cols else None
    with _span("conv_fwd", 2, 2.0 * B * To * Fo * O * C * kh * kw):
        if drop is not None and drop[0] > 0.0:
            check(L.sa_conv2d_relu_dropout_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, C, T, F, O, kh, kw, s, ys[0], ys[1],
                                               ys[2], ptr(cols), ptr(ws), ws.numel(), float(drop[0]), int(drop[1]),
                                               int(drop[2]), cur_stream()), "sa_conv2d_relu_dropout_fwd")
        else:
            check(L.sa_conv2d_relu_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), B, C, T, F, O, kh, kw, s, ys[0], ys[1], ys[2],
                                       ptr(cols), ptr(ws), ws.numel(), cur_stream()), "sa_conv2d_relu_fwd")
    if keep_cols or direct:
        return y, ys, cols
    return y, ys


def conv2d_relu_bwd(x, w, y, dy, ys, s, need_dx, dw=None, db=None, cols=None, p_drop=0.0):
    """Gradients of relu(conv(x)) (p_drop: of dropout(relu(conv(x))), y being the dropped output): returns
    (dx or None, dw, dbias).  y / dy share the strides `ys`."""
    B, C, T, F = x.shape
    O, _, kh, kw = w.shape
    if dw is None:
        dw = torch.empty_like(w)
    if db is None:
        db = torch.empty(O, dtype=torch.float32, device=x.device)
    assert dw.is_contiguous() and dw.shape == w.shape
    dx = torch.empty_like(x) if need_dx else None
    L = _lib.lib()
    nbytes = L.sa_conv2d_bwd_workspace_bytes(B, C, T, F, O, kh, kw, s)
    ws = WORKSPACE.get(nbytes, x.device, "conv")
    with _span("conv_bwd", 5, 0.0):
        if p_drop > 0.0:
            check(L.sa_conv2d_relu_dropout_bwd(ptr(x), ptr(w), ptr(y), ptr(dy), ptr(dx), ptr(dw), ptr(db), B, C, T, F, O,
                                               kh, kw, s, ys[0], ys[1], ys[2], ptr(cols), ptr(ws), ws.numel(),
                                               float(p_drop), cur_stream()), "sa_conv2d_relu_dropout_bwd")
        else:
            check(L.sa_conv2d_relu_bwd(ptr(x), ptr(w), ptr(y), ptr(dy), ptr(dx), ptr(dw), ptr(db), B, C, T, F, O, kh,
                                       kw, s, ys[0], ys[1], ys[2], ptr(cols), ptr(ws), ws.numel(), cur_stream()),
                  "sa_conv2d_relu_bwd")
    return dx, dw, db


def gru_fwd(ai, w_hh, b_hh, h_out, stash, reverse):
    """ai (B,T,3H); h_out a (B,T,H) view (last stride 1) written in place; stash (B,T,5H) or None."""
    B, T, H3 = ai.shape
    H = H3 // 3
    assert ai.is_contiguous() and w_hh.is_contiguous() and h_out.shape == (B, T, H) and h_out.stride(2) == 1
    # algorithmic bytes per step launch (W_hh resident on chip): ai 3H + h_prev H + h H (+ stash 5H), fp32
    nbytes = 4.0 * B * H * (5 + (5 if stash is not None else 0))
    with _span("gru_fwd_step", T, nbytes * T):
        check(_lib.lib().sa_gru_fwd(ptr(ai), ptr(w_hh), ptr(b_hh), ptr(h_out), h_out.stride(0), h_out.stride(1),
                                    ptr(stash), B, T, H, int(reverse), cur_stream()), "sa_gru_fwd")


def gru_bwd(dh_out, h_out, stash, w_hh, dai, dah, reverse):
    """dh_out: (B,T,H) view of the gradient wrt h_out; fills dai, dah (B,T,3H)."""
    B, T, H = dh_out.shape
    assert dh_out.stride(2) == 1 and dai.is_contiguous() and dah.is_contiguous() and stash.is_contiguous()
    L = _lib.lib()
    nbytes = L.sa_gru_bwd_workspace_bytes(B, T, H)
    ws = WORKSPACE.get(nbytes, dh_out.device, "gru")
    # algorithmic bytes per step launch: read dah 3H + stash 5H + dh_out H + dh H, write dai 3H + dah 3H + dh H
    with _span("gru_bwd_step", T, 4.0 * B * H * 17 * T):
        check(L.sa_gru_bwd(ptr(dh_out), dh_out.stride(0), dh_out.stride(1), ptr(h_out), ptr(stash), ptr(w_hh),
                           ptr(dai), ptr(dah), B, T, H, int(reverse), ptr(ws), ws.numel(), cur_stream()),
              "sa_gru_bwd")


def colsum(a, out=None, accumulate=False):
    """Column sums of a 2-D view (bias gradients)."""
    assert a.dim() == 2 and a.stride(1) == 1
    M, N = a.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=a.device)
    L = _lib.lib()
    ws = WORKSPACE.get(L.sa_colsum_workspace_bytes(M, N), a.device, "colsum")
    with _span("colsum", 2, 4.0 * M * N):
        check(L.sa_colsum_f32(ptr(a), a.stride(0), M, N, ptr(out), int(accumulate), ptr(ws), ws.numel(),
                              cur_stream()), "sa_colsum_f32")
    return out


N_CHAINS = 1      # independent recurrence chains on separate HIP streams. Measured on MI355X (tools/gru_step_bench.py):
                  # 1 chain 7.2 ms, 2 chains 9.8 ms, 4 chains 11.4 ms -- step launches do not overlap across queues,
                  # so the default stays 1.
_AUX_STREAMS = {}


def _aux_streams(device, n):
    """n cached side streams for `device` as a ctypes array of raw hipStream_t handles."""
    key = str(device)
    pool = _AUX_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=device))
    arr = (ctypes.c_void_p * max(n, 1))()
    for i in range(n):
        arr[i] = pool[i].cuda_stream
    return arr


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = ptr(t)
    return arr


def gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash, chunk=0, drop=None):
    """x (T, B, I0) time-major.  Parameter lists hold L*D tensors (index l*D + d).  Returns (h_out list of L
    (T, B, D*H) tensors, stash list of L*D (T, B, 5H) tensors or None).
    drop = (p, seed, stream0): nn.GRU's inter-layer dropout (model.py:38) inside the library; a third value is then
    returned, the list of L-1 dropped layer outputs (what gru_stack_bwd(drop=..., h_drop=...) needs back)."""
    _f32(x, "x")
    assert x.is_contiguous()
    T, B, I0 = x.shape
    dev = x.device
    h_out = [torch.empty(T, B, D * H, dtype=torch.float32, device=dev) for _ in range(L)]
    stash = [torch.empty(T, B, 5 * H, dtype=torch.float32, device=dev) for _ in range(L * D)] if want_stash else None
    lib = _lib.lib()
    ws = WORKSPACE.get(lib.sa_gru_stack_fwd_workspace_bytes(L, D, B, T, H, I0), dev, "gru_stack")
    launches = (T + L - 1 if D == 1 else L * T)
    nbytes = 4.0 * B * H * (5 + (5 if want_stash else 0)) * T * L * D
    dropping = drop is not None and drop[0] > 0.0 and L > 1
    with _span("gru_fwd_stack", launches, nbytes):
        if dropping:
            h_drop = [torch.empty(T, B, D * H, dtype=torch.float32, device=dev) for _ in range(L - 1)]
            check(lib.sa_gru_stack_fwd_dropout(ptr(x), I0, _ptr_array(w_ih), _ptr_array(b_ih), _ptr_array(w_hh),
                                               _ptr_array(b_hh), _ptr_array(h_out), _ptr_array(h_drop + [None]),
                                               _ptr_array(stash) if stash else None, L, D, B, T, H, chunk, ptr(ws),
                                               ws.numel(), float(drop[0]), int(drop[1]), int(drop[2]), cur_stream()),
                  "sa_gru_stack_fwd_dropout")
        else:
            check(lib.sa_gru_stack_fwd(ptr(x), I0, _ptr_array(w_ih), _ptr_array(b_ih), _ptr_array(w_hh),
                                       _ptr_array(b_hh), _ptr_array(h_out), _ptr_array(stash) if stash else None, L, D,
                                       B, T, H, chunk, ptr(ws), ws.numel(), cur_stream(),
                                       _aux_streams(dev, N_CHAINS - 1), N_CHAINS - 1), "sa_gru_stack_fwd")
    if drop is not None:
        return h_out, stash, (h_drop if dropping else None)
    return h_out, stash


def gru_stack_bwd(dh_top, stash, w_ih, w_hh, L, D, H, I0, want_dx=True, chunk=0, wgrad=None, drop=None, h_drop=None):
    """dh_top (T, B, D*H).  Returns (dai list, dah list of L*D (T, B, 3H) tensors, dx (T, B, I0) or None).
    wgrad = (x, h_out, dw_ih, dw_hh, db_ih, db_hh): the stack input (T, B, I0), the L layer outputs of the forward
    pass, and L*D output tensors each for the weight / bias gradients -- the library then computes them itself and
    overlaps them with the recurrence (sa_gru_stack_bwd_wgrad: side-stream GEMMs beside the persistent kernels)."""
    T, B, _ = dh_top.shape
    dev = dh_top.device
    assert dh_top.is_contiguous()
    dai = [torch.empty(T, B, 3 * H, dtype=torch.float32, device=dev) for _ in range(L * D)]
    dah = [torch.empty(T, B, 3 * H, dtype=torch.float32, device=dev) for _ in range(L * D)]
    dx = torch.empty(T, B, I0, dtype=torch.float32, device=dev) if want_dx else None
    lib = _lib.lib()
    ws = WORKSPACE.get(lib.sa_gru_stack_bwd_workspace_bytes(L, D, B, T, H, I0), dev, "gru_stack_bwd")
    launches = (T + L - 1 if D == 1 else L * T)
    with _span("gru_bwd_stack", launches, 4.0 * B * H * 17 * T * L * D):
        if wgrad is None:
            assert drop is None or not drop[0], "inter-layer dropout needs the wgrad form of the backward pass"
            check(lib.sa_gru_stack_bwd(ptr(dh_top), _ptr_array(stash), _ptr_array(w_ih), _ptr_array(w_hh),
                                       _ptr_array(dai), _ptr_array(dah), ptr(dx), I0, L, D, B, T, H, chunk, ptr(ws),
                                       ws.numel(), cur_stream(), _aux_streams(dev, N_CHAINS - 1), N_CHAINS - 1),
                  "sa_gru_stack_bwd")
        else:
            x, h_out, dw_ih, dw_hh, db_ih, db_hh = wgrad
            assert x.is_contiguous() and x.shape == (T, B, I0) and len(h_out) == L
            for k in range(L * D):
                I = I0 if k // D == 0 else D * H
                assert dw_ih[k].is_contiguous() and dw_ih[k].shape == (3 * H, I) and dw_hh[k].is_contiguous()
                assert dw_hh[k].shape == (3 * H, H) and db_ih[k].numel() == 3 * H and db_hh[k].numel() == 3 * H
            if drop is not None and drop[0] > 0.0 and L > 1:
                assert h_drop is not None and len(h_drop) == L - 1
                check(lib.sa_gru_stack_bwd_wgrad_dropout(
                    ptr(dh_top), _ptr_array(stash), _ptr_array(w_ih), _ptr_array(w_hh), _ptr_array(dai),
                    _ptr_array(dah), ptr(dx), I0, L, D, B, T, H, chunk, ptr(x), _ptr_array(h_out),
                    _ptr_array(list(h_drop) + [None]), _ptr_array(dw_ih), _ptr_array(dw_hh), _ptr_array(db_ih),
                    _ptr_array(db_hh), ptr(ws), ws.numel(), float(drop[0]), int(drop[1]), int(drop[2]), cur_stream()),
                    "sa_gru_stack_bwd_wgrad_dropout")
            else:
                check(lib.sa_gru_stack_bwd_wgrad(ptr(dh_top), _ptr_array(stash), _ptr_array(w_ih), _ptr_array(w_hh),
                                                 _ptr_array(dai), _ptr_array(dah), ptr(dx), I0, L, D, B, T, H, chunk,
                                                 ptr(x), _ptr_array(h_out), _ptr_array(dw_ih), _ptr_array(dw_hh),
                                                 _ptr_array(db_ih), _ptr_array(db_hh), ptr(ws), ws.numel(),
                                                 cur_stream()), "sa_gru_stack_bwd_wgrad")
    return dai, dah, dx


def add_rows(a, b, out=None):
    """out = a + b for 2-D views with unit last stride (bidirectional sum, model.py:75-77)."""
    assert a.shape == b.shape and a.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    if out is None:
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)
    check(_lib.lib().sa_add_rows_f32(ptr(a), a.stride(0), ptr(b), b.stride(0), ptr(out), out.stride(0), a.shape[0],
                                     a.shape[1], cur_stream()), "sa_add_rows_f32")
    return out


# ---- gradients handed over by reference (Model.flatten_parameters_(by_reference=True)) --------------------------------------
# A backward node that wrote a parameter's gradient into its slot of the flat buffer and RETURNS that view makes autograd's
# accumulator clone it into a fresh p.grad: one device copy per parameter and step (and a second copy of the gradients in
# memory) that nothing reads -- the optimiser works on the flat buffer.  The encoder's node (speech_amd/encoder.py) has
# handed its slots over by reference since round 3; these three helpers do the same for every other node that owns
# parameters (the Seq2Seq decoder's 11, the Transducer's prediction network / embedding / joint, LinearND).
def slot_carry(params, slots):
    """A parameter whose .grad already IS its slot (a second backward without zero_grad, or a parameter two nodes use) expects
    '+=', but the kernels OVERWRITE the slot: keep what is there, to be added back by slot_hand_over()."""
    return [(slot, slot.clone()) for p, slot in zip(params, slots)
            if slot is not None and getattr(p, "_grad_by_ref", False) and p.grad is not None and
            p.grad.data_ptr() == slot.data_ptr()]


def slot_hand_over(params, slots, grads, carry=()):
    """grads with None where the gradient sits in the parameter's slot and p.grad is (made) that slot."""
    for slot, old in carry:
        add_rows(slot.view(1, -1), old.view(1, -1), out=slot.view(1, -1))
    out = list(grads)
    for i, (p, slot) in enumerate(zip(params, slots)):
        if slot is None or out[i] is not slot or not getattr(p, "_grad_by_ref", False) or not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = slot
            out[i] = None
        elif p.grad.data_ptr() == slot.data_ptr():
            out[i] = None
    return out


def clip_sgd_step(params, grads, momentum_buf, lr, momentum, max_norm, grad_scale=1.0, norm_out=None):
    """Fused train.py:32,35 on flat buffers.  Returns the device scalar holding the pre-clip gradient norm.
    `grads` may carry ONE extra trailing element (Model.flatten_parameters_ allocates it): the health flag written by
    stamp_health() and summed by the gradient all-reduce.  When it is non-zero the update is skipped on the device
    and the returned norm is NEGATIVE (see include/speech_amd.h, sa_gru_health_flag)."""
    _f32(params, "params"), _f32(grads, "grads")
    n = params.numel()
    assert grads.numel() in (n, n + 1) and params.is_contiguous() and grads.is_contiguous()
    flag = grads[n:] if grads.numel() == n + 1 else None
    if norm_out is None:
        norm_out = torch.empty(1, dtype=torch.float32, device=params.device)
    L = _lib.lib()
    ws = WORKSPACE.get(L.sa_sgd_workspace_bytes(n), params.device, "sgd")
    check(L.sa_clip_sgd_step(ptr(params), ptr(grads), ptr(momentum_buf), n, lr, momentum, max_norm, grad_scale,
                             ptr(norm_out), ptr(flag), ptr(ws), ws.numel(), cur_stream()), "sa_clip_sgd_step")
    return norm_out


def stamp_health(grads_with_flag):
    """Write the persistent kernels' health flag (0.0 fine, 1.0 failed) into the LAST element of the flat gradient
    message, stream-ordered behind the backward pass: call it after loss.backward() and before the all-reduce."""
    check(_lib.lib().sa_gru_health_flag(ptr(grads_with_flag[-1:]), cur_stream()), "sa_gru_health_flag")


def persist_status(wait=True):
    """OR of the persistent kernels' failure codes seen so far (0 = fine).  wait=True drains the outstanding copies."""
    return int(_lib.lib().sa_gru_persist_status()) if wait else 0


def persist_reset():
    """After a failure: drain, clear the device error word and the host state.  The persistent path stays switched
    off for the process; the caller re-runs the lost step (it goes to the step kernels).  Returns the failure code."""
    return int(_lib.lib().sa_gru_persist_reset())


class ScalarPipe:
    """Device scalars to the host WITHOUT stalling the launch queue (the reference's loop reads loss.data[0] every
    step, /root/reference/train.py:33: a full sync).  push(t) enqueues an async copy of the 1-element tensor into
    pinned memory behind the work queued so far and returns the values of EARLIER pushes; drain() waits for the rest.
    Values come back in push order.
      lag=None: whatever has landed by now (timing dependent: fine for a progress bar);
      lag=k   : exactly the pushes that are k or more pushes old, waiting for them if need be -- WHICH push a value is
                reported at is then a function of the push count alone, identical on every data-parallel rank (the
                training loop's failure replay must be: ranks that notice a skipped update at different iterations
                would issue different numbers of gradient all-reduces).  k <= depth - 1.
    CPU tensors are accepted (their value is taken at once, released under the same rules): the loop logic is tested
    without a GPU."""

    def __init__(self, depth=4):
        self._depth = depth
        self._slots = None  # pinned buffers + events, allocated at the first CUDA push
        self._pending = []  # (slot index or None, tag, value or None) in push order
        self._n = 0

    def _value(self, entry, wait):
        i, tag, val = entry
        if i is None:
            return val
        buf, ev = self._slots[i]
        if wait:
            ev.synchronize()
        elif not ev.query():
            return None
        return float(buf[0])

    def _collect(self, wait, keep=0):
        out = []
        while len(self._pending) > keep:
            v = self._value(self._pending[0], wait)
            if v is None:
                break
            out.append((self._pending[0][1], v))
            self._pending.pop(0)
        return out

    def push(self, t, tag=None, lag=None):
        if lag is None:
            done = self._collect(wait=False)
            if len(self._pending) == self._depth:  # ring full: the oldest copy is `depth` steps old
                done += self._collect(wait=True, keep=self._depth - 1)
        else:
            assert 0 <= lag < self._depth
            done = self._collect(wait=True, keep=max(lag - 1, 0)) if lag else self._collect(wait=True)
        if t.is_cuda:
            if self._slots is None:
                self._slots = [(torch.empty(1, dtype=torch.float32).pin_memory(), torch.cuda.Event())
                               for _ in range(self._depth)]
            i = self._n % self._depth
            self._n += 1
            buf, ev = self._slots[i]
            buf.copy_(t.detach().reshape(1), non_blocking=True)
            ev.record()
            self._pending.append((i, tag, None))
        else:
            self._pending.append((None, tag, float(t.detach().reshape(1)[0])))
        if lag == 0:
            done += self._collect(wait=True)
        return done

    def drain(self):
        return self._collect(wait=True)


_UNIT = {}
_UNIT_VERSION = {}


def unit_gradient(device):
    """The (1,) float32 tensor holding 1.0 on `device`, created once: what backward(loss) seeds autograd with."""
    key = str(device)
    if key not in _UNIT:
        _UNIT[key] = torch.ones(1, dtype=torch.float32, device=device)
        _UNIT_VERSION[key] = _UNIT[key]._version
    return _UNIT[key]


def is_unit_gradient(t):
    """`t` is the cached unit gradient of its device (the same storage cell), and nothing has written into that tensor
    since it was created (torch's version counter): only then may a node skip the multiplication by it."""
    key = str(t.device)
    u = _UNIT.get(key)
    return (u is not None and t.numel() == 1 and t.dtype == torch.float32 and t.data_ptr() == u.data_ptr()
            and u._version == _UNIT_VERSION[key])


def backward(loss):
    """loss.backward() (/root/reference/train.py:30) without autograd's own launches: torch seeds the walk with
    ones_like(loss) -- a fill kernel per step -- and the CTC node then multiplies its gradient by that 1 (a second launch);
    seeded with the cached unit gradient the walk starts with no launch and the node recognises the seed.  Plain
    loss.backward() keeps working (and keeps those two launches)."""
    if loss.numel() == 1 and loss.dtype == torch.float32:
        loss.backward(gradient=unit_gradient(loss.device).view_as(loss))
    else:
        loss.backward()
