"""speech_amd.encoder -- forward / backward orchestration of the conv + GRU encoder and the fc head over the HIP ops.

This is the host-side schedule of what /root/reference/speech/models/model.py:60-79 (Model.encode) and
ctc_model.py:25-32 (CTC.forward_impl) run through torch.nn / cuDNN, plus the backward pass autograd derives for the
reference (train.py:30).  One torch.autograd.Function covers the whole network, so the backward is an explicit,
fixed sequence of kernels writing every parameter gradient straight into its slot (no per-op autograd graph).

Inside the encoder every activation is TIME-MAJOR (T', B, .): the last conv's GEMM epilogue scatters straight into
that layout, a time chunk of any layer is then a contiguous row range (what the GRU layer wavefront needs), and the
logits come out as (T', B, V+1) -- warp-ctc's native layout -- and are handed to the caller as a (B, T', V+1) view.

Every tensor op below is a libspeech_amd.so call through speech_amd.ops; torch only allocates.
Dropout (config["dropout"] != 0, training mode; model.py:25-27 behind every conv ReLU, model.py:38 between the GRU
layers) happens INSIDE those kernels: each forward pass draws one 64-bit Philox key, the conv epilogue and the fused
recurrence kernels evaluate the mask of the element they write (or whose gradient they route) from (key, tensor, index)
-- no mask tensor, no extra pass, and the GRU stack stays one launch per direction of time (csrc/dropout.h).
"""

import torch

from . import ops


class EncoderPlan:
    """Static description of the network (built once per Model): conv specs and GRU geometry."""

    def __init__(self, input_dim, config):
        enc = config["encoder"]
        self.conv_cfg = [tuple(int(v) for v in c) for c in enc["conv"]]
        self.dropout = float(config["dropout"])
        rnn = enc["rnn"]
        self.H = int(rnn["dim"])
        self.layers = int(rnn["layers"])
        self.bidirectional = bool(rnn["bidirectional"])
        self.D = 2 if self.bidirectional else 1
        f = input_dim
        for out_c, h, w, s in self.conv_cfg:
            f = ops.conv_out_size(f, w, s)
        self.conv_out_dim = self.conv_cfg[-1][0] * f
        self.input_dim = input_dim
        self.chunk = 0  # time steps per wavefront chunk of the non-fused paths (0: the library's default)

        # Model.flatten_parameters_() opts in to by-reference gradient hand-over (see EncoderFunction.backward)
        self.grad_by_reference = True
        # tests pin the masks: a fixed key instead of a fresh one per forward pass (None: draw from torch's CPU generator)
        self.fixed_seed = None

    def time_out(self, t):
        for out_c, h, w, s in self.conv_cfg:
            t = ops.conv_out_size(t, h, s)
        return t


class EncoderFunction(torch.autograd.Function):
    """logits = fc(encode(x)).  params: [conv w, conv b, ..., per (layer, dir): w_ih, w_hh, b_ih, b_hh, ..., fc w, fc b]."""

    @staticmethod
    def forward(ctx, plan, training, x, *params):
        P = list(params)
        nconv = len(plan.conv_cfg)
        H, D, L = plan.H, plan.D, plan.layers
        conv_p = [(P[2 * i], P[2 * i + 1]) for i in range(nconv)]
        off = 2 * nconv
        w_ih = [P[off + 4 * k] for k in range(L * D)]
        w_hh = [P[off + 4 * k + 1] for k in range(L * D)]
        b_ih = [P[off + 4 * k + 2] for k in range(L * D)]
        b_hh = [P[off + 4 * k + 3] for k in range(L * D)]
        fc_w, fc_b = P[off + 4 * L * D], P[off + 4 * L * D + 1]
        # gradients are wanted whenever a parameter requires them and grad mode is on (also after model.eval():
        # fine-tuning / gradient checks with dropout off); `training` only selects dropout
        need_grad = any(ctx.needs_input_grad[3:])
        p_drop = plan.dropout if training else 0.0
        seed = 0
        if p_drop:
            seed = plan.fixed_seed if plan.fixed_seed is not None else ops.new_dropout_seed()
        B, T, F = x.shape

        # conv stack; the last conv writes time-major channel-major features (T', B, C*F') (model.py:66-71)
        a = x.contiguous().view(B, 1, T, F)
        conv_saved = []
        for i, ((w, b), (out_c, kh, kw, s)) in enumerate(zip(conv_p, plan.conv_cfg)):
            # the first conv keeps its im2col matrix for the backward pass (no dx there, so it is read-only)
            keep = need_grad and i == 0
            res = ops.conv2d_relu_fwd(a, w, b, s, "tbf" if i == nconv - 1 else "nchw", keep_cols=keep,
                                      drop=(p_drop, seed, ops.DROP_STREAM_CONV + i) if p_drop else None)
            y, ys = res[0], res[1]  # with dropout: the dropped output (what the next layer reads, what backward masks by)
            cols = res[2] if keep else None
            conv_saved.append((a, y, ys, cols))
            a = y
        feat = a  # (T', B, conv_out)
        Tp = feat.shape[0]

        # GRU stack: ONE call for all layers, nn.GRU's inter-layer dropout (every layer's output but the last) inside it
        gdrop = (p_drop, seed, ops.DROP_STREAM_GRU)
        h_out, stash, h_drop = ops.gru_stack_fwd(feat, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=need_grad,
                                                 chunk=plan.chunk, drop=gdrop)
        top2 = h_out[-1].view(Tp * B, D * H)
        enc = ops.add_rows(top2[:, :H], top2[:, H:]) if D == 2 else top2  # model.py:75-77
        logits_tm = ops.gemm(enc, fc_w, trans_b=True, bias=fc_b).view(Tp, B, fc_w.shape[0])

        if need_grad:
            # parameters re-homed by Model.flatten_parameters_() carry a `_grad_slot` view of the flat gradient
            # buffer: backward writes each gradient there and hands the view to autograd
            ctx.slots = [getattr(p, "_grad_slot", None) for p in params]
            ctx.param_refs = params
            ctx.plan = plan
            ctx.conv_p, ctx.w_ih, ctx.w_hh, ctx.fc_w = conv_p, w_ih, w_hh, fc_w
            ctx.conv_saved, ctx.enc = conv_saved, enc
            ctx.gru_saved = (feat, h_out, stash, h_drop, gdrop)
            ctx.p_drop = p_drop
            ctx.dims = (B, Tp)
        return logits_tm.transpose(0, 1)  # (B, T', V+1) view of the time-major buffer

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        B, Tp = ctx.dims
        H, D, L = plan.H, plan.D, plan.layers
        nconv = len(plan.conv_cfg)
        slots = ctx.slots
        dl = dlogits.transpose(0, 1).contiguous().view(Tp * B, -1)  # no copy when it is the CTC gradient
        # A parameter whose .grad already IS its slot (a second backward without zero_grad: gradient accumulation)
        # expects "+=", but the kernels below OVERWRITE the slots: keep what is there and add it back at the end
        by_ref = plan.grad_by_reference
        carry = [(slot, slot.clone()) for p, slot in zip(ctx.param_refs, slots)
                 if by_ref and slot is not None and p.grad is not None and p.grad.data_ptr() == slot.data_ptr()]
        fc_i = 2 * nconv + 4 * L * D
        grads = [None] * (fc_i + 2)
        grads[fc_i], grads[fc_i + 1] = ops.gemm_tn_colsum(dl, ctx.enc, out=slots[fc_i], colsum_out=slots[fc_i + 1])
        denc = ops.gemm(dl, ctx.fc_w)  # (T'*B, H)
        # both directions of the top layer receive denc (model.py:75-77)
        dtop = torch.cat([denc, denc], dim=1).view(Tp, B, 2 * H) if D == 2 else denc.view(Tp, B, H)
        feat, h_out, stash, h_drop, gdrop = ctx.gru_saved
        I0 = feat.shape[2]
        # the library computes the stack's parameter gradients itself (and overlaps them with the recurrence):
        # hand it the slots of the flat gradient buffer, or fresh tensors for parameters that have none
        outs = []
        for l in range(L):
            for d in range(D):
                gi = 2 * nconv + 4 * (l * D + d)
                for q in range(4):
                    slot = slots[gi + q]
                    if slot is None:
                        ref = (ctx.w_ih, ctx.w_hh)[q][l * D + d] if q < 2 else None
                        slot = torch.empty_like(ref) if ref is not None else \
                            torch.empty(3 * H, dtype=torch.float32, device=dtop.device)
                    grads[gi + q] = slot
                outs.append(gi)
        wg = (feat, h_out, [grads[g] for g in outs], [grads[g + 1] for g in outs], [grads[g + 2] for g in outs],
              [grads[g + 3] for g in outs])
        dai, dah, dtop = ops.gru_stack_bwd(dtop.contiguous(), stash, ctx.w_ih, ctx.w_hh, L, D, H, I0, want_dx=True,
                                           chunk=plan.chunk, wgrad=wg, drop=gdrop, h_drop=h_drop)
        # conv stack
        dy = dtop
        for i in range(nconv - 1, -1, -1):
            a, y, ys, cols = ctx.conv_saved[i]
            s = plan.conv_cfg[i][3]
            dx, dw, db = ops.conv2d_relu_bwd(a, ctx.conv_p[i][0], y, dy.contiguous(), ys, s, need_dx=(i > 0),
                                             dw=slots[2 * i], db=slots[2 * i + 1], cols=cols, p_drop=ctx.p_drop)
            grads[2 * i], grads[2 * i + 1] = dw, db
            dy = dx
        # A gradient that sits in its slot of the flat buffer is handed over by reference (flatten_parameters_'s opt-in):
        # autograd's accumulator would otherwise CLONE every returned view into a fresh p.grad (20 device copies per
        # step, and a second copy of the whole gradient in memory).  A parameter that holds some OTHER gradient tensor
        # takes the ordinary (accumulating) path.
        for slot, old in carry:
            ops.add_rows(slot.view(1, -1), old.view(1, -1), out=slot.view(1, -1))
        out = list(grads)
        for i, (p, slot) in enumerate(zip(ctx.param_refs, slots)):
            if not by_ref or slot is None or out[i] is not slot or not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = slot
                out[i] = None
            elif p.grad.data_ptr() == slot.data_ptr():
                out[i] = None
        return (None, None, None) + tuple(out)
