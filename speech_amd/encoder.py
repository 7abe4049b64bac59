"""speech_amd.encoder -- forward / backward orchestration of the conv + GRU encoder and the fc head over the HIP ops.

This is the host-side schedule of what /root/reference/speech/models/model.py:60-79 (Model.encode) and
ctc_model.py:25-32 (CTC.forward_impl) run through torch.nn / cuDNN, plus the backward pass autograd derives for the
reference (train.py:30).  One torch.autograd.Function covers the whole network, so the backward is an explicit,
fixed sequence of kernels writing every parameter gradient straight into its slot (no per-op autograd graph).

Every tensor op below is a libspeech_amd.so call through speech_amd.ops; torch only allocates.
Dropout (config["dropout"] != 0, training mode) multiplies by a torch-generated Bernoulli mask between kernels.
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

    def time_out(self, t):
        for out_c, h, w, s in self.conv_cfg:
            t = ops.conv_out_size(t, h, s)
        return t


def _drop_mask(t, p):
    return (torch.rand_like(t) >= p).to(t.dtype) / (1.0 - p)


class EncoderFunction(torch.autograd.Function):
    """logits = fc(encode(x)).  params: flat list [conv w, conv b, ..., per (layer, dir): w_ih, w_hh, b_ih, b_hh, ...,
    fc w, fc b] in that order."""

    @staticmethod
    def forward(ctx, plan, training, x, *params):
        P = list(params)
        nconv = len(plan.conv_cfg)
        conv_p = [(P[2 * i], P[2 * i + 1]) for i in range(nconv)]
        off = 2 * nconv
        gru_p = []
        for l in range(plan.layers):
            row = []
            for d in range(plan.D):
                row.append(tuple(P[off:off + 4]))  # w_ih, w_hh, b_ih, b_hh
                off += 4
            gru_p.append(row)
        fc_w, fc_b = P[off], P[off + 1]
        need_grad = training and any(ctx.needs_input_grad[3:])
        p_drop = plan.dropout if training else 0.0
        B, T, F = x.shape
        H, D = plan.H, plan.D

        # conv stack: the last conv writes the GRU-ready (B, T', C*F') channel-major layout (model.py:66-71)
        a = x.contiguous().view(B, 1, T, F)
        conv_saved = []
        for i, ((w, b), (out_c, kh, kw, s)) in enumerate(zip(conv_p, plan.conv_cfg)):
            last = i == nconv - 1
            y, ys = ops.conv2d_relu_fwd(a, w, b, s, feature_layout=last)
            mask = None
            if p_drop:
                mask = _drop_mask(y, p_drop)
                y = y * mask
            conv_saved.append((a, y, ys, mask))
            a = y
        feat = a  # (B, T', conv_out)
        Tp = feat.shape[1]

        # GRU stack
        inp = feat
        gru_saved = []
        for l in range(plan.layers):
            hbuf = torch.empty(B, Tp, D * H, dtype=torch.float32, device=x.device)
            stashes = []
            for d in range(D):
                w_ih, w_hh, b_ih, b_hh = gru_p[l][d]
                ai = ops.gemm(inp.view(B * Tp, inp.shape[2]), w_ih, trans_b=True, bias=b_ih).view(B, Tp, 3 * H)
                stash = torch.empty(B, Tp, 5 * H, dtype=torch.float32, device=x.device) if need_grad else None
                ops.gru_fwd(ai, w_hh, b_hh, hbuf[:, :, d * H:(d + 1) * H], stash, reverse=(d == 1))
                stashes.append(stash)
            mask = None
            out = hbuf
            if p_drop and l + 1 < plan.layers:  # nn.GRU: dropout on every layer's output except the last
                mask = _drop_mask(hbuf, p_drop)
                out = hbuf * mask
            gru_saved.append((inp, hbuf, stashes, mask))
            inp = out
        if D == 2:  # model.py:75-77
            enc = ops.add_rows(inp.view(B * Tp, 2 * H)[:, :H], inp.view(B * Tp, 2 * H)[:, H:])
        else:
            enc = inp.view(B * Tp, H)
        logits = ops.gemm(enc, fc_w, trans_b=True, bias=fc_b).view(B, Tp, fc_w.shape[0])

        if need_grad:
            ctx.plan = plan
            ctx.conv_p, ctx.gru_p, ctx.fc_w = conv_p, gru_p, fc_w
            ctx.conv_saved, ctx.gru_saved, ctx.enc = conv_saved, gru_saved, enc
            ctx.dims = (B, Tp)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        plan = ctx.plan
        B, Tp = ctx.dims
        H, D = plan.H, plan.D
        dl = dlogits.contiguous().view(B * Tp, -1)
        grads_fc = [ops.gemm(dl, ctx.enc, trans_a=True), ops.colsum(dl)]
        denc = ops.gemm(dl, ctx.fc_w)  # (B*T', H)
        # gradient wrt the top layer's (B, T', D*H) output: both directions receive denc (model.py:75-77)
        dout = denc.view(B, Tp, H)
        dout_views = [dout, dout] if D == 2 else [dout]
        grads_gru = [None] * plan.layers
        dai = torch.empty(B, Tp, 3 * H, dtype=torch.float32, device=dl.device)
        dah = torch.empty(B, Tp, 3 * H, dtype=torch.float32, device=dl.device)
        for l in range(plan.layers - 1, -1, -1):
            inp, hbuf, stashes, mask = ctx.gru_saved[l]
            I = inp.shape[2]
            dinp = torch.empty(B * Tp, I, dtype=torch.float32, device=dl.device)
            row = []
            for d in range(D):
                w_ih, w_hh, b_ih, b_hh = ctx.gru_p[l][d]
                stash = stashes[d]
                ops.gru_bwd(dout_views[d], hbuf[:, :, d * H:(d + 1) * H], stash, w_hh, dai, dah, reverse=(d == 1))
                dai2, dah2 = dai.view(B * Tp, 3 * H), dah.view(B * Tp, 3 * H)
                g_wih = ops.gemm(dai2, inp.view(B * Tp, I), trans_a=True)
                g_whh = ops.gemm(dah2, stash.view(B * Tp, 5 * H)[:, 4 * H:], trans_a=True)
                g_bih, g_bhh = ops.colsum(dai2), ops.colsum(dah2)
                ops.gemm(dai2, w_ih, out=dinp, beta=(1.0 if d == 1 else 0.0))
                row.append((g_wih, g_whh, g_bih, g_bhh))
            grads_gru[l] = row
            if l > 0:
                prev_mask = ctx.gru_saved[l - 1][3]
                dprev = dinp.view(B, Tp, I)
                if prev_mask is not None:
                    dprev = dprev * prev_mask
                dout_views = [dprev[:, :, d * H:(d + 1) * H] for d in range(D)]
        # conv stack
        dy = dinp.view(B, Tp, -1)
        grads_conv = [None] * len(plan.conv_cfg)
        for i in range(len(plan.conv_cfg) - 1, -1, -1):
            a, y, ys, mask = ctx.conv_saved[i]
            if mask is not None:
                dy = dy * mask
            s = plan.conv_cfg[i][3]
            dx, dw, db = ops.conv2d_relu_bwd(a, ctx.conv_p[i][0], y, dy.contiguous(), ys, s, need_dx=(i > 0))
            grads_conv[i] = (dw, db)
            dy = dx
        flat = []
        for dw, db in grads_conv:
            flat += [dw, db]
        for row in grads_gru:
            for g in row:
                flat += list(g)
        flat += grads_fc
        return (None, None, None) + tuple(flat)
