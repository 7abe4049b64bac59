"""
oracle/encoder_np.py -- NumPy restatement of the reference's conv + GRU encoder and the CTC head,
forward AND backward, in a caller-chosen dtype (float64 = the checker, float32 = the timed CPU port).
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline); never on the product path.

Restates
  * /root/reference/speech/models/model.py:19-29   Conv2d(in_c, out_c, (h, w), stride=(s, s), padding=0) + ReLU stack
  * /root/reference/speech/models/model.py:44-52   conv_out_size: n = ceil((n - k + 1) / s) per conv, along dim
  * /root/reference/speech/models/model.py:60-79   encode: unsqueeze(1) -> conv -> transpose(1,2) -> view(B,T',C*F')
                                                    -> nn.GRU(batch_first) -> (bidirectional: sum of the two halves)
  * /root/reference/speech/models/model.py:35-39   nn.GRU(input_size, hidden, layers, batch_first, bidirectional)
                                                    (equations: SURVEY.md Appendix B; gate row order [r; z; n];
                                                    b_hn sits inside the r * (.) term)
  * /root/reference/speech/models/model.py:126-133 LinearND: flatten leading dims, nn.Linear, reshape
  * /root/reference/speech/models/ctc_model.py:19,25-32  fc to |V|+1 classes; forward_impl = encode -> fc
  * /root/reference/train.py:32,35                 clip_grad_norm(params, 200) then SGD step

Parameters use the reference's state-dict names: conv.{i}.weight/bias, rnn.weight_ih_l{k}[_reverse],
rnn.weight_hh_l{k}[_reverse], rnn.bias_ih_l{k}[_reverse], rnn.bias_hh_l{k}[_reverse], fc.fc.weight, fc.fc.bias.

Pinned against the live reference Model/CTC classes by oracle/gen_golden.py -> tests/golden/encoder_*.npz,
and (forward + backward) against torch.nn float64 autograd in tests/test_oracle_encoder.py.
Dropout is restated for p == 0 only (the parity configuration).
"""
import math

import numpy as np


def conv_out_size(n, conv_cfg, dim):
    """model.py:44-52.  conv_cfg rows are [out_c, h, w, s]; dim 0 = time (kernel h), dim 1 = freq (kernel w)."""
    for out_c, h, w, s in conv_cfg:
        k = h if dim == 0 else w
        n = int(math.ceil((n - k + 1) / s))
    return n


def conv_param_names(cfg):
    """Sequential indices of the Conv2d modules (model.py:19-29): conv, relu[, dropout] per layer."""
    step = 3 if cfg["dropout"] != 0 else 2
    return [step * i for i in range(len(cfg["encoder"]["conv"]))]


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _im2col(x, h, w, s):
    """x (B,C,T,F) -> cols (B, T', F', C*h*w) for a valid conv with stride (s, s)."""
    win = np.lib.stride_tricks.sliding_window_view(x, (h, w), axis=(2, 3))  # (B,C,T-h+1,F-w+1,h,w)
    win = win[:, :, ::s, ::s]
    B, C, To, Fo = win.shape[:4]
    return win.transpose(0, 2, 3, 1, 4, 5).reshape(B, To, Fo, C * h * w)


def conv_relu_fwd(x, wgt, bias, s):
    O, C, h, w = wgt.shape
    cols = _im2col(x, h, w, s)
    y = cols @ wgt.reshape(O, -1).T + bias  # (B,T',F',O)
    y = np.maximum(y, 0.0)
    return np.ascontiguousarray(y.transpose(0, 3, 1, 2)), cols  # (B,O,T',F')


def conv_relu_bwd(dy, y, cols, x_shape, wgt, s, need_dx):
    O, C, h, w = wgt.shape
    d = (dy * (y > 0)).transpose(0, 2, 3, 1)  # (B,T',F',O)
    dW = np.tensordot(d, cols, axes=([0, 1, 2], [0, 1, 2])).reshape(wgt.shape)
    db = d.sum(axis=(0, 1, 2))
    dx = None
    if need_dx:
        dcols = (d @ wgt.reshape(O, -1)).reshape(d.shape[0], d.shape[1], d.shape[2], C, h, w)
        dx = np.zeros(x_shape, dtype=dy.dtype)
        To, Fo = d.shape[1], d.shape[2]
        for i in range(h):
            for j in range(w):
                dx[:, :, i:i + s * To:s, j:j + s * Fo:s] += dcols[:, :, :, :, i, j].transpose(0, 3, 1, 2)
    return dx, dW, db


def gru_dir_fwd(x, Wih, Whh, bih, bhh, reverse):
    """One direction of one layer.  x (B,T,I) -> h (B,T,H).  App. B equations."""
    B, T, _ = x.shape
    H = Whh.shape[1]
    ai = x @ Wih.T + bih  # (B,T,3H)
    hs = np.zeros((B, T, H), dtype=x.dtype)
    r_ = np.zeros_like(hs); z_ = np.zeros_like(hs); n_ = np.zeros_like(hs); q_ = np.zeros_like(hs)
    h = np.zeros((B, H), dtype=x.dtype)
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        ah = h @ Whh.T + bhh
        r = _sigmoid(ai[:, t, :H] + ah[:, :H])
        z = _sigmoid(ai[:, t, H:2 * H] + ah[:, H:2 * H])
        q = ah[:, 2 * H:]  # W_hn h + b_hn
        n = np.tanh(ai[:, t, 2 * H:] + r * q)
        h = (1.0 - z) * n + z * h
        hs[:, t] = h; r_[:, t] = r; z_[:, t] = z; n_[:, t] = n; q_[:, t] = q
    return hs, (x, hs, r_, z_, n_, q_, reverse)


def gru_dir_bwd(dhs, cache, Wih, Whh):
    """Backward of gru_dir_fwd.  dhs (B,T,H) -> dx, dWih, dWhh, dbih, dbhh."""
    x, hs, r_, z_, n_, q_, reverse = cache
    B, T, _ = x.shape
    H = Whh.shape[1]
    dai = np.zeros((B, T, 3 * H), dtype=x.dtype)  # grad wrt i2h pre-activations
    dah = np.zeros((B, T, 3 * H), dtype=x.dtype)  # grad wrt h2h pre-activations
    dh = np.zeros((B, H), dtype=x.dtype)
    order = range(T) if reverse else range(T - 1, -1, -1)  # reverse of the forward order
    hprev_all = np.zeros_like(hs)
    if reverse:
        hprev_all[:, :-1] = hs[:, 1:]
    else:
        hprev_all[:, 1:] = hs[:, :-1]
    for t in order:
        dh = dh + dhs[:, t]
        r = r_[:, t]; z = z_[:, t]; n = n_[:, t]; q = q_[:, t]; hp = hprev_all[:, t]
        dn = dh * (1.0 - z)
        dz = dh * (hp - n)
        dpn = dn * (1.0 - n * n)          # wrt (a_in + r*q)
        dr = dpn * q
        dpr = dr * r * (1.0 - r)
        dpz = dz * z * (1.0 - z)
        dai[:, t, :H] = dpr; dai[:, t, H:2 * H] = dpz; dai[:, t, 2 * H:] = dpn
        dah[:, t, :H] = dpr; dah[:, t, H:2 * H] = dpz; dah[:, t, 2 * H:] = dpn * r
        dh = dh * z + dah[:, t] @ Whh
    dx = dai @ Wih
    dWih = np.tensordot(dai, x, axes=([0, 1], [0, 1]))
    dWhh = np.tensordot(dah, hprev_all, axes=([0, 1], [0, 1]))
    return dx, dWih, dWhh, dai.sum(axis=(0, 1)), dah.sum(axis=(0, 1))


def _sfx(layer, rev):
    return "_l%d%s" % (layer, "_reverse" if rev else "")


def model_fwd(params, x, cfg, dtype=np.float64):
    """CTC.forward_impl (ctc_model.py:25-32) without softmax: x (B,T,F) -> logits (B,T',|V|+1), cache."""
    P = {k: np.asarray(v, dtype=dtype) for k, v in params.items()}
    conv_cfg = cfg["encoder"]["conv"]
    rnn_cfg = cfg["encoder"]["rnn"]
    a = np.asarray(x, dtype=dtype)[:, None]  # unsqueeze(1)
    conv_cache = []
    for idx, (out_c, h, w, s) in zip(conv_param_names(cfg), conv_cfg):
        xin = a
        a, cols = conv_relu_fwd(xin, P["conv.%d.weight" % idx], P["conv.%d.bias" % idx], s)
        conv_cache.append((xin.shape, cols, a))
    B, C, To, Fo = a.shape
    feat = np.ascontiguousarray(a.transpose(0, 2, 1, 3)).reshape(B, To, C * Fo)  # model.py:66-71 (channel-major)
    bi = bool(rnn_cfg["bidirectional"])
    gru_cache = []
    inp = feat
    for l in range(rnn_cfg["layers"]):
        outs, caches = [], []
        for rev in ([False, True] if bi else [False]):
            sfx = _sfx(l, rev)
            hs, c = gru_dir_fwd(inp, P["rnn.weight_ih" + sfx], P["rnn.weight_hh" + sfx],
                                P["rnn.bias_ih" + sfx], P["rnn.bias_hh" + sfx], rev)
            outs.append(hs); caches.append(c)
        gru_cache.append(caches)
        inp = np.concatenate(outs, axis=2) if bi else outs[0]
    H = rnn_cfg["dim"]
    enc = inp[:, :, :H] + inp[:, :, H:] if bi else inp  # model.py:75-77
    logits = enc @ P["fc.fc.weight"].T + P["fc.fc.bias"]
    cache = dict(P=P, cfg=cfg, conv=conv_cache, a_shape=(B, C, To, Fo), gru=gru_cache, enc=enc, bi=bi)
    return logits, cache


def model_bwd(cache, dlogits):
    """Gradients of sum(logits * dlogits) wrt every parameter, keyed by state-dict name."""
    P, cfg, bi = cache["P"], cache["cfg"], cache["bi"]
    rnn_cfg = cfg["encoder"]["rnn"]
    H = rnn_cfg["dim"]
    dl = np.asarray(dlogits, dtype=cache["enc"].dtype)
    G = {}
    G["fc.fc.weight"] = np.tensordot(dl, cache["enc"], axes=([0, 1], [0, 1]))
    G["fc.fc.bias"] = dl.sum(axis=(0, 1))
    denc = dl @ P["fc.fc.weight"]
    dout = np.concatenate([denc, denc], axis=2) if bi else denc
    for l in range(rnn_cfg["layers"] - 1, -1, -1):
        dins = []
        for d, rev in enumerate([False, True] if bi else [False]):
            sfx = _sfx(l, rev)
            dhs = dout[:, :, d * H:(d + 1) * H]
            dx, dWih, dWhh, dbih, dbhh = gru_dir_bwd(dhs, cache["gru"][l][d], P["rnn.weight_ih" + sfx],
                                                     P["rnn.weight_hh" + sfx])
            G["rnn.weight_ih" + sfx] = dWih; G["rnn.weight_hh" + sfx] = dWhh
            G["rnn.bias_ih" + sfx] = dbih; G["rnn.bias_hh" + sfx] = dbhh
            dins.append(dx)
        dout = dins[0] + dins[1] if bi else dins[0]
    B, C, To, Fo = cache["a_shape"]
    da = np.ascontiguousarray(dout.reshape(B, To, C, Fo).transpose(0, 2, 1, 3))
    idxs = conv_param_names(cfg)
    conv_cfg = cfg["encoder"]["conv"]
    for i in range(len(conv_cfg) - 1, -1, -1):
        x_shape, cols, y = cache["conv"][i]
        s = conv_cfg[i][3]
        da, dW, db = conv_relu_bwd(da, y, cols, x_shape, P["conv.%d.weight" % idxs[i]], s, need_dx=(i > 0))
        G["conv.%d.weight" % idxs[i]] = dW
        G["conv.%d.bias" % idxs[i]] = db
    return G


def init_params(input_dim, output_dim, cfg, seed=2017, dtype=np.float32):
    """Random parameters with nn.GRU's U(-1/sqrt(H), 1/sqrt(H)) scale (App. B) and fan-in scaling elsewhere."""
    rng = np.random.RandomState(seed)
    conv_cfg = cfg["encoder"]["conv"]
    rnn_cfg = cfg["encoder"]["rnn"]
    P = {}
    in_c = 1
    for idx, (out_c, h, w, s) in zip(conv_param_names(cfg), conv_cfg):
        k = 1.0 / math.sqrt(in_c * h * w)
        P["conv.%d.weight" % idx] = rng.uniform(-k, k, (out_c, in_c, h, w)).astype(dtype)
        P["conv.%d.bias" % idx] = rng.uniform(-k, k, (out_c,)).astype(dtype)
        in_c = out_c
    H = rnn_cfg["dim"]
    bi = bool(rnn_cfg["bidirectional"])
    isz = in_c * conv_out_size(input_dim, conv_cfg, 1)
    k = 1.0 / math.sqrt(H)
    for l in range(rnn_cfg["layers"]):
        for rev in ([False, True] if bi else [False]):
            sfx = _sfx(l, rev)
            P["rnn.weight_ih" + sfx] = rng.uniform(-k, k, (3 * H, isz)).astype(dtype)
            P["rnn.weight_hh" + sfx] = rng.uniform(-k, k, (3 * H, H)).astype(dtype)
            P["rnn.bias_ih" + sfx] = rng.uniform(-k, k, (3 * H,)).astype(dtype)
            P["rnn.bias_hh" + sfx] = rng.uniform(-k, k, (3 * H,)).astype(dtype)
        isz = H * (2 if bi else 1)
    k = 1.0 / math.sqrt(H)
    P["fc.fc.weight"] = rng.uniform(-k, k, (output_dim + 1, H)).astype(dtype)
    P["fc.fc.bias"] = rng.uniform(-k, k, (output_dim + 1,)).astype(dtype)
    return P


def clip_and_sgd(P, G, lr, max_norm=200.0, momentum=0.0, bufs=None):
    """train.py:32,35: clip_grad_norm(params, 200) (scale by max_norm/(norm+1e-6) when that is < 1) then SGD."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in G.values()))
    coef = max_norm / (total + 1e-6)
    scale = coef if coef < 1.0 else 1.0
    out = {}
    for k, p in P.items():
        g = G[k] * scale
        if momentum != 0.0:
            if bufs is None:
                raise ValueError("momentum needs bufs")
            bufs[k] = g.copy() if k not in bufs else momentum * bufs[k] + g
            g = bufs[k]
        out[k] = p - lr * g
    return out, total
