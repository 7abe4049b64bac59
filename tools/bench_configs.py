#!/usr/bin/env python
"""tools/bench_configs.py -- every workload SURVEY.md 8(d) names, measured on one MI355X (GPU legs only).

  M-CTC   CTC loss fwd+bwd on (B, 1000, 29) logits: B=32 L=100; B=32 L~U{50..150}; B=4096 L=100 (throughput regime)
  M-STEP  full train step, S-LIBRI unidirectional (the bench.py workload) and its bidirectional variant
  M-TIMIT shipped examples/timit/ctc_config.json shapes (F=161, V+1=49, B=8, T=300, 4 x biGRU-256... as shipped)
          and the BASELINE-described 2 x GRU-256, F=40, |V|=61 model at B=32, T=1000
  M-DEC   prefix beam search on softmax(4 randn) (32, 498, 29), beam 1 and beam 8; greedy decode

Prints one JSON object; bench.py stays the one-line headline.  Inputs resident in HBM, synthetic, seed 2017.
"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from speech_amd import decoder, ops  # noqa: E402
from speech_amd.ctc import CTCLabels, CTCLoss, ctc_loss_raw  # noqa: E402
from speech_amd.models import CTC  # noqa: E402

DEV = torch.device("cuda", 0)


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def m_ctc(B, T, K, lens, iters):
    rng = np.random.RandomState(2017)
    logits = torch.from_numpy(rng.randn(B, T, K).astype(np.float32)).to(DEV)
    lens = np.asarray(lens, np.int32)
    labels = rng.randint(0, K - 1, int(lens.sum())).astype(np.int32)
    lab = CTCLabels(labels, np.full(B, T, np.int32), lens, DEV)
    sec = timed(lambda: ctc_loss_raw(logits, lab, blank=K - 1), iters)
    alg = 2 * B * T * K * 4 + labels.nbytes + B * 4
    return {"B": B, "T": T, "K": K, "L": "%d..%d" % (lens.min(), lens.max()), "ms": sec * 1e3,
            "utt_per_s": B / sec, "algorithmic_GBps": alg / sec / 1e9}


def m_step(name, F, V, cfg, B, T, L, iters):
    torch.manual_seed(2017)
    model = CTC(F, V, cfg).cuda()
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    rng = np.random.RandomState(2017)
    x = torch.from_numpy(rng.randn(B, T, F).astype(np.float32)).to(DEV)
    Tp = model.conv_out_size(T, 0)
    lab = CTCLabels(rng.randint(0, V, B * L).astype(np.int32), np.full(B, Tp, np.int32), np.full(B, L, np.int32), DEV)
    loss_fn = CTCLoss(denom=B)
    norm = torch.zeros(1, device=DEV)
    out = {}

    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model.forward_impl(x), lab, None, None)
        ops.backward(loss)  # train.py:51 (loss.backward() without autograd's two scalar launches)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        out["loss"] = loss

    sec = timed(step, iters, warmup=3)
    with torch.no_grad():
        model.set_eval()
        fwd = timed(lambda: model.forward_impl(x), iters, warmup=1)
    return {"workload": name, "B": B, "T": T, "T_out": Tp, "F": F, "classes": V + 1,
            "params": int(flat_p.numel()), "train_step_ms": sec * 1e3, "train_utt_per_s": B / sec,
            "forward_ms": fwd * 1e3, "loss": float(out["loss"].item())}


def m_transducer(iters, dropout=0.0):
    """BASELINE config 5: RNN-Transducer on the S-LIBRI encoder (4 x GRU-512 uni), 1-layer prediction network,
    B=32, T=1000 -> T'=498, U=100: the loss alone on a (32, 498, 101, 29) lattice, and the full train step."""
    from speech_amd.models import Transducer
    from speech_amd.transducer import TransducerLabels, transducer_loss_raw
    B, T, F, V, L = 32, 1000, 80, 28, 100
    cfg = {"dropout": dropout, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": False}},
           "decoder": {"embedding_dim": 256, "layers": 1}}
    torch.manual_seed(2017)
    model = Transducer(F, V, cfg).cuda()
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    rng = np.random.RandomState(2017)
    inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
    labels = tuple(rng.randint(0, V, L) for _ in range(B))
    Tp = model.conv_out_size(T, 0)
    lat = torch.log_softmax(torch.from_numpy(rng.randn(B, Tp, L + 1, V + 1).astype(np.float32)).to(DEV), dim=3)
    lab = TransducerLabels(np.concatenate(labels).astype(np.int32), np.full(B, Tp, np.int32), np.full(B, L, np.int32), DEV)
    loss_ms = timed(lambda: transducer_loss_raw(lat, lab), 20) * 1e3
    norm = torch.zeros(1, device=DEV)
    out = {}

    def step():
        model.zero_grad(set_to_none=True)
        loss = model.loss((inputs, labels))
        ops.backward(loss)  # train.py:51 (loss.backward() without autograd's two scalar launches)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        out["loss"] = loss

    sec = timed(step, iters, warmup=4)
    alg = 2 * lat.numel() * 4
    return {"workload": "RNN-T S-LIBRI (config 5)" + (", dropout %g" % dropout if dropout else ""), "B": B, "T_out": Tp, "U": L, "classes": V + 1,
            "params": int(flat_p.numel()), "loss_fwd_bwd_ms": loss_ms, "loss_algorithmic_GBps": alg / loss_ms / 1e6,
            "train_step_ms": sec * 1e3, "train_utt_per_s": B / sec, "loss": float(out["loss"].item()),
            "note": "train step includes the host-side collate of the reference's loss(batch) API"}


def m_seq2seq(iters, dropout=0.0):
    """BASELINE config 4: examples/wsj/seq2seq_config.json shapes (conv [[32,5,8,2],[32,5,8,2]], 4 x biGRU-256, F=161,
    batch 16, log_t, sample_prob 0.2), T=800 frames -> T'=197, 100 output tokens; train step, greedy infer, beam 8."""
    import random
    from speech_amd.models import Seq2Seq
    B, T, F, V, U = 16, 800, 161, 30, 100
    cfg = {"dropout": dropout, "encoder": {"conv": [[32, 5, 8, 2], [32, 5, 8, 2]],
                                       "rnn": {"dim": 256, "layers": 4, "bidirectional": True}},
           "decoder": {"sample_prob": 0.2, "embedding_dim": 256, "log_t": True, "layers": 1}}
    torch.manual_seed(2017)
    random.seed(2017)
    model = Seq2Seq(F, V + 2, cfg).cuda()     # + start and end tokens
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    rng = np.random.RandomState(2017)
    inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
    labels = tuple([V + 1] + list(rng.randint(0, V, U - 2)) + [V] for _ in range(B))
    norm = torch.zeros(1, device=DEV)
    out = {}

    def step():
        model.zero_grad(set_to_none=True)
        loss = model.loss((inputs, labels))
        ops.backward(loss)  # train.py:51 (loss.backward() without autograd's two scalar launches)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        out["loss"] = loss

    sec = timed(step, iters, warmup=4)
    model.set_eval()
    one = (inputs[:1], labels[:1])
    greedy = timed(lambda: model.infer((inputs, labels), max_len=U), 2, warmup=1)
    beam = timed(lambda: model.beam_search(one, beam_size=8, max_len=U), 1, warmup=1)
    return {"workload": "Seq2Seq WSJ config shapes (config 4)" + (", dropout %g" % dropout if dropout else ""), "B": B, "T": T, "T_out": model.conv_out_size(T, 0),
            "tokens": U, "params": int(flat_p.numel()), "train_step_ms": sec * 1e3, "train_utt_per_s": B / sec,
            "greedy_infer_batch_ms": greedy * 1e3, "beam8_one_utt_ms": beam * 1e3, "loss": float(out["loss"].item())}


def m_dec(beam, iters):
    rng = np.random.RandomState(2017)
    z = torch.from_numpy((4.0 * rng.randn(32, 498, 29)).astype(np.float32)).to(DEV)
    probs = torch.softmax(z, dim=2)
    sec = timed(lambda: decoder.beam_decode(probs, beam_size=beam, blank=28), iters, warmup=1)
    return {"beam": beam, "B": 32, "T": 498, "S": 29, "ms": sec * 1e3, "utt_per_s": 32 / sec}


def main():
    # --only M-STEP,M-TIMIT,... restricts the run (GPU minutes are budgeted)
    only = None
    for i, a in enumerate(sys.argv):
        if a == "--only" and i + 1 < len(sys.argv):
            only = set(sys.argv[i + 1].split(","))
    want = lambda k: only is None or k in only  # noqa: E731
    res = {"device": torch.cuda.get_device_name(0)}
    rng = np.random.RandomState(7)
    if want("M-CTC"):
        res["M-CTC"] = [m_ctc(32, 1000, 29, [100] * 32, 50),
                        m_ctc(32, 1000, 29, rng.randint(50, 151, 32), 50),
                        m_ctc(4096, 1000, 29, [100] * 4096, 5)]
    uni = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                       "rnn": {"dim": 512, "layers": 4, "bidirectional": False}}}
    bi = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                      "rnn": {"dim": 512, "layers": 4, "bidirectional": True}}}
    # examples/timit/ctc_config.json:18-32 (conv 2 layers stride 2 then 1, 4 x biGRU-256... shapes per SURVEY 8d)
    timit = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]],
                                         "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}
    small = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                         "rnn": {"dim": 256, "layers": 2, "bidirectional": False}}}
    # the configs AS SHIPPED train with dropout (examples/timit/ctc_config.json:20 0.4, timit/transducer_config.json:20
    # 0.5, wsj/seq2seq_config.json:20 0.2): every workload is measured without it (round-2 comparability) AND with it
    drop = lambda cfg, p: dict(cfg, dropout=p)  # noqa: E731
    if want("M-STEP"):
        res["M-STEP"] = [m_step("S-LIBRI uni (bench.py)", 80, 28, uni, 32, 1000, 100, 5),
                         m_step("S-LIBRI uni, dropout 0.2", 80, 28, drop(uni, 0.2), 32, 1000, 100, 5),
                         m_step("S-LIBRI bidirectional", 80, 28, bi, 32, 1000, 100, 5),
                         m_step("S-LIBRI bidirectional, dropout 0.2", 80, 28, drop(bi, 0.2), 32, 1000, 100, 5)]
    if want("M-TIMIT"):
        res["M-TIMIT"] = [m_step("timit ctc_config shapes", 161, 48, timit, 8, 300, 40, 5),
                          m_step("timit ctc_config AS SHIPPED (dropout 0.4)", 161, 48, drop(timit, 0.4), 8, 300, 40, 5),
                          m_step("2xGRU-256 F=40 |V|=61", 40, 61, small, 32, 1000, 100, 5)]
    if want("M-RNNT"):
        res["M-RNNT"] = [m_transducer(8), m_transducer(8, 0.5)]
    if want("M-S2S"):
        res["M-S2S"] = [m_seq2seq(8), m_seq2seq(8, 0.2)]
    if want("M-DEC"):
        rng = np.random.RandomState(2017)
        z = torch.from_numpy((4.0 * rng.randn(32, 498, 29)).astype(np.float32)).to(DEV)
        greedy = timed(lambda: decoder.greedy_decode(z, blank=28), 20)
        res["M-DEC"] = [m_dec(1, 5), m_dec(8, 3), {"greedy": True, "ms": greedy * 1e3, "utt_per_s": 32 / greedy}]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
