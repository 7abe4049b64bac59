"""
oracle/gen_golden.py -- generate tests/golden/*.npz by running the LIVE reference (/root/reference) on CPU.
Run in the build container only (/root/reference does not exist on the GPU box):
    python oracle/gen_golden.py
The fixtures pin oracle/encoder_np.py and oracle/decoder_ref.py (and, through them, the HIP path) to the
reference's own Model.encode / CTC.forward / CTC.infer / ctc_decoder.decode outputs.

Import recipe: the reference imports four third-party modules that are absent here
(editdistance, soundfile, functions.ctc = warp-ctc binding, transducer); they are stubbed in sys.modules.
Nothing is copied from the reference; it is only executed.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    for name in ["editdistance", "soundfile", "functions", "functions.ctc", "transducer", "transducer.decoders",
                 "transducer.functions", "transducer.functions.transducer"]:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["functions.ctc"].CTCLoss = object
    sys.modules["functions"].ctc = sys.modules["functions.ctc"]
    sys.modules["transducer.functions.transducer"].TransducerLoss = object
    sys.modules["transducer.decoders"].decode_static = None
    sys.modules["transducer"].decoders = sys.modules["transducer.decoders"]
    sys.modules["transducer"].functions = sys.modules["transducer.functions"]
    sys.modules["transducer.functions"].transducer = sys.modules["transducer.functions.transducer"]
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import speech.models as models  # noqa
    from speech.models import ctc_decoder  # noqa
    return models, ctc_decoder


def encoder_case(models, name, freq_dim, vocab, cfg, B, T, seed):
    """Reference CTC model (random init under a fixed seed) on a seeded fake batch (tests/shared.py:18-26 shapes)."""
    torch.manual_seed(seed)
    np.random.seed(seed)
    model = models.CTC(freq_dim, vocab, cfg)
    model.eval()
    inputs = tuple(np.random.randn(T, freq_dim) for _ in range(B))
    labels = tuple(np.random.randint(0, vocab, 20) for _ in range(B))
    with torch.no_grad():
        logits = model((inputs, labels))                      # CTC.forward -> collate -> forward_impl
        x = torch.FloatTensor(models.model.zero_pad_concat(inputs))
        enc = model.encode(x)
        preds = model.infer((inputs, labels))                 # beam_size=1 prefix search, blank = vocab
    out = {"param." + k: v.numpy() for k, v in model.state_dict().items()}
    out["x"] = x.numpy()
    out["logits"] = logits.numpy()
    out["enc"] = enc.numpy()
    out["t_out"] = np.array(model.conv_out_size(T, 0))
    out["f_out"] = np.array(model.conv_out_size(freq_dim, 1))
    out["infer_flat"] = np.array([l for p in preds for l in p], dtype=np.int32)
    out["infer_lens"] = np.array([len(p) for p in preds], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "logits", logits.shape, "enc", enc.shape, "infer lens", out["infer_lens"])


def decoder_cases(ctc_decoder):
    cases = {}

    def add(name, probs, beam, blank):
        probs = np.asarray(probs)
        with np.errstate(divide="ignore"):
            labels, nll = ctc_decoder.decode(probs, beam_size=beam, blank=blank)
        cases[name + ".probs"] = probs
        cases[name + ".meta"] = np.array([beam, blank], dtype=np.int32)
        cases[name + ".labels"] = np.array(labels, dtype=np.int32)
        cases[name + ".nll"] = np.array(float(nll), dtype=np.float64)
        print(name, "beam", beam, "blank", blank, "len", len(labels), "nll", float(nll))

    # the reference's own demo vector (ctc_decoder.py:115-126): float64 probs, beam 10, blank 0
    np.random.seed(3)
    probs = np.random.rand(50, 20)
    probs = probs / np.sum(probs, axis=1, keepdims=True)
    add("demo", probs, 10, 0)

    rng = np.random.RandomState(2017)

    def softmax(z):
        z = z - z.max(axis=1, keepdims=True)
        e = np.exp(z)
        return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)

    for i, (T, S, scale, beam, blank) in enumerate([
            (60, 11, 1.0, 1, 10), (60, 11, 4.0, 1, 10), (80, 29, 4.0, 1, 28), (80, 29, 1.0, 1, 28),
            (60, 11, 4.0, 4, 10), (80, 29, 4.0, 8, 28), (80, 29, 2.0, 8, 28), (50, 20, 3.0, 10, 0),
            (40, 6, 8.0, 3, 5), (120, 29, 6.0, 8, 28), (120, 29, 6.0, 1, 28), (30, 5, 1.0, 16, 4)]):
        add("rand%d" % i, softmax(scale * rng.randn(T, S)), beam, blank)
    # hard zeros (log -> -inf) and exact ties
    p = softmax(3.0 * rng.randn(40, 8))
    p[::3, 2] = 0.0
    p = (p / p.sum(axis=1, keepdims=True)).astype(np.float32)
    add("zeros", p, 4, 7)
    add("zeros_b1", p, 1, 7)
    add("uniform", np.full((20, 5), 0.2, dtype=np.float32), 3, 4)
    add("uniform_b1", np.full((20, 5), 0.2, dtype=np.float32), 1, 4)
    onehot = np.zeros((12, 4), dtype=np.float32)
    onehot[np.arange(12), [0, 0, 3, 1, 1, 3, 1, 2, 2, 3, 3, 0]] = 1.0
    add("onehot", onehot, 2, 3)
    add("onehot_b1", onehot, 1, 3)
    add("single_frame", softmax(rng.randn(1, 7)), 5, 6)
    np.savez_compressed(os.path.join(OUT, "decoder.npz"), **cases)


def specgram_case():
    """The reference's own log_specgram (speech/loader.py:156-166) on a seeded synthetic int16 signal."""
    import speech.loader as ref_loader
    rng = np.random.RandomState(2017)
    n = 16000 + 377
    t = np.arange(n) / 16000.0
    audio = (4000 * np.sin(2 * np.pi * 440 * t) + 1500 * np.sin(2 * np.pi * 1234.5 * t) + 300 * rng.randn(n))
    audio = audio.astype(np.int16)
    feats = ref_loader.log_specgram(audio, 16000)
    np.savez_compressed(os.path.join(OUT, "specgram.npz"), audio=audio, feats=feats, sample_rate=np.array(16000))
    print("specgram", feats.shape, feats.dtype)


def transducer_case(models):
    """The reference's own Transducer (transducer_model.py) under a fixed seed: parameters, a seeded fake batch
    (tests/shared.py:18-26 shapes) and the LOG-softmax lattice Transducer.forward returns.  The loss and the decoder
    live in the un-vendored `transducer` package and are NOT exercised (parity unpinned there)."""
    from speech.models import transducer_model
    torch.manual_seed(2017)
    np.random.seed(2017)
    cfg = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2]], "rnn": {"dim": 16, "bidirectional": True, "layers": 2}},
           "decoder": {"embedding_dim": 12, "layers": 2}}
    freq_dim, vocab, B, T = 40, 10, 3, 61
    model = transducer_model.Transducer(freq_dim, vocab, cfg)
    model.eval()
    inputs = tuple(np.random.randn(T - 7 * i, freq_dim) for i in range(B))
    labels = tuple(np.random.randint(0, vocab, 6 + 2 * i) for i in range(B))
    with torch.no_grad():
        out = model((inputs, labels))
    res = {"param." + k: v.numpy() for k, v in model.state_dict().items()}
    res["x"] = models.model.zero_pad_concat(inputs)
    res["y_mat"] = model.label_collate(labels).numpy()
    res["labels_flat"] = np.array([l for lab in labels for l in lab], dtype=np.int32)
    res["label_lens"] = np.array([len(l) for l in labels], dtype=np.int32)
    res["out"] = out.numpy()
    np.savez_compressed(os.path.join(OUT, "transducer_tiny.npz"), **res)
    print("transducer_tiny out", out.shape)


def seq2seq_case(models):
    """The reference's own Seq2Seq (speech/models/seq2seq.py) under the seeds of its test (tests/seq2seq_test.py:15-16):
    parameters, a seeded fake batch with start / end tokens, the teacher-forced logits of Seq2Seq.forward, the loss of
    Seq2Seq.loss, every parameter gradient from the reference's own autograd, the alignments, and the greedy decode of
    Seq2Seq.infer.  Everything on this path is in the reference tree, so these vectors PIN it."""
    from speech.models import seq2seq
    torch.manual_seed(1337)
    np.random.seed(1337)
    cfg = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2]], "rnn": {"dim": 16, "bidirectional": True, "layers": 2}},
           "decoder": {"embedding_dim": 16, "layers": 1, "log_t": True}}
    freq_dim, vocab, B, T = 40, 10, 3, 61
    model = seq2seq.Seq2Seq(freq_dim, vocab + 1, cfg)
    model.train()
    inputs = tuple(np.random.randn(T - 7 * i, freq_dim).astype(np.float32) for i in range(B))
    # start token = vocab (never predicted, seq2seq.py:33-35), end token = vocab - 1 (last item of every label)
    labels = tuple([vocab] + list(np.random.randint(0, vocab - 1, 5 + 2 * i)) + [vocab - 1] for i in range(B))
    x, y = model.collate(inputs, labels)
    out, alis = model.forward_impl(x, y)
    loss = model.loss((inputs, labels))
    loss.backward()
    res = {"param." + k: v.numpy() for k, v in model.state_dict().items()}
    for n, p in model.named_parameters():
        res["grad." + n] = p.grad.numpy()
    res["x"], res["y"] = x.numpy(), y.numpy()
    res["out"], res["aligns"], res["loss"] = out.detach().numpy(), alis.detach().numpy(), np.array(float(loss))
    model.set_eval()
    with torch.no_grad():
        res["infer"] = np.array(model.infer((inputs, labels), max_len=12), dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "seq2seq_tiny.npz"), **res)
    print("seq2seq_tiny out", out.shape, "loss", float(loss), "infer", res["infer"].shape)


def seq2seq_beam_cases(models):
    """The reference's own Seq2Seq.beam_search (speech/models/seq2seq.py:180-227) run LIVE on CPU, beam 1 / 4 / 8 / 10.
    The one line that does not run on python 3 -- `filter(...)[:beam_size]` (:213-214) -- is served by a module-level
    `filter` that returns the list python 2 returned; nothing else is touched.  Models: random init (flat
    distributions: long searches that end at max_len), the same with the output layer scaled up (peaked: hypotheses
    complete and the stopping rule fires), and with DUPLICATED output rows (two classes -- one of them the end token in
    the last model -- get bit-identical log-probabilities in any implementation, so the result depends on the
    reference's tie order).  Stored per case: the hypothesis, its score (recomputed along the hypothesis with the
    reference's decode_step), and the restatement's diagnostics; oracle/seq2seq_beam_ref.py driven by the live
    decode_step must reproduce every hypothesis (asserted here)."""
    import builtins
    from speech.models import seq2seq
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import seq2seq_beam_ref as R
    seq2seq.filter = lambda f, xs: list(builtins.filter(f, xs))
    res = {}
    names = []

    def make(tag, freq_dim, vocab, dim, conv, T, seed, fc_scale=1.0, dup=()):
        torch.manual_seed(seed)
        rng = np.random.RandomState(seed)
        cfg = {"dropout": 0.0, "encoder": {"conv": conv, "rnn": {"dim": dim, "bidirectional": True, "layers": 2}},
               "decoder": {"embedding_dim": dim, "layers": 1, "log_t": True}}
        model = seq2seq.Seq2Seq(freq_dim, vocab + 1, cfg)
        with torch.no_grad():
            model.fc.fc.weight.mul_(fc_scale)
            model.fc.fc.bias.mul_(fc_scale)
            for a, b in dup:                       # class b := class a
                model.fc.fc.weight[b] = model.fc.fc.weight[a]
                model.fc.fc.bias[b] = model.fc.fc.bias[a]
        model.set_eval()
        inputs = (rng.randn(T, freq_dim).astype(np.float32),)
        labels = ([vocab, 0, vocab - 1],)           # start token = vocab, end token = vocab - 1 (only these are read)
        for k, v in model.state_dict().items():
            res["%s.param.%s" % (tag, k)] = v.numpy()
        res[tag + ".x"] = inputs[0]
        res[tag + ".cfg"] = np.array([freq_dim, vocab, dim, T], dtype=np.int64)
        res[tag + ".conv"] = np.array(conv, dtype=np.int64)
        return model, (inputs, labels)

    def run(tag, model, batch, beam, max_len):
        name = "%s.b%d.m%d" % (tag, beam, max_len)
        with torch.no_grad():
            hyp = model.beam_search(batch, beam_size=beam, max_len=max_len)[0]
        hyp = tuple(int(t) for t in hyp)
        x, y = model.collate(*batch)
        with torch.no_grad():
            enc = model.encode(x)
        start, end = int(y[0, 0]), int(y[0, -1])
        hyp2, score, info = R.beam_search(R.torch_step_fn(model, enc), start, end, beam, max_len)
        assert hyp2 == hyp, (name, hyp, hyp2)
        res[name + ".hyp"] = np.array(hyp, dtype=np.int64)
        res[name + ".score"] = np.array(score, dtype=np.float64)
        res[name + ".info"] = np.array([info["steps"], info["n_complete"]], dtype=np.int64)
        res[name + ".min_margin"] = np.array(info["min_margin"], dtype=np.float64)
        names.append(name)
        print(name, "len", len(hyp), "score %.4f" % score, info)

    conv = [[8, 5, 11, 2]]
    m, b = make("flat", 40, 10, 16, conv, 61, 1337)
    for beam, ml in ((1, 12), (4, 12), (8, 20), (10, 15)):
        run("flat", m, b, beam, ml)
    m, b = make("peaked", 40, 10, 16, conv, 61, 7, fc_scale=40.0)
    for beam, ml in ((1, 30), (4, 30), (8, 30), (10, 30)):
        run("peaked", m, b, beam, ml)
    m, b = make("tie", 40, 10, 16, conv, 75, 11, fc_scale=30.0, dup=((2, 5), (1, 7)))
    for beam, ml in ((1, 25), (4, 25), (8, 25)):
        run("tie", m, b, beam, ml)
    m, b = make("tie_end", 40, 10, 16, conv, 75, 23, fc_scale=30.0, dup=((9, 3),))   # the end token (9) ties class 3
    for beam, ml in ((1, 25), (4, 25), (8, 25)):
        run("tie_end", m, b, beam, ml)
    m, b = make("wide", 40, 30, 32, [[8, 5, 11, 2]], 90, 5, fc_scale=25.0)
    for beam, ml in ((4, 40), (8, 40), (10, 40)):
        run("wide", m, b, beam, ml)
    res["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "seq2seq_beam.npz"), **res)


def main():
    os.makedirs(OUT, exist_ok=True)
    models, ctc_decoder = import_reference()
    transducer_case(models)
    seq2seq_case(models)
    seq2seq_beam_cases(models)
    specgram_case()
    sys.path.insert(0, os.path.join(REF, "tests"))
    import shared  # the reference's own test config (tests/shared.py:4-16)
    encoder_case(models, "encoder_tiny", 40, 10, shared.model_config, B=4, T=100, seed=2017)
    cfg_bi = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2], [8, 3, 7, 1]],
                                          "rnn": {"dim": 24, "bidirectional": True, "layers": 2}}}
    encoder_case(models, "encoder_bi2", 40, 12, cfg_bi, B=3, T=61, seed=7)
    cfg_uni3 = {"dropout": 0.0, "encoder": {"conv": [[16, 5, 32, 2]],
                                            "rnn": {"dim": 32, "bidirectional": False, "layers": 3}}}
    encoder_case(models, "encoder_uni3", 80, 28, cfg_uni3, B=2, T=75, seed=11)
    decoder_cases(ctc_decoder)


if __name__ == "__main__":
    main()
