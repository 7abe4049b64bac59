"""End-to-end parity AT the BASELINE configurations themselves: the full speech_amd.models.CTC
(default kernel selection: fused forward wavefront + XCD-local persistent backward where eligible) against
oracle/torch_ref.TorchRefCTC -- the reference's own torch.nn CPU modules (/root/reference/speech/models/model.py:12-79,
ctc_model.py:17-40) + the C CTC restatement -- built from the SAME state dict, on the same seeded batch.

  (i)   S-LIBRI headline (BASELINE.json metric config): conv [32,5,32,2], 4 x GRU-512 uni, F=80, |V|+1=29, B=32, T=1000
  (ii)  the shipped TIMIT config (/root/reference/examples/timit/ctc_config.json:18-32): 2 convs, 4 x biGRU-256,
        F=161, |V|+1=49, B=8, T=300
  (iii) BASELINE config 2: 2 x GRU-256, F=40, |V|+1=62, B=32, T=1000

  (iv)  S-LIBRI bidirectional (SURVEY 8d "report also the bidirectional variant"): 4 x biGRU-512, 18.2 M parameters
  (v)   the configs AS SHIPPED, i.e. with their dropout: examples/timit/ctc_config.json:20 trains with
        dropout 0.4; S-LIBRI with 0.2 (examples/wsj/seq2seq_config.json:20's value).  The masks are generated inside the
        HIP kernels from a Philox key (csrc/dropout.h); the oracle gets the SAME masks from oracle/philox_ref.py.

Tolerances (fp32 HIP kernels vs fp32 torch CPU, both summing in different orders): loss rtol 1e-4 (north_star), every
parameter gradient within 1e-3 of that tensor's max magnitude AND within 1e-3 in relative L2 norm (a wrong
small-magnitude row cannot hide under the max), logits within 2e-4 of the logit range."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

def _with_dropout(case, p):
    c = dict(case)
    c["cfg"] = dict(case["cfg"], dropout=p)
    return c


CASES = {
    "s_libri": dict(F=80, V=28, B=32, T=1000, L=100,
                    cfg={"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                                     "rnn": {"dim": 512, "layers": 4, "bidirectional": False}}}),
    "timit_shipped": dict(F=161, V=48, B=8, T=300, L=40,
                          cfg={"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]],
                                                           "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}),
    "config2": dict(F=40, V=61, B=32, T=1000, L=100,
                    cfg={"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                                     "rnn": {"dim": 256, "layers": 2, "bidirectional": False}}}),
    "s_libri_bi": dict(F=80, V=28, B=32, T=1000, L=100,
                       cfg={"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                                        "rnn": {"dim": 512, "layers": 4, "bidirectional": True}}}),
}
CASES["timit_shipped_dropout0.4"] = _with_dropout(CASES["timit_shipped"], 0.4)
CASES["s_libri_dropout0.2"] = _with_dropout(CASES["s_libri"], 0.2)
MASK_SEED = (1 << 50) + 2017


def _masks(case):
    from oracle import philox_ref
    from speech_amd.ops import conv_out_size
    cfg = case["cfg"]
    if not cfg["dropout"]:
        return None
    shapes, t, f = [], case["T"], case["F"]
    for out_c, kh, kw, s in cfg["encoder"]["conv"]:
        t, f = conv_out_size(t, kh, s), conv_out_size(f, kw, s)
        shapes.append((case["B"], out_c, t, f))
    rnn = cfg["encoder"]["rnn"]
    D = 2 if rnn["bidirectional"] else 1
    return philox_ref.encoder_masks(cfg["dropout"], MASK_SEED, shapes, (case["B"], t, D * rnn["dim"]), rnn["layers"])


def _reference_step(case, state, x, labels, label_lens):
    from oracle.torch_ref import TorchRefCTC, _CTCRef
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, prev))
    try:
        ref = TorchRefCTC(case["F"], case["V"], case["cfg"])
        ref.load_state_dict(state)
        logits = ref(torch.from_numpy(x), _masks(case))
        B, Tp, _ = logits.shape
        loss = _CTCRef.apply(logits, labels, np.full(B, Tp, np.int32), label_lens, ref.blank, 0)
        loss.backward()
        grads = {k: p.grad.detach().numpy() for k, p in ref.named_parameters()}
        return float(loss.item()), logits.detach().numpy(), grads
    finally:
        torch.set_num_threads(prev)


@pytest.mark.parametrize("name", sorted(CASES))
def test_loss_and_gradients_at_baseline_config(name):
    from speech_amd import _lib
    from speech_amd.models import CTC
    case = CASES[name]
    F, V, B, T, L = case["F"], case["V"], case["B"], case["T"], case["L"]
    torch.manual_seed(2017)
    model = CTC(F, V, case["cfg"])
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    rng = np.random.RandomState(2017)
    x = rng.randn(B, T, F).astype(np.float32)
    labels = tuple(rng.randint(0, V, L) for _ in range(B))
    batch = (tuple(x[b] for b in range(B)), labels)
    model.set_train()
    model._plan.fixed_seed = MASK_SEED  # dropout cases: the masks the oracle is given
    out = model(batch)
    loss = model.loss(batch)
    loss.backward()
    assert _lib.lib().sa_gru_persist_status() == 0  # the XCD-local persistent kernels ran clean
    flat = np.concatenate(labels).astype(np.int32)
    want_loss, want_logits, want_grads = _reference_step(case, state, x, flat, np.full(B, L, np.int32))
    got_logits = out.detach().cpu().numpy()
    assert got_logits.shape == want_logits.shape
    span = float(want_logits.max() - want_logits.min())
    assert np.abs(got_logits - want_logits).max() <= 2e-4 * span
    assert abs(float(loss.item()) - want_loss) <= 1e-4 * abs(want_loss), (float(loss.item()), want_loss)
    assert set(want_grads) == {k for k, _ in model.named_parameters()}
    for k, p in model.named_parameters():
        got = p.grad.cpu().numpy()
        scale = max(float(np.abs(want_grads[k]).max()), 1e-10)
        err = float(np.abs(got - want_grads[k]).max())
        rel = float(np.linalg.norm((got - want_grads[k]).ravel()) / max(np.linalg.norm(want_grads[k].ravel()), 1e-20))
        assert err <= 1e-3 * scale and rel <= 1e-3, (name, k, err, scale, rel)


# ---- BASELINE configs 4 and 5 at their real sizes (VERDICT r03 missing #2) --------------------------------------------
def _check_grads(name, model, want):
    assert set(want) == {k for k, _ in model.named_parameters()}
    for k, p in model.named_parameters():
        got = p.grad.cpu().numpy()
        if k == "attend.nn.1.fc.bias":
            # a constant added to every score before the softmax over time (seq2seq.py:350-354): its gradient is exactly
            # zero; the oracle returns 1e-17 (fp64), the HIP path fp32 rounding noise of the score gradients' sum
            assert np.abs(want[k]).max() < 1e-12 and np.abs(got).max() < 1e-6, (name, k, got, want[k])
            continue
        scale = max(float(np.abs(want[k]).max()), 1e-10)
        err = float(np.abs(got - want[k]).max())
        rel = float(np.linalg.norm((got - want[k]).ravel()) / max(np.linalg.norm(want[k].ravel()), 1e-20))
        # The location-aware attention's 1-D conv (seq2seq.py:331-348): its gradient is a sum over B x U x T' score gradients of
        # both signs that all but cancels -- max 3e-5 where the scores' own gradients are 1e-2 -- so fp32 rounding of the terms
        # (5e-8) is 1.5e-3 of its max while every other tensor sits below 1e-4 of theirs.  Its relative L2 error (6e-4) is
        # held to the common 1e-3; only the max-norm bound of THIS tensor is 3e-3 (it sat at 0.7e-3 - 1e-3 of max with the f32
        # forward kernels; the planes kernels move the encoder states by 1e-7, inside their own fp64 budget).
        max_tol = 3e-3 if k == "attend.conv.weight" else 1e-3
        assert err <= max_tol * scale and rel <= 1e-3, (name, k, err, scale, rel)


S2S_WSJ = dict(F=161, V=30, B=16, T=800, U=100,
               cfg={"dropout": 0.2, "encoder": {"conv": [[32, 5, 8, 2], [32, 5, 8, 2]],
                                                "rnn": {"dim": 256, "layers": 4, "bidirectional": True}},
                    "decoder": {"sample_prob": 0.2, "embedding_dim": 256, "log_t": True, "layers": 1}})


@pytest.mark.parametrize("sampling", [False, True])
def test_seq2seq_at_the_shipped_wsj_config(sampling):
    """BASELINE config 4 = /root/reference/examples/wsj/seq2seq_config.json:18-38 AS SHIPPED: 2 convs [32,5,8,2],
    4 x biGRU-256, dropout 0.2 (shared Philox masks), NNAttention with log_t, 16 utterances of 800 frames (T' = 197),
    100 tokens; teacher forced, and with the config's scheduled sampling (sample_prob 0.2) on the SAME draws of python's
    RNG (seq2seq.py:91-96: token t > 0 reads the argmax of token t-1's logits where random.random() < sample_prob).
    Against oracle/torch_ref.TorchRefSeq2Seq in fp64 (pinned to the live reference, tests/test_oracle_seq2seq.py): loss rtol 1e-4,
    logits within 2e-4 of their range, every parameter gradient within 1e-3 of its max and 1e-3 in relative L2."""
    import random
    from oracle.torch_ref import TorchRefSeq2Seq
    from speech_amd import _lib
    from speech_amd.models import Seq2Seq
    c = S2S_WSJ
    F, V, B, T, U = c["F"], c["V"], c["B"], c["T"], c["U"]
    cfg = dict(c["cfg"], decoder=dict(c["cfg"]["decoder"], sample_prob=0.2 if sampling else 0))
    torch.manual_seed(2017)
    model = Seq2Seq(F, V + 2, cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    rng = np.random.RandomState(2017)
    x = rng.randn(B, T, F).astype(np.float32)
    labels = tuple([V + 1] + list(rng.randint(0, V, U - 2)) + [V] for _ in range(B))
    batch = (tuple(x[b] for b in range(B)), labels)
    model.set_train()
    assert model.scheduled_sampling == sampling
    model._plan.fixed_seed = MASK_SEED
    random.seed(99)
    flags = [0] + [1 if random.random() < 0.2 else 0 for _ in range(U - 2)] if sampling else None
    random.seed(99)
    xs, ys = model.collate(*batch)
    out, _ = model.forward_impl(xs.cuda(), ys.cuda())
    random.seed(99)
    loss = model.loss(batch)
    loss.backward()
    assert _lib.lib().sa_gru_persist_status() == 0
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, prev))
    try:
        # fp64 oracle: the location filter's gradient is a sum of ~3e5 terms that cancel to 3e-5 of their size -- against
        # an fp32 CPU sum both sides' rounding shows (1.3e-3 of the maximum); against fp64 only the HIP path's does
        ref = TorchRefSeq2Seq(F, V + 2, cfg).double()
        ref.load_state_dict({k: v.double() for k, v in state.items()})
        ref.train()
        masks = _masks(dict(cfg=cfg, T=T, F=F, B=B))
        xd = torch.from_numpy(x).double()
        want_out, _ = ref(xd, ys, masks, flags)
        want_loss = ref.loss(xd, ys, masks, flags)
        want_loss.backward()
    finally:
        torch.set_num_threads(prev)
    want_out = want_out.detach().numpy()
    got_out = out.detach().cpu().numpy()
    span = float(want_out.max() - want_out.min())
    assert np.abs(got_out - want_out).max() <= 2e-4 * span, np.abs(got_out - want_out).max() / span
    assert abs(float(loss.item()) - float(want_loss)) <= 1e-4 * abs(float(want_loss)), (float(loss.item()), float(want_loss))
    _check_grads("s2s_wsj", model, {k: p.grad.numpy() for k, p in ref.named_parameters()})


def test_transducer_at_baseline_config_5():
    """BASELINE config 5: RNN-Transducer on the S-LIBRI encoder (conv [32,5,32,2], 4 x GRU-512 uni), 1-layer prediction
    network (embedding 256), B=32, T=1000 -> T'=498, U=100, |V|+1=29, with the dropout the reference's only Transducer
    config trains with (/root/reference/examples/timit/transducer_config.json:20: 0.5; shared Philox masks).  The whole
    model -- (32, 498, 101, 29) lattice, loss, every parameter gradient -- against oracle/torch_ref.TorchRefTransducer
    (fp32 torch CPU modules; lattice pinned to the live reference at the tiny fixture) + the fp64 C loss."""
    from oracle import torch_ref
    from speech_amd import _lib
    from speech_amd.models import Transducer
    F, V, B, T, L = 80, 28, 32, 1000, 100
    cfg = {"dropout": 0.5, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": False}},
           "decoder": {"embedding_dim": 256, "layers": 1}}
    torch.manual_seed(2017)
    model = Transducer(F, V, cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    rng = np.random.RandomState(2017)
    x = rng.randn(B, T, F).astype(np.float32)
    labels = tuple(rng.randint(0, V, L) for _ in range(B))
    batch = (tuple(x[b] for b in range(B)), labels)
    model.set_train()
    model._plan.fixed_seed = MASK_SEED
    loss = model.loss(batch)
    loss.backward()
    assert _lib.lib().sa_gru_persist_status() == 0
    got_loss = float(loss.item())
    got = {k: p.grad.cpu().numpy() for k, p in model.named_parameters()}
    y_mat = model.label_collate(labels)
    Tp = model.conv_out_size(T, 0)
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, prev))
    try:
        ref = torch_ref.TorchRefTransducer(F, V, cfg)
        ref.load_state_dict(state)
        ref.train()
        masks = _masks(dict(cfg=cfg, T=T, F=F, B=B))
        want_loss = torch_ref.transducer_loss(ref, torch.from_numpy(x), y_mat, np.concatenate(labels).astype(np.int32),
                                              np.full(B, Tp, np.int32), np.full(B, L, np.int32), masks)
        want_loss.backward()
    finally:
        torch.set_num_threads(prev)
    want_loss = float(want_loss.item())
    assert abs(got_loss - want_loss) <= 1e-4 * abs(want_loss), (got_loss, want_loss)
    want = {k: p.grad.numpy() for k, p in ref.named_parameters()}
    assert set(want) == set(got)
    for k in want:
        scale = max(float(np.abs(want[k]).max()), 1e-10)
        err = float(np.abs(got[k] - want[k]).max())
        rel = float(np.linalg.norm((got[k] - want[k]).ravel()) / max(np.linalg.norm(want[k].ravel()), 1e-20))
        assert err <= 1e-3 * scale and rel <= 1e-3, ("rnnt", k, err, scale, rel)


def test_seq2seq_decode_at_the_shipped_wsj_config():
    """BASELINE config 4 is "seq2seq attention decoder, beam = 8": the decode paths at the shipped WSJ shapes (2 convs, 4 x
    biGRU-256, T = 800 -> T' = 197, 31 classes, eval mode).  Greedy `infer` of 4 utterances against
    oracle/torch_ref.TorchRefSeq2Seq.infer, and `beam_search` (beam 8 and the reference's default 10) of one utterance against
    oracle/seq2seq_beam_ref.py driven by the same restated model -- token for token.  The classifier is scaled up so that the
    distributions are peaked (a random-init model's are flat: every cut would sit inside fp32 noise); cases whose smallest
    score gap at a selection cut is below 2e-5 are not decidable at this precision and are skipped."""
    from oracle import seq2seq_beam_ref as R
    from oracle.torch_ref import TorchRefSeq2Seq
    from speech_amd.models import Seq2Seq
    c = S2S_WSJ
    F, V, T = c["F"], c["V"], c["T"]
    cfg = dict(c["cfg"], dropout=0.0, decoder=dict(c["cfg"]["decoder"], sample_prob=0))
    torch.manual_seed(2017)
    model = Seq2Seq(F, V + 2, cfg)
    with torch.no_grad():
        model.fc.fc.weight.mul_(30.0)
        model.fc.fc.bias.mul_(30.0)
        model.fc.fc.bias[V] += 8.0   # the end token becomes likely enough that hypotheses complete and the stopping rule
                                     # fires (17 search steps, 11 - 12 completed hypotheses, cuts 1e-3 apart on the CPU side)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    model.set_eval()
    ref = TorchRefSeq2Seq(F, V + 2, cfg)
    ref.load_state_dict(state)
    ref.eval()
    rng = np.random.RandomState(5)
    B = 4
    x = rng.randn(B, T, F).astype(np.float32)
    labels = tuple([V + 1, 0, V] for _ in range(B))   # start = V + 1, end = V
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, prev))
    try:
        got = model.infer((tuple(x[b] for b in range(B)), labels), max_len=40)
        with torch.no_grad():
            want = ref.infer(torch.from_numpy(x), torch.full((B, 1), V + 1, dtype=torch.int64), V, 40).numpy()
        assert np.array_equal(np.array(got), want), (got, want)
        decided = 0
        for beam in (8, 10):
            for b in range(2):
                with torch.no_grad():
                    enc = ref.encode(torch.from_numpy(x[b:b + 1]))
                hyp, score, info = R.beam_search(R.torch_step_fn(ref, enc), V + 1, V, beam, 40)
                if info["min_margin"] < 2e-5:
                    continue
                decided += 1
                got_h = model.beam_search(((x[b],), (labels[b],)), beam_size=beam, max_len=40)[0]
                assert got_h == hyp, (beam, b, got_h, hyp, info)
                assert abs(model.last_beam_score - score) < 2e-4 * max(1.0, abs(score))
                assert list(model.last_beam_info) == [info["steps"], info["n_complete"]]
        assert decided >= 2
    finally:
        torch.set_num_threads(prev)
