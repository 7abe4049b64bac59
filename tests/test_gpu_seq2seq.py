"""speech_amd.models.Seq2Seq (the reference's seq2seq.py on the HIP ops).  Nothing on this path is un-vendored, so the
LIVE reference pins it (tests/golden/seq2seq_tiny.npz, made by oracle/gen_golden.py): teacher-forced logits,
alignments, the loss, EVERY parameter gradient from the reference's own autograd, and the greedy decode.  Other shapes /
options go against oracle/torch_ref.TorchRefSeq2Seq (itself pinned to the same fixture, tests/test_oracle_seq2seq.py).
Tolerances: logits / alignments rtol 2e-4; loss rtol 1e-4; gradients within 1e-3 of the tensor's max magnitude."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import torch_ref

pytestmark = pytest.mark.gpu

CFG = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 11, 2]], "rnn": {"dim": 16, "bidirectional": True, "layers": 2}},
       "decoder": {"embedding_dim": 16, "layers": 1, "log_t": True}}


def fixture():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "seq2seq_tiny.npz"))


def build(g, cfg=CFG, flatten=False):
    from speech_amd.models import Seq2Seq
    m = Seq2Seq(40, 11, cfg)
    m.load_state_dict({k[len("param."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("param.")})
    m = m.cuda()
    if flatten:
        m.flatten_parameters_()
    return m


def batch_of(g):
    x, y = g["x"], g["y"]
    B = x.shape[0]
    # the fixture's padded tensors re-enter through the reference's batch format (inputs, labels)
    return tuple(x[b] for b in range(B)), tuple(list(y[b]) for b in range(B))


def test_logits_alignments_loss_gradients_match_live_reference():
    g = fixture()
    for flatten in (False, True):
        m = build(g, flatten=flatten)
        m.set_train()
        out, alis = m.forward_impl(torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda())
        np.testing.assert_allclose(out.detach().cpu().numpy(), g["out"], rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(alis.cpu().numpy(), g["aligns"], rtol=2e-4, atol=1e-6)
        m.zero_grad(set_to_none=True)
        loss = m.loss(batch_of(g))
        assert abs(float(loss.item()) - float(g["loss"])) < 1e-4 * float(g["loss"])
        loss.backward()
        for name, p in m.named_parameters():
            w = g["grad." + name]
            got = (p._grad_slot if flatten else p.grad).cpu().numpy()
            assert np.abs(got - w).max() < 1e-3 * max(np.abs(w).max(), 1e-3), (name, np.abs(got - w).max())


def test_decode_step_and_greedy_infer_match_live_reference():
    g = fixture()
    m = build(g)
    m.set_eval()
    x, y = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["y"]).cuda()
    with torch.no_grad():
        enc = m.encode(x)
        out, _ = m.forward_impl(x, y)
    # the reference's own test (tests/seq2seq_test.py:32-45): token-by-token decode_step reproduces forward
    state, outs = None, []
    for t in range(y.shape[1] - 1):
        o, state = m.decode_step(enc, y[:, t:t + 1], state=state)
        outs.append(o)
    np.testing.assert_allclose(torch.stack(outs, dim=1).cpu().numpy(), out.cpu().numpy(), rtol=1e-5, atol=1e-6)
    seqs = m.infer(batch_of(g), max_len=12)
    assert np.array_equal(np.array(seqs), g["infer"])
    assert len(m.predict(batch_of(g))) == 3


def test_other_shapes_against_restatement_and_beam_search():
    cfg = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 9, 2]], "rnn": {"dim": 24, "bidirectional": False, "layers": 1}},
           "decoder": {"embedding_dim": 24, "layers": 1, "log_t": False}}
    torch.manual_seed(5)
    ref = torch_ref.TorchRefSeq2Seq(20, 9, cfg)
    from speech_amd.models import Seq2Seq
    m = Seq2Seq(20, 9, cfg)
    m.load_state_dict(ref.state_dict())
    m = m.cuda()
    rng = np.random.RandomState(7)
    B, T = 5, 90   # T' = 43 > 15-tap window; 5 utterances of ragged label length
    inputs = tuple(rng.randn(T - 3 * i, 20).astype(np.float32) for i in range(B))
    labels = tuple([8] + list(rng.randint(0, 7, 3 + i)) + [7] for i in range(B))
    x, y = m.collate(inputs, labels)
    ref.train()
    loss_ref = ref.loss(x, y)
    loss_ref.backward()
    m.set_train()
    loss = m.loss((inputs, labels))
    loss.backward()
    assert abs(float(loss.item()) - float(loss_ref)) < 1e-4 * float(loss_ref)
    want = dict(ref.named_parameters())
    for name, p in m.named_parameters():
        w = want[name].grad.numpy()
        assert np.abs(p.grad.cpu().numpy() - w).max() < 1e-3 * max(np.abs(w).max(), 1e-3), name


def _beam_model(g, tag):
    from speech_amd.models import Seq2Seq
    freq_dim, vocab, dim, _ = [int(v) for v in g[tag + ".cfg"]]
    cfg = {"dropout": 0.0, "encoder": {"conv": g[tag + ".conv"].tolist(),
                                       "rnn": {"dim": dim, "bidirectional": True, "layers": 2}},
           "decoder": {"embedding_dim": dim, "layers": 1, "log_t": True}}
    m = Seq2Seq(freq_dim, vocab + 1, cfg)
    pre = tag + ".param."
    m.load_state_dict({k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)})
    m = m.cuda()
    m.set_eval()
    return m, vocab


def test_device_beam_search_equals_live_reference_hypotheses():
    """Seq2Seq.beam_search (sa_s2s_beam_search: the whole search on the device) against the hypotheses the LIVE
    reference's own beam_search returned (tests/golden/seq2seq_beam.npz, oracle/gen_golden.py; seq2seq.py:180-227):
    beam 1 / 4 / 8 / 10 on flat, peaked and EXACTLY tied output distributions (duplicated output rows: the winner then
    depends on the reference's (hypothesis rank, class) tie order; in `tie_end` the end token itself ties a class).
    Hypotheses must be identical; scores (double sums of float32 log-probabilities) within 1e-4; the number of search
    steps and of completed hypotheses -- i.e. the stopping rule -- identical.  Each case's smallest non-zero score gap at
    a selection cut is in the fixture (min_margin, 7.7e-5 at worst): well above the fp32 noise of the decoder step."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "seq2seq_beam.npz"))
    models = {}
    for name in g["names"]:
        name = str(name)
        tag, b, ml = name.split(".")
        if tag not in models:
            models[tag] = _beam_model(g, tag)
        m, vocab = models[tag]
        batch = ((g[tag + ".x"],), ([vocab, 0, vocab - 1],))
        for check_every in (8, 0):   # polling the stop flag every 8 tokens / never synchronising: same search
            from speech_amd import seq2seq as s2s
            x, y = m.collate(*batch)
            with torch.no_grad():
                enc = m.encode(x.cuda())
            hyp, score, info = s2s.beam_search(enc[0], m._param_dict(), m.attend.log_t, vocab, vocab - 1, int(b[1:]),
                                               int(ml[1:]), check_every=check_every)
            assert hyp == tuple(g[name + ".hyp"].tolist()), (name, check_every, hyp, g[name + ".hyp"].tolist())
            assert abs(score - float(g[name + ".score"])) < 1e-4, (name, score, float(g[name + ".score"]))
            assert list(info) == g[name + ".info"].tolist(), (name, info, g[name + ".info"].tolist())
        assert m.beam_search(batch, beam_size=int(b[1:]), max_len=int(ml[1:]))[0] == tuple(g[name + ".hyp"].tolist())


def test_device_beam_search_equals_restatement_on_other_shapes():
    """Shapes the fixtures do not hold (unidirectional encoder, no log_t, 33 classes, beams up to 16, a search that
    completes early) against oracle/seq2seq_beam_ref.py driven by oracle/torch_ref.TorchRefSeq2Seq."""
    from oracle import seq2seq_beam_ref as R
    from speech_amd.models import Seq2Seq
    for seed, dim, vocab, log_t, fc_scale in ((3, 24, 9, False, 20.0), (4, 32, 33, True, 30.0), (6, 64, 20, True, 12.0)):
        cfg = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 9, 2]],
                                           "rnn": {"dim": dim, "bidirectional": seed % 2 == 0, "layers": 1}},
               "decoder": {"embedding_dim": dim, "layers": 1, "log_t": log_t}}
        torch.manual_seed(seed)
        ref = torch_ref.TorchRefSeq2Seq(20, vocab + 1, cfg)
        with torch.no_grad():
            ref.fc.fc.weight.mul_(fc_scale)
            ref.fc.fc.bias.mul_(fc_scale)
        ref.eval()
        m = Seq2Seq(20, vocab + 1, cfg)
        m.load_state_dict(ref.state_dict())
        m = m.cuda()
        m.set_eval()
        rng = np.random.RandomState(seed)
        x = rng.randn(70 + 10 * seed, 20).astype(np.float32)
        with torch.no_grad():
            enc = ref.encode(torch.from_numpy(x)[None])
        for beam in (1, 3, 8, 16):
            want, score, info = R.beam_search(R.torch_step_fn(ref, enc), vocab, vocab - 1, beam, 30)
            got = m.beam_search(((x,), ([vocab, 0, vocab - 1],)), beam_size=beam, max_len=30)[0]
            if info["min_margin"] < 2e-5:
                continue   # a cut inside fp32 noise: the winner is not defined at this precision
            assert got == want, (seed, beam, got, want, info)
            assert abs(m.last_beam_score - score) < 1e-4 and list(m.last_beam_info) == [info["steps"], info["n_complete"]]


def test_device_beam_search_takes_wide_beams_and_word_piece_vocabularies():
    """the reference's beam_search takes any beam width and vocabulary (seq2seq.py:180); the device search
    kept the candidates' scores in 64 KB of LDS and refused beam_size > 32 or beam_size * classes > 8192 -- a word-piece
    vocabulary of 1000 with the default beam of 10 raised.  Now: a lane per live hypothesis (beam_size <= 64) and the
    candidate scores in the workspace beyond the LDS budget.  Against oracle/seq2seq_beam_ref.py."""
    from oracle import seq2seq_beam_ref as R
    from speech_amd.models import Seq2Seq
    for seed, dim, vocab, beams, fc_scale in ((11, 32, 1000, (10, 12), 40.0), (12, 32, 40, (33, 40, 64), 25.0)):
        cfg = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 9, 2]], "rnn": {"dim": dim, "bidirectional": False, "layers": 1}},
               "decoder": {"embedding_dim": dim, "layers": 1, "log_t": True}}
        torch.manual_seed(seed)
        ref = torch_ref.TorchRefSeq2Seq(20, vocab + 1, cfg)
        with torch.no_grad():
            ref.fc.fc.weight.mul_(fc_scale)
            ref.fc.fc.bias.mul_(fc_scale)
        ref.eval()
        m = Seq2Seq(20, vocab + 1, cfg)
        m.load_state_dict(ref.state_dict())
        m = m.cuda()
        m.set_eval()
        x = np.random.RandomState(seed).randn(90, 20).astype(np.float32)
        with torch.no_grad():
            enc = ref.encode(torch.from_numpy(x)[None])
        checked = 0
        for beam in beams:
            want, score, info = R.beam_search(R.torch_step_fn(ref, enc), vocab, vocab - 1, beam, 12)
            got = m.beam_search(((x,), ([vocab, 0, vocab - 1],)), beam_size=beam, max_len=12)[0]
            if info["min_margin"] < 2e-5:
                continue   # a cut inside fp32 noise: the winner is not defined at this precision
            checked += 1
            assert got == want, (seed, beam, got, want, info)
            assert abs(m.last_beam_score - score) < 1e-4 and list(m.last_beam_info) == [info["steps"], info["n_complete"]]
        assert checked > 0, (seed, "every case fell inside fp32 noise: pick another seed")
    with pytest.raises(Exception):  # beyond a wave's 64 lanes: a clear error, not a wrong search
        m.beam_search(((x,), ([vocab, 0, vocab - 1],)), beam_size=65, max_len=4)


def test_scheduled_sampling_consumes_the_rng_like_the_reference_and_trains():
    g = fixture()
    cfg = dict(CFG, decoder=dict(CFG["decoder"], sample_prob=0.5))
    m = build(g, cfg)
    m.set_train()
    assert m.scheduled_sampling
    random.seed(3)
    loss = m.loss(batch_of(g))
    drawn = random.random()
    random.seed(3)
    for _ in range(g["y"].shape[1] - 2):   # one draw per token after the first (seq2seq.py:91-92)
        random.random()
    assert drawn == random.random()
    loss.backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    m.set_eval()
    assert not m.scheduled_sampling


def test_greedy_infer_in_the_library_equals_the_step_by_step_loop():
    """Seq2Seq.infer runs the token loop inside the library (sa_s2s_greedy_decode: arg-max, feedback and the "every row
    emitted the end token" stopping rule in a kernel); infer_decode is the reference's step-by-step loop (seq2seq.py:140-158)
    on the same decoder step.  Same tokens, same number of columns -- on a model whose rows stop early and on one that
    runs to max_len."""
    from speech_amd.models import Seq2Seq
    for seed, fc_scale, max_len in ((3, 25.0, 40), (5, 1.0, 9)):
        cfg = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 9, 2]], "rnn": {"dim": 32, "bidirectional": True, "layers": 1}},
               "decoder": {"embedding_dim": 32, "layers": 1, "log_t": True}}
        torch.manual_seed(seed)
        m = Seq2Seq(20, 9, cfg)
        with torch.no_grad():
            m.fc.fc.weight.mul_(fc_scale)
            m.fc.fc.bias.mul_(fc_scale)
            m.fc.fc.bias[7] += 4.0 * fc_scale / 25.0   # the end token (7) is likely: rows stop, not always together
        m = m.cuda()
        m.set_eval()
        rng = np.random.RandomState(seed)
        B = 4
        inputs = tuple(rng.randn(80 - 5 * i, 20).astype(np.float32) for i in range(B))
        labels = tuple([8, 1, 7] for _ in range(B))
        got = m.infer((inputs, labels), max_len=max_len)
        x, y = m.collate(inputs, labels)
        with torch.no_grad():
            enc = m.encode(x.cuda())
            _, want = m.infer_decode(enc, y[:, 0:1].cuda(), 7, max_len)
        assert np.array_equal(np.array(got), want.cpu().numpy()), (seed, got, want)


def test_mfma_layout_attention_kernels_against_the_round4_kernels():
    """(r6, option s2s.kernels) bit 0: attention_bwd_main2_kernel (d ax, the softmax and the score network of a token in ONE launch:
    s = d_sx . sx + sum_t ax d_ax_next instead of a pass over the utterance, the tap sums on f32 MFMAs) against the round-4 stages
    ON THE SAME FORWARD (3 vs 2): every gradient within 2e-5 of the tensor's max magnitude.  Bit 1: the forward's score network in
    the MFMA layout and the context on four workgroups per utterance against the round-4 kernels throughout (3 vs 0): the loss
    to 1e-6, the gradients to 1e-3 of max -- the encoder states' contexts move by an ulp, and one pre-activation of the score
    network within an ulp of zero changes sides: a whole term of the heavily cancelling attention gradients, not a rounding.
    At a tiny shape (H = 16: one MFMA per wave, a last chunk of 11 steps) and at the shipped WSJ width (H = 256, T' = 197)."""
    from speech_amd import _lib
    from speech_amd.models import Seq2Seq
    for (dim, F, T, B, U, conv) in ((16, 20, 90, 3, 7, [[4, 5, 9, 2]]), (256, 40, 400, 4, 12, [[8, 5, 8, 2]])):
        cfg = {"dropout": 0.0, "encoder": {"conv": conv, "rnn": {"dim": dim, "bidirectional": True, "layers": 1}},
               "decoder": {"embedding_dim": dim, "layers": 1, "log_t": True}}
        torch.manual_seed(11)
        m = Seq2Seq(F, 12, cfg).cuda()
        m.set_train()
        rng = np.random.RandomState(3)
        inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
        labels = tuple([11] + list(rng.randint(0, 10, U - 2)) + [10] for _ in range(B))
        got, loss = {}, {}
        try:
            for kernels in (3, 2, 0):
                _lib.set_option("s2s.kernels", kernels)
                m.zero_grad(set_to_none=True)
                lo = m.loss((inputs, labels))
                lo.backward()
                loss[kernels] = float(lo.item())
                got[kernels] = {n: p.grad.detach().cpu().numpy().copy() for n, p in m.named_parameters()}
        finally:
            _lib.set_option("s2s.kernels", 3)
        assert loss[3] == loss[2] and abs(loss[3] - loss[0]) <= 1e-6 * abs(loss[0]), loss
        for n in got[3]:
            a, w, w0 = got[3][n], got[2][n], got[0][n]
            assert np.isfinite(a).all(), n
            # the score bias' gradient is analytically zero (the softmax is shift invariant): every path delivers rounding noise
            zero = n.endswith("attend.nn.1.fc.bias")
            tol = 1e-6 if zero else 2e-5 * max(np.abs(w).max(), 1e-3)
            assert np.abs(a - w).max() <= tol, (dim, n, np.abs(a - w).max(), np.abs(w).max())
            # (the location conv's gradients all but cancel, tests/test_gpu_baseline_configs.py: 3e-3 / 3e-2 of their max)
            loose = 1e-6 if zero else (3e-2 if "attend.conv" in n else 1e-3) * max(np.abs(w0).max(), 1e-3)
            assert np.abs(a - w0).max() <= loose, (dim, n, np.abs(a - w0).max(), np.abs(w0).max())


def test_flat_buffer_gradients_are_handed_over_by_reference_by_every_node():
    """After flatten_parameters_() every parameter's .grad IS its slot of the flat buffer after a backward pass -- the decoder's 11
    as well as the encoder's (autograd cloned the decoder's into fresh tensors every step before r6) -- and a second backward
    without zero_grad accumulates (2 x the gradient), as autograd's own accumulation would."""
    g = fixture()
    m = build(g, flatten=True)
    m.set_train()
    m.zero_grad(set_to_none=True)
    m.loss(batch_of(g)).backward()
    once = {}
    for name, p in m.named_parameters():
        assert p.grad is not None and p.grad.data_ptr() == p._grad_slot.data_ptr(), name
        once[name] = p.grad.detach().cpu().numpy().copy()
    m.loss(batch_of(g)).backward()
    for name, p in m.named_parameters():
        got = p.grad.detach().cpu().numpy()
        assert np.abs(got - 2 * once[name]).max() <= 1e-5 * max(np.abs(once[name]).max(), 1e-3), name


def test_edge_shapes_of_the_attention_kernels():
    """tools/s2s_shape_sweep.py: widths that are not a multiple of 16 (4, 8, 20, 36, 200), one-chunk utterances, one utterance, one
    token, tap counts 1 .. 15 -- the MFMA-layout kernels against the round-4 backward on the same forward (2e-5 of max) and against
    the round-4 kernels throughout; the narrow bidirectional encoders of the first two cases killed the process with SIGFPE
    before round 6 (an integer division by zero in the host's XCD arithmetic of sa_gru_stack_fwd)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "s2s_shape_sweep.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "all shapes agree" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
