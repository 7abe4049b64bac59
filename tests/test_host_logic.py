"""Host-side logic that needs no GPU: the loader / io / score mirrors of the reference API, the import-path shims,
collate / conv_out_size of the model classes, and that a CPU model refuses to compute."""
import json
import os

import numpy as np
import pytest
import scipy.io.wavfile
import torch


@pytest.fixture()
def tiny_dataset(tmp_path):
    """Two synthetic 16 kHz wavs and an 8-line JSON-lines file (the shape of /root/reference/tests/test.json)."""
    rng = np.random.RandomState(0)
    files = []
    for i, n in enumerate([17622, 25130]):
        t = np.arange(n) / 16000.0
        audio = (3000 * np.sin(2 * np.pi * (200 + 150 * i) * t) + 300 * rng.randn(n)).astype(np.int16)
        path = str(tmp_path / ("utt%d.wav" % i))
        scipy.io.wavfile.write(path, 16000, audio)
        files.append((path, n / 16000.0))
    lines = []
    for k in range(8):
        path, dur = files[k % 2]
        lines.append({"text": list("hello world" if k % 2 == 0 else "hello hi"), "duration": dur, "audio": path})
    js = str(tmp_path / "data.json")
    with open(js, "w") as fid:
        for l in lines:
            fid.write(json.dumps(l) + "\n")
    return js


def test_loader_surface(tiny_dataset):
    # mirrors /root/reference/tests/loader_test.py:5-35
    import speech.loader as loader
    preproc = loader.Preprocessor(tiny_dataset)
    assert preproc.vocab_size == 11  # 9 characters + </s> + <s>
    assert preproc.int_to_char[preproc.vocab_size - 1] == preproc.START
    assert preproc.input_dim == 161  # 20 ms window at 16 kHz -> 320 // 2 + 1 bins
    ldr = loader.make_loader(tiny_dataset, preproc, batch_size=2, num_workers=0)
    n = 0
    for inputs, labels in ldr:
        assert inputs[0].shape == inputs[1].shape and inputs[0].shape[1] == preproc.input_dim
        assert inputs[0].dtype == np.float32 and len(labels) == 2
        n += len(inputs)
        _ = (inputs, labels)[1]  # a materialised tuple: indexable and re-iterable (reference App. C defect)
    assert n == 8
    text = preproc.decode(preproc.encode("hello"))
    assert "".join(text) == "hello"
    # the device-feed switch belongs to the loader's dataset, never to the (pickled) Preprocessor: a checkpoint trained
    # with data.device_features must evaluate with eval.py's forked workers
    import pickle
    dev_ldr = loader.make_loader(tiny_dataset, preproc, batch_size=2, device_features=True)
    assert dev_ldr.dataset.device_features and dev_ldr.num_workers == 0 and "device_features" not in preproc.__dict__
    preproc.device_features = True  # what a round-2 object looked like
    assert "device_features" not in pickle.loads(pickle.dumps(preproc)).__dict__


def test_io_and_score_round_trip(tiny_dataset, tmp_path):
    # mirrors /root/reference/tests/io_test.py:8-31
    import speech
    import speech.loader as loader
    from speech.models import CTC
    cfg = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": False, "layers": 1}}}
    preproc = loader.Preprocessor(tiny_dataset)
    model = CTC(preproc.input_dim, preproc.vocab_size, cfg)
    speech.save(model, preproc, str(tmp_path / "ckpt"), tag="best")
    m2, p2 = speech.load(str(tmp_path / "ckpt"), tag="best")
    for attr in ("mean", "std", "int_to_char", "char_to_int"):
        assert hasattr(p2, attr)
    assert set(m2.state_dict().keys()) == set(model.state_dict().keys())
    for k, v in model.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])
    assert hasattr(m2, "encoder_dim") and not m2.is_cuda
    assert speech.compute_cer([("abc", "abc"), ("abcd", "abxd")]) == 1 / 7


def test_model_host_methods_and_cpu_refusal():
    from speech_amd import _lib
    from speech_amd.models import CTC, zero_pad_concat
    cfg = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": False, "layers": 1}}}
    model = CTC(40, 10, cfg)
    assert model.conv_out_size(100, 0) == 48 and model.conv_out_size(40, 1) == 5 and model.blank == 10
    assert sorted(model.state_dict()) == sorted(["conv.0.weight", "conv.0.bias", "rnn.weight_ih_l0", "rnn.weight_hh_l0",
                                                 "rnn.bias_ih_l0", "rnn.bias_hh_l0", "fc.fc.weight", "fc.fc.bias"])
    inputs = (np.ones((7, 40)), np.ones((5, 40)))
    x, y, x_lens, y_lens = model.collate(inputs, ([1, 2, 3], [4]))
    assert x.shape == (2, 7, 40) and x.dtype == torch.float32 and float(x[1, 5:].abs().sum()) == 0
    assert y.tolist() == [1, 2, 3, 4] and y_lens.tolist() == [3, 1]
    assert x_lens.tolist() == [model.conv_out_size(7, 0)] * 2  # ctc_model.py:43-45: padded length for everyone
    assert zero_pad_concat(inputs).shape == (2, 7, 40)
    assert CTC.max_decode([1, 2, 2, 0, 0, 0, 2, 1], 0) == [1, 2, 2, 1] and CTC.max_decode([0, 0, 0], 0) == []
    with pytest.raises(_lib.SpeechAmdError):
        model.loss((tuple(np.random.randn(100, 40) for _ in range(2)), ([1, 2], [3])))  # CPU model: no fallback


def test_import_shims_match_reference_paths():
    import functions.ctc as ctc
    import speech
    import speech.models as models
    from speech_amd.ctc import CTCLoss
    assert ctc.CTCLoss is CTCLoss and hasattr(models, "CTC") and hasattr(models, "Model")
    assert callable(speech.save) and callable(speech.load) and callable(speech.compute_cer)
    assert ctc.CTCLoss().blank is None  # no-arg constructor, as ctc_model.py:38 uses it


def test_transducer_and_seq2seq_host_methods_and_cpu_refusal():
    """The other two model classes of the reference: constructor surface and state-dict keys
    (transducer_model.py:14-34, seq2seq.py:14-36), collate / label_collate / end_pad_concat, scheduled-sampling
    switches, the `transducer` import shim, and that a CPU model refuses to compute."""
    from speech.models import Seq2Seq, Transducer
    from speech_amd.models import end_pad_concat
    from speech_amd._lib import SpeechAmdError
    import transducer.decoders as td
    import transducer.functions.transducer as tf
    cfg = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": True, "layers": 2}},
           "decoder": {"embedding_dim": 16, "layers": 1, "sample_prob": 0.3, "log_t": True}}
    rng = np.random.RandomState(0)
    inputs = tuple(rng.randn(60 - 5 * i, 40).astype(np.float32) for i in range(3))

    t = Transducer(40, 10, cfg)
    keys = set(t.state_dict().keys())
    assert {"embedding.weight", "dec_rnn.weight_ih_l0", "fc1.fc.weight", "fc2.fc.bias", "rnn.weight_hh_l1_reverse"} <= keys
    assert t.blank == 10 and t.fc2.fc.out_features == 11
    labels = ([1, 2, 3], [4, 5], [6, 7, 8, 9])
    x, y, x_lens, y_lens = t.collate(inputs, labels)
    assert x.shape == (3, 60, 40) and y.tolist() == [1, 2, 3, 4, 5, 6, 7, 8, 9]
    assert x_lens.tolist() == [28, 28, 28] and y_lens.tolist() == [3, 2, 4]      # ceil((60 - 5 + 1) / 2) = 28
    ym = t.label_collate(labels)
    assert ym.dtype == torch.int64 and ym.tolist() == [[1, 2, 3, 3], [4, 5, 3, 3], [6, 7, 8, 9]]  # pad = labels[0][-1]
    with pytest.raises(SpeechAmdError):
        t.loss((inputs, labels))
    assert tf.TransducerLoss.__name__ == "TransducerLoss" and callable(td.decode_static)

    s = Seq2Seq(40, 11, cfg)
    keys = set(s.state_dict().keys())
    assert {"embedding.weight", "dec_rnn.weight_ih", "attend.conv.weight", "attend.nn.1.fc.weight", "fc.fc.bias"} <= keys
    assert s.fc.fc.out_features == 10 and tuple(s.attend.conv.weight.shape) == (16, 1, 15)
    assert s.scheduled_sampling and s.sample_prob == 0.3
    s.set_eval()
    assert not s.scheduled_sampling and s.volatile
    s.set_train()
    assert s.scheduled_sampling and not s.volatile
    seqs = ([10, 1, 2, 9], [10, 3, 9], [10, 4, 5, 6, 9])
    assert end_pad_concat(seqs).tolist() == [[10, 1, 2, 9, 9], [10, 3, 9, 9, 9], [10, 4, 5, 6, 9]]
    xs, ys = s.collate(inputs, seqs)
    assert xs.shape == (3, 60, 40) and ys.dtype == torch.int64 and ys.shape == (3, 5)
    with pytest.raises(SpeechAmdError):
        s.loss((inputs, seqs))
    with pytest.raises(AssertionError):   # the context vector is added to the embedding: the two widths must match
        Seq2Seq(40, 11, dict(cfg, decoder={"embedding_dim": 8, "layers": 1}))


def test_batch_sampler_is_private_sharded_and_in_step(tiny_dataset):
    """Data-parallel contract of loader.BatchRandomSampler (batch order must not depend on the global RNG
    after construction; SURVEY 8e: rank r takes utterances [r*B/W, (r+1)*B/W) of every global batch)."""
    import random
    import speech.loader as loader
    random.seed(2017)
    preproc = loader.Preprocessor(tiny_dataset)
    ds = loader.AudioDataset(tiny_dataset, preproc, 4)

    def as_process(**kw):
        random.seed(99)  # every rank is its own process, seeds alike (train.py:137) and builds its loaders alike
        return loader.BatchRandomSampler(ds, 3, **kw)
    r0, r1, whole = as_process(world=2, rank=0), as_process(world=2, rank=1), as_process()
    for epoch in range(3):
        if epoch == 1:
            random.random()  # rank 0 alone consumes the global RNG (its dev pass): must not matter
        a = list(r0)
        random.seed(epoch)   # ... nor must anything else that happens to it
        b, w = list(r1), list(whole)
        assert len(a) == len(b) == len(w) == len(whole) == 2
        for sa, sb, sw in zip(a, b, w):
            assert sa + sb == sw and len(sa) == 2 and len(sb) == 1  # remainder goes to the first ranks
    # a global batch smaller than the world: trailing ranks get an empty shard and the loader still yields it
    tiny = loader.BatchRandomSampler(ds, 2, world=4, rank=3)
    assert all(s == [] for s in tiny)
    ldr = loader.make_loader(tiny_dataset, preproc, batch_size=2, num_workers=0, world=4, rank=3)
    assert [b for b in ldr] == [((), ())] * 4


def test_collate_pads_to_the_global_batch_shape():
    from speech_amd.models import CTC, Seq2Seq, zero_pad_concat, end_pad_concat
    cfg = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": False, "layers": 1}},
           "decoder": {"embedding_dim": 16, "layers": 1}}
    inputs = (np.ones((30, 40), np.float32), np.ones((25, 40), np.float32))
    labels = ([1, 2, 3], [4, 5])
    model = CTC(40, 10, cfg)
    x, y, xl, yl = model.collate(inputs, labels)
    assert x.shape == (2, 30, 40) and xl.tolist() == [13, 13]
    model.set_global_batch(8, 41, 7)     # the global batch's longest utterance has 41 frames
    x, y, xl, yl = model.collate(inputs, labels)
    assert x.shape == (2, 41, 40) and xl.tolist() == [model.conv_out_size(41, 0)] * 2 and float(x[:, 30:].abs().sum()) == 0
    assert model.loss_denominator == 8 == model.ctc_denominator
    model.set_global_batch()
    assert model.collate(inputs, labels)[0].shape == (2, 30, 40) and model.loss_denominator is None
    s2s = Seq2Seq(40, 12, cfg)
    s2s.set_global_batch(8, 41, 7)
    xs, ys = s2s.collate(inputs, ([11, 1, 2, 10], [11, 3, 10]))
    assert xs.shape == (2, 41, 40) and ys.shape == (2, 7) and ys[1].tolist() == [11, 3, 10, 10, 10, 10, 10]
    assert zero_pad_concat(inputs, 0).shape == (2, 30, 40) and end_pad_concat(([1, 2], [3]), 0).shape == (2, 2)
