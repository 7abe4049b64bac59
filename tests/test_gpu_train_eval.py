"""End-to-end drop-in drivers on the GPU: train.py (the reference's loop on the HIP path, fused clip+SGD) for two
epochs on a tiny synthetic dataset, checkpoint through speech.save, then eval.py (speech.load -> CTC.infer)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import scipy.io.wavfile
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _dataset(tmp_path):
    rng = np.random.RandomState(0)
    lines = []
    for i in range(8):
        n = 9000 + 800 * (i % 2)
        t = np.arange(n) / 16000.0
        audio = (3000 * np.sin(2 * np.pi * (200 + 60 * i) * t) + 200 * rng.randn(n)).astype(np.int16)
        path = str(tmp_path / ("utt%d.wav" % i))
        scipy.io.wavfile.write(path, 16000, audio)
        lines.append({"text": list("ab" if i % 2 else "ba"), "duration": n / 16000.0, "audio": path})
    js = str(tmp_path / "data.json")
    with open(js, "w") as fid:
        for l in lines:
            fid.write(json.dumps(l) + "\n")
    return js


def test_train_then_eval(tmp_path, monkeypatch):
    import random
    js = _dataset(tmp_path)
    cfg = {"seed": 2017, "save_path": str(tmp_path / "ckpt"),
           "data": {"train_set": js, "dev_set": js, "start_and_end": False},
           "optimizer": {"batch_size": 4, "epochs": 2, "learning_rate": 1e-3, "momentum": 0.9},
           "model": {"class": "CTC", "dropout": 0.0,
                     "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 32, "bidirectional": True, "layers": 2}}}}
    random.seed(cfg["seed"])
    torch.manual_seed(cfg["seed"])
    train = _load("train")
    import speech.loader as loader
    real = loader.make_loader
    monkeypatch.setattr(loader, "make_loader", lambda *a, **k: real(*a, **dict(k, num_workers=0)))
    train.run(cfg)
    assert os.path.exists(os.path.join(cfg["save_path"], "model")) and os.path.exists(
        os.path.join(cfg["save_path"], "best_model"))
    ev = _load("eval")
    out = str(tmp_path / "pred.jsonl")
    cer = ev.run(cfg["save_path"], js, batch_size=4, tag="best", out_file=out)
    assert 0.0 <= cer <= 2.0
    rows = [json.loads(l) for l in open(out)]
    assert len(rows) == 8 and all(set(r) == {"prediction", "label"} for r in rows)


@pytest.mark.parametrize("model_cfg,start_and_end", [
    ({"class": "Transducer", "dropout": 0.0,
      "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 32, "bidirectional": True, "layers": 2}},
      "decoder": {"embedding_dim": 16, "layers": 1}}, False),
    ({"class": "Seq2Seq", "dropout": 0.0,
      "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 32, "bidirectional": True, "layers": 2}},
      "decoder": {"embedding_dim": 32, "layers": 1, "log_t": True, "sample_prob": 0.2}}, True),
])
def test_other_model_families_train_then_eval(tmp_path, monkeypatch, model_cfg, start_and_end):
    """The same drivers on the reference's other two model classes (examples/timit/transducer_config.json and
    examples/wsj/seq2seq_config.json structure): two epochs of train.py, checkpoint, eval.py."""
    import random
    js = _dataset(tmp_path)
    cfg = {"seed": 2017, "save_path": str(tmp_path / "ckpt"),
           "data": {"train_set": js, "dev_set": js, "start_and_end": start_and_end},
           "optimizer": {"batch_size": 4, "epochs": 2, "learning_rate": 1e-3, "momentum": 0.0},
           "model": model_cfg}
    random.seed(cfg["seed"])
    torch.manual_seed(cfg["seed"])
    train = _load("train")
    import speech.loader as loader
    real = loader.make_loader
    monkeypatch.setattr(loader, "make_loader", lambda *a, **k: real(*a, **dict(k, num_workers=0)))
    train.run(cfg)
    assert os.path.exists(os.path.join(cfg["save_path"], "best_model"))
    ev = _load("eval")
    cer = ev.run(cfg["save_path"], js, batch_size=4, tag="best", out_file=str(tmp_path / "pred.jsonl"))
    # two epochs do not train anything: an untrained Seq2Seq greedy-decodes max_len tokens against 2-character labels,
    # so only the plumbing is asserted here (parity is tests/test_gpu_seq2seq.py / test_gpu_transducer_model.py)
    assert np.isfinite(cer) and cer >= 0.0
