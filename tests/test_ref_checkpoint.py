"""speech.load() reads the REFERENCE's own checkpoints (/root/reference/speech/utils/io.py:15-26: torch.save of the
whole module, so the class paths speech.models.ctc_model.CTC ... are part of the on-disk format).  The files under
tests/golden/ref_ckpt/ were written by the live reference's speech.save() (oracle/gen_ref_checkpoint.py)."""
import os

import numpy as np
import pytest
import torch

NAMES = {"ctc": "CTC", "ctc_bi": "CTC", "seq2seq": "Seq2Seq", "transducer": "Transducer"}


def _load(golden_dir, name):
    import speech
    return speech.load(os.path.join(golden_dir, "ref_ckpt"), tag=name)


@pytest.mark.parametrize("name", sorted(NAMES))
def test_reference_whole_module_checkpoint_loads(golden_dir, name):
    import speech_amd.models as M
    z = np.load(os.path.join(golden_dir, "ref_ckpt.npz"))
    model, preproc = _load(golden_dir, name)
    assert type(model) is getattr(M, NAMES[name])            # a real speech_amd model, not the half-initialised pickle
    assert sorted(model.state_dict().keys()) == list(z[name + ".keys"])
    assert hasattr(model, "_plan") and model.encoder_dim in (16, 24) and not model.is_cuda
    # the pickled values, not a fresh initialisation
    raw = torch.load(os.path.join(golden_dir, "ref_ckpt", name + "_model"), map_location="cpu", weights_only=False)
    for k, v in torch.nn.Module.state_dict(raw).items():
        assert torch.equal(model.state_dict()[k], v), k
    # reference Preprocessor pickle: same attributes (io_test.py:21-24), same arithmetic
    for attr in ("mean", "std", "int_to_char", "char_to_int"):
        assert hasattr(preproc, attr)
    np.testing.assert_array_equal(np.asarray(preproc.mean), z["preproc.mean"])
    assert preproc.vocab_size == int(z["preproc.vocab"]) and preproc.input_dim == z["preproc.mean"].shape[0]
    assert preproc.decode(preproc.encode("hello")) == list("hello")


def test_rebuilt_constructor_arguments(golden_dir):
    model, _ = _load(golden_dir, "ctc_bi")
    freq, out, cfg = model._ctor_args
    assert (freq, out) == (40, 12) and cfg["dropout"] == 0.25
    assert cfg["encoder"]["conv"] == [[8, 5, 11, 2], [8, 3, 7, 1]]
    assert cfg["encoder"]["rnn"] == {"dim": 24, "layers": 2, "bidirectional": True}
    s2s, _ = _load(golden_dir, "seq2seq")
    assert s2s._ctor_args[2]["decoder"]["sample_prob"] == 0.1 and s2s.attend.log_t is True
    tr, _ = _load(golden_dir, "transducer")
    assert tr._ctor_args[2]["decoder"] == {"embedding_dim": 12, "layers": 2} and tr.blank == 9


def test_submodule_import_paths_of_the_reference():
    """/root/reference/speech/models/__init__.py, speech/utils/: every module path the reference exposes resolves."""
    import importlib
    for mod, names in (("speech.models.model", ["Model", "LinearND", "zero_pad_concat"]),
                       ("speech.models.ctc_model", ["CTC"]), ("speech.models.ctc_decoder", ["decode"]),
                       ("speech.models.seq2seq", ["Seq2Seq", "NNAttention", "end_pad_concat"]),
                       ("speech.models.transducer_model", ["Transducer"]),
                       ("speech.utils.io", ["save", "load", "get_names", "MODEL", "PREPROC"]),
                       ("speech.utils.score", ["compute_cer"]),
                       ("speech.utils.wave", ["array_from_wave", "wav_duration"]),
                       ("speech.loader", ["Preprocessor", "AudioDataset", "BatchRandomSampler", "make_loader",
                                          "log_specgram", "log_specgram_from_file", "read_data_json",
                                          "compute_mean_std"])):
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(NAMES))
def test_loaded_reference_checkpoint_encodes_like_the_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "ref_ckpt.npz"))
    model, _ = _load(golden_dir, name)
    model = model.cuda()
    model.set_eval()
    with torch.no_grad():
        enc = model.encode(torch.from_numpy(z[name + ".x"]).cuda())
    np.testing.assert_allclose(enc.cpu().numpy(), z[name + ".enc"], rtol=2e-4, atol=2e-5)
