"""
oracle/gen_ref_checkpoint.py -- write checkpoints with the LIVE reference's own speech.save() (/root/reference/
speech/utils/io.py:15-19: torch.save of the WHOLE module + a pickled Preprocessor) so that tests can prove
speech.load() of this repo reads the reference's on-disk format.  Build container only:
    python oracle/gen_ref_checkpoint.py
Outputs tests/golden/ref_ckpt/<name>_model, <name>_preproc.pyc (the reference's file naming with tag = <name>) and
tests/golden/ref_ckpt.npz (a seeded input batch and the reference's outputs for it).
Nothing is copied from the reference; it is only executed (import recipe: oracle/gen_golden.py).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden import import_reference, REF  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden", "ref_ckpt")

CTC_CFG = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": False, "layers": 2}}}
BI_CFG = {"dropout": 0.25, "encoder": {"conv": [[8, 5, 11, 2], [8, 3, 7, 1]],
                                       "rnn": {"dim": 24, "bidirectional": True, "layers": 2}}}
S2S_CFG = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": False, "layers": 1}},
           "decoder": {"embedding_dim": 16, "layers": 1, "sample_prob": 0.1, "log_t": True}}
TR_CFG = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 16, "bidirectional": False, "layers": 1}},
          "decoder": {"embedding_dim": 12, "layers": 2}}


def main():
    models, _ = import_reference()
    import scipy.io.wavfile
    import soundfile  # the stub module gen_golden installed

    def read(name, dtype="int16"):
        rate, audio = scipy.io.wavfile.read(name)
        return audio, rate
    soundfile.read = read
    import speech
    import speech.loader
    import random
    os.makedirs(OUT, exist_ok=True)
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "tests"))  # test.json names its wave files relative to that directory
    random.seed(2017)
    preproc = speech.loader.Preprocessor("test.json")
    os.chdir(cwd)
    arrays = {}
    rng = np.random.RandomState(5)
    for name, cls, args in (("ctc", models.CTC, (40, 10, CTC_CFG)), ("ctc_bi", models.CTC, (40, 12, BI_CFG)),
                            ("seq2seq", models.Seq2Seq, (40, 11, S2S_CFG)),
                            ("transducer", models.Transducer, (40, 9, TR_CFG))):
        torch.manual_seed(11)
        model = cls(*args)
        model.eval()
        speech.save(model, preproc, OUT, tag=name)   # the reference's writer: whole-module pickle
        x = rng.randn(3, 60, 40).astype(np.float32)
        with torch.no_grad():
            enc = model.encode(torch.from_numpy(x))
        arrays[name + ".x"] = x
        arrays[name + ".enc"] = enc.numpy()
        arrays[name + ".keys"] = np.array(sorted(model.state_dict().keys()))
        print(name, "saved;", type(model).__module__, "enc", tuple(enc.shape))
    arrays["preproc.mean"] = np.asarray(preproc.mean)
    arrays["preproc.vocab"] = np.array(preproc.vocab_size)
    np.savez_compressed(os.path.join(OUT, "..", "ref_ckpt.npz"), **arrays)


if __name__ == "__main__":
    main()
