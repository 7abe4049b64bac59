"""speech_amd.io / score -- checkpoint I/O and CER of the reference (/root/reference/speech/utils/io.py:15-26,
speech/utils/score.py:7-18), kept so eval.py / train.py stay drop-in.

save() writes the same two files (<path>/[tag_]model, <path>/[tag_]preproc.pyc).  The model file holds
{"class", "args", "state_dict"} instead of a whole-module pickle: state-dict keys are the reference's
(conv.0.weight, rnn.weight_ih_l0, fc.fc.weight, ...), so a reference checkpoint's state_dict loads unchanged, and
torch >= 2.6's weights_only default no longer rejects the file."""
import os
import pickle

import torch

MODEL = "model"
PREPROC = "preproc.pyc"


def get_names(path, tag):
    tag = tag + "_" if tag else ""
    return os.path.join(path, tag + MODEL), os.path.join(path, tag + PREPROC)


def save(model, preproc, path, tag=""):
    model_n, preproc_n = get_names(path, tag)
    os.makedirs(path, exist_ok=True)
    blob = {"class": type(model).__name__, "args": getattr(model, "_ctor_args", None),
            "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}}
    torch.save(blob, model_n)
    with open(preproc_n, "wb") as fid:
        pickle.dump(preproc, fid)


def load(path, tag=""):
    from . import models
    model_n, preproc_n = get_names(path, tag)
    blob = torch.load(model_n, map_location="cpu", weights_only=False)
    if isinstance(blob, torch.nn.Module):  # a whole-module pickle (the reference's format)
        model = blob
    else:
        cls = getattr(models, blob["class"])
        model = cls(*blob["args"])
        model.load_state_dict(blob["state_dict"])
    with open(preproc_n, "rb") as fid:
        preproc = pickle.load(fid)
    return model, preproc


def edit_distance(a, b):
    """Levenshtein distance (the reference uses the `editdistance` package, score.py:5)."""
    a, b = list(a), list(b)
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def compute_cer(results):
    """score.py:7-18: total edit distance over total label length, results = [(label, prediction), ...]."""
    dist = sum(edit_distance(label, pred) for label, pred in results)
    total = sum(len(label) for label, _ in results)
    return dist / total
