"""speech_amd.io / score -- checkpoint I/O and CER of the reference (/root/reference/speech/utils/io.py:15-26,
speech/utils/score.py:7-18), kept so eval.py / train.py stay drop-in.

save() writes the same two files (<path>/[tag_]model, <path>/[tag_]preproc.pyc).  The model file holds
{"class", "args", "state_dict"} instead of a whole-module pickle: state-dict keys are the reference's
(conv.0.weight, rnn.weight_ih_l0, fc.fc.weight, ...), and torch >= 2.6's weights_only default no longer rejects
the file.

load() also reads the REFERENCE's own checkpoints: io.py:15-19 pickles the whole module, so the class path
(speech.models.ctc_model.CTC, speech.models.model.LinearND, speech.loader.Preprocessor ...) is part of the on-disk
format.  The speech/ shim package provides those module paths; unpickling then yields a half-initialised object (the
pickle restores the reference's attributes, not this implementation's), from which module_to_model() reads the
constructor arguments back (conv specs, GRU geometry, dropout, decoder settings) and rebuilds a real
speech_amd model with the pickled state_dict."""
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
        model = module_to_model(blob)
    else:
        cls = getattr(models, blob["class"])
        model = cls(*blob["args"])
        model.load_state_dict(blob["state_dict"])
    with open(preproc_n, "rb") as fid:
        preproc = pickle.load(fid)
    return model, preproc


def module_to_model(mod):
    """Rebuild a speech_amd model from an unpickled reference module (see the module docstring).  Everything is read
    from the torch.nn containers the reference builds (model.py:12-42, ctc_model.py:15-19, seq2seq.py:16-38,
    transducer_model.py:16-34), which is exactly what their constructors derive from the config."""
    from . import models
    nn = torch.nn
    sub = mod._modules
    convs = [c for c in sub["conv"]._modules.values() if isinstance(c, nn.Conv2d)]
    drops = [c.p for c in sub["conv"]._modules.values() if isinstance(c, nn.Dropout)]
    rnn = sub["rnn"]
    cfg = {"dropout": float(drops[0]) if drops else float(rnn.dropout),
           "encoder": {"conv": [[c.out_channels, c.kernel_size[0], c.kernel_size[1], c.stride[0]] for c in convs],
                       "rnn": {"dim": rnn.hidden_size, "layers": rnn.num_layers,
                               "bidirectional": bool(rnn.bidirectional)}}}
    freq_dim = mod.__dict__["input_dim"]
    name = type(mod).__name__
    if name == "CTC":
        args = (freq_dim, sub["fc"]._modules["fc"].out_features - 1, cfg)
    elif name == "Transducer":
        cfg["decoder"] = {"embedding_dim": sub["embedding"].embedding_dim, "layers": sub["dec_rnn"].num_layers}
        args = (freq_dim, sub["embedding"].num_embeddings, cfg)
    elif name == "Seq2Seq":
        cfg["decoder"] = {"embedding_dim": sub["embedding"].embedding_dim, "layers": 1,
                          "sample_prob": mod.__dict__.get("sample_prob", 0),
                          "log_t": bool(sub["attend"].__dict__.get("log_t", False))}
        args = (freq_dim, sub["embedding"].num_embeddings, cfg)
    elif name == "Model":
        args = (freq_dim, cfg)
    else:
        raise TypeError("cannot rebuild a %s from a whole-module checkpoint" % name)
    model = getattr(models, name)(*args)
    model.load_state_dict(nn.Module.state_dict(mod))
    return model


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


def cer_counts(results):
    """(total edit distance, total label length) of results = [(label, prediction), ...]: the two sums of score.py:7-18,
    kept apart so that data-parallel ranks can add theirs before dividing."""
    return (sum(edit_distance(label, pred) for label, pred in results), sum(len(label) for label, _ in results))


def compute_cer(results):
    """score.py:7-18: total edit distance over total label length."""
    dist, total = cer_counts(results)
    return dist / total
