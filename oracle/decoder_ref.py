"""
oracle/decoder_ref.py -- CPU restatement of the reference's CTC decoders.
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline); never on the product path.

Restates
  * /root/reference/speech/models/ctc_decoder.py:38-113  decode(probs, beam_size, blank) -- CTC prefix beam search
  * /root/reference/speech/models/ctc_decoder.py:27-36   logsumexp (all -inf guard, max-shifted)
  * /root/reference/speech/models/ctc_model.py:62-70     CTC.max_decode -- greedy collapse of an argmax path

The reference mixes np.float32 scalars (np.log of a float32 array, ctc_decoder.py:52) with Python floats.
Under NumPy 2 (NEP 50, the stack that runs here) a Python float is "weak": every `np.float32 + float`
is computed in float32, while math.exp/math.log run in double on float32-rounded arguments.  This file
makes that arithmetic explicit so the restatement does not depend on the NumPy version:
    lse(args) = f32(a_max) (+f32) f32( log_double( sum_double_k exp_double( f32(a_k - a_max) ) ) )
Tie-breaking is the reference's: candidates are held in first-touch (dict insertion) order -- vocab-major,
beam-minor (ctc_decoder.py:65,71) -- and the sort is stable, descending (ctc_decoder.py:107-110).

Pinned against the live reference decoder by oracle/gen_golden.py -> tests/golden/decoder_*.npz
(including the reference's own __main__ demo vector, ctc_decoder.py:115-126) and the three greedy KATs
of /root/reference/tests/ctc_test.py:31-43.
"""
import math

import numpy as np

NEG_INF = -float("inf")


def _make_ops(F):
    """Arithmetic in the array's scalar type F (np.float32 on the model path, ctc_model.py:57-58)."""

    def lse(*args):
        """ctc_decoder.py:27-36 with the NumPy-2 F/double mixture spelled out."""
        if all(a == NEG_INF for a in args):
            return NEG_INF
        a_max = max(args)  # first maximal element, as Python's max
        tot = 0.0
        for a in args:
            d = F(a) - F(a_max)  # difference rounded to F (or -inf)
            tot += math.exp(float(d))
        return F(a_max) + F(math.log(tot))

    def add(a, p):
        """`p_b + p` at ctc_decoder.py:77,89,94,102: an F add; the initial Python 0.0 / -inf are weak scalars."""
        return F(a) + F(p)

    return lse, add


def decode(probs, beam_size=10, blank=0):
    """Prefix beam search.  probs: (T, S) post-softmax float32.  Returns (labels tuple, nll)."""
    probs = np.asarray(probs)
    if probs.dtype != np.float64:
        probs = probs.astype(np.float32, copy=False)
    _lse, _add = _make_ops(probs.dtype.type)
    T, S = probs.shape
    with np.errstate(divide="ignore"):
        logp = np.log(probs)
    beam = [(tuple(), (0.0, NEG_INF))]
    for t in range(T):
        nxt = {}  # insertion-ordered, like the reference's defaultdict

        def get(key):
            if key not in nxt:
                nxt[key] = (NEG_INF, NEG_INF)
            return nxt[key]

        for s in range(S):
            p = logp[t, s]
            for prefix, (p_b, p_nb) in beam:
                if s == blank:
                    n_p_b, n_p_nb = get(prefix)
                    nxt[prefix] = (_lse(n_p_b, _add(p_b, p), _add(p_nb, p)), n_p_nb)
                    continue
                end_t = prefix[-1] if prefix else None
                n_prefix = prefix + (s,)
                n_p_b, n_p_nb = get(n_prefix)
                if s != end_t:
                    n_p_nb = _lse(n_p_nb, _add(p_b, p), _add(p_nb, p))
                else:
                    n_p_nb = _lse(n_p_nb, _add(p_b, p))
                nxt[n_prefix] = (n_p_b, n_p_nb)
                if s == end_t:
                    n_p_b, n_p_nb = get(prefix)
                    nxt[prefix] = (n_p_b, _lse(n_p_nb, _add(p_nb, p)))
        beam = sorted(nxt.items(), key=lambda x: _lse(*x[1]), reverse=True)[:beam_size]
    best = beam[0]
    return best[0], -_lse(*best[1])


def max_decode(pred, blank):
    """ctc_model.py:62-70: keep p iff p != blank and p != previous frame's label."""
    pred = list(pred)
    if not pred:
        return []
    prev = pred[0]
    seq = [prev] if prev != blank else []
    for p in pred[1:]:
        if p != blank and p != prev:
            seq.append(p)
        prev = p
    return seq


def greedy(logits_or_probs, blank):
    """argmax over classes (first maximal index, as np.argmax) then max_decode."""
    path = np.argmax(np.asarray(logits_or_probs), axis=-1)
    return max_decode([int(v) for v in path], blank)
