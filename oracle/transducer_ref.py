"""
oracle/transducer_ref.py -- ctypes face of oracle/transducer_ref.c (RNN-Transducer loss, fp64) plus NumPy restatements
of the two other pieces the reference takes from the un-vendored `transducer` package: brute-force path enumeration
(the pin for the C code) and the static beam-search decoder.  PARITY UNPINNED -- see transducer_ref.c's header.
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), tools' cpu legs).
"""
import ctypes
import itertools
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libtransducer_ref.so")
    src = os.path.join(_HERE, "transducer_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libtransducer_ref.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        ip = ctypes.POINTER(ctypes.c_int)
        L.transducer_ref_f64.restype = ctypes.c_int
        L.transducer_ref_f64.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ip, ip,
                                         ip, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        _LIB = L
    return _LIB


def transducer_loss(log_probs, labels, act_lens, label_lens, blank=None, want_grad=True):
    """log_probs: float32 (B, T, U+1, K) log-softmax lattice.  Returns (costs[B] f64, grads f64 like log_probs)."""
    lp = np.ascontiguousarray(log_probs, dtype=np.float32)
    B, T, U1, K = lp.shape
    if blank is None:
        blank = K - 1
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    act_lens = np.ascontiguousarray(act_lens, dtype=np.int32)
    label_lens = np.ascontiguousarray(label_lens, dtype=np.int32)
    assert labels.size == int(label_lens.sum())
    ip = ctypes.POINTER(ctypes.c_int)
    costs = np.zeros(B)
    grads = np.zeros(lp.shape) if want_grad else None
    rc = _lib().transducer_ref_f64(lp.ctypes.data, B, T, U1, K, labels.ctypes.data_as(ip),
                                   label_lens.ctypes.data_as(ip), act_lens.ctypes.data_as(ip), blank,
                                   costs.ctypes.data, grads.ctypes.data if want_grad else None)
    if rc != 0:
        raise ValueError("transducer_ref: invalid arguments")
    return costs, grads


def enumerate_log_prob(lp, y, blank):
    """log p(y | lattice) by summing every alignment: a path is an ordering of T-1 blank moves and U label moves
    from (0,0) to (T-1,U), then the final blank.  lp: (T, U+1, K) float64; for tiny T, U only."""
    T, U1, _ = lp.shape
    U = U1 - 1
    assert len(y) == U
    total = -math.inf
    for labels_at in itertools.combinations(range(T - 1 + U), U):
        t = u = 0
        s = 0.0
        moves = set(labels_at)
        for i in range(T - 1 + U):
            if i in moves:
                s += lp[t, u, y[u]]
                u += 1
            else:
                s += lp[t, u, blank]
                t += 1
        s += lp[T - 1, U, blank]
        total = s if total == -math.inf else max(total, s) + math.log1p(math.exp(-abs(total - s)))
    return total


def decode_static(log_probs, beam_size=1, blank=0):
    """Static beam search over a precomputed lattice, the call the reference makes at transducer_model.py:98
    (`td.decode_static(lp, beam_size, blank=self.blank)[0]`, lp = out[e, :T, :U, :]).  The callee lives in the
    un-vendored package; this is its published behaviour **[recalled, unpinned]**: the search does not re-run the
    prediction network, so a hypothesis with u labels reads row u of the lattice, and it looks for hypotheses of
    exactly U-1 labels.  Step i of T+U-2 extends every beam entry (hyp, score), u = len(hyp), t = i - u, by a blank
    (if t < T-1: same hyp) or by any label (if u < U-1); equal hypotheses merge by log-sum-exp; the best `beam_size`
    survive (stable sort, descending).  Returns (best hyp, its score + lp[T-1, U-1, blank])."""
    lp = np.asarray(log_probs)
    T, U, V = lp.shape
    beam = [((), 0.0)]
    for i in range(T + U - 2):
        new_beam = {}
        for hyp, score in beam:
            u = len(hyp)
            t = i - u
            if t < 0 or t > T - 1:
                continue
            for v in range(V):
                if v == blank:
                    if t >= T - 1:
                        continue
                    new_hyp = hyp
                elif u < U - 1:
                    new_hyp = hyp + (v,)
                else:
                    continue
                new_score = score + float(lp[t, u, v])
                old = new_beam.get(new_hyp)
                if old is None:
                    new_beam[new_hyp] = new_score
                else:
                    m = max(old, new_score)
                    new_beam[new_hyp] = m + math.log(math.exp(old - m) + math.exp(new_score - m))
        ranked = sorted(new_beam.items(), key=lambda kv: kv[1], reverse=True)
        beam = ranked[:beam_size]
    hyp, score = beam[0]
    return hyp, score + float(lp[T - 1, U - 1, blank])
