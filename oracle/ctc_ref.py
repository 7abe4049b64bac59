"""
oracle/ctc_ref.py -- ctypes face of oracle/ctc_ref.c (the CPU CTC restatement; see that file's header:
PARITY UNPINNED against the un-vendored warp-ctc, pinned by enumeration / finite differences / torch CPU).
TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libctc_ref.so")
    src = os.path.join(_HERE, "ctc_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libctc_ref.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        ip = ctypes.POINTER(ctypes.c_int)
        L.ctc_ref_f64.restype = ctypes.c_int
        L.ctc_ref_f64.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ip, ip, ip, ctypes.c_int,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        L.ctc_ref_f32.restype = ctypes.c_int
        L.ctc_ref_f32.argtypes = L.ctc_ref_f64.argtypes + [ctypes.c_int]
        L.ctc_ref_max_threads.restype = ctypes.c_int
        _LIB = L
    return _LIB


def max_threads():
    return _lib().ctc_ref_max_threads()


def ctc_loss(acts, labels, act_lens, label_lens, blank=None, batch_first=True, want_grad=True, dtype=np.float64,
             num_threads=0):
    """acts: float32 (B,T,K) if batch_first else (T,B,K).  Returns (costs[B], grads like acts or None)."""
    acts = np.ascontiguousarray(acts, dtype=np.float32)
    if batch_first:
        B, T, K = acts.shape
        st, sb = K, T * K
    else:
        T, B, K = acts.shape
        st, sb = B * K, K
    if blank is None:
        blank = K - 1
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    act_lens = np.ascontiguousarray(act_lens, dtype=np.int32)
    label_lens = np.ascontiguousarray(label_lens, dtype=np.int32)
    assert act_lens.shape == (B,) and label_lens.shape == (B,) and labels.size == int(label_lens.sum())
    ip = ctypes.POINTER(ctypes.c_int)
    costs = np.zeros(B, dtype=dtype)
    grads = np.zeros(acts.shape, dtype=dtype) if want_grad else None
    gp = grads.ctypes.data if want_grad else None
    args = [acts.ctypes.data, st, sb, labels.ctypes.data_as(ip), label_lens.ctypes.data_as(ip),
            act_lens.ctypes.data_as(ip), K, B, blank, T, costs.ctypes.data, gp]
    if dtype == np.float64:
        rc = _lib().ctc_ref_f64(*args)
    elif dtype == np.float32:
        rc = _lib().ctc_ref_f32(*args, num_threads)
    else:
        raise TypeError(dtype)
    if rc != 0:
        raise ValueError("ctc_ref: invalid arguments (status %d)" % rc)
    return costs, grads
