"""K_W (one wave per utterance, forced) on alignment-shaped logits: costs / flags per kernel mode against the fp64 oracle.
python tools/ctc_aligned_check.py [margin] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import ctc_ref
from speech_amd import _lib
from speech_amd.ctc import ctc_loss_raw
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_ctc import aligned_case, flags_of
margin = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 141
B, T, K, L = 6, 1000, 29, 100
acts, labs, al, ll = aligned_case(seed, B, T, K, L, margin)
co, go = ctc_ref.ctc_loss(acts, labs, al, ll)
print("oracle costs", np.array2string(co, precision=6))
_lib.lib()
for wide in (1, 0):
    _lib.set_option("ctc.wide", wide)
    for prob in (-1, 0, 3, 2):
        _lib.set_option("ctc.prob", prob)
        a = torch.from_numpy(acts).cuda()
        c, g = ctc_loss_raw(a, torch.from_numpy(labs), torch.from_numpy(al), torch.from_numpy(ll))
        torch.cuda.synchronize()
        g = g.cpu().numpy()
        print("wide=%d prob=%2d costs %s flags %s max|grad err| %.3g finite %s" % (wide, prob, np.array2string(c.cpu().numpy(), precision=6),
              flags_of(B, T, K, L).tolist(), float(np.nanmax(np.abs(g - go))), bool(np.isfinite(g).all())))
