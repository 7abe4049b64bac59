import sys, ctypes
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from speech_amd import ops, _lib
hip = ctypes.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = ctypes.c_char_p
def last():
    e = hip.hipPeekAtLastError(); return e, hip.hipGetErrorString(e)
rng = np.random.RandomState(0)
for (B,T,F,O) in [(3,60,40,8),(3,60,40,8),(2,40,40,8)]:
    x = torch.from_numpy(rng.randn(B,1,T,F).astype(np.float32)).cuda()
    w = torch.from_numpy(rng.randn(O,1,5,32).astype(np.float32)).cuda()
    b = torch.zeros(O).cuda()
    try:
        y, ys = ops.conv2d_relu_fwd(x, w, b, 2, True)
        torch.cuda.synchronize(); print("ok", B,T,F,O, y.shape)
    except Exception as e:
        print("FAIL", B,T,F,O, e, last())
