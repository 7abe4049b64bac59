"""Data parallelism with the HIP kernels doing the per-rank math (the multi-rank tests of tests/test_dist_gloo.py
run the CPU oracle behind the product's collate, so the HIP path had never produced a W > 1 gradient).

Two processes share the ONE GPU of the test box; the gradient message goes through torch.distributed with the gloo backend
(RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Each rank takes its shard of a RAGGED global
batch (speech_amd.dist.shard_batch), pads it to the global longest utterance, divides its CTC loss by the global batch size
(Model.set_global_batch), runs forward + loss + backward on the HIP library, stamps the health word and SUM-all-reduces the
flat gradient buffer (speech_amd.dist.allreduce_gradients) -- train.py's step up to the optimiser.  The result must be the
single-process gradient of the whole batch (same kernels, same weights) up to fp32 summation order: a row's arithmetic
does not depend on which batch it sits in, only the order in which rows are added into the weight gradients does.
SA_GRU_PERSIST=0 in the workers: two processes' persistent recurrence launches would each want every CU of the one GPU
(on a real node every rank has its own)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"dropout": 0.0, "encoder": {"conv": [[8, 5, 32, 2]], "rnn": {"dim": 128, "bidirectional": False, "layers": 2}}}
F, V = 40, 10
LENS = [131, 96, 120, 77, 104]  # ragged; the longest utterance lives in rank 0's shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch():
    rng = np.random.RandomState(11)
    inputs = tuple(rng.randn(t, F).astype(np.float32) for t in LENS)
    labels = tuple(list(rng.randint(0, V, 3 + i % 4)) for i in range(len(LENS)))
    return inputs, labels


def _flat_grad(batch, shape):
    """Flat gradient and loss of `batch` as a shard of a global batch of `shape`, on the HIP path."""
    from speech_amd import ops
    from speech_amd.models import CTC
    torch.manual_seed(5)
    model = CTC(F, V, CFG).cuda()
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    model.set_global_batch(*shape)
    model.zero_grad(set_to_none=True)
    loss = model.loss(batch)
    loss.backward()
    ops.stamp_health(flat_g)
    return flat_g, float(loss.item())


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK="0", SA_GRU_PERSIST="0")
    from speech_amd import dist
    w, r, _ = dist.init(backend="gloo")
    assert (w, r) == (world, rank)
    shard, shape = dist.shard_batch(_make_batch(), world, rank)
    flat_g, loss = _flat_grad(shard, shape)
    on_device = True
    try:
        dist.allreduce_gradients(flat_g)          # gloo stages a device tensor through the host itself
    except RuntimeError:                           # a torch build whose gloo takes host tensors only
        on_device = False
        host = flat_g.cpu()
        dist.allreduce_gradients(host)
        flat_g.copy_(host)
    total = dist.allreduce_sum_host([loss])[0]
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        torch.save({"flat": flat_g.cpu(), "shape": tuple(shape), "n": len(shard[0]), "loss": total, "on_device": on_device}, out)


def test_two_ranks_on_the_hip_path_reproduce_the_single_process_gradient(tmp_path, monkeypatch):
    monkeypatch.setenv("SA_GRU_PERSIST", "0")  # the single-process side on the same kernels as the workers
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["n"] == 3 and got["shape"] == (5, max(LENS), 6)
    whole = _make_batch()
    want, want_loss = _flat_grad(whole, (None, None, None))
    want = want.cpu()
    # the last slot of the flat buffer is the health flag (0 = healthy on both ranks and here)
    assert float(got["flat"][-1]) == 0.0 and float(want[-1]) == 0.0
    rel = float((got["flat"] - want).norm() / want.norm())
    assert rel <= 2e-6, rel
    torch.testing.assert_close(got["flat"], want, rtol=1e-4, atol=2e-6 * float(want.abs().max()))
    assert abs(got["loss"] - want_loss) <= 1e-6 * abs(want_loss)
