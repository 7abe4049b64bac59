"""Data-parallel logic on CPU: 2 processes, gloo backend (the GPU path uses the same code with backend "nccl" = RCCL).

Checks the contract of speech_amd/dist.py: sharding a global batch over W ranks with the CTC loss divided by the
GLOBAL batch size and a SUM all-reduce of the flat gradient buffer reproduces the single-process gradient of the same
global batch.  The per-rank math is the CPU oracle (oracle/torch_ref.py) -- the HIP kernels need a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 16, 2]], "rnn": {"dim": 8, "bidirectional": False, "layers": 2}}}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch():
    rng = np.random.RandomState(0)
    inputs = tuple(rng.randn(40, 20).astype(np.float32) for _ in range(5))
    labels = tuple(list(rng.randint(0, 6, 3)) for _ in range(5))
    return inputs, labels


def _grad_of(batch, denom):
    from oracle.torch_ref import TorchRefCTC, _CTCRef
    torch.manual_seed(1)
    model = TorchRefCTC(20, 6, CFG)
    x = torch.from_numpy(np.stack(batch[0]))
    logits = model(x)
    B, Tp, _ = logits.shape
    labs = np.concatenate([np.asarray(l, np.int32) for l in batch[1]]).astype(np.int32)
    loss = _CTCRef.apply(logits, labs, np.full(B, Tp, np.int32), np.full(B, 3, np.int32), 6, 1) * (B / denom)
    loss.backward()
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()]), float(loss.item())


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from speech_amd import dist
    w, r, _ = dist.init(backend="gloo")
    assert (w, r) == (world, rank)
    shard, global_b = dist.shard_batch(_make_batch(), world, rank)
    flat, loss = _grad_of(shard, global_b)
    dist.allreduce_gradients(flat)
    total_loss = dist.max_over_ranks(loss, torch.device("cpu"))
    dist.barrier()
    if rank == 0:
        torch.save({"flat": flat, "n": len(shard[0]), "global_b": global_b, "max_loss": total_loss}, out)


def test_shard_bounds_cover_batch_exactly():
    from speech_amd import dist
    for n in (1, 5, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [dist.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_two_rank_allreduce_equals_single_process(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    want, _ = _grad_of(_make_batch(), 5)
    assert got["n"] == 3 and got["global_b"] == 5  # rank 0 takes the remainder utterance
    torch.testing.assert_close(got["flat"], want, rtol=1e-5, atol=1e-7)
