"""Data-parallel logic on CPU: 2-3 processes, gloo backend (the GPU path runs the same code with backend "nccl" = RCCL).

Checks the contract of speech_amd/dist.py on RAGGED batches: a global batch sharded over W
ranks -- every shard zero-padded to the GLOBAL longest utterance, the CTC loss divided by the GLOBAL batch size -- and
a SUM all-reduce of the flat gradient buffer reproduce the single-process gradient of the same global batch.  The
reference scores the padded frames (every act_len is the padded length, /root/reference/speech/models/
ctc_model.py:43-45), so padding a shard only to its own maximum would NOT (also asserted).
The per-rank math is the CPU oracle (oracle/torch_ref.py) behind speech_amd.models.CTC's own host-side collate --
the HIP kernels need a GPU."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = {"dropout": 0.0, "encoder": {"conv": [[4, 5, 16, 2]], "rnn": {"dim": 8, "bidirectional": False, "layers": 2}}}
LENS = [52, 40, 47, 33, 44]  # ragged: the longest utterance lives in rank 0's shard, rank 1's shard is shorter


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(lens=LENS):
    rng = np.random.RandomState(0)
    inputs = tuple(rng.randn(t, 20).astype(np.float32) for t in lens)
    labels = tuple(list(rng.randint(0, 6, 2 + i % 3)) for i in range(len(lens)))
    return inputs, labels


def _grad_of(batch, shape):
    """Flat gradient + loss of `batch` treated as a shard of a global batch of `shape` = (size, frames, label len).
    Host side = the product's own collate (speech_amd.models.CTC.collate / set_global_batch); math = CPU oracle."""
    from oracle.torch_ref import TorchRefCTC, _CTCRef
    from speech_amd.models import CTC
    torch.manual_seed(1)
    ref = TorchRefCTC(20, 6, CFG)
    flat = torch.zeros(sum(p.numel() for p in ref.parameters()))
    if len(batch[0]) == 0:
        return flat, 0.0
    host = CTC(20, 6, CFG)  # never computes: collate only
    host.set_global_batch(*shape)
    x, y, x_lens, y_lens = host.collate(*batch)
    logits = ref(x)
    assert logits.shape[1] == int(x_lens[0])
    B = logits.shape[0]
    loss = _CTCRef.apply(logits, y.numpy(), x_lens.numpy(), y_lens.numpy(), 6, 1) * (B / host.loss_denominator)
    loss.backward()
    return torch.cat([p.grad.reshape(-1) for p in ref.parameters()]), float(loss.item())


def _worker(rank, world, port, out, lens, sharded_loader):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from speech_amd import dist
    w, r, _ = dist.init(backend="gloo")
    assert (w, r) == (world, rank)
    whole = _make_batch(lens)
    if sharded_loader:  # the rank only ever sees its own utterances (loader.make_loader(world, rank)): agree by collective
        lo, hi = dist.shard_bounds(len(lens), world, rank)
        shard = (whole[0][lo:hi], whole[1][lo:hi])
        shape = dist.global_shape(shard)
    else:
        shard, shape = dist.shard_batch(whole, world, rank)
    flat, loss = _grad_of(shard, shape)
    dist.allreduce_gradients(flat)
    total_loss = dist.allreduce_sum_host([loss])[0]
    # the scalar reductions bench.py's multi-GPU legs use, and the look-ahead iterator with a model's stage_ahead hook
    cpu = torch.device("cpu")
    ssum, smax = dist.sum_over_ranks(float(rank + 1), cpu), dist.max_over_ranks(float(rank + 1), cpu)

    class Recorder:  # stands in for Model.stage_ahead: resolves the batch's global shape on ANOTHER thread
        def __init__(self):
            import concurrent.futures
            self.pool, self.seen = concurrent.futures.ThreadPoolExecutor(1), []

        def stage_ahead(self, batch, handle):
            self.seen.append(self.pool.submit(handle.result))
    rec = Recorder()
    lo, hi = dist.shard_bounds(len(lens), world, rank)
    mine = (whole[0][lo:hi], whole[1][lo:hi])
    shapes = [sh for _, sh in dist.with_global_shapes(iter([mine, mine, mine]), rec)]
    ahead = [f.result() for f in rec.seen]
    dist.barrier()
    if rank == 0:
        torch.save({"flat": flat, "n": len(shard[0]), "shape": shape, "loss": total_loss, "ssum": ssum, "smax": smax,
                    "shapes": shapes, "ahead": ahead}, out)


def test_shard_bounds_cover_batch_exactly():
    from speech_amd import dist
    for n in (1, 5, 32, 33):
        for w in (1, 2, 3, 8):
            spans = [dist.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


@pytest.mark.parametrize("sharded_loader", [False, True])
def test_two_rank_allreduce_equals_single_process_on_a_ragged_batch(tmp_path, sharded_loader):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out, LENS, sharded_loader), nprocs=2, join=True)
    got = torch.load(out)
    whole = _make_batch()
    want, want_loss = _grad_of(whole, (5, max(LENS), 4))
    assert got["n"] == 3 and tuple(got["shape"]) == (5, 52, 4)  # rank 0 takes the remainder utterance
    torch.testing.assert_close(got["flat"], want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))  # fp32 sum order
    assert abs(got["loss"] - want_loss) <= 1e-5 * abs(want_loss)
    assert got["ssum"] == 3.0 and got["smax"] == 2.0
    # the look-ahead iterator: every batch's global shape, also as the collate-ahead thread saw it (one wait, two readers)
    assert got["shapes"] == [(5, 52, 4)] * 3 and got["ahead"] == [(5, 52, 4)] * 3
    # the round-1 behaviour -- each shard padded to its OWN longest utterance -- is a different function
    s1 = (whole[0][3:], whole[1][3:])
    local, _ = _grad_of(s1, (5, max(LENS[3:]), 4))
    glob, _ = _grad_of(s1, (5, max(LENS), 4))
    assert float((local - glob).abs().max()) > 1e-4 * float(glob.abs().max())


def test_global_batch_smaller_than_the_world(tmp_path):
    """B < W: the trailing rank's shard is empty; it contributes zeros and still joins every collective."""
    out = str(tmp_path / "r0.pt")
    lens = LENS[:2]
    mp.spawn(_worker, args=(3, _free_port(), out, lens, True), nprocs=3, join=True)
    got = torch.load(out)
    want, want_loss = _grad_of(_make_batch(lens), (2, max(lens), 3))
    assert tuple(got["shape"]) == (2, 52, 3)
    torch.testing.assert_close(got["flat"], want, rtol=1e-5, atol=1e-5 * float(want.abs().max()))  # fp32 sum order
    assert abs(got["loss"] - want_loss) <= 1e-5 * abs(want_loss)


# ---- failure replay stays in lock-step (train.py) ------------------------------------------------------------
def _replay_worker(rank, world, port, out_dir, fail_rank, fail_at, n_batches):
    """train.run_epoch with host-only stand-ins for the step and the reset: `step` does what train_step's collective
    structure does (one SUM all-reduce per call carrying this rank's sticky health flag, norm negative when the sum is
    non-zero), so the number and order of collectives per rank is what the real loop would issue."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    import time
    import torch.distributed as td
    import train
    from speech_amd import dist
    dist.init(backend="gloo")
    state = {"word": 0, "calls": [], "resets": 0, "updates": []}

    def step(batch_, shape):
        batch = int(batch_[0][0].shape[0]) - 1      # the batch's index is encoded in its frame count
        assert shape == (world, batch + 1, 1)        # ... and the shape exchange (one step ahead) paired the SAME batches
        if rank == fail_rank and batch == fail_at and state["resets"] == 0:
            state["word"] = 1                      # a persistent kernel on THIS rank failed: sticky until the reset
        if rank == 1:
            time.sleep(0.002)                      # ranks do not run at the same speed
        msg = torch.tensor([float(batch), float(state["word"])])
        td.all_reduce(msg, op=td.ReduceOp.SUM)     # the gradient message with the health flag behind it
        assert float(msg[0]) == world * batch      # every rank is in THIS collective with the same batch
        ok = float(msg[1]) == 0.0
        state["calls"].append(batch)
        if ok:
            state["updates"].append(batch)
        return torch.tensor([1.0 + batch]), torch.tensor([2.0 if ok else -2.0])

    def reset():
        state["resets"] += 1
        state["word"] = 0
        return 1

    batches = [((np.zeros((k + 1, 1), np.float32),), ([0],)) for k in range(n_batches)]
    it, _ = train.run_epoch(None, (torch.zeros(1),), None, batches, 0, 0.0, world, rank, step_fn=step, reset_fn=reset)
    dist.barrier()
    torch.save({"it": it, **state}, os.path.join(out_dir, "r%d.pt" % rank))


@pytest.mark.parametrize("fail_at", [4, 9])  # mid-epoch, and inside the epoch's last LAG steps (caught by the final drain)
def test_failure_replay_is_lock_step_across_ranks(tmp_path, fail_at):
    n = 10
    mp.spawn(_replay_worker, args=(2, _free_port(), str(tmp_path), 1, fail_at, n), nprocs=2, join=True)
    r0, r1 = torch.load(str(tmp_path / "r0.pt")), torch.load(str(tmp_path / "r1.pt"))
    assert r0["calls"] == r1["calls"]                      # same collectives, same order, on both ranks
    assert r0["resets"] == r1["resets"] == 1               # both ranks noticed (the flag rides the all-reduce) -- once
    assert r0["it"] == r1["it"] == n
    assert sorted(r0["updates"]) == list(range(n)) and r0["updates"] == r1["updates"]  # every batch applied exactly once
    import train
    first_seen = min(fail_at + train.LAG, n - 1)           # detection: LAG steps later, or the end-of-epoch drain
    assert r0["calls"] == list(range(first_seen + 1)) + list(range(fail_at, first_seen + 1)) + list(range(first_seen + 1, n))
