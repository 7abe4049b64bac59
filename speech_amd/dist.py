"""speech_amd.dist -- data-parallel training of the CTC hot path: one process per GPU, torch.distributed backend
"nccl" (= RCCL over xGMI on ROCm).  New functionality relative to the reference, which has no parallelism at all
(SURVEY.md 2.3); the hook sits where SURVEY 8(e) puts it: between loss.backward() (/root/reference/train.py:30) and
clip_grad_norm (train.py:32).

Partitioning: rank r of W takes utterances [r*B/W, (r+1)*B/W) of every GLOBAL batch (all ranks derive the same
batch order from the config seed, train.py:137); the CTC loss divides by the GLOBAL batch size
(CTC.ctc_denominator), so the SUM all-reduce of the flat gradient buffer reproduces the single-GPU gradient of the
same global batch exactly (up to fp32 summation order).  One collective per step, one flat message (27 MB for the
S-LIBRI model): on 7 x ~153 GB/s point-to-point xGMI links this is far below the step time, so it is not bucketed.
"""
import os

import torch
import torch.distributed as td


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise the process group from the torchrun environment (no-op when WORLD_SIZE == 1)."""
    world, rank, local = env_world()
    if world > 1 and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def shard_bounds(n, world, rank):
    """Utterances [lo, hi) of a global batch of n that belong to `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, world, rank):
    """batch = (inputs, labels) as the reference's loader yields it (loader.py:148).  Returns this rank's slice and the
    global batch size (to be stored in CTC.ctc_denominator)."""
    inputs, labels = batch
    lo, hi = shard_bounds(len(inputs), world, rank)
    return (tuple(inputs[lo:hi]), tuple(labels[lo:hi])), len(inputs)


def allreduce_gradients(flat_grads):
    """SUM all-reduce of the flat fp32 gradient buffer, in place (RCCL ring/tree chosen by the library)."""
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.all_reduce(flat_grads, op=td.ReduceOp.SUM)
    return flat_grads


def barrier():
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.barrier()


def max_over_ranks(value, device):
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())
    return value
