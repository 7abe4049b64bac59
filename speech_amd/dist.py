"""speech_amd.dist -- data-parallel training of the CTC hot path: one process per GPU, torch.distributed backend
"nccl" (= RCCL over xGMI on ROCm).  New functionality relative to the reference, which has no parallelism at all
(SURVEY.md 2.3); the hook sits where SURVEY 8(e) puts it: between loss.backward() (/root/reference/train.py:30) and
clip_grad_norm (train.py:32).

Partitioning: rank r of W takes utterances [r*B/W, (r+1)*B/W) of every GLOBAL batch (all ranks derive the same
batch order: loader.BatchRandomSampler owns an RNG seeded from the config seed, train.py:137).  For the SUM all-reduce
of the flat gradient buffer to reproduce the single-GPU gradient of the same global batch (up to fp32 summation
order) a shard must be treated exactly as its rows are inside the whole batch:
  * the loss divides by the GLOBAL batch size (Model.loss_denominator);
  * inputs are zero-padded to the GLOBAL batch's longest utterance (Model.pad_frames): the reference scores the padded
    frames -- every act_len is the padded length (/root/reference/speech/models/ctc_model.py:43-45) -- so a shard
    padded only to its own maximum would see a different T' and a different loss;
  * Seq2Seq labels are end-padded to the global maximum (Model.pad_labels; the padded positions are scored,
    seq2seq.py:61-63).
`global_shape()` agrees on those three numbers with one tiny MAX/SUM all-reduce on a host-side (gloo) group, so a rank
only ever loads its own shard; `shard_batch()` is the single-loader variant (every rank holds the whole batch).
A rank whose shard is empty (global batch smaller than the world) contributes a zero gradient and still takes part in
the collective.

One gradient collective per step, one flat message (27 MB for the S-LIBRI model): on 7 x ~153 GB/s point-to-point xGMI
links this is far below the step time, so it is not bucketed.
"""
import os
import threading

import torch
import torch.distributed as td

_META_GROUP = None


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, force=False):
    """Initialise the process group from the torchrun environment (no-op when WORLD_SIZE == 1, unless `force`: a
    one-rank group still sends the gradient message through RCCL -- used to test that path on a 1-GPU box)."""
    world, rank, local = env_world()
    if (world > 1 or force) and not td.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        td.init_process_group(backend=backend, rank=rank, world_size=world)
    return world, rank, local


def active():
    return td.is_available() and td.is_initialized()


def _meta_group():
    """Host-side group for batch metadata: a gloo all-reduce of three integers never touches the GPU streams, so
    agreeing on the next batch's shape does not wait for the previous step's kernels."""
    global _META_GROUP
    if _META_GROUP is None:
        _META_GROUP = td.group.WORLD if td.get_backend() == "gloo" else td.new_group(backend="gloo")
    return _META_GROUP


def shard_bounds(n, world, rank):
    """Utterances [lo, hi) of a global batch of n that belong to `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def batch_shape(batch):
    """(utterances, longest input in frames, longest label sequence) of a batch = (inputs, labels)."""
    inputs, labels = batch
    return (len(inputs), max([int(i.shape[0]) for i in inputs], default=0), max([len(l) for l in labels], default=0))


def shard_batch(batch, world, rank):
    """Single-loader variant: batch = (inputs, labels) is the WHOLE global batch as the reference's loader yields it
    (loader.py:148).  Returns (this rank's slice, global shape) -- pass the shape to Model.set_global_batch(*shape)."""
    inputs, labels = batch
    lo, hi = shard_bounds(len(inputs), world, rank)
    return (tuple(inputs[lo:hi]), tuple(labels[lo:hi])), batch_shape(batch)


def global_shape(shard):
    """Sharded-loader variant (loader.make_loader(..., world, rank)): every rank holds only its own utterances.
    Returns the GLOBAL (batch size, max frames, max label length) -- one SUM and one MAX all-reduce of host integers."""
    n, frames, lab = batch_shape(shard)
    if not active():
        return n, frames, lab
    g = _meta_group()
    total = torch.tensor([n], dtype=torch.int64)
    most = torch.tensor([frames, lab], dtype=torch.int64)
    td.all_reduce(total, op=td.ReduceOp.SUM, group=g)
    td.all_reduce(most, op=td.ReduceOp.MAX, group=g)
    return int(total[0]), int(most[0]), int(most[1])


class _Shape:
    """Result handle of global_shape_async(): .result() waits for the exchange (normally long finished)."""

    def __init__(self, local, works=None, tensors=None):
        self._local, self._works, self._tensors = local, works, tensors
        self._lock, self._value = threading.Lock(), None

    def result(self):
        if self._works is None:
            return self._local
        with self._lock:  # (the loop and a model's collate-ahead thread may both ask: the works are waited for once)
            if self._value is None:
                for w in self._works:
                    w.wait()
                total, most = self._tensors
                self._value = (int(total[0]), int(most[0]), int(most[1]))
        return self._value


def global_shape_async(shard):
    """global_shape() without the wait: the two host all-reduces are started (gloo, async_op) and a handle comes back.
    The training loop starts the exchange for batch k+1 as soon as the loader has produced it -- before it enqueues
    step k -- and collects the result when step k+1 begins, so the agreement on the next batch's padded shape is never
    on the launch path of a step.  Every rank must call it for the same batches in the same order."""
    local = batch_shape(shard)
    if not active():
        return _Shape(local)
    g = _meta_group()
    total = torch.tensor([local[0]], dtype=torch.int64)
    most = torch.tensor([local[1], local[2]], dtype=torch.int64)
    works = [td.all_reduce(total, op=td.ReduceOp.SUM, group=g, async_op=True),
             td.all_reduce(most, op=td.ReduceOp.MAX, group=g, async_op=True)]
    return _Shape(local, works, (total, most))


def with_global_shapes(batches, model=None):
    """Iterate (batch, global shape) with the shape exchange of batch k+1 in flight while batch k is being used.
    `model` (optional): its stage_ahead(batch, handle) is called for batch k+1 at the same moment -- the host-side padding of
    the next batch into pinned memory (a 10 MB copy at S-LIBRI) then runs on a worker thread while the loop enqueues step k,
    instead of on the launch path of step k+1."""
    prev = None
    for batch in batches:
        handle = global_shape_async(batch)
        if model is not None and hasattr(model, "stage_ahead"):
            model.stage_ahead(batch, handle)
        if prev is not None:
            yield prev[0], prev[1].result()
        prev = (batch, handle)
    if prev is not None:
        yield prev[0], prev[1].result()


def allreduce_gradients(flat_grads):
    """SUM all-reduce of the flat fp32 gradient buffer, in place (RCCL ring/tree chosen by the library)."""
    if active():
        td.all_reduce(flat_grads, op=td.ReduceOp.SUM)
    return flat_grads


def allreduce_sum_host(values):
    """SUM of a short list of host numbers over the ranks (dev-set loss / edit-distance totals)."""
    if not active():
        return list(values)
    t = torch.tensor(list(values), dtype=torch.float64)
    td.all_reduce(t, op=td.ReduceOp.SUM, group=_meta_group())
    return t.tolist()


def barrier():
    if active():
        td.barrier()


def max_over_ranks(value, device):
    if active():
        t = torch.tensor([value], dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        return float(t.item())
    return value


def sum_over_ranks(value, device):
    if active():
        t = torch.tensor([value], dtype=torch.float64, device=device)
        td.all_reduce(t, op=td.ReduceOp.SUM)
        return float(t.item())
    return value
