#!/usr/bin/env python
"""train.py -- the reference's training driver (/root/reference/train.py) on the MI355X hot path.

Same CLI (`python train.py config.json [--deterministic]`), same JSON config schema (train.py:72-97), same loop
(run_epoch :21-49, eval_dev :51-70, checkpoint every epoch + "best" :115-121).  Differences:
  * data parallel: launched under torch.distributed.run every rank loads ITS shard of each global batch
    (loader.make_loader(world, rank)), pads it to the global batch's shape (Model.set_global_batch) and SUM
    all-reduces the flat gradient buffer (RCCL) between backward and the clip (speech_amd/dist.py); the dev set is
    evaluated by all ranks on their shards and the totals are summed, so no rank idles in a collective;
  * clip_grad_norm(200) + SGD run as the fused flat-buffer kernel (speech_amd.ops.clip_sgd_step);
  * the loss / gradient norm of a step are read back asynchronously (ops.ScalarPipe) instead of with the reference's
    per-step `loss.data[0]` sync (train.py:33): the progress bar shows them one or two steps late and the launch
    queue never drains;
  * a failed persistent-kernel hand-off cannot reach the parameters: the optimiser skips the update on the device
    (negative norm, carried to every rank by the gradient all-reduce); this loop reads the norms at a FIXED lag, so all
    ranks notice at the same iteration, reset the library and replay the same lost batches on the step kernels;
  * py3 / torch-2 fixes of SURVEY.md App. C (materialised batches, tensorboard optional).
"""
import argparse
import collections
import json
import random
import time

import torch
import tqdm

import speech
import speech.loader as loader
import speech.models as models
from speech_amd import dist, io as sa_io, ops

try:
    import tensorboard_logger as tb
except ImportError:  # optional, as in requirements.txt of the reference
    tb = None


def train_step(model, flat, opt_cfg, batch, shape=None):
    """One train.py:28-35 step on this rank's shard.  Returns (loss tensor or None for an empty shard, norm tensor).
    `shape`: the GLOBAL batch's (size, frames, label length), normally exchanged a step ahead (dist.with_global_shapes)."""
    flat_p, flat_g, mom = flat
    model.set_global_batch(*(shape if shape is not None else dist.global_shape(batch)))
    model.zero_grad(set_to_none=True)
    loss = None
    if len(batch[0]) == 0:
        flat_g.zero_()  # global batch smaller than the world: this rank adds nothing but still joins the collective
        model.skipped_step()  # ... and keeps its random generator in step with the ranks that did run a forward pass
    else:
        loss = model.loss(batch)
        ops.backward(loss)
    ops.stamp_health(flat_g)
    dist.allreduce_gradients(flat_g)
    norm = ops.clip_sgd_step(flat_p, flat_g, mom, opt_cfg["learning_rate"], opt_cfg["momentum"], 200.0)
    return loss, norm


LAG = 3  # a step's loss / norm are read back exactly LAG steps later (ops.ScalarPipe, lag mode)


def run_epoch(model, flat, opt_cfg, train_ldr, it, avg_loss, world, rank, step_fn=None, reset_fn=None):
    """train.py:21-49.  step_fn(batch, global_shape) / reset_fn default to train_step / ops.persist_reset (tests inject
    host-only stand-ins)."""
    step_fn = step_fn or (lambda batch, shape: train_step(model, flat, opt_cfg, batch, shape))
    reset_fn = reset_fn or ops.persist_reset
    model_t = 0.0
    data_t = 0.0
    end_t = time.time()
    tq = tqdm.tqdm(train_ldr, disable=rank != 0)
    losses, norms = ops.ScalarPipe(LAG + 1), ops.ScalarPipe(LAG + 1)
    recent = collections.deque(maxlen=LAG + 2)  # batches whose update has not been confirmed yet
    zero = torch.zeros(1, device=flat[0].device)
    shown = {"loss": 0.0, "norm": 0.0}

    def consume(done_losses, done_norms):
        nonlocal avg_loss
        lost = None
        for (tag, v), (_, n) in zip(done_losses, done_norms):
            if n < 0:  # the optimiser skipped this step: a persistent kernel (on some rank) reported a failure
                lost = tag if lost is None else lost
                continue
            shown["loss"], shown["norm"] = v, n
            avg_loss = 0.99 * avg_loss + 0.01 * v
            if tb is not None and rank == 0:
                tb.log_value("train_loss", v, tag)
        return lost

    def push(loss, norm, tag):
        return consume(losses.push(loss if loss is not None else zero, tag, lag=LAG), norms.push(norm, tag, lag=LAG))

    def replay(lost):
        """Every rank gets here at the SAME iteration with the SAME `lost`: a norm's sign comes from the health flag
        summed by the gradient all-reduce (identical on every rank) and is read at a fixed lag, not "when the copy has
        landed" -- ranks noticing at different iterations would replay different spans and pair their all-reduces
        with different batches.  The steps after `lost` were skipped too (the error word is sticky until the reset)."""
        consume(losses.drain(), norms.drain())
        code = reset_fn()
        if rank == 0:
            print("persistent GRU kernels failed (code %d) at iteration %d: replaying %d step(s) on the step kernels"
                  % (code, lost, len([rb for rb in recent if rb[0] >= lost])))
        bad = None
        for tag, b in [rb for rb in recent if rb[0] >= lost]:
            l2, n2 = step_fn(*b)
            r = push(l2, n2, tag)
            bad = r if bad is None else bad
        r = consume(losses.drain(), norms.drain())
        bad = r if bad is None else bad
        if bad is not None:
            raise RuntimeError("update of iteration %d skipped again after the reset (step kernels): giving up" % bad)

    # (batch, global shape) pairs: the shape exchange of batch k+1 runs while step k is being enqueued
    for batch, shape in dist.with_global_shapes(tq, model):
        start_t = time.time()
        recent.append((it, (batch, shape)))
        loss, norm = step_fn(batch, shape)
        lost = push(loss, norm, it)
        if lost is not None:
            replay(lost)
        prev_end_t = end_t
        end_t = time.time()
        model_t += end_t - start_t
        data_t += start_t - prev_end_t
        tq.set_postfix(iter=it, loss=shown["loss"], avg_loss=avg_loss, grad_norm=shown["norm"], model_time=model_t,
                       data_time=data_t)
        it += 1
    lost = consume(losses.drain(), norms.drain())  # a failure in the epoch's last LAG steps
    if lost is not None:
        replay(lost)
    return it, avg_loss


def eval_dev(model, ldr, preproc, rank=0):
    """train.py:51-70 on this rank's shards; loss and CER totals are summed over the ranks."""
    loss_sum, n_batches, results = 0.0, 0, []
    model.set_eval()
    for batch in tqdm.tqdm(ldr, disable=rank != 0):
        model.set_global_batch(*dist.global_shape(batch))
        n_batches += 1
        if len(batch[0]) == 0:
            continue
        preds = model.infer(batch)
        # this rank's share of the global-mean loss; forward-only, so there is no optimiser gate behind it: ask the
        # library whether the persistent kernels ran clean and recompute on the step kernels if not (Model._healthy)
        loss_sum += model._healthy(lambda: float(model.loss(batch).item()))
        results.extend((preproc.decode(l), preproc.decode(p)) for l, p in zip(batch[1], preds))
    model.set_train()
    model.set_global_batch()
    edits, length = sa_io.cer_counts(results)
    loss_sum, edits, length = dist.allreduce_sum_host([loss_sum, edits, length])
    loss = loss_sum / max(n_batches, 1)
    cer = edits / max(length, 1)
    if rank == 0:
        print("Dev: Loss {:.3f}, CER {:.3f}".format(loss, cer))
    return loss, cer


def run(config):
    world, rank, local = dist.init()
    opt_cfg, data_cfg, model_cfg = config["optimizer"], config["data"], config["model"]
    batch_size = opt_cfg["batch_size"]
    preproc = loader.Preprocessor(data_cfg["train_set"], start_and_end=data_cfg["start_and_end"])
    device_features = bool(data_cfg.get("device_features", False))  # featurise on the GPU (speech_amd.features)
    train_ldr = loader.make_loader(data_cfg["train_set"], preproc, batch_size, world=world, rank=rank,
                                   device_features=device_features)
    dev_ldr = loader.make_loader(data_cfg["dev_set"], preproc, batch_size, world=world, rank=rank,
                                 device_features=device_features)
    model_class = getattr(models, model_cfg["class"])
    model = model_class(preproc.input_dim, preproc.vocab_size, model_cfg)
    model.cuda()  # there is no CPU path
    flat_p, flat_g = model.flatten_parameters_()
    mom = torch.zeros_like(flat_p) if opt_cfg["momentum"] else None
    run_state = (0, 0)
    best_so_far = float("inf")
    for e in range(opt_cfg["epochs"]):
        start = time.time()
        run_state = run_epoch(model, (flat_p, flat_g, mom), opt_cfg, train_ldr, *run_state, world, rank)
        if rank == 0:
            print("Epoch {} completed in {:.2f} (s).".format(e, time.time() - start))
        dev_loss, dev_cer = eval_dev(model, dev_ldr, preproc, rank)
        if rank != 0:
            continue
        if tb is not None:
            tb.log_value("dev_loss", dev_loss, e)
            tb.log_value("dev_cer", dev_cer, e)
        speech.save(model, preproc, config["save_path"])
        if dev_cer < best_so_far:
            best_so_far = dev_cer
            speech.save(model, preproc, config["save_path"], tag="best")


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Train a speech model.")
    parser.add_argument("config", help="A json file with the training configuration.")
    parser.add_argument("--deterministic", default=False, action="store_true",
                        help="Kept for CLI compatibility: the HIP path is deterministic by construction.")
    args = parser.parse_args()
    with open(args.config, "r") as fid:
        config = json.load(fid)
    random.seed(config["seed"])
    torch.manual_seed(config["seed"])
    if tb is not None:
        tb.configure(config["save_path"])
    run(config)
