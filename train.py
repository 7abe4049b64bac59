#!/usr/bin/env python
"""train.py -- the reference's training driver (/root/reference/train.py) on the MI355X hot path.

Same CLI (`python train.py config.json [--deterministic]`), same JSON config schema (train.py:72-97), same loop
(run_epoch :21-49, eval_dev :51-70, checkpoint every epoch + "best" :115-121).  Differences:
  * data parallel: launched under torch.distributed.run it shards every global batch over the ranks and SUM
    all-reduces the flat gradient buffer (RCCL) between backward and the clip (speech_amd/dist.py);
  * clip_grad_norm(200) + SGD run as the fused flat-buffer kernel (speech_amd.ops.clip_sgd_step);
  * py3 / torch-2 fixes of SURVEY.md App. C (loss.item(), materialised batches, tensorboard optional).
"""
import argparse
import json
import random
import time

import torch
import tqdm

import speech
import speech.loader as loader
import speech.models as models
from speech_amd import dist, ops

try:
    import tensorboard_logger as tb
except ImportError:  # optional, as in requirements.txt of the reference
    tb = None


def run_epoch(model, flat, opt_cfg, train_ldr, it, avg_loss, world, rank):
    flat_p, flat_g, mom = flat
    model_t = 0.0
    data_t = 0.0
    end_t = time.time()
    tq = tqdm.tqdm(train_ldr, disable=rank != 0)
    for batch in tq:
        start_t = time.time()
        batch, global_b = dist.shard_batch(batch, world, rank)
        model.ctc_denominator = model.loss_denominator = global_b  # CTC / Transducer: mean over the GLOBAL batch
        model.zero_grad(set_to_none=True)
        loss = model.loss(batch)
        loss.backward()
        dist.allreduce_gradients(flat_g)
        grad_norm = ops.clip_sgd_step(flat_p, flat_g, mom, opt_cfg["learning_rate"], opt_cfg["momentum"], 200.0)
        loss = float(loss.item())  # this rank's share of the global-mean loss
        prev_end_t = end_t
        end_t = time.time()
        model_t += end_t - start_t
        data_t += start_t - prev_end_t
        exp_w = 0.99
        avg_loss = exp_w * avg_loss + (1 - exp_w) * loss
        if tb is not None and rank == 0:
            tb.log_value("train_loss", loss, it)
        tq.set_postfix(iter=it, loss=loss, avg_loss=avg_loss, grad_norm=float(grad_norm), model_time=model_t,
                       data_time=data_t)
        it += 1
    return it, avg_loss


def eval_dev(model, ldr, preproc):
    losses, all_preds, all_labels = [], [], []
    model.set_eval()
    model.ctc_denominator = model.loss_denominator = None
    for batch in tqdm.tqdm(ldr):
        preds = model.infer(batch)
        losses.append(float(model.loss(batch).item()))
        all_preds.extend(preds)
        all_labels.extend(batch[1])
    model.set_train()
    loss = sum(losses) / len(losses)
    results = [(preproc.decode(l), preproc.decode(p)) for l, p in zip(all_labels, all_preds)]
    cer = speech.compute_cer(results)
    print("Dev: Loss {:.3f}, CER {:.3f}".format(loss, cer))
    return loss, cer


def run(config):
    world, rank, local = dist.init()
    opt_cfg, data_cfg, model_cfg = config["optimizer"], config["data"], config["model"]
    batch_size = opt_cfg["batch_size"]
    preproc = loader.Preprocessor(data_cfg["train_set"], start_and_end=data_cfg["start_and_end"])
    train_ldr = loader.make_loader(data_cfg["train_set"], preproc, batch_size)
    dev_ldr = loader.make_loader(data_cfg["dev_set"], preproc, batch_size)
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
        if rank != 0:
            continue
        print("Epoch {} completed in {:.2f} (s).".format(e, time.time() - start))
        dev_loss, dev_cer = eval_dev(model, dev_ldr, preproc)
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
