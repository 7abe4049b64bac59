#!/usr/bin/env python
"""bench.py -- the reference's headline metric on MI355X: utterances/s of one full CTC train step
(BASELINE.json: T=1000, B=32, F=80, |V|+1=29; SURVEY.md 8d "M-STEP / S-LIBRI": conv [32,5,32,2] -> T'=498,
4 x GRU-512 unidirectional, fc -> 29; SGD lr 1e-3, clip 200), plus the CTC-loss-only step time (M-CTC).

    python bench.py --gpus N --steps K --warmup W
N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL): weak scaling, every rank
trains B=32 utterances per step and the flat gradient buffer is SUM all-reduced once per step.  Run by hand without
a torchrun environment, `python bench.py --gpus N` launches those N ranks itself (it re-executes under
torch.distributed.run on 127.0.0.1).

One step = zero_grad + forward + CTC loss + backward + health stamp + (all-reduce) + clip_grad_norm(200) + SGD +
the loss read-back of the shipped loop (train.py: asynchronous, ops.ScalarPipe), nothing skipped; inputs and labels are
resident in HBM before the timed region (synthetic, seed 2017).  Rank 0 prints ONE JSON line.

Beside the headline (SURVEY 8d excludes data loading from the metric) the line carries `train_loop_utt_s`: an extra,
separately timed leg that runs the loop a user of train.py actually runs (/root/reference/train.py:21-49) -- per step
`model.loss(batch)` on HOST batches (collate: zero-pad into pinned memory, flat labels, H2D copy), backward, health
stamp, gradient all-reduce, clip + SGD, lagged loss read-back, and under --gpus N the agreement on the global batch's
padded shape (dist.with_global_shapes: exchanged one step ahead on a host-side group) -- everything except reading
audio files and featurising them.

Matched loss (north_star: "at matched CTC loss (rtol 1e-4)"): before any update the GPU model's loss on the batch is
recorded as `loss_step0`; the CPU baseline copies the SAME initial weights into oracle/torch_ref.py and its first
step's loss is `cpu_baseline.loss_step0`; `loss_rel_err` is their relative difference.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

S_LIBRI = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2]],
                                       "rnn": {"dim": 512, "layers": 4, "bidirectional": False}}}
B, T, F, V, L = 32, 1000, 80, 28, 100
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
MFMA_F32_PEAK_TFS = 157.3    # f32-input MFMA = fp32 vector peak
BF16_MFMA_PEAK_TFS = 2500.0  # dense bf16 MFMA (MI355X_MICROARCH.md)


def synthetic(rank):
    rng = np.random.RandomState(2017 + rank)
    x = rng.randn(B, T, F).astype(np.float32)
    labels = rng.randint(0, V, B * L).astype(np.int32)
    return x, labels


def make_step(model, flat_p, flat_g, x, labels, loss_fn, norm, pipe, last):
    """One full train step (module docstring) on resident inputs, as a closure."""
    from speech_amd import ops, dist

    def step():
        model.zero_grad(set_to_none=True)
        logits = model.forward_impl(x)
        with ops._span("ctc_loss", 3, 0.0):
            loss = loss_fn(logits, labels, None, None)
        ops.backward(loss)  # loss.backward() seeded with the cached unit gradient: no autograd fill launch
        ops.stamp_health(flat_g)
        dist.allreduce_gradients(flat_g)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        for _, v in pipe.push(loss):  # train.py's read-back of the loss: asynchronous, one or two steps late
            last["loss_host"] = v
        last["loss"] = loss
    return step


def timed_steps(step, steps, warmup, dev):
    """`warmup` untimed steps, then exactly `steps` steps bracketed by barrier + synchronize; MAX over ranks (seconds)."""
    from speech_amd import dist
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    return dist.max_over_ranks(time.perf_counter() - t0, dev)


def strong_scaling_leg(args, model, flat_p, flat_g, world, rank, dev):
    """SURVEY 8(e) "report both": the SAME global batch of 32 utterances sharded over the ranks (B / world each, the loss
    averaged over the global batch, one gradient all-reduce per step) -- total work fixed as N grows."""
    from speech_amd import ops
    from speech_amd.ctc import CTCLabels, CTCLoss
    if B % world:
        return None
    Bs = B // world
    x_h, lab_h = synthetic(0)  # every rank draws the same global batch and keeps its rows
    x = torch.from_numpy(x_h[rank * Bs:(rank + 1) * Bs]).to(dev)
    Tp = model.conv_out_size(T, 0)
    labels = CTCLabels(lab_h[rank * Bs * L:(rank + 1) * Bs * L], np.full(Bs, Tp, np.int32), np.full(Bs, L, np.int32), dev)
    norm, last, pipe = torch.zeros(1, device=dev), {}, ops.ScalarPipe()
    step = make_step(model, flat_p, flat_g, x, labels, CTCLoss(denom=B), norm, pipe, last)
    dt = timed_steps(step, args.steps, args.warmup, dev)
    pipe.drain()
    return {"value": B * args.steps / dt, "unit": "utt/s", "ms_per_step": dt / args.steps * 1e3, "global_batch": B,
            "per_gpu_batch": Bs, "n_gpus": world, "loss": float(last["loss"].item())}


def ctc_chain_spans(T_want):
    """Device-clock spans (us) of the alpha / beta kernel of the profiled calls whose padded length is T_want
    (sa_ctc_profile_*: earliest workgroup entry to latest workgroup exit of the launch)."""
    import ctypes
    from speech_amd import _lib
    lib = _lib.lib()
    out = []
    for i in range(lib.sa_ctc_profile_count()):
        us, t, b = ctypes.c_double(0.0), ctypes.c_int(0), ctypes.c_int(0)
        if lib.sa_ctc_profile_read(i, ctypes.byref(us), ctypes.byref(t), ctypes.byref(b)) == 0 and t.value == T_want and us.value > 0:
            out.append(us.value)
    return out


def gpu_leg(args, world, rank, local):
    from speech_amd import ops, dist
    from speech_amd.ctc import CTCLabels, CTCLoss, ctc_loss_raw
    from speech_amd.models import CTC
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    torch.manual_seed(2017)
    model = CTC(F, V, S_LIBRI)
    state0 = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None  # for the CPU leg
    model = model.cuda()
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    strong = args.scaling == "strong"
    Bl = B // world if strong else B  # utterances per rank
    x_h, lab_h = synthetic(0 if strong else rank)
    r0 = rank * Bl if strong else 0
    x = torch.from_numpy(x_h[r0:r0 + Bl]).to(dev)
    Tp = model.conv_out_size(T, 0)
    labels = CTCLabels(lab_h[r0 * L:(r0 + Bl) * L], np.full(Bl, Tp, np.int32), np.full(Bl, L, np.int32), dev)
    loss_fn = CTCLoss(denom=B if strong else B * world)  # mean over the GLOBAL batch
    norm = torch.zeros(1, device=dev)
    last = {}
    pipe = ops.ScalarPipe()
    with torch.no_grad():  # the loss of the initial weights on this batch, before any update
        loss_step0 = float(loss_fn(model.forward_impl(x), labels, None, None).item())
        # (a rank's loss is its shard's share of the global mean: the GLOBAL loss is the sum over ranks -- what a single
        # process computes on the whole global batch, tests/test_gpu_rccl_two_gpus.py)
        loss_step0 = dist.sum_over_ranks(loss_step0, dev)
    step = make_step(model, flat_p, flat_g, x, labels, loss_fn, norm, pipe, last)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    from speech_amd import _lib
    if rank == 0:
        _lib.lib().sa_gru_profile_configure(1)  # device-side clock stamps in every step launch (see speech_amd.h):
                                                # no host work, so they stay on inside the timed region
        _lib.lib().sa_ctc_profile_configure(1)  # ... and the span of the CTC loss's alpha / beta kernel (one atomic per workgroup)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    chain_in_step = ctc_chain_spans(Tp) if rank == 0 else []
    if rank == 0:
        _lib.lib().sa_ctc_profile_configure(0)
    step_us = {}
    if rank == 0:
        import ctypes
        for kind, name in ((0, "gru_fwd_step_kernel"), (1, "gru_bwd_step_kernel")):
            vi, vk = ctypes.c_float(0.0), ctypes.c_float(0.0)
            n = _lib.lib().sa_gru_profile_read(kind, ctypes.byref(vi), ctypes.byref(vk))
            step_us[name] = (float(vi.value), float(vk.value), int(n),
                             int(_lib.lib().sa_gru_profile_steps_per_launch(kind)))
        _lib.lib().sa_gru_profile_configure(0)
    # per-op HIP-event spans (two event records per op on the host) are taken in a separate, untimed pass right after:
    # on a slow host they would otherwise sit on the launch path of the timed steps
    prof, prof_steps = {}, min(3, args.steps)
    ops.PROFILE = ops.Profile() if rank == 0 else None
    for _ in range(prof_steps):  # every rank steps (the gradient all-reduce is a collective); rank 0 records
        step()
    torch.cuda.synchronize()
    if rank == 0:
        prof = ops.PROFILE.summary()
    ops.PROFILE = None
    dist.barrier()
    dt = dist.max_over_ranks(dt, dev)
    pipe.drain()
    res = {"dt": dt, "loss": float(last["loss"].item()), "grad_norm": float(norm.item()), "prof": prof,
           "loss_step0": loss_step0, "state0": state0, "persist_status": ops.persist_status(),
           "step_us": step_us, "prof_steps": prof_steps,
           "params": int(flat_p.numel()), "Tp": Tp, "per_gpu_batch": Bl,
           "ctc_chain_in_step_us": float(np.mean(chain_in_step)) if chain_in_step else None}
    if args.scaling == "both" and world > 1:  # the weak-scaling headline above; the same global batch sharded, beside it
        try:
            res["strong"] = strong_scaling_leg(args, model, flat_p, flat_g, world, rank, dev)
        except Exception as exc:  # (an extra leg must never cost the headline line)
            res["strong"] = {"error": repr(exc)}

    if args.headline_only:  # A/B experiments (tools/ab_env.sh): the timed region and the device clocks, nothing else
        res.update({"train_loop_dt": 1.0, "train_loop_steps": 0})
        return res
    res.update(train_loop_leg(model, flat_p, flat_g, world, rank, dev, max(5, min(args.steps, 20))))

    if rank == 0:  # CTC-loss-only step time (M-CTC: logits (32, 1000, 29), L = 100), fwd + grad
        res.update(ctc_legs(dev))
        res["stack_gemm"] = stack_gemm_rates(dev, Tp)
        res.update(bidirectional_leg(dev))
        res["infer_ms"] = infer_legs(dev)
    return res


def aligned_logits(Bw, labels, margin, dev, rng):
    """Logits a TRAINED CTC model emits (tools/ctc_flags_probe.py): N(0, 1) noise plus `margin` on the class of one monotonic
    alignment per utterance -- every label held for 1 - 3 frames at a random position, blanks elsewhere."""
    a = torch.randn(Bw, T, V + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(7))
    cls = np.full((Bw, T), V, np.int64)  # blank = V
    for b in range(Bw):
        starts = np.sort(rng.choice(T // 4, L, replace=False)) * 4
        for i, t0 in enumerate(starts):
            cls[b, t0:t0 + rng.randint(1, 4)] = labels[b * L + i]
    a.scatter_add_(2, torch.from_numpy(cls).to(dev).unsqueeze(2), torch.full((Bw, T, 1), float(margin), device=dev))
    return a


def ctc_call_ms(acts, lab, n, warm=2):
    """HIP events around n back-to-back forward + gradient calls (ms per call), and the utterances the probability-domain
    pass handed to the log-domain kernels in the last one."""
    from speech_amd import _lib
    from speech_amd.ctc import ctc_loss_raw
    for _ in range(warm):
        ctc_loss_raw(acts, lab)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        ctc_loss_raw(acts, lab)
    e1.record()
    torch.cuda.synchronize()
    Bw = acts.shape[0]
    off = _lib.lib().sa_ctc_flags_offset(T, L, V + 1, Bw)
    ws = _lib.WORKSPACE.get(off + 4 * Bw, acts.device, "ctc")
    flagged = int((ws.view(torch.uint8)[off:off + 4 * Bw].view(torch.int32) != 0).sum().item())
    return e0.elapsed_time(e1) / n, flagged


def ctc_legs(dev):
    """M-CTC (T=1000, |V|+1=29, L=100), forward + gradient per call, timed DIRECTLY (HIP events over back-to-back calls):
    the latency regime (B = 32: one workgroup per utterance) and the throughput regime (B = 4096: one wave per utterance),
    each on N(0, 1) logits and on alignment-shaped logits with margin 20 (what a trained model emits: the probability-domain
    pass flags more of those for the exact log-domain kernels).  For B = 32 also the device-clock span of the alpha / beta
    kernel alone (sa_ctc_profile_*) -- the serial chain, as ns per lattice step."""
    from speech_amd import _lib
    from speech_amd.ctc import CTCLabels
    out = {}
    for Bw, n in ((B, 20), (4096, 5)):
        rng = np.random.RandomState(2017)
        lab_h = rng.randint(0, V, Bw * L).astype(np.int32)
        lab = CTCLabels(lab_h, np.full(Bw, T, np.int32), np.full(Bw, L, np.int32), dev)
        if Bw == B:
            acts = torch.from_numpy(np.random.RandomState(2017).randn(Bw, T, V + 1).astype(np.float32)).to(dev)
            _lib.lib().sa_ctc_profile_configure(1)
        else:
            acts = torch.randn(Bw, T, V + 1, device=dev, generator=torch.Generator(device=dev).manual_seed(2017))
        ms, flagged = ctc_call_ms(acts, lab, n)
        tag = "b32" if Bw == B else "b%d" % Bw
        out["ctc_%s_ms" % tag], out["ctc_%s_flagged" % tag] = ms, flagged
        if Bw == B:
            spans = ctc_chain_spans(T)
            _lib.lib().sa_ctc_profile_configure(0)
            out["ctc_chain_us"] = float(np.mean(spans)) if spans else None
        del acts
        acts = aligned_logits(Bw, lab_h, 20.0, dev, np.random.RandomState(2017))
        ms, flagged = ctc_call_ms(acts, lab, n)
        out["ctc_%s_aligned20_ms" % tag], out["ctc_%s_aligned20_flagged" % tag] = ms, flagged
        del acts, lab
    return out


def infer_legs(dev):
    """CTC.infer (/root/reference/eval.py:12-18: encoder forward + greedy collapse, here the device beam-1 decode) at the batch
    sizes the reference evaluates with (eval.py:20-22 defaults to 8, examples/timit/README.md:56-58 recommends 1): ms per
    call on host batches, S-LIBRI (T = 1000, 4 x GRU-512 unidirectional) and the shipped TIMIT model (2 conv, 4 x biGRU-256,
    T = 300 frames of 161 bins)."""
    from speech_amd.models import CTC
    timit = {"dropout": 0.0, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 32, 1]],
                                         "rnn": {"dim": 256, "layers": 4, "bidirectional": True}}}
    out = {}
    for name, cfg, freq, vocab, frames in (("slibri", S_LIBRI, F, V, T), ("timit", timit, 161, 48, 300)):
        torch.manual_seed(2017)
        model = CTC(freq, vocab, cfg).cuda()
        model.set_eval()
        rng = np.random.RandomState(11)
        for bs in (1, 8):
            batch = (tuple(rng.randn(frames, freq).astype(np.float32) for _ in range(bs)), tuple([0, 1] for _ in range(bs)))
            for _ in range(10):  # (the pinned rings of the host path grow slot by slot: eight calls until every slot exists)
                model.infer(batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 20
            for _ in range(n):
                model.infer(batch)
            torch.cuda.synchronize()
            out["%s_b%d" % (name, bs)] = (time.perf_counter() - t0) / n * 1e3
        del model
    return out


def bidirectional_leg(dev, steps=8, warm=3):
    """Untimed-by-the-headline extra leg: S-LIBRI BIDIRECTIONAL (4 x biGRU-512, 18.2 M parameters -- every shipped config
    that sets the key uses bidirectional: true, /root/reference/examples/timit/ctc_config.json:28) with dropout 0.2, one
    full train step (fwd + CTC loss + bwd + clip 200 + SGD), inputs resident, ms per step."""
    from speech_amd import ops
    from speech_amd.ctc import CTCLabels, CTCLoss
    from speech_amd.models import CTC
    cfg = {"dropout": 0.2, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 4, "bidirectional": True}}}
    torch.manual_seed(2017)
    model = CTC(F, V, cfg).cuda()
    model.set_train()
    flat_p, flat_g = model.flatten_parameters_()
    x_h, lab_h = synthetic(0)
    x = torch.from_numpy(x_h).to(dev)
    Tp = model.conv_out_size(T, 0)
    labels = CTCLabels(lab_h, np.full(B, Tp, np.int32), np.full(B, L, np.int32), dev)
    loss_fn = CTCLoss()
    norm = torch.zeros(1, device=dev)

    def step():
        model.zero_grad(set_to_none=True)
        loss = loss_fn(model.forward_impl(x), labels, None, None)
        ops.backward(loss)
        ops.stamp_health(flat_g)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        return loss

    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"bi_ms": ms, "bi_loss": float(loss.item()), "bi_params": int(flat_p.numel()),
            "bi_status": ops.persist_status()}


def train_loop_leg(model, flat_p, flat_g, world, rank, dev, steps):
    """The shipped loop on host batches (see the module docstring): `steps` steps after 3 untimed ones, bracketed by
    barrier + synchronize, max over ranks.  Four distinct host batches are cycled (a real loader hands over fresh
    arrays every step; the padded copy must not be a cache hit)."""
    from speech_amd import dist, ops
    rng = np.random.RandomState(4242 + rank)
    host = []
    for _ in range(4):
        inputs = tuple(rng.randn(T, F).astype(np.float32) for _ in range(B))
        labels = tuple(rng.randint(0, V, L).tolist() for _ in range(B))
        host.append((inputs, labels))
    norm = torch.zeros(1, device=dev)
    pipe = ops.ScalarPipe()
    warm = 3

    def batches():
        for k in range(warm + steps):
            yield host[k % len(host)]

    t0 = None
    for k, (batch, shape) in enumerate(dist.with_global_shapes(batches(), model)):
        if k == warm:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        model.set_global_batch(*shape)
        model.zero_grad(set_to_none=True)
        loss = model.loss(batch)
        ops.backward(loss)
        ops.stamp_health(flat_g)
        dist.allreduce_gradients(flat_g)
        ops.clip_sgd_step(flat_p, flat_g, None, 1e-3, 0.0, 200.0, norm_out=norm)
        pipe.push(loss, k, lag=3)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = dist.max_over_ranks(time.perf_counter() - t0, dev)
    pipe.drain()
    model.set_global_batch()
    return {"train_loop_dt": dt, "train_loop_steps": steps}


CPU_THREADS = 16  # best of {8, 16, 32, 64, 128} on the GPU box's 256 host threads (tools/cpu_baseline_threads.py:
                  # 2.22, 1.84, 3.21, 7.87, 19.98 s/step) -- torch's default of 128 oversubscribes the small GRU GEMMs


def cpu_baseline(steps=4, state0=None):
    """The reference's CPU path restated (oracle/torch_ref.py: the same torch.nn CPU modules + C CTC restatement),
    timed on this box's host cores on a bounded sample: `steps` full B=32 train steps after one warm-up.
    `state0`: the GPU leg's initial weights -- the warm-up step then starts from the same point and its loss is the
    CPU side of the matched-loss check."""
    from oracle.torch_ref import TorchRefCTC, train_step
    threads = min(CPU_THREADS, os.cpu_count() or CPU_THREADS)
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        torch.manual_seed(2017)
        model = TorchRefCTC(F, V, S_LIBRI)
        if state0 is not None:
            model.load_state_dict(state0)
        opt = torch.optim.SGD(model.parameters(), lr=1e-3, momentum=0.0)
        x_h, lab_h = synthetic(0)
        x = torch.from_numpy(x_h)
        ll = np.full(B, L, np.int32)
        loss0, _ = train_step(model, opt, x, lab_h, ll, threads=threads)
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, _ = train_step(model, opt, x, lab_h, ll, threads=threads)
        dt = (time.perf_counter() - t0) / steps
    finally:
        torch.set_num_threads(prev)
    return {"value": B / dt, "unit": "utt/s", "cores": threads, "kind": "port",
            "sample": "%d full train steps (B=32, T=1000) of oracle/torch_ref.py after 1 warm-up on %d threads; "
                      "%.2f s/step" % (steps, threads, dt), "loss": loss, "loss_step0": loss0}


def stack_gemm_rates(dev, Tp):
    """The GEMM families INSIDE the GRU stack calls (the bulk of the step's 443 GEMM-shaped GFLOP; the per-op profile cannot
    see them: a stack call is one library entry point): each shape timed as standalone launches right after the timed
    region (HIP events over 10 launches), weighted by how often the step runs it."""
    from speech_amd import ops
    H, rows = 512, Tp * B
    fams = [  # name, M, N, K, trans_a, trans_b, launches per step
        ("i2h layer 0", rows, 3 * H, 800, False, True, 1),
        ("dX layer 0", rows, 800, 3 * H, False, False, 1),  # (the upper layers' dX runs inside the recurrence kernel)
        ("dW_hh / dW_ih (H)", 3 * H, H, rows, True, False, 7),
        ("dW_ih layer 0", 3 * H, 800, rows, True, False, 1),
    ]
    out, flops, secs = [], 0.0, 0.0
    for name, M, N, K, ta, tb, count in fams:
        a = torch.randn((K, M) if ta else (M, K), device=dev)
        b = torch.randn((N, K) if tb else (K, N), device=dev)
        c = torch.empty(M, N, device=dev)
        for _ in range(2):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(a, b, trans_a=ta, trans_b=tb, out=c)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
        out.append({"shape": name, "M": M, "N": N, "K": K, "us": ms * 1e3, "TFLOP/s": tf, "per_step": count})
        flops += 2.0 * M * N * K * count
        secs += ms * 1e-3 * count
    ach = flops / secs / 1e12
    # These shapes run the split-bf16 kernel on packed operands (csrc/gemm_f32.hip): three bf16 pieces per fp32 operand,
    # six piece products per fp32 product on v_mfma_f32_32x32x16_bf16, fp32 accumulate.  `achieved` counts the ALGORITHMIC
    # fp32 flops (2 M N K) over the whole call (pack launches included); the roofline that bounds it is the dense bf16
    # MFMA peak divided by the six MFMA flops an fp32 flop costs -- not the f32-input MFMA peak, which it exceeds on
    # some shapes.
    peak = BF16_MFMA_PEAK_TFS / 6.0
    return {"kernel": "gemm_pk_kernel / gemm_pk256_kernel + pack kernels (the GEMM families inside the GRU stack, standalone "
                      "calls incl. their pack launches, weighted by their count per step)", "bound": "mfma",
            "achieved": ach, "peak": peak, "unit": "TFLOP/s (fp32-equivalent: 2*M*N*K per call)", "frac": ach / peak,
            "peak_note": "2500 TFLOP/s dense bf16 MFMA / 6 piece products per fp32 product",
            "frac_of_f32_mfma_peak": ach / MFMA_F32_PEAK_TFS, "traffic": None, "gflop_per_step": flops / 1e9,
            "shapes": out}


def kernel_source_sha():
    """Hash of the recurrence kernels' source (what profiles/hbm_traffic.json entries are stamped with)."""
    import hashlib
    h = hashlib.sha256()
    for f in ("gru.hip", "dropout.h", "common.h"):
        h.update(open(os.path.join(ROOT, "speech_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def planes_forward():
    from speech_amd import _lib
    try:
        return _lib.get_option("gru.fwd_planes") != 0 and _lib.get_option("gru.fused") != 0
    except Exception:
        return False


def elements16_backward():
    """The one-launch backward of the headline stack (H = 512, in-kernel packed gate operands) runs gru_bwd_fused16_kernel
    unless option gru.exp bit 7 asks for the three-tile exchange."""
    from speech_amd import _lib
    try:
        return (_lib.get_option("gru.exp") & 128) == 0 and _lib.get_option("gru.pack_in_kernel") != 0
    except Exception:
        return False


def roofline(prof, step_us, steps):
    """Roofline of the dominant kernel: the GRU backward recurrence (largest share of the step in every rocprof summary
    under profiles/).  Duration = the kernel's own entry-to-exit device clock inside the timed region (HIP events around
    single launches perturb the stream by several us); work = the algorithmic bytes one launch moves -- DESIGN.md 3.3's
    count, 17 B H 4 bytes per backward layer-step and 10 B H 4 per forward one (SURVEY 8(d) prices the class of kernel in
    bytes but gives no figure for the backward pass).  Because the recurrence kernels are not bandwidth-bound in any
    useful sense, every entry also carries its matrix-pipe view: `mfma_frac` = the launch's algorithmic flops / its
    duration / the peak of the arithmetic it actually runs (f32-input MFMA 157.3 TF for the backward kernel; 2500 / 6 =
    416.7 fp32-equivalent TF for the forward kernel's split-bf16 products), and `pipe_busy` = SQ_VALU_MFMA_BUSY_CYCLES /
    (4 x SQ_WAVE_CYCLES) from a separate rocprofv3 --pmc pass (profiles/pipe_busy.json, stamped like the traffic)."""
    out = {}
    try:
        busy = json.load(open(os.path.join(ROOT, "profiles", "pipe_busy.json")))
    except Exception:
        busy = {}
    # HBM bytes per launch come from separate rocprofv3 --pmc passes (they cannot run inside this process):
    # profiles/hbm_traffic.json, each entry stamped with the hash of the kernel source it was measured on
    # (tools/pmc_traffic.py --stamp).  An entry whose stamp does not match the tree being benchmarked is STALE and is
    # not reported (traffic = null, traffic_stale = true) -- a number from another kernel would be worse than none.
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json")))
    except Exception:
        traffic = {}
    sha_now = kernel_source_sha()
    for name, per_job in (("gru_bwd_step_kernel", 17), ("gru_fwd_step_kernel", 10)):
        us, kern_us, n, nsteps = step_us.get(name, (0.0, 0.0, 0, 1))
        if n == 0:
            continue
        if nsteps > 1:
            # the recurrence ran as persistent chunk kernels (one launch = `nsteps` time steps x 4 layer-jobs, launches
            # separated by the chunk's GEMMs): duration = the kernel's own entry-to-exit clock
            # (more than 64 steps in one launch: the fused kernels that run the whole stack's recurrence in one launch)
            name, us = name.replace("_step_", "_fused_" if nsteps > 64 else "_persist_"), kern_us
            if name == "gru_fwd_fused_kernel" and planes_forward():
                name = "gru_fwd_planes_kernel"  # the default one-launch forward since round 6
            if name == "gru_bwd_fused_kernel" and elements16_backward():
                name = "gru_bwd_fused16_kernel"  # the default one-launch backward at H = 512 since round 6 (16-byte elements)
        if us <= 0:
            continue
        nbytes = 4.0 * B * 512 * per_job * 4 * nsteps  # 4 layer-jobs x nsteps steps per launch, B x H fp32 each
        ach = nbytes / (us * 1e-6) / 1e9
        entry = traffic.get(name, {})
        stale = entry.get("source_sha") != sha_now
        tr = None if stale else entry.get("bytes_per_launch")
        key = name.replace("_persist_", "_step_").replace("_fused16_", "_step_").replace("_fused_", "_step_").replace("_planes_", "_step_")
        # the matrix-pipe view: 2 B 3H H flops per product and layer-step; 4 recurrent + 3 second / input products per step
        gflop = 2.0 * B * 1536 * 512 * 7 * nsteps / 1e9
        fwd = "fwd" in name
        mpeak = BF16_MFMA_PEAK_TFS / 6.0 if fwd else MFMA_F32_PEAK_TFS
        pb = busy.get(name, {})
        pb_ok = pb.get("source_sha") == sha_now
        # which roof binds: the launch's arithmetic intensity (algorithmic flops / algorithmic bytes) against the ridge point of
        # the arithmetic it runs (peak flops / 8 TB/s).  79 flop/B against a ridge of 19.7 for the backward kernel (134 against
        # 52 for the forward one): both sit on the MATRIX roof, so `bound` / `achieved` / `peak` / `frac` are the MFMA view
        # and the HBM view rides along as hbm_* (the line called a kernel "hbm"-bound that is not).
        mfma_tfs = gflop / us * 1e3
        on_mfma = (gflop * 1e9 / nbytes) > (mpeak * 1e12 / (HBM_PEAK_GBS * 1e9))
        out[key] = {"kernel": name, "bound": "mfma" if on_mfma else "hbm",
                    "achieved": mfma_tfs if on_mfma else ach, "peak": mpeak if on_mfma else HBM_PEAK_GBS,
                    "unit": "TFLOP/s" if on_mfma else "GB/s",
                    "frac": mfma_tfs / mpeak if on_mfma else ach / HBM_PEAK_GBS,
                    "arithmetic_intensity_flop_per_byte": gflop * 1e9 / nbytes,
                    "ridge_flop_per_byte": mpeak * 1e12 / (HBM_PEAK_GBS * 1e9),
                    "hbm_achieved_gbs": ach, "hbm_peak_gbs": HBM_PEAK_GBS, "hbm_frac": ach / HBM_PEAK_GBS,
                    "traffic": tr, "traffic_stale": bool(stale and entry),
                    "avg_launch_us": us, "kernel_us": kern_us,
                    "kernel_us_note": "entry of the first block to exit of the LAST block (device clock, atomic max)",
                    "samples": n, "bytes_per_launch": nbytes, "time_steps_per_launch": nsteps,
                    "gflop_per_launch": gflop, "mfma_tflops": gflop / us * 1e3, "mfma_peak_tflops": mpeak,
                    "mfma_frac": gflop / us * 1e3 / mpeak,
                    "mfma_note": ("both products of a step as six bf16 piece products per fp32 product (gru_fwd_planes_kernel): "
                                  "peak = 2500 TF dense bf16 / 6" if fwd else
                                  "exact fp32 products on v_mfma_f32_16x16x4_f32: peak = the f32-input MFMA rate"),
                    "pipe_busy": pb.get("pipe_busy") if pb_ok else None,
                    "pipe_busy_source": pb.get("source") if pb_ok else None}
    g = prof.get("gemm")
    gemm = None
    if g and g["ms"] > 0:
        ach = g["work"] / (g["ms"] * 1e-3) / 1e12
        gemm = {"kernel": "gemm_f32_kernel (calls outside the GRU stack)", "bound": "mfma", "achieved": ach,
                "peak": MFMA_F32_PEAK_TFS, "unit": "TFLOP/s", "frac": ach / MFMA_F32_PEAK_TFS, "traffic": None}
    main_entry = out.get("gru_bwd_step_kernel") or out.get("gru_fwd_step_kernel") or gemm
    if main_entry is not None and "hbm_frac" in main_entry:
        main_entry = dict(main_entry, note="in time a latency-bound recurrence (a cross-CU hand-off of the state every time "
                          "step); by the roofline model on the matrix roof; see DESIGN.md 3.3")
    return main_entry, {"gru_fwd_step_kernel": out.get("gru_fwd_step_kernel"), "gemm_f32_kernel": gemm}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra legs (train loop, M-CTC, GEMM rates, "
                    "bidirectional): A/B experiments")
    ap.add_argument("--scaling", choices=("both", "weak", "strong"), default="both",
                    help="weak: B=32 per GPU (the headline); strong: the SAME global batch of 32 sharded over the GPUs is the "
                         "headline; both (default): the weak headline plus, under --gpus N > 1, a strong-scaling leg in "
                         "`strong_scaling`")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # run by hand: become the launcher (one rank per GPU over RCCL, rendezvous on the loopback address)
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    from speech_amd import dist
    world, rank, local = dist.init()
    if world != args.gpus:
        if rank == 0:
            print("error: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    r = gpu_leg(args, world, rank, local)
    if rank != 0:
        return
    ms = r["dt"] / args.steps * 1e3
    strong = args.scaling == "strong"
    gbatch = B if strong else B * world
    chain = r.get("ctc_chain_us")
    out = {
        "metric": "utterances/sec + CTC-loss step time, T=1000 B=32 F=80 |V|=29, 1/2/4/8 GPUs",
        "value": gbatch * args.steps / r["dt"], "unit": "utt/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong" if strong else "weak",
        "vs_baseline": None,
        "dtype": "f32", "data": "synthetic (seed 2017; random-init weights)",
        "dtype_note": "fp32 storage and accumulation throughout; recurrences, convolutions and small products on the "
                      "f32-input MFMA; the large GEMMs multiply exact three-piece bf16 splits of their fp32 operands on "
                      "the bf16 MFMA (six piece products per fp32 product, dropped terms < 2^-26 of it)",
        "config": {"workload": "S-LIBRI M-STEP: full CTC train step (fwd + CTC loss + bwd + clip 200 + SGD), "
                               "B=%d per GPU, T=1000, F=80, |V|+1=29, L=100, conv [32,5,32,2] -> T'=%d, "
                               "4xGRU-512 uni, fc->29, %d params" % (r["per_gpu_batch"], r["Tp"], r["params"]),
                   "global_batch": gbatch, "parallelism": "dp%d" % world},
        "strong_scaling": r.get("strong"),
        "strong_scaling_note": "SURVEY 8(e) reports both: `value` is %s scaling; `strong_scaling` (under --gpus N > 1 with the "
                               "default --scaling both) is the SAME global batch of 32 sharded over the GPUs, B/N each, one "
                               "gradient all-reduce per step; at N = 1 the two coincide" % ("strong" if strong else "weak"),
        "ctc_loss_step_ms": r.get("ctc_b32_ms"),
        "ctc_chain_us": chain, "ctc_chain_ns_per_step": chain * 1e3 / T if chain else None,
        "ctc_chain_in_step_us": r.get("ctc_chain_in_step_us"),
        "ctc_chain_in_step_ns_per_step": r["ctc_chain_in_step_us"] * 1e3 / r["Tp"] if r.get("ctc_chain_in_step_us") else None,
        "ctc_note": "M-CTC (B=32, T=1000, |V|+1=29, L=100) forward + gradient per call, timed directly: ctc_loss_step_ms = HIP "
                    "events over twenty calls back to back (log-softmax + alpha/beta + gradient rows + the gated log-domain "
                    "launch); ctc_chain_us = the device-clock span of the alpha/beta kernel alone in those calls (earliest "
                    "workgroup in to latest out), ctc_chain_in_step_us the same inside the timed train steps (T'=498); "
                    "kernel_time_ms_per_step.ctc_loss = HIP events around the call inside a train step.  (Rounds 3-4 also "
                    "printed a 'busy chip' figure, t(GEMM + loss) - t(GEMM): a subtraction artefact, removed.)",
        "ctc_b32_aligned20_ms": r.get("ctc_b32_aligned20_ms"),
        "ctc_flagged": {k[4:]: r.get(k) for k in ("ctc_b32_flagged", "ctc_b32_aligned20_flagged", "ctc_b4096_flagged",
                                                  "ctc_b4096_aligned20_flagged")},
        "ctc_aligned_note": "aligned20 = alignment-shaped logits, margin 20 (what a trained model emits); ctc_flagged = "
                            "utterances the probability-domain pass handed to the exact log-domain kernels in that call",
        "ctc_b4096_aligned20_ms": r.get("ctc_b4096_aligned20_ms"),
        "infer_ms": r.get("infer_ms"),
        "infer_note": "CTC.infer per call on host batches (collate + H2D + encoder forward + device beam-1 decode + labels back), "
                      "batch 1 and 8: S-LIBRI (T=1000, 4xGRU-512 uni) and the shipped TIMIT model (4xbiGRU-256, T=300)",
        "ctc_b4096_ms": r.get("ctc_b4096_ms"),
        "bi_ms_per_step": r.get("bi_ms"),
        "bi_note": "untimed-by-the-headline extra leg: S-LIBRI BIDIRECTIONAL (4 x biGRU-512, %s params) with dropout 0.2, full "
                   "train step, B=32, inputs resident; loss %s, persist status %s"
                   % (r.get("bi_params"), r.get("bi_loss"), r.get("bi_status")),
        "loss": r["loss"], "grad_norm": r["grad_norm"],
        "loss_step0": r["loss_step0"], "loss_rel_err": None, "persist_status": r["persist_status"],
        "train_loop_utt_s": B * world * r["train_loop_steps"] / r["train_loop_dt"] if r["train_loop_steps"] else None,
        "train_loop_ms_per_step": r["train_loop_dt"] / r["train_loop_steps"] * 1e3 if r["train_loop_steps"] else None,
        "train_loop_note": "untimed-by-the-headline extra leg: train.py's loop on HOST batches (model.loss(batch): collate "
                           "into pinned memory + H2D, backward, all-reduce, clip+SGD, lagged read-back; global-shape "
                           "exchange one step ahead under --gpus N); %d steps" % r["train_loop_steps"],
        "roofline": None,
        "kernel_time_ms_per_step": {k: v["ms"] / r["prof_steps"] for k, v in sorted(r["prof"].items())},
    }
    out["roofline"], out["roofline_other"] = roofline(r["prof"], r["step_us"], r["prof_steps"])
    # The step against the arithmetic it ACTUALLY uses: the backward recurrence on the f32-input MFMA
    # (157.3 TF), the stack's GEMMs and the forward recurrence on split-bf16 products (2500 / 6 = 416.7 fp32-equivalent TF),
    # everything else (convolutions, classifier, small products: 13.6 GFLOP) on the f32-input MFMA.
    rec = 2.0 * B * 1536 * 512 * 7 * r["Tp"] / 1e9
    gem = (r.get("stack_gemm") or {}).get("gflop_per_step") or 292.949
    rest = 13.6
    floor_ms = rec / MFMA_F32_PEAK_TFS + (gem + rec) / (BF16_MFMA_PEAK_TFS / 6.0) + rest / MFMA_F32_PEAK_TFS
    out["step_floor_ms"] = floor_ms
    out["step_floor_frac"] = floor_ms / ms
    out["step_floor_note"] = ("matrix-pipe time of the step's %.1f GFLOP at the peak of the arithmetic each part runs: backward "
                              "recurrence %.1f GFLOP at 157.3 TF, forward recurrence %.1f + stack GEMMs %.1f GFLOP at 416.7 "
                              "fp32-equivalent TF (six bf16 piece products per fp32 product), the rest %.1f GFLOP at 157.3 TF; "
                              "step_floor_frac = floor / ms_per_step" % (2 * rec + gem + rest, rec, rec, gem, rest))
    out["roofline_other"]["gemm_f32_small_calls"] = out["roofline_other"].pop("gemm_f32_kernel")  # fc / misc products
    out["roofline_other"]["gemm_f32_kernel"] = r.get("stack_gemm")
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cb = cpu_baseline(state0=r["state0"])
        out["loss_rel_err"] = abs(r["loss_step0"] - cb["loss_step0"]) / abs(cb["loss_step0"])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
