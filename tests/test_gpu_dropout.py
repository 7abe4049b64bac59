"""Dropout INSIDE the HIP kernels (speech_amd/csrc/dropout.h) against the CPU oracle run on the SAME masks.

Reference: nn.Dropout(p) behind every conv ReLU and nn.GRU(dropout=p) between the GRU layers
(/root/reference/speech/models/model.py:25-27,35-39); every shipped config trains with p = 0.2 .. 0.5.
The kernels derive each mask bit from (seed, mask stream, element index) with Philox4x32-10; oracle/philox_ref.py
restates that in NumPy, so (i) the device masks are compared with it BIT FOR BIT, and (ii) the whole train-mode
forward + backward of speech_amd.models.CTC is compared with oracle/torch_ref.TorchRefCTC -- the reference's own
torch.nn CPU modules -- fed those masks: logits within 2e-4 of their range, loss rtol 1e-4 (north_star), every parameter
gradient within 1e-3 of its maximum and 1e-3 in relative L2.  The cases walk every place a mask is applied: the direct
conv epilogue, the generic conv path, the fused one-launch GRU kernels (forward write, backward d h_out), the chunked
wavefront (per-chunk element-wise launches with index offsets), the step kernels, and bidirectional stacks."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_device_masks_are_bit_identical_to_the_restatement():
    from oracle import philox_ref
    from speech_amd import ops
    dev = torch.device("cuda:0")
    for n, p, seed, stream, idx0 in [(1000, 0.4, 2017, 0, 0), (4099, 0.2, (1 << 61) + 12345, 64, 0),
                                     (777, 0.5, 3, 66, 1234567), (5, 0.3, 9, 1, 3), (100000, 0.25, 42, 65, 2 ** 33 + 5)]:
        got = ops.dropout_mask(n, p, seed, stream, dev, idx0=idx0).cpu().numpy()
        np.testing.assert_array_equal(got, philox_ref.mask(n, p, seed, stream, idx0))
    x = torch.randn(3, 17, 5, device=dev)
    y = ops.dropout_apply(x, 0.4, 11, 7)
    np.testing.assert_array_equal(y.cpu().numpy().ravel(), x.cpu().numpy().ravel() * philox_ref.mask(x.numel(), 0.4, 11, 7))
    assert torch.equal(ops.dropout_apply(x, 0.0, 11, 7), x)


def _conv_shapes(cfg, B, T, F):
    from speech_amd.ops import conv_out_size
    shapes, t, f = [], T, F
    for out_c, kh, kw, s in cfg["encoder"]["conv"]:
        t, f = conv_out_size(t, kh, s), conv_out_size(f, kw, s)
        shapes.append((B, out_c, t, f))
    return shapes, t


def run_case(F, V, B, T, L, cfg, seed=2017, mask_seed=(1 << 40) + 77, grad_tol=1e-3):
    """Train-mode forward + loss + backward of the product at a FIXED mask key vs the torch CPU oracle on those masks.
    Returns the worst relative-L2 gradient error (for the record)."""
    from oracle import philox_ref
    from oracle.torch_ref import TorchRefCTC, _CTCRef
    from speech_amd import _lib
    from speech_amd.models import CTC
    torch.manual_seed(seed)
    model = CTC(F, V, cfg)
    state = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.cuda()
    rng = np.random.RandomState(seed)
    x = rng.randn(B, T, F).astype(np.float32)
    labels = tuple(rng.randint(0, V, L) for _ in range(B))
    batch = (tuple(x[b] for b in range(B)), labels)
    model.set_train()
    model._plan.fixed_seed = mask_seed
    out = model(batch)
    loss = model.loss(batch)
    loss.backward()
    assert _lib.lib().sa_gru_persist_status() == 0

    p = cfg["dropout"]
    rnn = cfg["encoder"]["rnn"]
    D = 2 if rnn["bidirectional"] else 1
    shapes, Tp = _conv_shapes(cfg, B, T, F)
    masks = philox_ref.encoder_masks(p, mask_seed, shapes, (B, Tp, D * rnn["dim"]), rnn["layers"])
    prev = torch.get_num_threads()
    torch.set_num_threads(min(16, prev))
    try:
        ref = TorchRefCTC(F, V, cfg)
        ref.load_state_dict(state)
        logits = ref(torch.from_numpy(x), masks)
        flat = np.concatenate(labels).astype(np.int32)
        rl = _CTCRef.apply(logits, flat, np.full(B, Tp, np.int32), np.full(B, L, np.int32), ref.blank, 0)
        rl.backward()
    finally:
        torch.set_num_threads(prev)
    want_logits = logits.detach().numpy()
    got_logits = out.detach().cpu().numpy()
    span = float(want_logits.max() - want_logits.min())
    assert np.abs(got_logits - want_logits).max() <= 2e-4 * span
    assert abs(float(loss.item()) - float(rl.item())) <= 1e-4 * abs(float(rl.item()))
    worst = 0.0
    want = {k: q.grad.detach().numpy() for k, q in ref.named_parameters()}
    for k, q in model.named_parameters():
        got = q.grad.cpu().numpy()
        scale = max(float(np.abs(want[k]).max()), 1e-10)
        err = float(np.abs(got - want[k]).max())
        rel = float(np.linalg.norm((got - want[k]).ravel()) / max(np.linalg.norm(want[k].ravel()), 1e-20))
        assert err <= grad_tol * scale and rel <= grad_tol, (k, err, scale, rel)
        worst = max(worst, rel)
    return worst


UNI256 = {"dropout": 0.3, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 256, "layers": 3, "bidirectional": False}}}


def test_fused_one_launch_kernels_apply_the_masks():
    # conv: direct kernel epilogue; GRU: gru_fwd_fused_kernel writes h * mask, gru_bwd_fused_kernel<.., DROP> masks d h_out;
    # B = 20: a ragged second batch tile; T' = 60
    run_case(F=40, V=20, B=20, T=150, L=8, cfg=UNI256)


@pytest.mark.parametrize("env", [{"SA_GRU_FUSED": "0"}, {"SA_GRU_BWD_ONE": "0"}, {"SA_GRU_FUSE_DX": "0"},
                                 {"SA_GRU_PERSIST": "0"}, {"SA_GRU_TILED": "0"}, {"SA_GEMM_EXACT": "0"},
                                 {"SA_GEMM_EXACT": "0", "SA_GRU_SHARED_PACK": "0"}, {"SA_GEMM_EXACT": "1"}],
                         ids=["chunked_fwd", "chunked_fused_bwd", "gemm_dx", "step_kernels", "round1_bwd",
                              "every_gemm_packed_split_bf16", "packed_split_bf16_one_pack_per_product",
                              "every_gemm_exact_f32"])
def test_every_gru_path_applies_the_same_masks(monkeypatch, env):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    run_case(F=40, V=20, B=20, T=150, L=8, cfg=UNI256)


@pytest.mark.parametrize("B,H,layers,bi", [(9, 128, 4, False), (32, 128, 3, False), (48, 256, 2, False), (16, 256, 3, True),
                                           (32, 256, 2, True)])
def test_shared_packed_operands_give_the_per_product_weight_gradients(monkeypatch, B, H, layers, bi):
    # WGradIssuer::issue_shared (gru.hip): the gate gradients of a layer packed ONCE as a 4H-row operand that dW_ih reads
    # as rows [0, 3H) and dW_hh as rows [0, 2H) + [3H, 4H) (a row-block jump), bias gradients from row sums through the
    # same row map -- against one pack per product.  With B a multiple of 16 the backward recurrence kernel writes that
    # operand itself (gru_bwd_fused_kernel<PACKG>: LDS transpose, split, bias sums in registers); SA_GRU_PACK_IN_KERNEL=0
    # leaves it to the pack launch.  Same pieces, same product kernels: the three agree to fp32 summation-order noise
    # (the weight gradients of the two shared forms bit for bit).  B = 9: a reduction length (522) that is not a multiple
    # of the 16-wide packed k tile and a ragged batch tile; B = 48: three batch tiles (two passes of the one-launch kernel
    # at H = 256 would be four).  Bidirectional stacks: the same, layer by layer and per direction (issue_shared_bi: both
    # directions' dW_ih read ONE packed copy of the layer's input; XCD-filtered launches beside the next layer's
    # recurrence), against the generic per-product path.
    from speech_amd.models import CTC
    cfg = {"dropout": 0.3, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": H, "layers": layers, "bidirectional": bi}}}
    monkeypatch.setenv("SA_GEMM_EXACT", "0")
    rng = np.random.RandomState(5)
    x = rng.randn(B, 120, 40).astype(np.float32)
    labels = tuple(rng.randint(0, 20, 6) for _ in range(B))
    batch = (tuple(x[b] for b in range(B)), labels)
    grads = {}
    for mode, env in (("kernel", {"SA_GRU_SHARED_PACK": "1", "SA_GRU_PACK_IN_KERNEL": "1"}),
                      ("shared", {"SA_GRU_SHARED_PACK": "1", "SA_GRU_PACK_IN_KERNEL": "0"}),
                      ("each", {"SA_GRU_SHARED_PACK": "0", "SA_GRU_PACK_IN_KERNEL": "0"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        torch.manual_seed(3)
        model = CTC(40, 20, cfg).cuda()
        model.set_train()
        model._plan.fixed_seed = 99
        model.loss(batch).backward()
        grads[mode] = {k: q.grad.clone() for k, q in model.named_parameters()}
    for k, want in grads["each"].items():
        for mode in ("kernel", "shared"):
            rel = float((grads[mode][k] - want).norm() / want.norm().clamp_min(1e-20))
            assert rel <= 2e-6, (mode, k, rel)
        if "weight" in k and not bi:
            assert torch.equal(grads["kernel"][k], grads["shared"][k]), k


@pytest.mark.parametrize("p_drop,B", [(0.0, 32), (0.3, 32), (0.0, 48)])
def test_backward_on_16_byte_elements_gives_the_three_tile_gradients(monkeypatch, p_drop, B):
    # gru_bwd_fused16_kernel (H = 512 with the in-kernel packed gate operand: a thread publishes {dpr, dpz, dqn, dpn} with one
    # 16-byte store, no r rows are read, an MFMA takes one gate of a fragment) against gru_bwd_fused_kernel<8, FUSE, DROP, PACKG>
    # (option gru.exp bit 7: three 1 KB tiles per step, dqn = dpn r formed by the consumer): the same products summed over k in
    # another order -- every parameter gradient to fp32 summation-order noise, with and without inter-layer dropout.  T' = 58
    # steps: the ring (4 slots) wraps many times; B = 32: two batch tiles; B = 48: three, in two passes of the launch.
    from speech_amd.models import CTC
    cfg = {"dropout": p_drop, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 512, "layers": 3, "bidirectional": False}}}
    monkeypatch.setenv("SA_GEMM_EXACT", "0")  # (every product packed, whatever its size: the path the in-kernel operand belongs to)
    rng = np.random.RandomState(11)
    x = rng.randn(B, 120, 40).astype(np.float32)
    labels = tuple(rng.randint(0, 20, 6) for _ in range(B))
    batch = (tuple(x[b] for b in range(B)), labels)
    grads = {}
    for mode, exp in (("elements16", "0"), ("tiles3", "128")):
        monkeypatch.setenv("SA_GRU_EXP", exp)
        torch.manual_seed(3)
        model = CTC(40, 20, cfg).cuda()
        model.set_train()
        model._plan.fixed_seed = 99
        loss = model.loss(batch)
        loss.backward()
        assert torch.isfinite(loss)
        grads[mode] = {k: q.grad.clone() for k, q in model.named_parameters()}
    for k, want in grads["tiles3"].items():
        got = grads["elements16"][k]
        assert torch.isfinite(got).all(), k
        rel = float((got - want).norm() / want.norm().clamp_min(1e-20))
        assert rel <= 2e-6, (k, rel)
    # (two different kernels did run: the same sums in another order do not agree bit for bit over 3.9 M weight gradients)
    assert any(not torch.equal(grads["elements16"][k], grads["tiles3"][k]) for k in grads["tiles3"])


@pytest.mark.parametrize("H,B", [(384, 32), (320, 20)])
def test_widths_between_the_kernel_templates(H, B):
    # the XCD-local kernels take every multiple of 64 from 128 to 512 (H / 16 unit tiles per sync group; the CUs of an XCD
    # beyond its last whole group idle): H = 384 (weight gradients on shared packed operands, packed by launches), H = 320
    # (one pack per product) -- whole train-mode forward + backward against the torch CPU oracle on the same masks
    cfg = {"dropout": 0.2, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": H, "layers": 3, "bidirectional": False}}}
    run_case(F=40, V=20, B=B, T=110, L=6, cfg=cfg)


def test_h128_stack_and_two_convs():
    # H = 128: fused forward, chunked persistent backward with GEMM d h_out (+ element-wise mask per chunk);
    # two direct convs (the second on 32 channels): both epilogues mask, both backward passes scale
    cfg = {"dropout": 0.4, "encoder": {"conv": [[32, 5, 32, 2], [32, 5, 8, 1]],
                                       "rnn": {"dim": 128, "layers": 4, "bidirectional": False}}}
    run_case(F=80, V=30, B=9, T=120, L=6, cfg=cfg)


def test_bidirectional_stack():
    cfg = {"dropout": 0.4, "encoder": {"conv": [[32, 5, 32, 2]], "rnn": {"dim": 256, "layers": 3, "bidirectional": True}}}
    run_case(F=40, V=20, B=8, T=120, L=8, cfg=cfg)


def test_generic_conv_path_and_step_kernels():
    # odd kernel widths -> im2col + GEMM convs (mask as one in-place pass); H = 24 -> step kernels, bidirectional
    cfg = {"dropout": 0.5, "encoder": {"conv": [[8, 5, 11, 2], [8, 3, 7, 1]],
                                       "rnn": {"dim": 24, "layers": 3, "bidirectional": True}}}
    run_case(F=40, V=12, B=3, T=70, L=4, cfg=cfg)


def test_fresh_masks_every_pass_and_eval_has_none():
    from speech_amd.models import CTC
    torch.manual_seed(1)
    model = CTC(40, 20, UNI256).cuda()
    rng = np.random.RandomState(0)
    x = rng.randn(4, 90, 40).astype(np.float32)
    batch = (tuple(x[b] for b in range(4)), tuple(rng.randint(0, 20, 5) for _ in range(4)))
    model.set_train()
    a, b = model(batch), model(batch)
    assert not torch.equal(a, b)  # a new Philox key per forward pass
    torch.manual_seed(5)
    c = model(batch)
    torch.manual_seed(5)
    d = model(batch)
    assert torch.equal(c, d)      # keys come from torch's CPU generator: manual_seed reproduces a run
    model.set_eval()
    e, f = model(batch), model(batch)
    assert torch.equal(e, f)
