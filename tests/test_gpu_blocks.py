"""HIP encoder building blocks (through the C ABI via speech_amd.ops) against the fp64 NumPy oracle
(oracle/encoder_np.py, itself pinned to the live reference by tests/golden/encoder_*.npz).

Tolerances: fp32 kernels vs fp64 oracle.  A length-K fp32 dot product of O(1) terms carries ~1e-7 * sqrt(K) .. 1e-7 * K
relative error; tests use rtol 2e-5 / atol scaled by the result magnitude unless noted."""
import numpy as np
import pytest
import torch

from oracle import encoder_np as E

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, want, rtol=2e-5, atol_scale=2e-6):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    atol = atol_scale * max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(got, want, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (130, 29, 29), (1, 1, 1), (257, 200, 100), (64, 1536, 800),
                                   (300, 512, 37)])
def test_gemm_forms(ta, tb, M, N, K):
    from speech_amd import ops
    rng = np.random.RandomState(M * 7 + N * 3 + K + ta * 2 + tb)
    a = rng.randn(K, M) if ta else rng.randn(M, K)
    b = rng.randn(N, K) if tb else rng.randn(K, N)
    bias = rng.randn(N)
    c = ops.gemm(dev(a), dev(b), trans_a=bool(ta), trans_b=bool(tb), bias=dev(bias))
    want = (a.T if ta else a) @ (b.T if tb else b) + bias
    close(c, want, rtol=1e-5, atol_scale=1e-6 * np.sqrt(K))


def test_gemm_is_exact_fp32_on_integers():
    # f32-input MFMA = an fp32 fma chain: products of small integers are exact, so the result is bit-exact
    from speech_amd import ops
    rng = np.random.RandomState(0)
    a = rng.randint(-8, 9, (200, 96)).astype(np.float32)
    b = rng.randint(-8, 9, (96, 72)).astype(np.float32)
    c = ops.gemm(dev(a), dev(b))
    assert np.array_equal(c.cpu().numpy(), a @ b)


@pytest.mark.parametrize("ta,tb,M,N,K", [(0, 1, 700, 300, 520), (0, 0, 333, 800, 1536), (1, 0, 1536, 512, 4000),
                                         (1, 1, 130, 257, 77)])
def test_split_bf16_gemm_error_budget(monkeypatch, ta, tb, M, N, K):
    """The GEMM kernel of the large products multiplies three-piece bf16 splits of the fp32 operands on the bf16 MFMA (six
    of the nine piece products, fp32 accumulate; operands packed once per call; csrc/gemm_f32.hip).  The dropped terms are < 2^-26 of each product -- below the
    rounding of the fp32 accumulation itself -- so against an fp64 product its error must sit where the f32-input MFMA
    kernel's (SA_GEMM_EXACT=1: exact fp32 products, fp32 accumulate) does: both within the fp32 accumulation bound,
    the split kernel within 3x of the exact kernel plus one part in 10^7."""
    from speech_amd import ops
    rng = np.random.RandomState(M + N + K)
    # wide dynamic range and mixed signs: mantissas of all lengths, cancellation in the sums
    a = (rng.randn(K, M) if ta else rng.randn(M, K)) * np.exp(rng.randn(1, M if ta else K))
    b = (rng.randn(N, K) if tb else rng.randn(K, N)) * np.exp(rng.randn(N if tb else K, 1) if tb else rng.randn(1, N))
    a, b = a.astype(np.float32), b.astype(np.float32)
    A, Bm = (a.T if ta else a).astype(np.float64), (b.T if tb else b).astype(np.float64)
    ref = A @ Bm
    scale = np.abs(A) @ np.abs(Bm)          # the scale rounding errors live on
    got = {}
    for mode in ("split", "exact"):
        monkeypatch.setenv("SA_GEMM_EXACT", "1" if mode == "exact" else "0")  # "0": the packed path whatever the size
        c = ops.gemm(dev(a), dev(b), trans_a=bool(ta), trans_b=bool(tb)).cpu().numpy().astype(np.float64)
        got[mode] = float((np.abs(c - ref) / scale).max()), float(np.linalg.norm(c - ref) / np.linalg.norm(ref))
    bound = 2.0 ** -24 * (2.0 + np.sqrt(K))  # generous statistical fp32 accumulation bound, relative to |A||B|
    assert got["exact"][0] <= bound and got["split"][0] <= bound, (got, bound)
    assert got["split"][0] <= 3.0 * got["exact"][0] + 1e-7 and got["split"][1] <= 3.0 * got["exact"][1] + 1e-7, got


def test_packed_gemm_all_forms_edges_and_exact_integers(monkeypatch):
    """Every transpose form, ragged M / N / K (zero-padded tiles), bias, alpha / beta and split-K through the packed
    split-bf16 path (forced: SA_GEMM_EXACT=0); products of small integers stay bit-exact (every piece product is exact)."""
    from speech_amd import ops
    monkeypatch.setenv("SA_GEMM_EXACT", "0")
    rng = np.random.RandomState(5)
    for ta, tb, M, N, K in [(0, 1, 200, 72, 96), (1, 0, 130, 300, 77), (0, 0, 64, 129, 250), (1, 1, 257, 65, 33)]:
        a = rng.randn(K, M) if ta else rng.randn(M, K)
        b = rng.randn(N, K) if tb else rng.randn(K, N)
        bias = rng.randn(N)
        c = ops.gemm(dev(a), dev(b), trans_a=bool(ta), trans_b=bool(tb), bias=dev(bias))
        close(c, (a.T if ta else a) @ (b.T if tb else b) + bias, rtol=1e-5, atol_scale=1e-6 * np.sqrt(K))
    ai = rng.randint(-8, 9, (200, 96)).astype(np.float32)
    bi = rng.randint(-8, 9, (96, 72)).astype(np.float32)
    assert np.array_equal(ops.gemm(dev(ai), dev(bi)).cpu().numpy(), ai @ bi)
    K, M, N = 5000, 96, 200   # split-K with alpha / beta
    a, b, c0 = rng.randn(K, M), rng.randn(K, N), rng.randn(M, N)
    out = dev(c0)
    ops.gemm(dev(a), dev(b), trans_a=True, out=out, alpha=0.5, beta=2.0)
    close(out, 0.5 * a.T @ b + 2.0 * c0, rtol=1e-5, atol_scale=1e-6 * np.sqrt(K))
    out2 = dev(c0)
    ops.gemm(dev(a), dev(b), trans_a=True, out=out2, alpha=0.5, beta=2.0)
    assert torch.equal(out, out2)  # deterministic


def test_gemm_split_k_alpha_beta_and_strided():
    from speech_amd import ops
    rng = np.random.RandomState(1)
    # weight-gradient shape: few output tiles, long K -> split-K path (deterministic reduce)
    K, M, N = 5000, 96, 200
    a, b = rng.randn(K, M), rng.randn(K, N)
    c0 = rng.randn(M, N)
    out = dev(c0)
    ops.gemm(dev(a), dev(b), trans_a=True, out=out, alpha=0.5, beta=2.0)
    close(out, 0.5 * a.T @ b + 2.0 * c0, rtol=1e-5, atol_scale=1e-6 * np.sqrt(K))
    out2 = dev(c0)
    ops.gemm(dev(a), dev(b), trans_a=True, out=out2, alpha=0.5, beta=2.0)
    assert torch.equal(out, out2)  # run-to-run deterministic
    # strided operands: a column slice (lda > K) and an output slice (ldc > N)
    big = dev(rng.randn(64, 300))
    w = dev(rng.randn(40, 100))
    outbig = torch.zeros(64, 90, device="cuda")
    ops.gemm(big[:, 100:200], w, trans_b=True, out=outbig[:, 10:50])
    close(outbig[:, 10:50], big[:, 100:200].cpu().numpy().astype(np.float64) @ w.cpu().numpy().astype(np.float64).T)
    assert float(outbig[:, :10].abs().max()) == 0 and float(outbig[:, 50:].abs().max()) == 0


# ------------------------------------------------------------------------------------------------------------- conv
@pytest.mark.parametrize("B,C,T,F,O,kh,kw,s", [(2, 1, 40, 40, 8, 5, 32, 2), (3, 4, 21, 17, 6, 5, 7, 1),
                                               (1, 1, 9, 33, 32, 5, 32, 2), (2, 3, 30, 30, 5, 3, 4, 3),
                                               # the stacked 32-channel conv of the shipped configs (narrow-kernel
                                               # col2im: 8 channels per wave, several channel passes when C > 32)
                                               (2, 32, 30, 37, 32, 5, 8, 2), (1, 40, 13, 21, 8, 5, 8, 2),
                                               # kw > 32: the one-lane-per-tap col2im
                                               (1, 2, 12, 40, 4, 3, 33, 2),
                                               # direct kernels with several input channels (<= 32, even kh * kw): the
                                               # stacked convs of the TIMIT / WSJ configs, stride 1 / 2 / other, a
                                               # ragged last slab, more positions than one accumulator round
                                               (2, 32, 20, 65, 32, 5, 32, 1), (2, 8, 24, 40, 32, 5, 32, 2),
                                               (1, 32, 50, 33, 32, 5, 8, 2), (3, 5, 23, 31, 7, 4, 6, 3),
                                               (1, 2, 70, 100, 3, 2, 2, 1)])
@pytest.mark.parametrize("feature_layout", [False, True])
def test_conv_relu_fwd_bwd(B, C, T, F, O, kh, kw, s, feature_layout):
    from speech_amd import ops
    rng = np.random.RandomState(B + C + T)
    x = rng.randn(B, C, T, F)
    w = rng.randn(O, C, kh, kw) / np.sqrt(C * kh * kw)
    b = rng.randn(O) * 0.1
    y_ref, cols = E.conv_relu_fwd(x, w, b, s)
    y, ys = ops.conv2d_relu_fwd(dev(x), dev(w), dev(b), s, "btf" if feature_layout else "nchw")
    Bo, Oo, To, Fo = y_ref.shape
    y_nchw = y.view(B, To, O, Fo).permute(0, 2, 1, 3) if feature_layout else y
    close(y_nchw, y_ref, rtol=1e-5, atol_scale=2e-6)
    dy = rng.randn(*y_ref.shape)
    dx_ref, dw_ref, db_ref = E.conv_relu_bwd(dy, y_ref, cols, x.shape, w, s, need_dx=True)
    dy_dev = dev(dy.transpose(0, 2, 1, 3).reshape(B, To, O * Fo)) if feature_layout else dev(dy)
    dx, dw, db = ops.conv2d_relu_bwd(dev(x), dev(w), y, dy_dev, ys, s, need_dx=True)
    close(dw, dw_ref, rtol=2e-5, atol_scale=1e-5)
    close(db, db_ref, rtol=2e-5, atol_scale=1e-5)
    close(dx, dx_ref, rtol=2e-5, atol_scale=1e-5)


@pytest.mark.parametrize("B,T,F,O,kh,kw,s", [(2, 70, 80, 32, 5, 32, 2),    # the first conv of every shipped config
                                             (1, 300, 161, 32, 5, 32, 2),  # ... at the TIMIT feature width
                                             (3, 41, 40, 8, 5, 11, 2),     # kw odd -> generic path (im2col + GEMM)
                                             (2, 37, 33, 16, 4, 7, 1),     # kw odd, kh * kw even -> generic path
                                             (2, 37, 33, 16, 3, 6, 1),     # kw / 2 odd: groups of one tap pair
                                             (2, 19, 50, 5, 2, 12, 3),     # groups of two, stride 3
                                             (2, 26, 44, 9, 5, 8, 2)])     # groups of four
@pytest.mark.parametrize("layout", ["nchw", "btf", "tbf"])
def test_first_conv_direct_kernels(B, T, F, O, kh, kw, s, layout):
    """One input channel: the direct MFMA kernels (no im2col matrix), forward and weight / bias gradient (need_dx=False,
    as the encoder calls the first conv), every output layout; an odd kernel width falls back to the generic path."""
    from speech_amd import ops, _lib
    rng = np.random.RandomState(T + F)
    x = rng.randn(B, 1, T, F)
    w = rng.randn(O, 1, kh, kw) / np.sqrt(kh * kw)
    b = rng.randn(O) * 0.1
    y_ref, cols = E.conv_relu_fwd(x, w, b, s)
    _, _, To, Fo = y_ref.shape
    assert bool(_lib.lib().sa_conv2d_is_direct(1, F, O, kh, kw, s)) == (kw % 2 == 0)
    res = ops.conv2d_relu_fwd(dev(x), dev(w), dev(b), s, layout, keep_cols=True)
    y, ys = res[0], res[1]
    to_nchw = {"nchw": lambda t: t, "btf": lambda t: t.view(B, To, O, Fo).permute(0, 2, 1, 3),
               "tbf": lambda t: t.view(To, B, O, Fo).permute(1, 2, 0, 3)}[layout]
    close(to_nchw(y), y_ref, rtol=1e-5, atol_scale=2e-6)
    dy = rng.randn(*y_ref.shape)
    _, dw_ref, db_ref = E.conv_relu_bwd(dy, y_ref, cols, x.shape, w, s, need_dx=False)
    dy_l = {"nchw": dy, "btf": dy.transpose(0, 2, 1, 3).reshape(B, To, O * Fo),
            "tbf": dy.transpose(2, 0, 1, 3).reshape(To, B, O * Fo)}[layout]
    dx, dw, db = ops.conv2d_relu_bwd(dev(x), dev(w), y, dev(dy_l), ys, s, need_dx=False, cols=res[2])
    assert dx is None
    close(dw, dw_ref, rtol=2e-5, atol_scale=1e-5)
    close(db, db_ref, rtol=2e-5, atol_scale=1e-5)


# -------------------------------------------------------------------------------------------------------------- GRU
def gru_case(B, T, I, H, reverse, seed):
    rng = np.random.RandomState(seed)
    k = 1.0 / np.sqrt(H)
    x = rng.randn(B, T, I)
    Wih, Whh = rng.uniform(-k, k, (3 * H, I)), rng.uniform(-k, k, (3 * H, H))
    bih, bhh = rng.uniform(-k, k, 3 * H), rng.uniform(-k, k, 3 * H)
    hs, cache = E.gru_dir_fwd(x, Wih, Whh, bih, bhh, reverse)
    return x, Wih, Whh, bih, bhh, hs, cache


@pytest.mark.parametrize("B,T,I,H", [(4, 12, 20, 16), (3, 9, 8, 24), (32, 6, 64, 512), (17, 7, 12, 36), (1, 1, 4, 4)])
@pytest.mark.parametrize("reverse", [False, True])
def test_gru_fwd_bwd(B, T, I, H, reverse):
    from speech_amd import ops
    x, Wih, Whh, bih, bhh, hs_ref, cache = gru_case(B, T, I, H, reverse, B * 100 + T + H)
    ai = ops.gemm(dev(x.reshape(B * T, I)), dev(Wih), trans_b=True, bias=dev(bih)).view(B, T, 3 * H)
    # write into one half of a (B, T, 2H) buffer to exercise the strided output
    hbuf = torch.zeros(B, T, 2 * H, device="cuda")
    h_out = hbuf[:, :, H:]
    stash = torch.empty(B, T, 5 * H, device="cuda")
    ops.gru_fwd(ai, dev(Whh), dev(bhh), h_out, stash, reverse)
    close(h_out, hs_ref, rtol=2e-5, atol_scale=2e-6)
    assert float(hbuf[:, :, :H].abs().max()) == 0
    _, _, r_, z_, n_, q_, _ = cache
    close(stash[:, :, :H], r_), close(stash[:, :, H:2 * H], z_), close(stash[:, :, 2 * H:3 * H], n_)
    close(stash[:, :, 3 * H:4 * H], q_)
    rng = np.random.RandomState(5)
    dhs = rng.randn(B, T, H)
    dx_ref, dWih_ref, dWhh_ref, dbih_ref, dbhh_ref = E.gru_dir_bwd(dhs, cache, Wih, Whh)
    dbuf = torch.zeros(B, T, 2 * H, device="cuda")
    dbuf[:, :, H:] = dev(dhs)
    dai = torch.empty(B, T, 3 * H, device="cuda")
    dah = torch.empty(B, T, 3 * H, device="cuda")
    ops.gru_bwd(dbuf[:, :, H:], h_out, stash, dev(Whh), dai, dah, reverse)
    dai2, dah2 = dai.view(B * T, 3 * H), dah.view(B * T, 3 * H)
    tol = dict(rtol=5e-5, atol_scale=2e-5)
    close(ops.gemm(dai2, dev(Wih)), dx_ref.reshape(B * T, I), **tol)
    close(ops.gemm(dai2, dev(x.reshape(B * T, I)), trans_a=True), dWih_ref, **tol)
    close(ops.gemm(dah2, stash.view(B * T, 5 * H)[:, 4 * H:], trans_a=True), dWhh_ref, **tol)
    close(ops.colsum(dai2), dbih_ref, **tol)
    close(ops.colsum(dah2), dbhh_ref, **tol)


def test_gru_long_sequence_stays_accurate():
    from speech_amd import ops
    B, T, I, H = 2, 300, 16, 32
    x, Wih, Whh, bih, bhh, hs_ref, _ = gru_case(B, T, I, H, False, 3)
    ai = ops.gemm(dev(x.reshape(B * T, I)), dev(Wih), trans_b=True, bias=dev(bih)).view(B, T, 3 * H)
    h = torch.empty(B, T, H, device="cuda")
    ops.gru_fwd(ai, dev(Whh), dev(bhh), h, None, False)
    close(h, hs_ref, rtol=5e-5, atol_scale=5e-6)


# ------------------------------------------------------------------------------------------------------ small helpers
def test_colsum_add_rows():
    from speech_amd import ops
    rng = np.random.RandomState(0)
    a, b = rng.randn(1000, 70), rng.randn(1000, 70)
    close(ops.colsum(dev(a)), a.sum(0), rtol=1e-5, atol_scale=1e-5)
    acc = dev(np.ones(70))
    ops.colsum(dev(a), out=acc, accumulate=True)
    close(acc, a.sum(0) + 1.0, rtol=1e-5, atol_scale=1e-5)
    wide = dev(np.concatenate([a, b], axis=1))
    close(ops.add_rows(wide[:, :70], wide[:, 70:]), a + b, rtol=1e-6)


@pytest.mark.parametrize("momentum", [0.0, 0.9])
@pytest.mark.parametrize("gscale", [1.0, 300.0])
def test_clip_sgd_step(momentum, gscale):
    from speech_amd import ops
    rng = np.random.RandomState(1)
    n = 100003
    p0, g0 = rng.randn(n), gscale * rng.randn(n) / np.sqrt(n)
    P, G = {"p": p0}, {"p": g0}
    bufs = {}
    p, g = dev(p0), dev(g0)
    mom = torch.zeros(n, device="cuda")
    for step in range(2):
        P, total = E.clip_and_sgd(P, G, 1e-2, 200.0, momentum, bufs)
        norm = ops.clip_sgd_step(p, g, mom if momentum else None, 1e-2, momentum, 200.0)
        assert abs(float(norm) - total) < 1e-5 * total
        close(p, P["p"], rtol=1e-5, atol_scale=1e-6)
    if gscale > 1:
        assert total > 200.0  # the clip branch was exercised


@pytest.mark.parametrize("n", [1, 3, 4, 5, 255, 1023, 1025, 65537, (1 << 21) + 7])
def test_clip_sgd_step_at_sizes_around_the_vector_and_block_boundaries(n):
    """The fused norm / clip / update on flat buffers whose length is not a multiple of 4 (16-byte loads), of a block, of the
    grid's stride -- and on an unaligned VIEW of a larger buffer (a parameter subset): against the numpy restatement."""
    from speech_amd import ops
    rng = np.random.RandomState(n % 1000)
    p0, g0 = rng.randn(n), 400.0 * rng.randn(n) / np.sqrt(n)
    for off in (0, 1):
        big_p, big_g = torch.zeros(n + 3, device="cuda"), torch.zeros(n + 3, device="cuda")
        p, g = big_p[off:off + n], big_g[off:off + n]
        p.copy_(dev(p0)); g.copy_(dev(g0))
        mom = torch.zeros(n + 3, device="cuda")[off:off + n]
        P, total = E.clip_and_sgd({"p": p0}, {"p": g0}, 1e-2, 200.0, 0.9, {})
        norm = ops.clip_sgd_step(p, g, mom, 1e-2, 0.9, 200.0)
        assert abs(float(norm) - total) < 1e-5 * total
        close(p, P["p"], rtol=1e-5, atol_scale=1e-6)
        assert float(big_p[off + n:].abs().max()) == 0.0 and (off == 0 or float(big_p[0]) == 0.0)   # nothing outside the view


# ------------------------------------------------------------------------------------------- the GRU stack (wavefront)
@pytest.mark.parametrize("L,D,B,T,I0,H,chunk", [(4, 1, 5, 23, 12, 16, 5), (3, 1, 32, 9, 40, 64, 4), (2, 2, 3, 11, 10, 24, 0),
                                              (1, 1, 2, 7, 6, 8, 32), (4, 1, 2, 40, 8, 16, 0),
                                              # bidirectional stacks narrower than one 16-unit group / wider than an XCD's
                                              # 32 groups: the host's XCD arithmetic divided by zero (SIGFPE) before r6
                                              (2, 2, 2, 5, 6, 8, 0), (1, 2, 1, 3, 4, 4, 0), (1, 2, 2, 4, 8, 528, 0)])
def test_gru_stack_matches_oracle(L, D, B, T, I0, H, chunk):
    """sa_gru_stack_fwd/bwd (chunked layer wavefront for D=1, paired directions for D=2) vs the per-layer oracle."""
    from speech_amd import ops
    rng = np.random.RandomState(L * 100 + T)
    k = 1.0 / np.sqrt(H)
    x = rng.randn(B, T, I0)
    P = []
    for l in range(L):
        I = I0 if l == 0 else D * H
        for d in range(D):
            P.append((rng.uniform(-k, k, (3 * H, I)), rng.uniform(-k, k, (3 * H, H)), rng.uniform(-k, k, 3 * H),
                      rng.uniform(-k, k, 3 * H)))
    # oracle forward
    inp, caches = x, []
    for l in range(L):
        outs, cs = [], []
        for d in range(D):
            Wih, Whh, bih, bhh = P[l * D + d]
            hs, c = E.gru_dir_fwd(inp, Wih, Whh, bih, bhh, d == 1)
            outs.append(hs), cs.append(c)
        caches.append(cs)
        inp = np.concatenate(outs, axis=2)
    top_ref = inp
    x_tm = dev(x.transpose(1, 0, 2))
    w_ih, w_hh = [dev(p[0]) for p in P], [dev(p[1]) for p in P]
    b_ih, b_hh = [dev(p[2]) for p in P], [dev(p[3]) for p in P]
    h_out, stash = ops.gru_stack_fwd(x_tm, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=True, chunk=chunk)
    close(h_out[-1].transpose(0, 1), top_ref, rtol=5e-5, atol_scale=5e-6)
    h_inf, none = ops.gru_stack_fwd(x_tm, w_ih, b_ih, w_hh, b_hh, L, D, H, want_stash=False, chunk=chunk)
    assert none is None and torch.equal(h_inf[-1], h_out[-1])
    # oracle backward
    dtop = rng.randn(B, T, D * H)
    dout, want = dtop, {}
    for l in range(L - 1, -1, -1):
        dins = []
        for d in range(D):
            Wih, Whh, _, _ = P[l * D + d]
            dx, dWih, dWhh, dbih, dbhh = E.gru_dir_bwd(dout[:, :, d * H:(d + 1) * H], caches[l][d], Wih, Whh)
            want[(l, d)] = (dWih, dWhh, dbih, dbhh)
            dins.append(dx)
        dout = dins[0] + dins[1] if D == 2 else dins[0]
    dai, dah, dx = ops.gru_stack_bwd(dev(dtop.transpose(1, 0, 2)), stash, w_ih, w_hh, L, D, H, I0, chunk=chunk)
    tol = dict(rtol=1e-4, atol_scale=5e-5)
    close(dx.transpose(0, 1), dout, **tol)
    for l in range(L):
        lay_in = x_tm if l == 0 else h_out[l - 1]
        for d in range(D):
            kk = l * D + d
            a2, h2 = dai[kk].view(T * B, 3 * H), dah[kk].view(T * B, 3 * H)
            close(ops.gemm(a2, lay_in.view(T * B, -1), trans_a=True), want[(l, d)][0], **tol)
            close(ops.gemm(h2, stash[kk].view(T * B, 5 * H)[:, 4 * H:], trans_a=True), want[(l, d)][1], **tol)
            close(ops.colsum(a2), want[(l, d)][2], **tol)
            close(ops.colsum(h2), want[(l, d)][3], **tol)


# ------------------------------------------------------------------- documented experimental paths stay bit-identical
def _stack_case(L, B, T, I0, H):
    torch.manual_seed(0)
    x = torch.randn(T, B, I0, device="cuda")
    k = 1.0 / H ** 0.5
    w_ih = [torch.empty(3 * H, I0 if l == 0 else H, device="cuda").uniform_(-k, k) for l in range(L)]
    w_hh = [torch.empty(3 * H, H, device="cuda").uniform_(-k, k) for l in range(L)]
    b_ih = [torch.empty(3 * H, device="cuda").uniform_(-k, k) for l in range(L)]
    b_hh = [torch.empty(3 * H, device="cuda").uniform_(-k, k) for l in range(L)]
    return x, w_ih, b_ih, w_hh, b_hh


def test_multi_stream_chains_are_bit_identical():
    """ops.N_CHAINS > 1 (batch rows split over HIP streams; measured slower) must not change results."""
    from speech_amd import ops
    L, B, T, I0, H = 3, 32, 40, 24, 64
    x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)
    dtop = torch.randn(T, B, H, device="cuda")
    outs = []
    try:
        for chains in (1, 2, 4):
            ops.N_CHAINS = chains
            h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True)
            dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, 1, H, I0)
            torch.cuda.synchronize()
            outs.append((h[-1].clone(), dx.clone(), dai[0].clone()))
    finally:
        ops.N_CHAINS = 1
    for o in outs[1:]:
        assert all(torch.equal(a, b) for a, b in zip(outs[0], o))


def test_xcd_local_persistent_kernels_are_bit_identical_and_healthy():
    """The default recurrence path at 512-wide unidirectional stacks (XCD-local persistent chunk kernels, forward and
    backward) against the one-launch-per-step kernels (SA_GRU_PERSIST=0): every output bit-identical,
    ragged last chunk included, and the kernels' error word clean.  (Same chunk length on both sides: the library's
    default differs per path, and the chunk sets the shape -- hence the split-K order -- of the projection GEMMs.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from tests.test_gpu_blocks import _stack_case\nfrom speech_amd import ops, _lib\n"
            "L, B, T, I0, H = 4, 32, 70, 48, 512\n"
            "x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)\n"
            "dtop = torch.randn(T, B, H, device='cuda')\n"
            "for _ in range(2):\n"
            "    h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True, chunk=16)\n"
            "    dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, 1, H, I0, chunk=16)\n"
            "torch.cuda.synchronize()\n"
            "assert _lib.lib().sa_gru_persist_status() == 0\n"
            "torch.save([t.cpu() for t in h + st + dai + dah + [dx]], sys.argv[1])\n") % (root, root)
    res = []
    for mode in ("0", "2", "3"):   # step kernels; XCD-local groups with arrival counters; flag-less hand-off (default)
        out = "/tmp/sa_xcd_%s.pt" % mode
        env = dict(os.environ, SA_GRU_PERSIST=mode, SA_GRU_FUSED="0", SA_GRU_TILED="0")  # the fused / tiled kernels sum in another order
        subprocess.run([sys.executable, "-c", code, out], env=env, check=True, timeout=180)
        res.append(torch.load(out))
    for other in res[1:]:
        assert len(res[0]) == len(other) and all(torch.equal(a, b) for a, b in zip(res[0], other))
    # narrower layers share an XCD (H = 256: two groups per XCD, H = 128: four), fewer groups than slots idle
    # ... and batches wider than the groups can host are walked in passes of batch tiles (4 x 512 at B = 48: 3 tiles,
    # 2 per pass; 4 x 256 at B = 80: 5 tiles, 4 per pass)
    # ... and widths whose unit tiles do not divide an XCD's 32 CUs (H = 384: 24 tiles, one group per XCD, 8 CUs idle;
    # H = 320: 20 tiles; H = 192: 12 tiles, two groups per XCD; H = 448: 28)
    for shape in ("2, 20, 33, 48, 512", "2, 32, 45, 40, 256", "3, 16, 37, 24, 128", "4, 64, 21, 24, 256",
                  "4, 48, 37, 40, 512", "4, 80, 19, 24, 256", "2, 32, 29, 40, 384", "3, 20, 21, 24, 320",
                  "3, 40, 17, 24, 192", "2, 16, 15, 16, 448"):
        code2 = code.replace("L, B, T, I0, H = 4, 32, 70, 48, 512", "L, B, T, I0, H = " + shape)
        res = []
        for mode in ("0", "2", "3"):
            out = "/tmp/sa_xcd2_%s.pt" % mode
            subprocess.run([sys.executable, "-c", code2, out],
                           env=dict(os.environ, SA_GRU_PERSIST=mode, SA_GRU_FUSED="0", SA_GRU_TILED="0"), check=True,
                           timeout=180)
            res.append(torch.load(out))
        for other in res[1:]:
            assert all(torch.equal(a, b) for a, b in zip(res[0], other)), shape


def test_fused_backward_input_gradient():
    """gru_bwd_fused_kernel (the default backward of eligible unidirectional stacks with H = 512 / 256: the lower layers'
    d h_out formed inside the recurrence kernel instead of by a grouped GEMM per wave of the layer wavefront) against
    the GEMM path (SA_GRU_FUSE_DX=0) and the round-1 kernels (SA_GRU_TILED=0): every gradient equal up to the summation order (the k-space is dealt out by gate
    and the product runs on 16x16x4 MFMAs), ragged chunks, several batch tiles and passes, one-step chunks.  (The
    oracle comparisons of the whole stack -- test_gru_stack_*, tests/test_gpu_baseline_configs.py -- run the default,
    i.e. this kernel, wherever it is eligible.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from tests.test_gpu_blocks import _stack_case\nfrom speech_amd import ops, _lib\n"
            "L, B, T, I0, H, CH = [int(v) for v in sys.argv[2:8]]\n"
            "x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)\n"
            "dtop = torch.randn(T, B, H, device='cuda')\n"
            "for _ in range(2):\n"
            "    h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True)\n"
            "    dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, 1, H, I0, chunk=CH)\n"
            "torch.cuda.synchronize()\nassert _lib.lib().sa_gru_persist_status() == 0\n"
            "torch.save([t.cpu() for t in dai + dah + [dx]], sys.argv[1])\n") % (root, root)
    for shape in ((4, 32, 70, 48, 512, 16), (4, 32, 90, 48, 512, 0), (2, 20, 33, 48, 512, 16), (2, 32, 45, 40, 256, 16),
                  (4, 64, 21, 24, 256, 16), (4, 48, 37, 40, 512, 16), (4, 80, 19, 24, 256, 16), (3, 5, 9, 16, 512, 1),
                  (2, 32, 7, 16, 256, 64), (3, 32, 41, 40, 384, 0), (2, 20, 33, 24, 320, 16), (3, 40, 25, 24, 192, 0),
                  (2, 16, 19, 16, 448, 0)):
        res = []
        # the round-1 kernels (row-major exchange); the tiled kernel with the GEMM; the tiled kernel with the product fused
        # ... launched chunk by chunk; the same as ONE launch for the whole recurrence (the default)
        for env in ({"SA_GRU_TILED": "0"}, {"SA_GRU_FUSE_DX": "0"}, {"SA_GRU_BWD_ONE": "0"}, {}):
            out = "/tmp/sa_fuse_dx_%d.pt" % len(res)
            subprocess.run([sys.executable, "-c", code, out] + [str(v) for v in shape],
                           env=dict(os.environ, **env), check=True, timeout=180)
            res.append(torch.load(out))
        for other in res[1:]:
            assert len(res[0]) == len(other)
            for a, b in zip(res[0], other):
                assert torch.isfinite(b).all(), shape
                assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(a.abs().max())), shape


def test_xcd_local_persistent_kernels_bidirectional():
    """Bidirectional layers (the shipped TIMIT / WSJ configs: 4 x biGRU-256) run one persistent launch per layer with
    the two directions as the two sync groups; bit-identical to the step kernels, forward and backward."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\nfrom speech_amd import ops, _lib\n"
            "L, B, T, I0, H = [int(v) for v in sys.argv[2:7]]\n"
            "torch.manual_seed(0)\nk = 1.0 / H ** 0.5\n"
            "x = torch.randn(T, B, I0, device='cuda')\n"
            "mk = lambda *s: torch.empty(*s, device='cuda').uniform_(-k, k)\n"
            "w_ih = [mk(3 * H, I0 if l == 0 else 2 * H) for l in range(L) for d in range(2)]\n"
            "w_hh = [mk(3 * H, H) for l in range(L) for d in range(2)]\n"
            "b_ih = [mk(3 * H) for l in range(L) for d in range(2)]\n"
            "b_hh = [mk(3 * H) for l in range(L) for d in range(2)]\n"
            "dtop = torch.randn(T, B, 2 * H, device='cuda')\n"
            "for _ in range(2):\n"
            "    h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 2, H, want_stash=True)\n"
            "    dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, 2, H, I0)\n"
            "torch.cuda.synchronize()\n"
            "assert _lib.lib().sa_gru_persist_status() == 0\n"
            "torch.save([t.cpu() for t in h + st + dai + dah + [dx]], sys.argv[1])\n") % (root,)
    for shape in ((2, 8, 40, 24, 256), (2, 20, 31, 24, 512), (3, 48, 17, 16, 128), (1, 1, 5, 8, 256), (2, 16, 23, 24, 384),
                  (2, 20, 13, 16, 320)):
        res = []
        for mode in ("0", "2", "3"):
            out = "/tmp/sa_xcd_bi_%s.pt" % mode
            # (the planes kernel has its own budget test below; SA_GRU_FWD_CHUNKS=1: the layer's projection as ONE product in
            # every mode -- since r6 a small product may be split along K by a rule of its own shape, so a projection cut into
            # time chunks sums in another order than the whole one)
            subprocess.run([sys.executable, "-c", code, out] + [str(v) for v in shape],
                           env=dict(os.environ, SA_GRU_PERSIST=mode, SA_GRU_TILED="0", SA_GRU_FWD_PLANES="0", SA_GRU_FWD_CHUNKS="1"),
                           check=True, timeout=180)
            res.append(torch.load(out))
        for other in res[1:]:
            assert len(res[0]) == len(other) and all(torch.equal(a, b) for a, b in zip(res[0], other)), shape
        # the default backward of H = 512 / 256 layers (tiled exchange, the k-space dealt out by gate: another summation
        # order) against the same
        out = "/tmp/sa_xcd_bi_tiled.pt"
        subprocess.run([sys.executable, "-c", code, out] + [str(v) for v in shape], env=dict(os.environ), check=True,
                       timeout=180)
        for a, b in zip(res[0], torch.load(out)):
            assert torch.isfinite(b).all(), shape
            assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(a.abs().max())), shape


def test_bidirectional_forward_projection_overlap():
    """Bidirectional forward with a layer's input projections cut into time chunks that run XCD-filtered on the side stream
    beside the chunked recurrence launches (the default from T >= 32; SA_GRU_FWD_CHUNKS=1 is the one-launch path): same
    hidden states and stash up to the summation order of the projection GEMMs, healthy, and the backward pass accepts
    the stash."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r)\nfrom speech_amd import ops, _lib\n"
            "L, B, T, I0, H = [int(v) for v in sys.argv[2:7]]\n"
            "torch.manual_seed(0)\nk = 1.0 / H ** 0.5\n"
            "x = torch.randn(T, B, I0, device='cuda')\n"
            "mk = lambda *s: torch.empty(*s, device='cuda').uniform_(-k, k)\n"
            "w_ih = [mk(3 * H, I0 if l == 0 else 2 * H) for l in range(L) for d in range(2)]\n"
            "w_hh = [mk(3 * H, H) for l in range(L) for d in range(2)]\n"
            "b_ih = [mk(3 * H) for l in range(L) for d in range(2)]\n"
            "b_hh = [mk(3 * H) for l in range(L) for d in range(2)]\n"
            "dtop = torch.randn(T, B, 2 * H, device='cuda')\n"
            "for _ in range(2):\n"
            "    h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 2, H, want_stash=True)\n"
            "    dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, 2, H, I0)\n"
            "torch.cuda.synchronize()\n"
            "assert _lib.lib().sa_gru_persist_status() == 0\n"
            "torch.save([t.cpu() for t in h + st + [dx]], sys.argv[1])\n") % (root,)
    for shape in ((2, 32, 96, 64, 256), (3, 20, 131, 40, 128)):
        res = []
        for chunks in ("1", "4", "7"):
            out = "/tmp/sa_bi_fwd_chunks_%s.pt" % chunks
            subprocess.run([sys.executable, "-c", code, out] + [str(v) for v in shape],
                           env=dict(os.environ, SA_GRU_FWD_CHUNKS=chunks), check=True, timeout=180)
            res.append(torch.load(out))
        for other in res[1:]:
            assert len(res[0]) == len(other)
            for a, b in zip(res[0], other):
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(a.abs().max())), shape


def test_fused_forward_is_bit_identical_for_every_progress_report_period(monkeypatch):
    """Round 5: a layer of the one-launch forward reports its progress to the layer above every `gru.fwd_report` steps (the
    per-step memory-side fetch_add was what coupled the layers: 2.69 -> 2.22 ms per S-LIBRI stack forward at 4).  The period
    only changes WHEN the layer above may read a row, never what it reads: every period gives the same bits -- also with
    T not a multiple of the period, T smaller than it, ragged batch tiles and inter-layer dropout."""
    from speech_amd import ops, _lib
    for (L, B, T, I0, H), drop in (((4, 32, 61, 40, 512), None), ((3, 20, 7, 24, 256), None), ((4, 32, 45, 40, 512), (0.3, 11, 64)),
                                   ((2, 48, 19, 24, 128), None)):
        x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)
        ref = None
        for period in (4, 1, 2, 8, 16):
            monkeypatch.setenv("SA_GRU_FWD_REPORT", str(period))
            out = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True, drop=drop)
            torch.cuda.synchronize()
            assert _lib.lib().sa_gru_persist_status() == 0
            got = [t.clone() for t in out[0] + out[1] + (out[2] if drop else [])]
            if ref is None:
                ref = got
            else:
                for a, b in zip(ref, got):
                    assert torch.equal(a, b), (L, B, T, H, period)
        monkeypatch.delenv("SA_GRU_FWD_REPORT")


def test_fused_forward_wavefront_matches_oracle_and_default():
    """gru_fwd_fused_kernel (the default forward of eligible unidirectional stacks: one launch, in-kernel input
    projections, weights resident in registers) against the NumPy oracle and against the chunked path; its stash feeds
    the backward pass.  The projection is summed in another order than the chunk GEMM: rtol 1e-4 on h."""
    import os
    import subprocess
    import sys
    from oracle import encoder_np as E
    from speech_amd import ops, _lib
    L, B, T, I0, H = 3, 20, 25, 24, 128
    x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)
    h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True)
    torch.cuda.synchronize()
    assert _lib.lib().sa_gru_persist_status() == 0
    inp = x.cpu().numpy().transpose(1, 0, 2).astype(np.float64)
    for l in range(L):
        inp, _ = E.gru_dir_fwd(inp, w_ih[l].cpu().numpy().astype(np.float64), w_hh[l].cpu().numpy().astype(np.float64),
                               b_ih[l].cpu().numpy().astype(np.float64), b_hh[l].cpu().numpy().astype(np.float64), False)
        np.testing.assert_allclose(h[l].cpu().numpy().transpose(1, 0, 2), inp, rtol=1e-4, atol=1e-5)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, torch; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from tests.test_gpu_blocks import _stack_case\nfrom speech_amd import ops, _lib\n"
            "L, B, T, I0, H = 4, 32, 90, 48, 512\n"
            "x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)\n"
            "dtop = torch.randn(T, B, H, device='cuda')\n"
            "h, st = ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True)\n"
            "dai, dah, dx = ops.gru_stack_bwd(dtop, st, w_ih, w_hh, L, 1, H, I0)\n"
            "torch.cuda.synchronize()\nassert _lib.lib().sa_gru_persist_status() == 0\n"
            "torch.save([t.cpu() for t in h + [dx]], sys.argv[1])\n") % (root, root)
    # B = 64: two passes of two batch tiles each; H = 384 / 320 / 192: widths whose unit tiles leave CUs of an XCD idle
    for shape in ("4, 32, 90, 48, 512", "4, 64, 40, 48, 512", "3, 32, 33, 40, 384", "2, 20, 27, 24, 320", "4, 16, 21, 24, 192"):
        code2 = code.replace("L, B, T, I0, H = 4, 32, 90, 48, 512", "L, B, T, I0, H = " + shape)
        res = []
        for fused in ("0", "1"):
            out = "/tmp/sa_fused_%s.pt" % fused
            subprocess.run([sys.executable, "-c", code2, out], env=dict(os.environ, SA_GRU_FUSED=fused), check=True,
                           timeout=180)
            res.append(torch.load(out))
        for a, b in zip(*res):
            assert float((a - b).abs().max()) < 2e-4 * max(1.0, float(a.abs().max())), shape


@pytest.mark.parametrize("L,B,T,I0,H,drop", [(4, 32, 120, 64, 512, None), (3, 20, 33, 24, 256, None), (2, 48, 19, 24, 128, None),
                                             (2, 32, 21, 40, 384, None), (4, 64, 30, 48, 512, None), (1, 7, 50, 16, 512, None),
                                             (4, 32, 45, 40, 512, (0.3, 11, 64)), (3, 20, 1, 24, 256, None)])
def test_planes_forward_error_budget(monkeypatch, L, B, T, I0, H, drop):
    """Round 6: the one-launch forward of eligible unidirectional stacks runs both products of a step on the bf16 MFMA -- h
    published by its PRODUCER as three exact bf16 planes, six piece products per fp32 product as in the split-bf16 GEMMs --
    with v_exp / v_rcp gates (gru_fwd_planes_kernel).  It is no longer the step kernels' bits.  The bar instead (VERDICT
    r05 item 1, as test_split_bf16_gemm_error_budget): against an fp64 restatement of the stack the planes kernel's error
    on every layer's output and on every stashed gate stays within 4 x the error of the f32-input-MFMA kernel (option
    gru.fwd_planes = 0, whose recurrence is the step kernels' bits: asserted below at L = 1) or 5e-7, whichever is larger -- |h| < 1, so this
    is an absolute bound of a few fp32 ulps -- and the two kernels agree to 1e-6.  Ragged batch tiles (rows beyond the
    batch publish zeros), two passes of batch tiles (B = 64 at L = 4), every eligible width, T = 1, inter-layer dropout
    (same masks: the dropped copies agree too)."""
    from speech_amd import ops, _lib
    x, w_ih, b_ih, w_hh, b_hh = _stack_case(L, B, T, I0, H)
    run = lambda: ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 1, H, want_stash=True, drop=drop)
    got = run()
    monkeypatch.setenv("SA_GRU_FWD_PLANES", "0")
    f32k = run()
    monkeypatch.setenv("SA_GRU_FUSED", "0")
    monkeypatch.setenv("SA_GRU_PERSIST", "0")
    step = run()
    monkeypatch.delenv("SA_GRU_FWD_PLANES"), monkeypatch.delenv("SA_GRU_FUSED"), monkeypatch.delenv("SA_GRU_PERSIST")
    torch.cuda.synchronize()
    assert _lib.lib().sa_gru_persist_status() == 0
    for a, b in zip(f32k[0] + f32k[1], step[0] + step[1]):
        if L == 1:                                   # the f32-input one-launch kernel's recurrence: the step kernels' bits
            assert torch.equal(a, b)                 # (upper layers: in-kernel projection vs a GEMM, another summation order)
        else:
            assert float((a - b).abs().max()) <= 2e-5
    # fp64 restatement (the dropped copies of the planes run feed the next layer: same masks everywhere)
    inp = x.double()
    for l in range(L):
        Wi, Wh, bi, bh = w_ih[l].double(), w_hh[l].double(), b_ih[l].double(), b_hh[l].double()
        ai = inp @ Wi.t() + bi
        h = torch.zeros(B, H, dtype=torch.float64, device="cuda")
        hs, gates = [], []
        for t in range(T):
            ah = h @ Wh.t() + bh
            r = torch.sigmoid(ai[t, :, :H] + ah[:, :H])
            z = torch.sigmoid(ai[t, :, H:2 * H] + ah[:, H:2 * H])
            n = torch.tanh(ai[t, :, 2 * H:] + r * ah[:, 2 * H:])
            gates.append(torch.cat([r, z, n, ah[:, 2 * H:], h], dim=1))
            h = (1 - z) * n + z * h
            hs.append(h)
        h64, st64 = torch.stack(hs), torch.stack(gates)
        for name, ref, a, b in (("h", h64, got[0][l], f32k[0][l]), ("stash", st64, got[1][l], f32k[1][l])):
            assert torch.isfinite(a).all()
            e_planes, e_f32 = float((a.double() - ref).abs().max()), float((b.double() - ref).abs().max())
            assert e_planes <= max(4.0 * e_f32, 5e-7), (l, name, e_planes, e_f32)
            assert float((a - b).abs().max()) <= 1e-6, (l, name)
        inp = h64
        if drop and l + 1 < L:   # the layer above reads the dropped copy: mask = dropped / plain of the f32 kernel's run
            assert float((got[2][l] - f32k[2][l]).abs().max()) <= 2e-6
            mask = torch.where(f32k[0][l] != 0, f32k[2][l] / f32k[0][l], torch.zeros_like(f32k[0][l])).double()
            inp = h64 * torch.round(mask * (1.0 - drop[0])) / (1.0 - drop[0])


@pytest.mark.parametrize("L,B,T,I0,H", [(2, 8, 40, 24, 256), (2, 20, 31, 24, 512), (3, 48, 17, 16, 128), (1, 1, 5, 8, 256),
                                        (2, 16, 23, 24, 384), (2, 32, 96, 64, 256)])
def test_planes_bidirectional_forward_error_budget(monkeypatch, L, B, T, I0, H):
    """The two directions of a bidirectional layer (every shipped config: 4 x biGRU-256) on gru_fwd_chunk_planes_kernel: the
    same budget as test_planes_forward_error_budget against an fp64 restatement of the bidirectional stack (layer l + 1 reads
    the concatenated directions, model.py:35-39), incl. the chunked projection overlap (T = 96) and a single row."""
    from speech_amd import ops, _lib
    torch.manual_seed(1)
    k = 1.0 / H ** 0.5
    mk = lambda *sh: torch.empty(*sh, device="cuda").uniform_(-k, k)
    x = torch.randn(T, B, I0, device="cuda")
    w_ih = [mk(3 * H, I0 if l == 0 else 2 * H) for l in range(L) for d in range(2)]
    w_hh = [mk(3 * H, H) for l in range(L) for d in range(2)]
    b_ih = [mk(3 * H) for l in range(L) for d in range(2)]
    b_hh = [mk(3 * H) for l in range(L) for d in range(2)]
    run = lambda: ops.gru_stack_fwd(x, w_ih, b_ih, w_hh, b_hh, L, 2, H, want_stash=True)
    got = run()
    monkeypatch.setenv("SA_GRU_FWD_PLANES", "0")
    f32k = run()
    monkeypatch.delenv("SA_GRU_FWD_PLANES")
    torch.cuda.synchronize()
    assert _lib.lib().sa_gru_persist_status() == 0
    inp = x.double()
    for l in range(L):
        outs = []
        for d in range(2):
            kk = 2 * l + d
            Wi, Wh, bi, bh = w_ih[kk].double(), w_hh[kk].double(), b_ih[kk].double(), b_hh[kk].double()
            ai = inp @ Wi.t() + bi
            h = torch.zeros(B, H, dtype=torch.float64, device="cuda")
            hs, gates = [None] * T, [None] * T
            for t in (range(T - 1, -1, -1) if d else range(T)):
                ah = h @ Wh.t() + bh
                r = torch.sigmoid(ai[t, :, :H] + ah[:, :H])
                z = torch.sigmoid(ai[t, :, H:2 * H] + ah[:, H:2 * H])
                n = torch.tanh(ai[t, :, 2 * H:] + r * ah[:, 2 * H:])
                gates[t] = torch.cat([r, z, n, ah[:, 2 * H:], h], dim=1)
                h = (1 - z) * n + z * h
                hs[t] = h
            outs.append(torch.stack(hs))
            st64 = torch.stack(gates)
            e_planes, e_f32 = float((got[1][kk].double() - st64).abs().max()), float((f32k[1][kk].double() - st64).abs().max())
            assert e_planes <= max(4.0 * e_f32, 5e-7), (l, d, "stash", e_planes, e_f32)
        h64 = torch.cat(outs, dim=2)
        assert torch.isfinite(got[0][l]).all()
        e_planes, e_f32 = float((got[0][l].double() - h64).abs().max()), float((f32k[0][l].double() - h64).abs().max())
        assert e_planes <= max(4.0 * e_f32, 5e-7), (l, "h", e_planes, e_f32)
        assert float((got[0][l] - f32k[0][l]).abs().max()) <= 1e-6
        inp = h64


def test_gemm_arithmetic_is_a_function_of_the_shape_not_of_the_workspace(monkeypatch):
    """include/speech_amd.h section 4: a product runs the split-bf16 path iff sa_gemm_is_split_bf16(M, N, K) (>= 8 GFLOP,
    K >= 256, M, N >= 64; option gemm.exact forces either way) -- a split-path product handed less workspace than
    sa_gemm_workspace_bytes returns CTC_STATUS_INVALID_VALUE instead of quietly running the other kernel; an exact-path
    product accepts no workspace at all (K is then simply not split)."""
    from speech_amd import _lib
    L = _lib.lib()
    monkeypatch.delenv("SA_GEMM_EXACT", raising=False)
    assert L.sa_gemm_is_split_bf16(4096, 1536, 800) == 1 and L.sa_gemm_is_split_bf16(512, 512, 512) == 0
    assert L.sa_gemm_is_split_bf16(1 << 20, 32, 4096) == 0 and L.sa_gemm_is_split_bf16(4096, 4096, 128) == 0
    M, N, K = 4096, 1536, 800
    a, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda")
    c = torch.empty(M, N, device="cuda")
    need = L.sa_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")

    def call(nbytes):
        return L.sa_gemm_f32(0, 1, M, N, K, 1.0, a.data_ptr(), K, b.data_ptr(), K, 0.0, c.data_ptr(), N, None,
                             ws.data_ptr() if nbytes else None, nbytes, _lib.cur_stream())
    assert call(need) == 0
    split = c.clone()
    assert call(need // 2) == 2 and call(0) == 2          # CTC_STATUS_INVALID_VALUE: never a silent change of arithmetic
    monkeypatch.setenv("SA_GEMM_EXACT", "1")
    L = _lib.lib()  # (the host applies the changed environment to the library's option table on its next access)
    assert L.sa_gemm_is_split_bf16(M, N, K) == 0
    assert call(0) == 0                                    # the exact path needs no workspace
    ref = a.double() @ b.double().t()
    for got in (split, c):
        assert float((got.double() - ref).norm() / ref.norm()) < 2e-6


@pytest.mark.parametrize("M,N,K", [(15936, 29, 512), (2050, 7, 64), (4096, 32, 1024), (3000, 17, 48)])
def test_thin_products_of_the_classifier(monkeypatch, M, N, K):
    """The fc layer's three products (ctc_model.py:19,29: (B T', H) x (H, |V| + 1), one dimension <= 32) run on the thin
    kernels of csrc/gemm_f32.hip (thin_nt / thin_nn / thin_tn: f32-input MFMA, the (B T', H) matrix streamed once).  Each
    against an fp64 product, within the fp32 accumulation bound, and against the tiled kernels (SA_GEMM_THIN=0); integer data
    bit-exact; beta / bias forms."""
    from speech_amd import ops
    rng = np.random.RandomState(M + N)
    x = rng.randn(M, K).astype(np.float32)          # the big matrix (encoder states)
    w = rng.randn(N, K).astype(np.float32)          # the classifier
    b = rng.randn(N).astype(np.float32)
    dl = rng.randn(M, N).astype(np.float32)         # gradient of the logits
    X, W, Bv, DL = dev(x), dev(w), dev(b), dev(dl)
    xd, wd, dld = x.astype(np.float64), w.astype(np.float64), dl.astype(np.float64)
    cases = {
        "fwd": (lambda: ops.gemm(X, W, trans_b=True, bias=Bv), xd @ wd.T + b, np.abs(xd) @ np.abs(wd.T) + np.abs(b), K),
        "dx": (lambda: ops.gemm(DL, W), dld @ wd, np.abs(dld) @ np.abs(wd), N),
        "dw": (lambda: ops.gemm(DL, X, trans_a=True), dld.T @ xd, np.abs(dld.T) @ np.abs(xd), M),
    }
    for name, (fn, ref, scale, red) in cases.items():
        monkeypatch.delenv("SA_GEMM_THIN", raising=False)
        got = fn().cpu().numpy().astype(np.float64)
        monkeypatch.setenv("SA_GEMM_THIN", "0")
        tiled = fn().cpu().numpy().astype(np.float64)
        bound = 2.0 ** -24 * (2.0 + np.sqrt(red))
        assert got.shape == ref.shape
        assert float((np.abs(got - ref) / scale).max()) <= bound, (name, float((np.abs(got - ref) / scale).max()), bound)
        assert float(np.linalg.norm(got - tiled) / np.linalg.norm(ref)) < 1e-6, name
    monkeypatch.delenv("SA_GEMM_THIN", raising=False)
    # weight and bias gradient in ONE call (sa_gemm_tn_colsum_f32: thin_tn_kernel<CS> forms the column sums of dl with one
    # more MFMA per k step): the product bit for bit the plain call's, the sums within the fp32 bound of the fp64 ones --
    # also through the tiled kernel's column-sum epilogue (SA_GEMM_THIN=0) and into views of a larger buffer
    flat = torch.full((N * K + N + 8,), 7.0, device="cuda")
    dw, db = ops.gemm_tn_colsum(DL, X, out=flat[4:4 + N * K].view(N, K), colsum_out=flat[4 + N * K:4 + N * K + N])
    assert torch.equal(dw, ops.gemm(DL, X, trans_a=True))
    cs_ref = dld.sum(0)
    cs_bound = 2.0 ** -24 * (2.0 + np.sqrt(M)) * np.abs(dld).sum(0)
    assert np.all(np.abs(db.cpu().numpy().astype(np.float64) - cs_ref) <= cs_bound)
    assert float(flat[:4].min()) == 7.0 and float(flat[4 + N * K + N:].min()) == 7.0
    monkeypatch.setenv("SA_GEMM_THIN", "0")
    dw_t, db_t = ops.gemm_tn_colsum(DL, X)
    assert float((dw_t - dw).norm() / dw.norm()) < 1e-6
    assert np.all(np.abs(db_t.cpu().numpy().astype(np.float64) - cs_ref) <= cs_bound)
    monkeypatch.delenv("SA_GEMM_THIN", raising=False)
    # exact on integers; beta accumulates; the thin dimension's edge columns / rows are left alone
    xi = rng.randint(-4, 5, (M, K)).astype(np.float32)
    wi = rng.randint(-4, 5, (N, K)).astype(np.float32)
    di = rng.randint(-4, 5, (M, N)).astype(np.float32)
    assert np.array_equal(ops.gemm(dev(xi), dev(wi), trans_b=True).cpu().numpy(), xi @ wi.T)
    assert np.array_equal(ops.gemm(dev(di), dev(wi)).cpu().numpy(), di @ wi)
    if M * 16 * 16 < 2 ** 24:   # the integer sums stay below 2^24: exact whatever the order
        assert np.array_equal(ops.gemm(dev(di), dev(xi), trans_a=True).cpu().numpy(), di.T @ xi)
    c0 = rng.randn(M, N).astype(np.float32)
    out = dev(c0.copy())
    ops.gemm(X, W, trans_b=True, bias=Bv, out=out, beta=0.5)
    np.testing.assert_allclose(out.cpu().numpy(), (xd @ wd.T + b + 0.5 * c0), rtol=2e-5, atol=2e-4)
    # out-of-range k / rows are read from CLAMPED rows and masked; a non-finite value in the last row must reach
    # exactly the sums it belongs to (+inf stays +inf, as on the tiled path), never 0 * inf = NaN through the mask
    x_inf, dl_pos = x.copy(), np.abs(dl) + 0.5
    x_inf[M - 1, 3] = np.inf
    dl_pos[M - 1, N - 1] = np.inf
    thin = ops.gemm(dev(dl_pos), dev(x_inf), trans_a=True).cpu().numpy()
    monkeypatch.setenv("SA_GEMM_THIN", "0")
    tiled = ops.gemm(dev(dl_pos), dev(x_inf), trans_a=True).cpu().numpy()
    monkeypatch.delenv("SA_GEMM_THIN", raising=False)
    assert np.array_equal(np.isnan(thin), np.isnan(tiled)) and np.array_equal(np.isposinf(thin), np.isposinf(tiled))
    assert np.isposinf(thin[:N - 1, 3]).all() and np.isfinite(thin[:N - 1, :3]).all()


@pytest.mark.parametrize("B,C,T,F,O,kh,kw,s", [(2, 8, 24, 40, 32, 5, 32, 2), (1, 32, 51, 33, 32, 5, 8, 2),
                                               (2, 32, 37, 41, 32, 4, 12, 2), (8, 32, 150, 65, 32, 5, 32, 2),
                                               (2, 16, 9, 10, 8, 2, 4, 2)])
@pytest.mark.parametrize("layout", ["nchw", "tbf"])
def test_stride2_input_gradient_by_phases_is_the_zero_stuffed_form_bit_for_bit(B, C, T, F, O, kh, kw, s, layout):
    """(r6) conv_dirc_phase_kernel: the input gradient of a stride-2 direct conv as four stride-1 phases in one launch (rows /
    columns of one parity meet the taps of that parity) against the zero-stuffed transposed conv it replaces (option
    conv.dx_phases = 0): the same non-zero products in the same order, so dx is BIT-identical (dw / db do not change path);
    odd and even extents, an even tap-row count, the TIMIT stacked conv's shape; and the oracle on top."""
    from speech_amd import ops, _lib
    rng = np.random.RandomState(T + F + kw)
    x = rng.randn(B, C, T, F)
    w = rng.randn(O, C, kh, kw) / np.sqrt(C * kh * kw)
    b = rng.randn(O) * 0.1
    y_ref, cols = E.conv_relu_fwd(x, w, b, s)
    _, _, To, Fo = y_ref.shape
    y, ys = ops.conv2d_relu_fwd(dev(x), dev(w), dev(b), s, layout)
    dy = rng.randn(*y_ref.shape)
    dy_l = dy if layout == "nchw" else dy.transpose(2, 0, 1, 3).reshape(To, B, O * Fo)
    got = {}
    try:
        for phases in (1, 0):
            _lib.set_option("conv.dx_phases", phases)
            dx, dw, db = ops.conv2d_relu_bwd(dev(x), dev(w), y, dev(dy_l), ys, s, need_dx=True)
            got[phases] = [t.cpu().numpy().copy() for t in (dx, dw, db)]
    finally:
        _lib.set_option("conv.dx_phases", 1)
    for a, o in zip(got[1], got[0]):
        assert np.array_equal(a, o)
    dx_ref, _, _ = E.conv_relu_bwd(dy, y_ref, cols, x.shape, w, s, need_dx=True)
    close(torch.from_numpy(got[1][0]), dx_ref, rtol=2e-5, atol_scale=1e-5)
