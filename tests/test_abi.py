"""The C ABI: the built library loads on a CPU-only host and exports every symbol include/speech_amd.h declares."""
import os
import re

import pytest

from speech_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "speech_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(?:ctcStatus_t|size_t|int|long|void|const char\s*\*)\s+([a-zA-Z_][a-zA-Z0-9_]*)\s*\(", src)
    return sorted(set(names))


def test_header_and_ctypes_table_agree():
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    L = _lib.lib()  # raises if the .so is missing or a symbol is absent
    for name in declared_symbols():
        assert hasattr(L, name), name
    assert L.get_warpctc_version() >= 2
    assert L.ctcGetStatusString(0) == b"no error"


def test_host_side_argument_checks_need_no_gpu():
    L = _lib.lib()
    assert L.sa_ctc_workspace_bytes(1000, 100, 29, 32) > 32 * 1000 * 29 * 4
    assert L.sa_ctc_workspace_bytes(10, 5, 0, 1) == 0
    # null pointers / bad sizes are rejected before anything touches a device
    assert L.sa_ctc_loss(None, None, 1, 1, None, None, None, 29, 32, 10, 5, 28, None, None, 0, None) == 2
    opts = _lib.ctcOptions()
    opts.loc = 0  # CTC_CPU: this library has no CPU path
    import ctypes
    n = ctypes.c_size_t(0)
    arr = (ctypes.c_int * 1)(5)
    assert L.get_workspace_size(ctypes.cast(arr, ctypes.c_void_p), ctypes.cast(arr, ctypes.c_void_p), 29, 1, opts,
                                ctypes.byref(n)) == 3


def test_gru_stack_workspaces_hold_what_the_kernels_lay_out():
    """The backward workspace must have room for the tiled exchange copy of the gate gradients per (layer, direction)
    (T * ceil(B / 16) * 16 * 3H floats: what gru_bwd_fused_kernel's blocks read from each other), the transposed
    weights and the d h_out of the lower layers; sizes grow with every dimension; bad arguments give 0."""
    L = _lib.lib()
    f = L.sa_gru_stack_bwd_workspace_bytes
    Ls, D, B, T, H, I0 = 4, 1, 32, 498, 512, 800
    w = f(Ls, D, B, T, H, I0)
    exchange = Ls * D * T * ((B + 15) // 16) * 16 * 3 * H * 4
    mid = (Ls - 1) * T * B * D * H * 4
    weights_t = Ls * D * 3 * H * H * 4 + (Ls - 1) * 3 * H * H * 4
    assert w >= exchange + mid + weights_t
    assert f(Ls, D, 33, T, H, I0) - w >= Ls * T * 16 * 3 * H * 4      # a ragged batch tile is padded to 16 rows
    assert f(Ls, 2, B, T, H, I0) > w and f(Ls, D, B, 2 * T, H, I0) > w and f(Ls + 1, D, B, T, H, I0) > w
    for bad in ((0, 1, 32, 498, 512, 800), (4, 1, 0, 498, 512, 800), (4, 1, 32, 0, 512, 800), (4, 1, 32, 498, 0, 800)):
        assert f(*bad) == 0
    assert L.sa_gru_stack_fwd_workspace_bytes(Ls, D, B, T, H, I0) >= Ls * D * T * B * 3 * H * 4


def test_shape_gates_of_the_fused_operators_need_no_gpu():
    """The host-side shape gates that choose between the fused and the general operators."""
    L = _lib.lib()
    # one-operator Transducer joint: H % 64 == 0, H <= 512, K <= 32
    assert L.sa_joint_fused_workspace_bytes(32, 498, 101, 512, 29) > 0
    assert L.sa_joint_fused_workspace_bytes(2, 9, 17, 64, 7) > 0
    for bad in ((2, 40, 18, 100, 29), (2, 40, 18, 1024, 29), (2, 40, 18, 128, 40), (0, 40, 18, 128, 29),
                (2, 0, 18, 128, 29), (2, 40, 0, 128, 29)):
        assert L.sa_joint_fused_workspace_bytes(*bad) == 0
    # the partial-sum workspace grows with every dimension it is indexed by
    w0 = L.sa_joint_fused_workspace_bytes(4, 100, 20, 256, 29)
    assert L.sa_joint_fused_workspace_bytes(8, 100, 20, 256, 29) > w0
    assert L.sa_joint_fused_workspace_bytes(4, 200, 20, 256, 29) > w0
    assert L.sa_joint_fused_workspace_bytes(4, 100, 40, 256, 29) > w0
    # null pointers are rejected before any launch
    assert L.sa_joint_fused_fwd(None, None, None, None, None, 2, 9, 17, 64, 7, None) == 2
    assert L.sa_joint_fused_bwd(None, None, None, None, None, None, None, None, None, 2, 9, 17, 64, 7, None, 0, None) == 2
    # Seq2Seq decoder (loop and single step): embedding dim == rnn dim, dims % 4 == 0, odd location kernel <= 15 taps
    assert L.sa_s2s_decoder_workspace_bytes(16, 197, 1, 256, 256, 15, 31) > 0
    assert L.sa_s2s_decoder_workspace_bytes(16, 197, 99, 256, 256, 15, 31) > L.sa_s2s_decoder_workspace_bytes(
        16, 197, 1, 256, 256, 15, 31)
    assert L.sa_s2s_decoder_workspace_bytes(16, 197, 1, 256, 128, 15, 31) == 0   # E != H
    assert L.sa_s2s_decoder_workspace_bytes(16, 197, 1, 256, 256, 14, 31) == 0   # even kernel
    assert L.sa_s2s_decoder_workspace_bytes(16, 197, 1, 256, 256, 17, 31) == 0   # > 15 taps
    assert L.sa_s2s_decoder_step(None, None, None, None, None, None, 16, 197, 256, 256, 15, 31, 1.0, None, None, None,
                                 None, None, 0, None) == 2


def test_cpu_tensor_fails_loudly():
    import torch
    from speech_amd.ctc import CTCLoss
    with pytest.raises(_lib.SpeechAmdError):
        CTCLoss()(torch.zeros(2, 5, 4, requires_grad=True), torch.IntTensor([1, 2]), torch.IntTensor([5, 5]),
                  torch.IntTensor([1, 1]))


def test_product_path_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under speech_amd/, speech/, functions/, train.py or eval.py may import,
    load or execute it (bench.py may, in its cpu_baseline leg only)."""
    import ast
    offenders = []
    files = [os.path.join(ROOT, f) for f in ("train.py", "eval.py")]
    for pkg in ("speech_amd", "speech", "functions", "transducer"):
        for dp, _, fs in os.walk(os.path.join(ROOT, pkg)):
            files += [os.path.join(dp, f) for f in fs if f.endswith(".py")]
    for f in files:
        src = open(f).read()
        for node in ast.walk(ast.parse(src)):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            if any(n == "oracle" or n.startswith("oracle.") for n in names):
                offenders.append(f)
        if "libctc_ref" in src:
            offenders.append(f)
    for dp, _, fs in os.walk(os.path.join(ROOT, "speech_amd", "csrc")):
        for f in fs:
            if "oracle" in open(os.path.join(dp, f), errors="ignore").read():
                offenders.append(os.path.join(dp, f))
    assert not offenders, offenders


def test_options_are_a_table_not_the_environment(monkeypatch):
    """the library read 28 SA_* environment variables with getenv() on every call.  Now it reads none
    (no getenv in csrc/ at all); the surviving switches are named integer options behind sa_set_option / sa_get_option, and
    the Python host applies SA_<NAME> variables to them (once at load; under tests whenever the environment changed)."""
    import ctypes
    import glob
    for src in glob.glob(os.path.join(ROOT, "speech_amd", "csrc", "*.h*")):
        code = re.sub(r"//.*", "", open(src).read())
        assert "getenv" not in code, src
    L = _lib.lib()
    names = _lib.option_names()
    assert len(names) == L.sa_option_count() >= 10 and len(set(names)) == len(names)
    for i, n in enumerate(names):
        assert re.fullmatch(r"[a-z0-9]+\.[a-z0-9_]+", n) and L.sa_option_help(i)
    assert L.sa_option_name(-1) is None and L.sa_option_name(len(names)) is None
    L.sa_reset_options()
    defaults = {n: _lib.get_option(n) for n in names}
    assert defaults["gru.fused"] == 1 and defaults["ctc.prob"] == -1 and defaults["gru.fwd_report"] == 8
    assert _lib.set_option("gru.fused", 0) == 1 and _lib.get_option("gru.fused") == 0
    v = ctypes.c_long(0)
    assert L.sa_set_option(b"no.such_option", 1) == 2 and L.sa_get_option(b"no.such_option", ctypes.byref(v)) == 2
    assert L.sa_set_option(None, 1) == 2
    with pytest.raises(_lib.SpeechAmdError):
        _lib.set_option("no.such_option", 1)
    L.sa_reset_options()
    assert {n: _lib.get_option(n) for n in names} == defaults
    # the host's bridge: SA_GRU_FUSED=0 -> gru.fused = 0, removed again -> the default is back
    monkeypatch.setenv("SA_GRU_FUSED", "0")
    monkeypatch.setenv("SA_CTC_PROB", "2")
    monkeypatch.setenv("SA_NOT_AN_OPTION", "7")
    _lib.lib()
    assert _lib.get_option("gru.fused") == 0 and _lib.get_option("ctc.prob") == 2
    # a value set through set_option() survives somebody else's environment change (only the options whose
    # variable changed are touched) ...
    _lib.set_option("gru.fwd_report", 16)
    monkeypatch.delenv("SA_GRU_FUSED")
    _lib.lib()
    assert _lib.get_option("gru.fused") == 1 and _lib.get_option("ctc.prob") == 2 and _lib.get_option("gru.fwd_report") == 16
    _lib.set_option("gru.fwd_report", 8)
    monkeypatch.delenv("SA_CTC_PROB")
    _lib.lib()
    assert {n: _lib.get_option(n) for n in names} == defaults
    assert _lib.option_defaults() == defaults and L.sa_option_default(-1) == 0
    # ... and a value that is not an integer is refused, not silently ignored
    monkeypatch.setenv("SA_GRU_PERSIST", "off")
    with pytest.raises(_lib.SpeechAmdError):
        _lib.lib()
    monkeypatch.delenv("SA_GRU_PERSIST")
    _lib.lib()
    assert {n: _lib.get_option(n) for n in names} == defaults


def test_every_library_option_is_documented():
    """The options table (sa_set_option / sa_get_option, csrc/options.hip) replaced the environment switches of rounds 1-4:
    every option the library knows has a help string and a row in DESIGN.md section 7, and a name the library does not
    know is refused."""
    import ctypes
    L = _lib.lib()
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    names = _lib.option_names()
    assert len(names) == L.sa_option_count() >= 18 and len(set(names)) == len(names)
    for i, name in enumerate(names):
        assert "`%s`" % name in design, name
        assert len(L.sa_option_help(i)) > 20, name
    v = ctypes.c_long(0)
    assert L.sa_get_option(b"no.such.option", ctypes.byref(v)) == 2 and L.sa_set_option(b"no.such.option", 1) == 2
