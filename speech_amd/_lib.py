"""ctypes face of libspeech_amd.so (the C ABI declared in include/speech_amd.h).

There is NO fallback: if the HIP library is missing or a call fails, the product path raises."""
import ctypes
import os

# torch bundles its own copy of the HIP runtime (same SONAME as /opt/rocm's).  It must be mapped BEFORE
# libspeech_amd.so so that the loader resolves our libamdhip64.so.7 dependency to that copy: two HIP runtimes in one
# process would not share streams (launches on torch's stream handle fail).
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPEECH_AMD_LIB") or os.path.join(_HERE, "libspeech_amd.so")  # override: experiment builds

STATUS_NAMES = {0: "CTC_STATUS_SUCCESS", 1: "CTC_STATUS_MEMOPS_FAILED", 2: "CTC_STATUS_INVALID_VALUE",
                3: "CTC_STATUS_EXECUTION_FAILED", 4: "CTC_STATUS_UNKNOWN_ERROR"}

c_int, c_long, c_size_t, c_float, c_void_p = ctypes.c_int, ctypes.c_long, ctypes.c_size_t, ctypes.c_float, ctypes.c_void_p
c_uint, c_ulonglong = ctypes.c_uint, ctypes.c_ulonglong


class ctcOptions(ctypes.Structure):
    class _U(ctypes.Union):
        _fields_ = [("num_threads", ctypes.c_uint), ("stream", c_void_p)]

    _anonymous_ = ("u",)
    _fields_ = [("loc", c_int), ("u", _U), ("blank_label", c_int)]


# name -> (restype, argtypes); every symbol include/speech_amd.h declares
SIGNATURES = {
    "get_warpctc_version": (c_int, []),
    "ctcGetStatusString": (ctypes.c_char_p, [c_int]),
    "compute_ctc_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p,
                                 ctcOptions]),
    "get_workspace_size": (c_int, [c_void_p, c_void_p, c_int, c_int, ctcOptions, ctypes.POINTER(c_size_t)]),
    "sa_ctc_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sa_ctc_flags_offset": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sa_ctc_loss": (c_int, [c_void_p, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                            c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_ctc_loss_reduced": (c_int, [c_void_p, c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                    c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_scale_by_device_scalar": (c_int, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "sa_ctc_beam_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sa_ctc_beam_decode": (c_int, [c_void_p, c_long, c_long, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_ctc_greedy_decode": (c_int, [c_void_p, c_long, c_long, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                     c_void_p, c_void_p]),
    "sa_gemm_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sa_gemm_is_split_bf16": (c_int, [c_int, c_int, c_int]),
    "sa_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, c_float, c_void_p, c_long, c_void_p, c_long, c_float,
                            c_void_p, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_conv2d_is_direct": (c_int, [c_int] * 6),
    "sa_conv2d_fwd_workspace_bytes": (c_size_t, [c_int] * 8),
    "sa_conv2d_relu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_int, c_int, c_long, c_long, c_long, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_conv2d_bwd_workspace_bytes": (c_size_t, [c_int] * 8),
    "sa_conv2d_relu_bwd": (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_long] * 3 + [c_void_p, c_void_p, c_size_t,
                                                                                c_void_p]),
    "sa_dropout_mask_f32": (c_int, [c_void_p, c_size_t, c_size_t, c_float, c_ulonglong, c_uint, c_void_p]),
    "sa_dropout_apply_f32": (c_int, [c_void_p, c_void_p, c_size_t, c_size_t, c_float, c_ulonglong, c_uint, c_void_p]),
    "sa_conv2d_relu_dropout_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                           c_int, c_int, c_int, c_long, c_long, c_long, c_void_p, c_void_p, c_size_t,
                                           c_float, c_ulonglong, c_uint, c_void_p]),
    "sa_conv2d_relu_dropout_bwd": (c_int, [c_void_p] * 7 + [c_int] * 8 + [c_long] * 3 + [c_void_p, c_void_p, c_size_t,
                                                                                        c_float, c_void_p]),
    "sa_gru_stack_fwd_dropout": (c_int, [c_void_p, c_int] + [c_void_p] * 7 + [c_int] * 6 +
                                 [c_void_p, c_size_t, c_float, c_ulonglong, c_uint, c_void_p]),
    "sa_gru_stack_bwd_wgrad_dropout": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p] * 7 +
                                       [c_void_p, c_size_t, c_float, c_ulonglong, c_uint, c_void_p]),
    "sa_gru_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_long, c_void_p, c_int, c_int, c_int,
                           c_int, c_void_p]),
    "sa_gru_bwd_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sa_gru_bwd": (c_int, [c_void_p, c_long, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                           c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sa_gru_stack_fwd_workspace_bytes": (c_size_t, [c_int] * 6),
    "sa_gru_stack_fwd": (c_int, [c_void_p, c_int] + [c_void_p] * 6 + [c_int] * 6 + [c_void_p, c_size_t, c_void_p,
                                                                                         c_void_p, c_int]),
    "sa_gru_stack_bwd_workspace_bytes": (c_size_t, [c_int] * 6),
    "sa_gru_stack_bwd": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p, c_size_t, c_void_p, c_void_p, c_int]),
    "sa_gru_stack_bwd_wgrad": (c_int, [c_void_p] * 7 + [c_int] * 7 + [c_void_p] * 6 + [c_void_p, c_size_t, c_void_p]),
    "sa_set_option": (c_int, [ctypes.c_char_p, c_long]),
    "sa_get_option": (c_int, [ctypes.c_char_p, ctypes.POINTER(c_long)]),
    "sa_reset_options": (None, []),
    "sa_option_count": (c_int, []),
    "sa_option_name": (ctypes.c_char_p, [c_int]),
    "sa_option_help": (ctypes.c_char_p, [c_int]),
    "sa_option_default": (c_long, [c_int]),
    "sa_ctc_profile_configure": (c_int, [c_int]),
    "sa_ctc_profile_count": (c_int, []),
    "sa_ctc_profile_read": (c_int, [c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "sa_gru_profile_configure": (None, [c_int]),
    "sa_gru_profile_read": (c_int, [c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float)]),
    "sa_gru_profile_steps_per_launch": (c_int, [c_int]),
    "sa_gru_persist_status": (c_int, []),
    "sa_gru_persist_reset": (c_int, []),
    "sa_gru_health_flag": (c_int, [c_void_p, c_void_p]),
    "sa_gemm_tn_colsum_f32": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_void_p,
                                      c_void_p, c_size_t, c_void_p]),
    "sa_colsum_workspace_bytes": (c_size_t, [c_int, c_int]),
    "sa_colsum_f32": (c_int, [c_void_p, c_long, c_int, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p]),
    "sa_add_rows_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_void_p]),
    "sa_specgram_frames": (c_int, [c_int, c_int, c_int]),
    "sa_specgram_build_dft": (c_int, [c_void_p, c_int, c_void_p]),
    "sa_log_specgram_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sa_log_specgram": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_float, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "sa_transducer_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sa_transducer_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int,
                                   c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_transducer_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sa_transducer_decode_static": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                            c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sa_embedding_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "sa_embedding_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sa_joint_relu_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sa_joint_relu_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                  c_void_p]),
    "sa_log_softmax_fwd": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "sa_log_softmax_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "sa_joint_fused_workspace_bytes": (c_size_t, [c_int] * 5),
    "sa_joint_fused_fwd": (c_int, [c_void_p] * 5 + [c_int] * 5 + [c_void_p]),
    "sa_joint_fused_bwd": (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p, c_size_t, c_void_p]),
    "sa_grucell_gates_fwd": (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p]),
    "sa_grucell_gates_bwd": (c_int, [c_void_p] * 6 + [c_int, c_int, c_void_p]),
    "sa_attention_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "sa_attention_fwd": (c_int, [c_void_p] * 7 + [c_float, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                                  c_size_t, c_void_p]),
    "sa_attention_bwd": (c_int, [c_void_p] * 7 + [c_float] + [c_void_p] * 10 + [c_int, c_int, c_int, c_int, c_void_p,
                                                                                 c_size_t, c_void_p]),
    "sa_s2s_decoder_workspace_bytes": (c_size_t, [c_int] * 7),
    "sa_s2s_decoder_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 7 + [c_float] + [c_void_p] * 7 +
                           [c_void_p, c_size_t, c_void_p]),
    "sa_s2s_decoder_step": (c_int, [c_void_p] * 6 + [c_int] * 6 + [c_float] + [c_void_p] * 5 + [c_size_t, c_void_p]),
    "sa_s2s_decoder_bwd": (c_int, [c_void_p] * 9 + [c_int] * 8 + [c_float, c_void_p, c_void_p, c_void_p, c_size_t,
                                                                  c_void_p]),
    "sa_s2s_beam_workspace_bytes": (c_size_t, [c_int] * 7),
    "sa_s2s_beam_search": (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [c_float] + [c_int] * 5 + [c_void_p] * 5 +
                           [c_size_t, c_void_p]),
    "sa_s2s_greedy_workspace_bytes": (c_size_t, [c_int] * 7),
    "sa_s2s_greedy_decode": (c_int, [c_void_p, c_void_p] + [c_int] * 6 + [c_float] + [c_int] * 3 + [c_void_p, c_void_p,
                                                                                                  c_void_p, c_size_t, c_void_p]),
    "sa_softmax_xent": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "sa_argmax_rows": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p]),
    "sa_sgd_workspace_bytes": (c_size_t, [c_size_t]),
    "sa_clip_sgd_step": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_float, c_float, c_float, c_float, c_void_p,
                                 c_void_p, c_void_p, c_size_t, c_void_p]),
}

_LIB = None


class SpeechAmdError(RuntimeError):
    pass


def lib():
    """Load the shared library once.  Raises if it has not been built -- there is no CPU path."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SpeechAmdError("libspeech_amd.so is missing (%s): build it with `python -m speech_amd.build` "
                                 "(hipcc, gfx950).  There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = L
        options_from_env()
    elif ENV_SYNC:
        options_from_env()
    return _LIB


# ---- run-time options (include/speech_amd.h: sa_set_option) -------------------------------------------------------------
# The C library never reads the environment.  This host does, ONCE, when it loads the library: SA_<NAME> sets the option
# <name> (SA_GRU_FUSED=0 -> "gru.fused" = 0), so `SA_GRU_TIMING=1 python tools/gru_fused_timing.py` keeps working.
# ENV_SYNC = True (tests/conftest.py) re-applies the environment whenever it changed since the last library access, which is
# what lets a test switch a kernel path with monkeypatch.setenv between two calls.
ENV_SYNC = False
_ENV_SEEN = None


def option_names():
    L = _LIB
    return [L.sa_option_name(i).decode() for i in range(L.sa_option_count())]


def _env_name(option):
    return "SA_" + option.replace(".", "_").upper()


def options_from_env():
    """Apply SA_<NAME> environment variables to the option table.  Only options whose variable CHANGED since the last look
    are touched (a variable that went away puts its option back to the default): values set through set_option() survive
    somebody else's monkeypatch.setenv.  A value that is not an integer is refused loudly."""
    global _ENV_SEEN
    seen = {k: v for k, v in os.environ.items() if k.startswith("SA_")}
    if seen == _ENV_SEEN:
        return
    before, _ENV_SEEN = (_ENV_SEEN or {}), seen
    defaults, bad = None, []
    for name in option_names():
        e = _env_name(name)
        v, was = seen.get(e), before.get(e)
        if v == was:
            continue
        if v is None:  # the variable was removed: back to the library's default for this one option
            if defaults is None:
                defaults = option_defaults()
            _LIB.sa_set_option(name.encode(), defaults[name])
            continue
        try:
            _LIB.sa_set_option(name.encode(), int(v))
        except ValueError:
            bad.append("%s=%r: library option %r takes an integer" % (e, v, name))
    if bad:
        raise SpeechAmdError("; ".join(bad))


def option_defaults():
    L = _LIB
    return {L.sa_option_name(i).decode(): int(L.sa_option_default(i)) for i in range(L.sa_option_count())}


def set_option(name, value):
    """Set a library option (see sa_option_help); returns the previous value."""
    L = lib()
    old = c_long(0)
    if L.sa_get_option(name.encode(), ctypes.byref(old)) != 0 or L.sa_set_option(name.encode(), int(value)) != 0:
        raise SpeechAmdError("unknown library option %r (known: %s)" % (name, ", ".join(option_names())))
    return old.value


def get_option(name):
    v = c_long(0)
    if lib().sa_get_option(name.encode(), ctypes.byref(v)) != 0:
        raise SpeechAmdError("unknown library option %r" % name)
    return v.value


def check(status, what):
    if status != 0:
        msg = lib().ctcGetStatusString(status).decode()
        raise SpeechAmdError("%s failed: %s (%s)" % (what, STATUS_NAMES.get(status, status), msg))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def cur_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(t, name):
    if not t.is_cuda:
        raise SpeechAmdError("%s must live on the GPU: speech_amd has no CPU compute path" % name)


class Workspace:
    """A grow-only byte buffer per device, reused across calls on the same stream."""

    def __init__(self):
        self._bufs = {}

    def get(self, nbytes, device, tag="default"):
        import torch
        key = (str(device), tag)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf


WORKSPACE = Workspace()


class PinnedRing:
    """A small ring of pinned host buffers for per-step host -> device traffic (the padded feature batch, the label /
    length integers).  torch copies a PAGEABLE source through a staging buffer and blocks the host until the stream has
    drained -- once per step that stalls the launch queue behind the whole forward pass; from pinned memory the copy is a
    real asynchronous DMA.  A slot is handed out again only after the copy that read it has completed (one event per
    slot), so the host fills batch k+1's slot while batch k's copy and kernels are still in flight."""

    def __init__(self, dtype, depth=4):
        import threading
        self._dtype = dtype
        self._bufs = [None] * depth
        self._events = [None] * depth
        self._next = 0
        self._by_ptr = {}
        self._lock = threading.Lock()  # (a model's collate-ahead thread takes slots while the loop marks copies)

    def get(self, shape):
        """A pinned tensor of `shape` (contents undefined)."""
        with self._lock:  # take the slot and its pending copy's event under the lock ...
            i = self._next
            self._next = (i + 1) % len(self._bufs)
            ev, self._events[i] = self._events[i], None
        if ev is not None:  # ... and wait for the device OUTSIDE it: the other thread's get() / copied() must not queue up
            ev.synchronize()  # behind a copy this thread is waiting for
        with self._lock:
            return self._claim(i, shape)

    def _claim(self, i, shape):
        n = 1
        for d in shape:
            n *= int(d)
        buf = self._bufs[i]
        if buf is None or buf.numel() < n:
            if buf is not None:
                self._by_ptr.pop(buf.data_ptr(), None)
            buf = torch.empty(int(n * 1.25) + 16, dtype=self._dtype).pin_memory()
            self._bufs[i] = buf
            self._by_ptr[buf.data_ptr()] = i
        return buf[:n].view(tuple(int(d) for d in shape))

    def copied(self, host_tensor, stream=None):
        """Call right after enqueueing (on `stream`, default the current one) the H2D copy of a tensor from get()."""
        with self._lock:
            i = self._by_ptr.get(host_tensor.data_ptr())
            if i is not None:
                ev = torch.cuda.Event()
                ev.record(stream if stream is not None else torch.cuda.current_stream())
                self._events[i] = ev


_INT_RING = None
_COPY_STREAMS = {}


def copy_stream(device):
    """The library's H2D stream of `device`: a step's input copy is issued there as soon as the host has padded the
    batch -- normally while the previous step's kernels still run -- and the compute stream waits for it."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    s = _COPY_STREAMS.get(key)
    if s is None:
        s = _COPY_STREAMS[key] = torch.cuda.Stream(device=key)
    return s


def h2d_async(host_tensor, device, ring=None):
    """host_tensor (pinned, from `ring`) -> a device tensor, copied on copy_stream(device); the CURRENT stream is made to
    wait for the copy, and the allocator is told that it uses the result."""
    cur = torch.cuda.current_stream(device)
    cs = copy_stream(device)
    with torch.cuda.stream(cs):
        out = host_tensor.to(device, non_blocking=True)
    if ring is not None:
        ring.copied(host_tensor, cs)
    cur.wait_stream(cs)
    out.record_stream(cur)
    return out


def ints_to_device(arr, device):
    """A small int32 host array -> device, through pinned memory (asynchronous; see PinnedRing)."""
    global _INT_RING
    if _INT_RING is None:
        _INT_RING = PinnedRing(torch.int32, depth=8)
    h = _INT_RING.get(arr.shape)
    h.numpy()[...] = arr
    return h2d_async(h, device, _INT_RING)
