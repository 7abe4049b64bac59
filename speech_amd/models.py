"""speech_amd.models -- the reference's model API on the HIP hot path.

Mirrors (names, arguments, return types, state-dict keys):
  /root/reference/speech/models/model.py      Model(input_dim, config), conv_out_size, encode, set_eval/set_train,
                                              is_cuda, encoder_dim; LinearND; zero_pad_concat
  /root/reference/speech/models/ctc_model.py  CTC(freq_dim, output_dim, config): forward, forward_impl, loss, collate,
                                              infer, max_decode, blank, fc
  /root/reference/speech/models/transducer_model.py  Transducer(freq_dim, vocab_size, config): forward, forward_impl,
                                              loss, decode, collate, label_collate, infer, blank, embedding, dec_rnn,
                                              fc1, fc2
  /root/reference/speech/models/seq2seq.py    Seq2Seq(freq_dim, vocab_size, config): forward, forward_impl, loss, decode,
                                              decode_step, predict, infer, infer_decode, beam_search, collate,
                                              set_eval / set_train (scheduled sampling), NNAttention, end_pad_concat

Parameters live in torch.nn containers (nn.Conv2d / nn.GRU / nn.Linear are used ONLY as parameter holders: they give
the reference's state-dict names -- conv.0.weight, rnn.weight_ih_l0, fc.fc.weight ... -- and, under the same
torch seed, the reference's initial values).  Their forward() is never called: all compute is the HIP library.
A CPU model raises on forward: there is no CPU path.

Deviations from the reference, all supersets:
  * `.volatile` is a dead torch-0.3 flag; set_eval()/infer() run under torch.no_grad() instead.
  * loss() returns a 1-element tensor, so both `loss.data[0]` (train.py:33) and `loss.item()` work.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib, ctc, decoder, seq2seq as _s2s, transducer as _tr
from .encoder import EncoderFunction, EncoderPlan


class Model(nn.Module):

    def __init__(self, input_dim, config):
        """model.py:12-42.  The modules built here are PARAMETER CONTAINERS (module docstring): what must match the
        reference is their order of construction (the initial values under one torch seed) and their positions inside
        `self.conv` -- [Conv2d, ReLU] per layer, plus a Dropout slot when the config has dropout -- because those
        positions are the state-dict keys (conv.0.weight, conv.2.weight / conv.3.weight, ..., rnn.weight_ih_l0)."""
        super().__init__()
        self.input_dim = input_dim
        self._ctor_args = (input_dim, config)
        enc, p_drop = config["encoder"], config["dropout"]
        slots, channels = [], 1
        for n_filters, k_time, k_freq, stride in enc["conv"]:
            slots.append(nn.Conv2d(channels, n_filters, (k_time, k_freq), stride=(stride, stride), padding=0))
            slots.append(nn.ReLU())
            if p_drop != 0:
                slots.append(nn.Dropout(p=p_drop))
            channels = n_filters
        self.conv = nn.Sequential(*slots)
        features = channels * self.conv_out_size(input_dim, 1)  # what one frame of the conv output hands to the GRU
        assert features > 0, "input_dim %d leaves no frequency bins after the conv stack %s" % (input_dim, enc["conv"])
        rnn = enc["rnn"]
        self.rnn = nn.GRU(input_size=features, hidden_size=rnn["dim"], num_layers=rnn["layers"], batch_first=True,
                          dropout=p_drop, bidirectional=rnn["bidirectional"])
        self._encoder_dim = rnn["dim"]
        self._plan = EncoderPlan(input_dim, config)
        self.volatile = False
        # data-parallel training (speech_amd.dist) describes the GLOBAL batch here so that a rank's shard is padded
        # and normalised exactly as the single-process run of the whole batch would be: see set_global_batch()
        self.loss_denominator = None
        self.pad_frames = None
        self.pad_labels = None
        self._stage = None  # pinned staging ring of collate(), created once the model sits on the GPU

    # ---- reference API -------------------------------------------------------------------------------------------
    def conv_out_size(self, n, dim):
        """Length left of `n` input positions along axis `dim` (0 = time, 1 = frequency) after the un-padded conv stack:
        every layer maps n -> ceil((n - kernel + 1) / stride)  (model.py:44-52)."""
        for layer in self.conv:
            if isinstance(layer, nn.Conv2d):
                span = n - layer.kernel_size[dim] + 1
                n = -(-span // layer.stride[dim])  # integer ceiling, also for span <= 0
        return n

    def forward(self, batch):
        raise NotImplementedError

    def loss(self, x, y):
        raise NotImplementedError

    def infer(self, x):
        raise NotImplementedError

    def set_eval(self):
        self.eval()
        self.volatile = True

    def set_train(self):
        self.train()
        self.volatile = False

    def set_global_batch(self, size=None, max_frames=None, max_label_len=None):
        """Data parallelism: this rank's batches are shards of a global batch of `size` utterances
        whose longest input has `max_frames` frames and whose longest label sequence has `max_label_len` entries.
        The reference scores the PADDED frames (every act_len is the padded length, ctc_model.py:43-45) and averages
        over the batch, so a shard must be padded to the global maximum and its loss divided by the global size for
        the summed shard gradients to equal the single-process gradient.  None (the default) = the batch is whole."""
        self.loss_denominator = size
        self.pad_frames = max_frames
        self.pad_labels = max_label_len

    def skipped_step(self):
        """A data-parallel rank whose shard of a global batch is empty calls this INSTEAD of loss(batch): it consumes the
        random draws a forward pass would have made (one dropout key per pass, ops.new_dropout_seed), so that every rank's
        generator -- seeded identically from the config, train.py:137-138 -- stays in lock-step with the others'."""
        from . import ops
        if self.training and self._plan.dropout and self._plan.fixed_seed is None:
            ops.new_dropout_seed()

    def _to_device(self, x):
        """The batch's H2D copy (ctc_model.py:26-27 `x.cuda()`): asynchronous when collate() staged it in pinned memory."""
        if x.is_cuda or not self.is_cuda:
            return x
        if x.is_pinned():  # on the copy stream: it runs while the previous step's kernels still hold the compute stream
            return _lib.h2d_async(x, self._first_param().device, self._stage)
        return x.cuda(non_blocking=True)

    def _collate_staged(self, *batch):
        """collate() for the model's OWN entry points (loss / forward / infer): the padded batch is staged in a small ring
        of pinned buffers, consumed by the H2D copy issued right behind.  The PUBLIC collate() (the reference's API,
        ctc_model.py:42-53) returns memory the caller owns, like the reference's: a caller that collates a whole dev
        set before using any of it must not find the first batches overwritten."""
        self._staged = True
        try:
            return self.collate(*batch)
        finally:
            self._staged = False

    def _pad_inputs(self, inputs):
        if self.is_cuda and getattr(self, "_staged", False) and not torch.is_tensor(inputs[0]):
            # stage_ahead() padded exactly this batch object on its worker thread?  (dist.with_global_shapes stages batch
            # k+1 BEFORE it hands out batch k, so the entry of the batch in use is normally the OLDER of two.)
            ahead = getattr(self, "_ahead", None)
            hit = ahead.pop(id(inputs), None) if ahead else None
            if hit is not None and hit[0] is inputs:
                staged = hit[1].result()
                want_t = max(max(int(i.shape[0]) for i in inputs), int(self.pad_frames or 0))
                if staged is not None and staged.shape[1] == want_t:
                    self.ahead_hits = getattr(self, "ahead_hits", 0) + 1
                    return staged
            if self._stage is None:
                self._stage = _PinnedStage()
            return zero_pad_concat(inputs, self.pad_frames, stage=self._stage)
        return _as_tensor(zero_pad_concat(inputs, self.pad_frames))

    def stage_ahead(self, batch, shape=None):
        """Start padding the NEXT batch into pinned memory on a worker thread (numpy releases the interpreter lock for the
        copy), while the caller enqueues the current step: dist.with_global_shapes(loader, model) calls this for batch k+1 as
        it hands out batch k.  `shape`: the batch's global-shape handle under data parallelism (its frames are what
        set_global_batch() will put into pad_frames before loss(batch)); None = the batch is whole.  loss() / forward() /
        infer() on exactly this batch object then find the padded tensor ready; anything else is padded as usual."""
        inputs = batch[0]
        if not self.is_cuda or len(inputs) == 0 or torch.is_tensor(inputs[0]):
            return
        if self._stage is None:
            self._stage = _PinnedStage()
        if getattr(self, "_ahead_pool", None) is None:
            import concurrent.futures
            self._ahead_pool = concurrent.futures.ThreadPoolExecutor(max_workers=1, thread_name_prefix="speech_amd-pad")
        def work():
            try:  # (the shape handle's frames are what set_global_batch() will put into pad_frames before loss(batch))
                frames = int(shape.result()[1]) if shape is not None and hasattr(shape, "result") else 0
                return zero_pad_concat(inputs, frames, stage=self._stage)
            except Exception:  # never fatal: the batch is then padded on the launch path, as before
                return None
        if getattr(self, "_ahead", None) is None:
            import collections
            self._ahead = collections.OrderedDict()  # id(inputs) -> (inputs, future); the batch in use + the next one
        while len(self._ahead) >= 2:  # a staged batch nobody used: its slot of the ring simply comes round again
            self._ahead.popitem(last=False)
        self._ahead[id(inputs)] = (inputs, self._ahead_pool.submit(work))

    def _healthy(self, fn):
        """Forward-only use (infer, dev-set loss): there is no optimiser whose device-side gate would catch a failed
        persistent-kernel hand-off, so ask the library once the results are on the host -- and, if it reports one,
        reset it and recompute on the step kernels (include/speech_amd.h, sa_gru_persist_status)."""
        from . import ops
        out = fn()
        if ops.persist_status():
            ops.persist_reset()
            out = fn()
        return out

    def _apply(self, fn, *args, **kwargs):  # .cuda() / .cpu() / .float(): drop the cached parameter references
        self.__dict__.pop("_p0", None)
        self.__dict__.pop("_enc_ps", None)
        return super()._apply(fn, *args, **kwargs)

    def _first_param(self):
        """The model's first parameter, found once: module.parameters() walks every sub-module on each call, and the launch
        path asked twice per batch (120 us of the 310 us a batch-1 CTC.infer spent on the host in front of its first kernel;
        r6).  nn.Module.cuda() / .cpu() keep the Parameter OBJECTS and swap their data, so the reference stays valid."""
        p = self.__dict__.get("_p0")
        if p is None:
            p = self.__dict__["_p0"] = next(self.parameters())
        return p

    @property
    def is_cuda(self):
        return self._first_param().is_cuda

    @property
    def encoder_dim(self):
        return self._encoder_dim

    # ---- HIP path --------------------------------------------------------------------------------------------------
    def _encoder_params(self):
        ps = self.__dict__.get("_enc_ps")  # (the same Parameter objects for the life of the model: see _first_param)
        if ps is not None:
            return list(ps)
        ps = []
        for c in self.conv.children():
            if type(c) == nn.Conv2d:
                ps += [c.weight, c.bias]
        for l in range(self.rnn.num_layers):
            for sfx in ([""] + (["_reverse"] if self.rnn.bidirectional else [])):
                ps += [getattr(self.rnn, "%s_l%d%s" % (n, l, sfx))
                       for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        self.__dict__["_enc_ps"] = tuple(ps)
        return ps

    def _run(self, x, head_w, head_b):
        """head(encode(x)) through the single fused autograd Function."""
        _lib.require_cuda(x, "input batch (move the model to the GPU: model.cuda())")
        return EncoderFunction.apply(self._plan, self.training, x, *(self._encoder_params() + [head_w, head_b]))

    def encode(self, x):
        """model.py:60-79: (B, T, F) -> (B, T', H).  Runs the encoder with an identity head."""
        H = self._encoder_dim
        eye = torch.eye(H, dtype=torch.float32, device=x.device)
        zero = torch.zeros(H, dtype=torch.float32, device=x.device)
        return self._run(x, eye, zero)

    def flatten_parameters_(self, by_reference=True):
        """Re-home every parameter into ONE flat fp32 buffer and give each a slot in ONE flat gradient buffer: the
        fused clip+SGD step and the single RCCL all-reduce both operate on these.  Call after .cuda().
        Per step: p.grad = None for all p (model.zero_grad(set_to_none=True)); backward then fills every slot.
        This call is the opt-in to FLAT-BUFFER gradient semantics (by_reference=True): the encoder's backward writes each
        gradient into its slot and makes p.grad that slot instead of returning it to autograd (which would clone all
        of them every step).  Accumulation still works -- a backward pass that finds p.grad already in its slot (no
        zero_grad in between) adds to it -- but torch.autograd.grad(loss, params) gets None for those parameters;
        by_reference=False keeps ordinary autograd hand-over (the slots are then just where the kernels write).
        The gradient buffer has ONE extra trailing element, the health flag (ops.stamp_health / ops.clip_sgd_step):
        it travels with the gradients through the all-reduce, so a failed persistent kernel on any rank stops every
        rank's update on the device.  Returns (flat_params [n], flat_grads [n + 1])."""
        ps = [p for p in self.parameters()]
        n = sum(p.numel() for p in ps)
        dev = ps[0].device
        flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        flat_g = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        off = 0
        for p in ps:
            k = p.numel()
            flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = flat_p[off:off + k].view(p.shape)
            p._grad_slot = flat_g[off:off + k].view(p.shape)  # EncoderFunction.backward writes gradients here
            p._grad_by_ref = bool(by_reference)               # ... and every other node that owns parameters (ops.slot_hand_over)
            p.grad = None
            off += k
        self._flat = (flat_p, flat_g)
        self._plan.grad_by_reference = bool(by_reference)
        return self._flat


class LinearND(nn.Module):
    """model.py:115-133: nn.Linear over the last dimension of an N-D input (parameter holder + HIP GEMM)."""

    def __init__(self, *args):
        super().__init__()
        self.fc = nn.Linear(*args)

    def forward(self, x):
        from . import ops
        size = list(x.size())
        out = _LinearFunction.apply(x.contiguous().view(-1, size[-1]), self.fc.weight, self.fc.bias)
        size[-1] = out.size(-1)
        return out.view(size)


class _LinearFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        from . import ops
        _lib.require_cuda(x, "x")
        ctx.save_for_backward(x, w)
        # parameters re-homed by Model.flatten_parameters_() carry a slot of the flat gradient buffer (one use of the
        # layer per step: the slot is written, not accumulated)
        ctx.slots = (getattr(w, "_grad_slot", None), getattr(b, "_grad_slot", None))
        ctx.param_refs = (w, b)
        return ops.gemm(x, w, trans_b=True, bias=b)

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, w = ctx.saved_tensors
        carry = ops.slot_carry(ctx.param_refs, ctx.slots)
        dy = dy.contiguous()
        n_out = dy.shape[1]
        if n_out % 4 and dy.shape[0] >= 4096:
            # A narrow, odd-width gradient (the Transducer's 29-class lattice: 1.6 M rows) defeats the GEMM's 16-byte
            # operand loads on both products: its rows are not 16-byte aligned (dx = dy w) and its columns are the
            # contiguous axis of a transposed operand (dw = dy^T x).  Two data movements fix that: zero-pad the class
            # axis to a multiple of 4 for dx, and hand dw a materialised dy^T whose rows are the long, aligned axis.
            pad = 4 - n_out % 4
            dx = None
            if ctx.needs_input_grad[0]:
                dy_p = torch.nn.functional.pad(dy, (0, pad))
                w_p = torch.nn.functional.pad(w.detach(), (0, 0, 0, pad))
                dx = ops.gemm(dy_p, w_p)
            rows = dy.shape[0]
            rows_p = (rows + 3) // 4 * 4  # dy^T (n_out, rows): pad the row count so that every row of dy^T is aligned
            dy_t = torch.zeros(n_out, rows_p, dtype=dy.dtype, device=dy.device)
            dy_t[:, :rows].copy_(dy.t())
            x_p = x if rows_p == rows else torch.nn.functional.pad(x, (0, 0, 0, rows_p - rows))
            dw = ops.gemm(dy_t, x_p, out=ctx.slots[0])
            return (dx,) + tuple(ops.slot_hand_over(ctx.param_refs, ctx.slots, (dw, ops.colsum(dy, out=ctx.slots[1])), carry))
        dx = ops.gemm(dy, w) if ctx.needs_input_grad[0] else None
        dw, db = ops.gemm_tn_colsum(dy, x, out=ctx.slots[0], colsum_out=ctx.slots[1])
        return (dx,) + tuple(ops.slot_hand_over(ctx.param_refs, ctx.slots, (dw, db), carry))


def zero_pad_concat(inputs, min_t=0, stage=None):
    """model.py:135-141.  `min_t`: pad at least this far (the global-batch maximum of a data-parallel shard).
    Inputs that already live on the GPU (loader.make_loader(..., device_features=True)) are padded there and come back
    as a float32 CUDA tensor; host arrays give the reference's float32 ndarray -- or, with `stage` (a _PinnedStage), a
    float32 tensor in PINNED host memory, so that the model's H2D copy is a real asynchronous DMA (a pageable source
    makes `.cuda(non_blocking=True)` a staged, blocking copy)."""
    max_t = max(max(inp.shape[0] for inp in inputs), int(min_t or 0))
    shape = (len(inputs), max_t, inputs[0].shape[1])
    if torch.is_tensor(inputs[0]):
        input_mat = torch.zeros(shape, dtype=torch.float32, device=inputs[0].device)
        for e, inp in enumerate(inputs):
            input_mat[e, :inp.shape[0], :] = inp
        return input_mat
    if stage is not None:
        out = stage.get(shape)
        input_mat = out.numpy()
        for e, inp in enumerate(inputs):  # only the padding tail is zeroed, not the whole batch twice
            input_mat[e, :inp.shape[0], :] = inp
            input_mat[e, inp.shape[0]:, :] = 0.0
        return out
    input_mat = np.zeros(shape, dtype=np.float32)
    for e, inp in enumerate(inputs):
        input_mat[e, :inp.shape[0], :] = inp
    return input_mat


class _PinnedStage(_lib.PinnedRing):
    """The pinned ring of the padded feature batch (train.py's loop: collate on the host, copy, launch)."""

    def __init__(self, depth=3):
        super().__init__(torch.float32, depth)


def _flat_labels(labels):
    """ctc_model.py:47-48 (`[l for label in labels for l in label]`) without a Python loop per label."""
    n = sum(len(l) for l in labels)
    out = np.empty(n, dtype=np.int32)
    o = 0
    for l in labels:
        k = len(l)
        out[o:o + k] = l
        o += k
    return torch.from_numpy(out)


def _as_tensor(x):
    return x if torch.is_tensor(x) else torch.from_numpy(x)


class CTC(Model):
    def __init__(self, freq_dim, output_dim, config):
        super().__init__(freq_dim, config)
        self._ctor_args = (freq_dim, output_dim, config)
        # include the blank token (ctc_model.py:17-19): blank is the LAST class
        self.blank = output_dim
        self.fc = LinearND(self.encoder_dim, output_dim + 1)

    @property
    def ctc_denominator(self):  # round-1 name of loss_denominator
        return self.loss_denominator

    @ctc_denominator.setter
    def ctc_denominator(self, v):
        self.loss_denominator = v

    def forward(self, batch):
        x, y, x_lens, y_lens = self._collate_staged(*batch)
        return self.forward_impl(x)

    def forward_impl(self, x, softmax=False):
        x = self._to_device(x)
        x = self._run(x, self.fc.fc.weight, self.fc.fc.bias)
        if softmax:
            return torch.nn.functional.softmax(x, dim=2)
        return x

    def loss(self, batch):
        x, y, x_lens, y_lens = self._collate_staged(*batch)
        with torch.set_grad_enabled(not self.volatile):
            out = self.forward_impl(x)
            loss_fn = ctc.CTCLoss(denom=self.loss_denominator)
            return loss_fn(out, y, x_lens, y_lens)

    def collate(self, inputs, labels):
        max_t = max(max(i.shape[0] for i in inputs), int(self.pad_frames or 0))
        max_t = self.conv_out_size(max_t, 0)
        x_lens = torch.IntTensor([max_t] * len(inputs))
        x = self._pad_inputs(inputs)
        y_lens = torch.IntTensor([len(l) for l in labels])
        y = _flat_labels(labels)
        return [x, y, x_lens, y_lens]

    def infer(self, batch):
        """ctc_model.py:55-60: prefix beam search with beam_size=1 over the full padded T' of every utterance --
        on the device, from the logits (the softmax of :30-31 is fused into the decode kernel)."""
        x, y, x_lens, y_lens = self._collate_staged(*batch)

        def run():
            with torch.no_grad():
                logits = self.forward_impl(x)
                return decoder.beam_decode(logits, beam_size=1, blank=self.blank, input_is_logits=True)[0]
        return self._healthy(run)

    @staticmethod
    def max_decode(pred, blank):
        """ctc_model.py:62-70 (host helper on a label path; the device version is decoder.greedy_decode)."""
        pred = list(pred)
        if not pred:
            return []
        prev = pred[0]
        seq = [prev] if prev != blank else []
        for p in pred[1:]:
            if p != blank and p != prev:
                seq.append(p)
            prev = p
        return seq


class Transducer(Model):
    """transducer_model.py:14-113 on the HIP ops.  The prediction network runs on the GRU recurrence kernels, fc1 is one
    sa_gemm_f32 over the stacked encoder / prediction rows, relu(xa + ya) -> fc2 -> log_softmax is ONE fused operator
    (sa_joint_fused_*: the (B, T', U+1, H) joint tensor -- 3.3 GB at B=32, T'=498, U=100, H=512 -- exists only inside
    its MFMA operands; shapes it does not take fall back to the unfused operators), the loss is sa_transducer_loss."""

    def __init__(self, freq_dim, vocab_size, config):
        super().__init__(freq_dim, config)
        self._ctor_args = (freq_dim, vocab_size, config)
        decoder_cfg = config["decoder"]
        rnn_dim = self.encoder_dim
        embed_dim = decoder_cfg["embedding_dim"]
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.dec_rnn = nn.GRU(input_size=embed_dim, hidden_size=rnn_dim, num_layers=decoder_cfg["layers"],
                              batch_first=True, dropout=config["dropout"])
        self._dec_dropout = float(config["dropout"])
        # include the blank token (transducer_model.py:31-32): blank is the LAST class
        self.blank = vocab_size
        self.fc1 = LinearND(rnn_dim, rnn_dim)
        self.fc2 = LinearND(rnn_dim, vocab_size + 1)

    def forward(self, batch):
        x, y, x_lens, y_lens = self._collate_staged(*batch)
        y_mat = self.label_collate(batch[1])
        return self.forward_impl(x, y_mat)

    def forward_impl(self, x, y):
        if self.is_cuda:
            x = self._to_device(x)
            y = y.cuda(non_blocking=True)
        x = self.encode(x)
        return self.decode(x, y)

    def loss(self, batch):
        x, y, x_lens, y_lens = self._collate_staged(*batch)
        y_mat = self.label_collate(batch[1])
        with torch.set_grad_enabled(not self.volatile):
            out = self.forward_impl(x, y_mat)
            loss_fn = _tr.TransducerLoss(denom=self.loss_denominator)
            return loss_fn(out, y, x_lens, y_lens)

    def skipped_step(self):
        super().skipped_step()
        if self.training and self._dec_dropout and self.dec_rnn.num_layers > 1:
            from . import ops
            ops.new_dropout_seed()   # the prediction network's inter-layer masks (decode)

    def _dec_params(self, l):
        return [getattr(self.dec_rnn, "%s_l%d" % (n, l)) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]

    def decode(self, x, y):
        """transducer_model.py:54-78.  x (B, T', H) encoder states, y (B, U) int64 labels -> (B, T', U+1, V+1)
        log-probabilities."""
        _lib.require_cuda(x, "x (move the model to the GPU: model.cuda())")
        H, L = self.encoder_dim, self.dec_rnn.num_layers
        emb = _tr.EmbeddingFunction.apply(y, self.embedding.weight)
        b, u, e = emb.shape
        # prepend zeros (:61-66)
        inp = torch.cat([torch.zeros((b, 1, e), dtype=torch.float32, device=emb.device), emb], dim=1)
        p_drop = self._dec_dropout if self.training else 0.0
        if p_drop and L > 1:  # nn.GRU's inter-layer dropout: one layer per call, the library's mask in between
            from . import ops
            seed = ops.new_dropout_seed()
            for l in range(L):
                inp = _tr.GRUStackFunction.apply(inp, H, *self._dec_params(l))
                if l + 1 < L:
                    inp = _tr.DropoutFunction.apply(inp, p_drop, seed, ops.DROP_STREAM_PRED + l)
            yd = inp
        else:
            yd = _tr.GRUStackFunction.apply(inp, H, *[p for l in range(L) for p in self._dec_params(l)])
        # fc1 on the encoder states and on the prediction states (:72): ONE product over the stacked rows
        B, T = x.shape[0], x.shape[1]
        U1 = yd.shape[1]
        rows = torch.cat([x.reshape(B * T, H), yd.reshape(B * U1, H)], dim=0)
        a = self.fc1(rows)
        xa, ya = a[:B * T].view(B, T, H), a[B * T:].view(B, U1, H)
        K = self.fc2.fc.weight.shape[0]
        if _tr.joint_fused_supported(B, T, U1, H, K):
            # (:73-76) as one operator: the (B, T, U1, H) joint tensor exists only inside the MFMA operands
            return _tr.FusedJointFunction.apply(xa, ya, self.fc2.fc.weight, self.fc2.fc.bias)
        z = _tr.JointFunction.apply(xa, ya)  # (:73)
        out = self.fc2(z)
        return _tr.LogSoftmaxFunction.apply(out)

    def collate(self, inputs, labels):
        max_t = max(max(i.shape[0] for i in inputs), int(self.pad_frames or 0))
        max_t = self.conv_out_size(max_t, 0)
        x_lens = torch.IntTensor([max_t] * len(inputs))
        x = self._pad_inputs(inputs)
        y_lens = torch.IntTensor([len(l) for l in labels])
        y = torch.IntTensor([int(l) for label in labels for l in label])
        return [x, y, x_lens, y_lens]

    def label_collate(self, labels):
        """transducer_model.py:101-113: pad with the first utterance's last label (ignored by the loss)."""
        batch_size = len(labels)
        end_tok = labels[0][-1]
        max_len = max(len(l) for l in labels)
        cat_labels = np.full((batch_size, max_len), fill_value=end_tok, dtype=np.int64)
        for e, l in enumerate(labels):
            cat_labels[e, :len(l)] = l
        return torch.LongTensor(cat_labels)

    def infer(self, batch, beam_size=4):
        """transducer_model.py:91-100: static beam search over each utterance's lattice out[e, :T, :len(l)+1]."""
        u1 = [len(l) + 1 for l in batch[1]]

        def run():
            with torch.no_grad():
                out = self(batch)
            return _tr.decode_static_batch(out, u1, beam_size=beam_size, blank=self.blank)[0]
        return self._healthy(run)


def end_pad_concat(labels, min_len=0):
    """seq2seq.py:239-248: pad with the first utterance's last item (assumed to be the end token).  `min_len`: pad at
    least this far (the global-batch maximum of a data-parallel shard: the padded positions are scored, :61-63)."""
    batch_size = len(labels)
    end_tok = labels[0][-1]
    max_len = max(max(len(l) for l in labels), int(min_len or 0))
    cat_labels = np.full((batch_size, max_len), fill_value=end_tok, dtype=np.int64)
    for e, l in enumerate(labels):
        cat_labels[e, :len(l)] = l
    return cat_labels


class NNAttention(nn.Module):
    """seq2seq.py:331-360 as a parameter holder (conv.{weight,bias}, nn.1.fc.{weight,bias}); the arithmetic is
    sa_attention_fwd / sa_attention_bwd."""

    def __init__(self, n_channels, kernel_size=15, log_t=False):
        super().__init__()
        assert kernel_size % 2 == 1, "Kernel size should be odd for 'same' conv."
        padding = (kernel_size - 1) // 2
        self.conv = nn.Conv1d(1, n_channels, kernel_size, padding=padding)
        self.nn = nn.Sequential(nn.ReLU(), LinearND(n_channels, 1))
        self.log_t = log_t

    def forward(self, eh, dhx, ax=None):
        """eh (B, T, H), dhx (B, 1, H) or (B, H), ax (B, T) or None -> (sx (B, 1, H), ax (B, T))."""
        sx, ax = _s2s.attention(eh.contiguous(), dhx.reshape(eh.shape[0], -1).contiguous(), ax,
                                self.conv.weight.detach().reshape(eh.shape[2], -1).contiguous(),
                                self.conv.bias.detach(), self.nn[1].fc.weight.detach().reshape(-1).contiguous(),
                                self.nn[1].fc.bias.detach(), self.log_t)
        return sx.unsqueeze(1), ax


class Seq2Seq(Model):
    """seq2seq.py:14-237 on the HIP ops: the shared encoder, then ONE autograd node for the attention decoder
    (speech_amd.seq2seq.DecoderFunction) and the summed cross-entropy (XentFunction)."""

    def __init__(self, freq_dim, vocab_size, config):
        super().__init__(freq_dim, config)
        self._ctor_args = (freq_dim, vocab_size, config)
        decoder_cfg = config["decoder"]
        rnn_dim = self.encoder_dim
        embed_dim = decoder_cfg["embedding_dim"]
        assert embed_dim == rnn_dim, "the context vector is added to the embedding (seq2seq.py:95): dims must match"
        self.embedding = nn.Embedding(vocab_size, embed_dim)
        self.dec_rnn = nn.GRUCell(input_size=embed_dim, hidden_size=rnn_dim)
        self.attend = NNAttention(rnn_dim, log_t=decoder_cfg.get("log_t", False))
        self.sample_prob = decoder_cfg.get("sample_prob", 0)
        self.scheduled_sampling = (self.sample_prob != 0)
        # *NB* vocab_size - 1 classes: the start-of-sequence token is never predicted (seq2seq.py:33-35)
        self.fc = LinearND(rnn_dim, vocab_size - 1)

    def set_eval(self):
        self.eval()
        self.volatile = True
        self.scheduled_sampling = False

    def set_train(self):
        self.train()
        self.volatile = False
        self.scheduled_sampling = (self.sample_prob != 0)

    def _dec_params(self):
        return [self.embedding.weight, self.dec_rnn.weight_ih, self.dec_rnn.weight_hh, self.dec_rnn.bias_ih,
                self.dec_rnn.bias_hh, self.attend.conv.weight, self.attend.conv.bias, self.attend.nn[1].fc.weight,
                self.attend.nn[1].fc.bias, self.fc.fc.weight, self.fc.fc.bias]

    def _param_dict(self):
        H = self.encoder_dim
        P = dict(zip(_s2s._NAMES, [p.detach() for p in self._dec_params()]))
        P["conv_w"] = P["conv_w"].reshape(H, -1).contiguous()
        P["nn_w"] = P["nn_w"].reshape(H).contiguous()
        return P

    def loss(self, batch):
        x, y = self._collate_staged(*batch)
        if self.is_cuda:
            x = self._to_device(x)
            y = y.cuda(non_blocking=True)
        with torch.set_grad_enabled(not self.volatile):
            out, alis = self.forward_impl(x, y)
            batch_size, _, out_dim = out.size()
            out = out.reshape(-1, out_dim)
            tgt = y[:, 1:].contiguous().view(-1)
            # sum / batch_size (:61-63); a data-parallel shard divides by the GLOBAL batch size
            return _s2s.XentFunction.apply(out, tgt, 1.0 / (self.loss_denominator or batch_size))

    def forward_impl(self, x, y):
        x = self.encode(x)
        return self.decode(x, y)

    def forward(self, batch):
        x, y = self._collate_staged(*batch)
        if self.is_cuda:
            x = self._to_device(x)
            y = y.cuda(non_blocking=True)
        return self.forward_impl(x, y)[0]

    def decode(self, x, y):
        """seq2seq.py:77-112.  x (B, T', H) encoder states, y (B, U) labels -> (logits (B, U-1, V-1), aligns)."""
        sp = self.sample_prob if self.scheduled_sampling else 0
        return _s2s.DecoderFunction.apply(x, y, self.attend.log_t, sp, *self._dec_params())

    def decode_step(self, x, y, state=None, softmax=False):
        """seq2seq.py:114-138.  y (B, 1) labels; state None or (hx, ax, sx) as returned by the previous call."""
        with torch.no_grad():
            st = None if state is None else (state[0], state[1], state[2].reshape(x.shape[0], -1))
            out, (hx, ax, sx) = _s2s.step(x.contiguous(), y.reshape(-1).to(x.device), st, self._param_dict(),
                                          self.attend.log_t)
            if softmax:
                out = _tr.LogSoftmaxFunction.apply(out)
        return out, (hx, ax, sx.unsqueeze(1))

    def predict(self, batch):
        probs = self(batch)
        B, U1, K = probs.shape
        return _s2s.argmax_rows(probs.reshape(B * U1, K).contiguous()).view(B, U1).cpu().numpy().tolist()

    def infer_decode(self, x, y, end_tok, max_len):
        probs = []
        argmaxs = [y]
        state = None
        for e in range(max_len):
            out, state = self.decode_step(x, y, state=state)
            probs.append(out)
            y = _s2s.argmax_rows(out).unsqueeze(1)
            argmaxs.append(y)
            if int(torch.sum(y == end_tok)) == y.numel():
                break
        return torch.cat(probs), torch.cat(argmaxs, dim=1)

    def infer(self, batch, max_len=200):
        """seq2seq.py:160-178: greedy decode from the start tokens (no beam search)."""
        x, y = self._collate_staged(*batch)
        end_tok = int(y[0, -1])

        def run():
            with torch.no_grad():
                enc = self.encode(self._to_device(x))
                # the whole token loop in the library (no device -> host round trip per token); infer_decode remains the
                # step-by-step form of the reference's API
                toks = _s2s.greedy_decode(enc, y[:, 0], self._param_dict(), self.attend.log_t, end_tok, max_len)
            return [seq.tolist() for seq in toks.numpy()]
        return self._healthy(run)

    def beam_search(self, batch, beam_size=10, max_len=200):
        """seq2seq.py:180-229 for a batch of ONE utterance (the reference's loop indexes row 0 only), on the device:
        encoder, then ONE library call (speech_amd.seq2seq.beam_search -> sa_s2s_beam_search) that runs the whole search
        -- hypotheses are rows of a batched decoder step, selection / completion / stopping rule in a kernel -- and
        one copy of the winner back.  Returns [hypothesis tuple incl. the start token]; `last_beam_score` /
        `last_beam_info` keep the score and (search steps, completed hypotheses) of the last call."""
        x, y = self._collate_staged(*batch)
        if x.shape[0] != 1:
            raise ValueError("beam_search decodes a batch of one utterance (seq2seq.py:196-201), got %d" % x.shape[0])
        start_tok, end_tok = int(y[0, 0]), int(y[0, -1])

        def run():
            with torch.no_grad():
                enc = self.encode(self._to_device(x))
                return _s2s.beam_search(enc[0], self._param_dict(), self.attend.log_t, start_tok, end_tok, beam_size,
                                        max_len)
        hyp, self.last_beam_score, self.last_beam_info = self._healthy(run)
        return [hyp]

    def collate(self, inputs, labels):
        inputs = self._pad_inputs(inputs)
        labels = end_pad_concat(labels, self.pad_labels)
        return inputs, torch.from_numpy(labels)
