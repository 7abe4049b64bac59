"""oracle/seq2seq_beam_ref.py -- TEST INFRASTRUCTURE ONLY (imported by tests/ and oracle/gen_golden.py, never by the
product).  A CPU restatement of the reference's attention-decoder beam search,
/root/reference/speech/models/seq2seq.py:180-227 (`Seq2Seq.beam_search`), written as an explicit state machine over a
step function so that what the device kernel has to reproduce is spelled out:

  * a hypothesis is (tokens, score, decoder state); the search starts from ((start,), 0, None)            (:192)
  * one search step expands EVERY live hypothesis by EVERY class i of the decoder's log-softmax row, in the order
    (hypothesis rank, class index) -- that order is the tie order, because Python's `sorted(..., reverse=True)` is
    stable                                                                                               (:196-206)
  * scores are Python floats: score + float(float32 log-probability), i.e. DOUBLE sums of float32 values (:203)
  * the first `beam_size` candidates of the sorted list that end in the end token are appended to `complete`   (:209-211)
  * the next beam is the first `beam_size` candidates of the WHOLE sorted list that do not end in the end token (the
    reference's py2 `filter(...)[:beam_size]`, :213-214) -- not only those inside the top `beam_size`
  * stop when the beam is empty, or when `complete` holds at least `beam_size` hypotheses whose score is strictly
    greater than the best live one's (:216-223), or after `max_len` steps
  * result: the best of `complete` under a stable descending sort, i.e. the EARLIEST appended among equal scores; the
    live beam's best if nothing completed                                                               (:225-229)

Pinned by tests/golden/seq2seq_beam.npz: hypotheses of the LIVE reference's own `beam_search` (oracle/gen_golden.py runs
it with the py2 `filter` shimmed to a list), reproduced exactly by this function driven (i) by the live reference's
`decode_step` at generation time and (ii) by oracle/torch_ref.TorchRefSeq2Seq in tests/test_oracle_seq2seq.py.
"""
import numpy as np


def beam_search(step_fn, start_tok, end_tok, beam_size=10, max_len=200, trace=None):
    """step_fn(tokens: list[int], states: list[state or None]) -> (logp float32 (n, K), new_states list) runs ONE decoder
    step for n hypotheses (log_softmax of the fc output, seq2seq.py:133-137).  Returns (hyp tuple, score float,
    info dict).  `trace`, if a list, receives one dict per search step (beam tokens, parents, scores)."""
    beam = [((int(start_tok),), 0.0, None)]
    complete = []
    min_margin = np.inf     # smallest NON-ZERO score gap across a selection cut (how robust the result is to fp noise)
    steps = 0
    for _ in range(max_len):
        steps += 1
        logp, states = step_fn([h[-1] for h, _, _ in beam], [st for _, _, st in beam])
        logp = np.asarray(logp, dtype=np.float32)
        cands = []          # (hyp, score, state, parent rank) in (hypothesis rank, class) order
        for r, (hyp, score, _) in enumerate(beam):
            row = logp[r].tolist()       # float32 -> Python float, exactly as `.numpy().tolist()` does (:201)
            for i, p in enumerate(row):
                cands.append((hyp + (i,), score + p, states[r], r))
        order = sorted(range(len(cands)), key=lambda j: cands[j][1], reverse=True)  # stable: ties keep (rank, class)
        top = order[:beam_size]
        for j in top:
            if cands[j][0][-1] == end_tok:
                complete.append(cands[j])
        live = [j for j in order if cands[j][0][-1] != end_tok][:beam_size]
        # margins (diagnostics only): the gaps at the two cuts
        sc = [cands[j][1] for j in order]
        if len(sc) > beam_size and sc[beam_size - 1] != sc[beam_size]:
            min_margin = min(min_margin, sc[beam_size - 1] - sc[beam_size])
        for a, b in zip(live[:-1], live[1:]):
            if cands[a][1] != cands[b][1]:
                min_margin = min(min_margin, cands[a][1] - cands[b][1])
        beam = [(cands[j][0], cands[j][1], cands[j][2]) for j in live]
        if trace is not None:
            trace.append({"tokens": [cands[j][0][-1] for j in live], "parents": [cands[j][3] for j in live],
                          "scores": [cands[j][1] for j in live], "n_complete": len(complete)})
        if len(beam) == 0:
            break
        if sum(c[1] > beam[0][1] for c in complete) >= beam_size:
            break
    done = sorted(complete, key=lambda c: c[1], reverse=True)
    if len(done) == 0:
        done = beam
    hyp, score = done[0][0], done[0][1]
    return tuple(int(t) for t in hyp), float(score), {"steps": steps, "n_complete": len(complete),
                                                      "min_margin": float(min_margin)}


def torch_step_fn(model, enc):
    """A step function over a torch model exposing decode_step(x, y, state) -> (logits, (hx, ax, sx)) with batch-1
    states (the reference's Seq2Seq, or oracle/torch_ref.TorchRefSeq2Seq).  One call per hypothesis, as the reference
    does (:199-200)."""
    import torch

    def fn(tokens, states):
        rows, new = [], []
        with torch.no_grad():
            for tok, st in zip(tokens, states):
                y = torch.tensor([[int(tok)]], dtype=torch.int64)
                out, st2 = model.decode_step(enc, y, st)
                rows.append(torch.log_softmax(out, dim=1).numpy()[0])
                new.append(st2)
        return np.stack(rows), new
    return fn
