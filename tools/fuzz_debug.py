"""Re-runs ONE trial of tools/fuzz_generate.py with per-step logits traces of both sides (oracle ``trace=``, product ``logits_hook=``)
and prints where they part: logit differences, the margin at the top-k threshold, the beams' tokens.  TEST INFRASTRUCTURE."""
import json
import random
import sys

import torch

import fuzz_generate as F
from oracle import ref_path as R


def main():
    seed, i = int(sys.argv[1]), int(sys.argv[2])
    rng = random.Random(seed * 100003 + i)
    traces = {"o": [], "p": []}
    R._trace_step = lambda trace, logits: trace.append(logits.detach().clone()) if trace is not None else None
    odraw = R.BeamBook.draw

    def draw(self, scores, n):
        st = torch.get_rng_state()
        pr = torch.softmax(scores / self.t, dim=-1)
        q = pr / torch.empty_like(pr).exponential_(1)
        torch.set_rng_state(st)
        got = odraw(self, scores, n)
        top = q.reshape(-1, q.shape[-1]).topk(min(n + 1, q.shape[-1]), dim=-1).values
        rel = ((top[:, :-1] - top[:, 1:]) / top[:, :-1].clamp_min(1e-30))
        rel = rel[top[:, :-1] > 0]
        print(f"  oracle draw {tuple(scores.shape)} k={n}: smallest relative gap between consecutive winners p/E = {rel.min().item():.3e}")
        return got
    R.BeamBook.draw = draw
    orig_l, orig_t = R.lstm_decoder_generate, R.transformer_generate
    R.lstm_decoder_generate = lambda *a, **k: orig_l(*a, **dict(k, trace=traces["o"]))
    R.transformer_generate = lambda *a, **k: orig_t(*a, **dict(k, trace=traces["o"]))
    import deephumor_amd.models.rnn_models as rm
    import deephumor_amd.models.transformers as tm
    for cls in (rm.LSTMDecoder, tm.TransformerDecoder, tm.SelfAttentionTransformerDecoder):
        og = cls.generate

        def gen(self, *a, _og=og, **k):
            return _og(self, *a, **dict(k, logits_hook=lambda pos, lg: traces["p"].append((pos, lg.detach().float().cpu().clone()))))
        cls.generate = gen
    cfg, want, got = F.one_trial(rng, i)
    print(json.dumps(cfg))
    print("want", want)
    print("got ", got)
    k = cfg["top_k"]
    for s, (o, (pos, p)) in enumerate(zip(traces["o"], traces["p"])):
        o = o if torch.is_tensor(o) else torch.as_tensor(o)
        n = min(o.shape[0], p.shape[0])
        d = (o[:n] - p[:n, :o.shape[1]]).abs().max().item()
        top = o[:n].topk(min(k + 1, o.shape[1]), dim=-1).values
        gap = (top[:, k - 1] - top[:, k]).min().item() if o.shape[1] > k else float("nan")
        unk_rows = [r for r in range(n) if 1 in o[r].topk(min(k, o.shape[1])).indices.tolist()]
        print(f"step {s} pos {pos}: <unk> in the top-k of rows {unk_rows};  rows {o.shape[0]} / {p.shape[0]}  max|dlogit| {d:.3e}  min gap at the top-k threshold {gap:.3e}")


if __name__ == "__main__":
    main()
