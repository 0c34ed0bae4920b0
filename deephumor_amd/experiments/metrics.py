"""Evaluation metric of the reference (deephumor/experiments/metrics.py:4-9) on the HIP kernels."""
from .. import hip


def sequence_perplexity(logits, targets, lengths, pad_index=0):
    """Per-sequence perplexity ``[bs]``: exp(-sum over non-pad targets of log p(target) / length).

    ``logits`` fp32 ``[bs, L, V]`` (the models' ``forward`` output cut to the targets' length, trainer.py:75),
    ``targets`` int64 ``[bs, L]``, ``lengths`` ``[bs]``.  ``dh_token_logprob`` reads every logit once."""
    bs, l, v = logits.shape
    logp = hip.token_logprob(logits.reshape(bs * l, v), targets.reshape(-1))
    return hip.seq_perplexity(logp.view(bs, l), targets, lengths.to(targets.device).long(), pad_index)


def perplexity(logits, targets, lengths, pad_index=0):
    """``deephumor.experiments.metrics.perplexity``: mean of the per-sequence perplexities (0-D tensor)."""
    pp = sequence_perplexity(logits, targets, lengths, pad_index)
    return pp.sum() / pp.numel()


def sequence_perplexity_from_hidden(hidden, cls_weight, cls_bias, targets, lengths, pad_index=0):
    """Same quantity from the decoder's pre-classifier hidden states ``[bs, L, D]`` (bf16): the classifier GEMM leaves
    only per-group log-sum-exp partials and the target logit (``dh_vocab_logprob``) -- the ``[bs*L, V]`` fp32 logits
    (146 KB per position at V = 36,541) are never written or re-read."""
    bs, l, d = hidden.shape
    logp = hip.vocab_logprob(hidden.reshape(bs * l, d), cls_weight, cls_bias, targets.reshape(-1))
    return hip.seq_perplexity(logp.view(bs, l), targets, lengths.to(targets.device).long(), pad_index)
