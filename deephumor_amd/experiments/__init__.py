"""Mirror of the inference-side pieces of ``deephumor.experiments``: the
perplexity metric on the HIP kernels, corpus scoring with per-template encoder caching, and the
text <-> token helpers.  The Trainer / tensorboard harness is out of scope."""
from .metrics import perplexity, sequence_perplexity
from .scoring import score_captions
from .inference import text_to_seq, seq_to_text, split_caption

__all__ = ["perplexity", "sequence_perplexity", "score_captions", "text_to_seq", "seq_to_text", "split_caption"]
