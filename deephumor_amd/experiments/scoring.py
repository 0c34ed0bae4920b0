"""Teacher-forced scoring of (template, caption) pairs -- SURVEY.md section 8(f) rank 1.

The corpus has 300 template images and 900,000 captions (README.md:26,37): the ResNet-50 encoder output
depends on the template only, so it is computed ONCE per distinct template and gathered per caption;
the decoders then run their teacher-forced ``forward`` (same incremental engine as ``generate``) and the
perplexity kernels score every caption (call shape of trainer.py:63-81)."""
import torch

from .. import hip
from .metrics import sequence_perplexity, sequence_perplexity_from_hidden


def score_captions(model, template_images, template_index, captions, lengths, labels=None, batch_size=256,
                   pad_index=0):
    """Per-caption perplexity ``[n]`` for ``captions`` int64 ``[n, L]`` (tokens + <eos>, padded with ``pad_index``,
    as ``MemeDataset``/``pad_collate`` produce them) of templates ``template_index`` int64 ``[n]`` into
    ``template_images`` ``[T, 3, H, W]``.  ``lengths`` ``[n]`` = non-pad tokens per caption (trainer.py:67).  Works for every captioning model of this package;
    label models additionally take ``labels`` int64 ``[T, l]`` (one label per template)."""
    enc = model.encoder
    with torch.no_grad():
        if labels is not None:
            feats = enc(template_images, labels)
        else:
            feats = enc(template_images)
        spatial = None
        if isinstance(feats, tuple):
            feats, spatial = feats
        out = []
        for lo in range(0, captions.shape[0], batch_size):
            hi = min(lo + batch_size, captions.shape[0])
            idx = template_index[lo:hi].to(feats.device)
            emb = feats.index_select(0, idx)
            tgt = captions[lo:hi]                       # tokens + <eos>, zero padded (datasets.py:72-79, no <bos>)
            inp = tgt[:, :-1]                           # trainer.py:69-73: model(images, captions[:, :-1], lengths)
            dec = model.decoder
            width = dec.classifier.in_features
            if feats.dtype in hip.HALF_DTYPES and width % 64 == 0 and width >= 128:       # dh_vocab_logprob's contract; else the logits route
                # bf16 path: hidden states -> fused classifier + log-softmax gather, the [rows, V] logits never exist
                if hasattr(dec, "lstm"):
                    hidden, _, _ = dec.hidden_states(emb, inp, None)
                else:
                    hidden = dec._forward(inp, None if spatial is None else spatial.index_select(0, idx), emb,
                                          num_positions=tgt.shape[1], return_hidden=True)
                plan = dec._get_plan()
                hidden = hidden[:, :tgt.shape[1]].contiguous()
                out.append(sequence_perplexity_from_hidden(hidden, plan["cls_w"], plan["cls_b"], tgt.contiguous(),
                                                           lengths[lo:hi], pad_index))
                continue
            if spatial is not None:
                logits = dec(inp, enc_out=spatial.index_select(0, idx), start_emb=emb, num_positions=tgt.shape[1])
            elif hasattr(dec, "lstm"):
                logits = dec(emb, inp, None)
            else:
                logits = dec(inp, start_emb=emb)
            # output position p (0 = image slot) predicts caption token p: pred[:, :max_len] (trainer.py:75)
            logits = logits[:, :tgt.shape[1]].contiguous()
            out.append(sequence_perplexity(logits, tgt.contiguous(), lengths[lo:hi], pad_index))
        return torch.cat(out)
