"""CPU oracle: a functional restatement of the reference's image->caption path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``deephumor_amd/`` imports this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may.
It is the *checker* for the HIP path, never the thing shipped or measured as the product.

Every function works on a flat ``state_dict`` (``{key: fp32 CPU tensor}``) with the
reference's key layout and follows the reference's algorithm, including its quirks:
per-image (batch-1) ``generate``, a full decoder re-forward for every generated token (no KV
cache), stochastic ``torch.multinomial`` selection drawing from the global CPU generator in
the reference's order, the LSTM hidden-state misalignment after a beam has ended, and the
discarded last Transformer step.  Citations are ``file:line`` under ``/root/reference``.

Pinning: ``oracle/make_golden.py`` runs the REAL reference (imported in the build container
through ``oracle/_standin``) on synthetic weights and commits its outputs under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against them
(token ids bit-exact, logits <= 1e-5).  Third-party arithmetic not in the reference tree:
torchvision's ``resnet50`` (requirements.txt:3, unpinned) -- restated here from the
published architecture (v1.5 stride placement); equivalence with real torchvision weights is
unverifiable without network and is stated as such in DESIGN.md.
"""
import math

import torch
import torch.nn.functional as F

PAD, UNK, BOS, EOS = 0, 1, 2, 3          # deephumor/data/vocab.py:5-12
RESNET50_STAGES = ((4, 64, 3, 1), (5, 128, 4, 2), (6, 256, 6, 2), (7, 512, 3, 2))


# --------------------------------------------------------------------------------------
# encoders (deephumor/models/encoders.py)
# --------------------------------------------------------------------------------------
def _bn(sd, p, x, eps=1e-5):
    """Eval-mode BatchNorm (running statistics), nn.BatchNorm{1,2}d default eps."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], False, 0.0, eps)


def _bottleneck(sd, p, x, stride):
    idt = x
    if (p + ".downsample.0.weight") in sd:
        idt = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride))
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
    y = F.relu(_bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], stride=stride, padding=1)))
    y = _bn(sd, p + ".bn3", F.conv2d(y, sd[p + ".conv3.weight"]))
    return F.relu(y + idt)


def resnet50_trunk(sd, p, images, taps=None):
    """``nn.Sequential(*list(resnet50.children())[:-2])`` (encoders.py:34-38):
    0 conv 7x7/2, 1 bn, 2 relu, 3 maxpool 3x3/2, 4..7 bottleneck stages [3,4,6,3]."""
    x = F.conv2d(images, sd[p + ".0.weight"], stride=2, padding=3)
    x = F.relu(_bn(sd, p + ".1", x))
    x = F.max_pool2d(x, 3, stride=2, padding=1)
    if taps is not None:
        taps["stem"] = x
    for idx, _planes, n_blocks, stride in RESNET50_STAGES:
        for b in range(n_blocks):
            x = _bottleneck(sd, f"{p}.{idx}.{b}", x, stride if b == 0 else 1)
        if taps is not None:
            taps[f"stage{idx}"] = x
    return x


def image_encoder(sd, p, images, spatial):
    """ImageEncoder.forward in eval mode (encoders.py:46-70).  Dropout is identity."""
    feats = resnet50_trunk(sd, p + ".resnet", images)
    bs, dim = feats.shape[:2]
    pooled = feats.mean(dim=(2, 3))                                  # AdaptiveAvgPool2d(1), :60
    emb = _bn(sd, p + ".bn", F.linear(pooled, sd[p + ".linear.weight"], sd[p + ".linear.bias"]))
    if not spatial:
        return emb
    grid = feats.reshape(bs, dim, -1).transpose(2, 1)                # :65-66
    return emb, F.linear(grid, sd[p + ".linear.weight"], sd[p + ".linear.bias"])   # no BN, :67


def label_encoder(sd, p, labels):
    """LabelEncoder.forward (encoders.py:96-106): mean over ALL label positions, pads included."""
    return sd[p + ".embedding.weight"][labels].mean(dim=1)


def image_label_encoder(sd, p, images, labels):
    """ImageLabelEncoder.forward (encoders.py:129-144)."""
    both = torch.cat([image_encoder(sd, p + ".image_encoder", images, False),
                      label_encoder(sd, p + ".label_encoder", labels)], dim=1)
    return F.linear(both, sd[p + ".linear.weight"], sd[p + ".linear.bias"])


# --------------------------------------------------------------------------------------
# beam bookkeeping (deephumor/models/beam.py)
# --------------------------------------------------------------------------------------
class BeamBook:
    """Restates BeamSearchHelper (beam.py:4-112) with explicit per-beam loops."""

    def __init__(self, temperature, beam_size, top_k, unk_index=UNK, eos_index=EOS):
        assert beam_size <= top_k, "`beam_size` should be less than `top_k`"       # beam.py:9
        self.t, self.b, self.k = temperature, beam_size, top_k
        self.unk, self.eos = unk_index, eos_index
        self.ended = torch.zeros(beam_size, dtype=torch.bool)                       # beam.py:22

    def keep_top_k(self, logits):
        """beam.py:32-37.  Strict ``<`` keeps ties at the threshold; unk always dropped;
        mutates ``logits`` in place like the reference."""
        kth = torch.topk(logits, self.k, dim=-1).values[:, -1:]
        drop = logits < kth
        drop[:, self.unk] = True
        logits[drop] = float("-inf")
        return logits

    def draw(self, scores, n):
        """beam.py:39-48: multinomial WITHOUT replacement on softmax(scores / T)."""
        return torch.multinomial(torch.softmax(scores / self.t, dim=-1), n)

    def expand(self, logits, seqs, vals):
        """beam.py:55-108.  Returns candidate ``(prev_seqs, prev_vals, new_ind, new_val, parent)``.

        A live beam contributes ``beam_size`` candidates, an ended beam exactly one with token 0
        and score increment 0 (beam.py:83-95).  ``self.ended`` becomes per-candidate."""
        logits = self.keep_top_k(logits)
        picks = self.draw(logits, self.b)                                           # [B, B]
        pick_val = torch.gather(logits, 1, picks).log_softmax(-1)                   # beam.py:79
        new_ind, new_val, parent, ended = [], [], [], []
        for beam in range(logits.size(0)):
            was_ended = bool(self.ended[beam])
            for j in range(1 if was_ended else self.b):
                tok = 0 if was_ended else int(picks[beam, j])
                new_ind.append(tok)
                new_val.append(0.0 if was_ended else float(pick_val[beam, j]))
                parent.append(beam)
                ended.append(was_ended or tok == self.eos)                           # beam.py:98
        self.ended = torch.tensor(ended)
        parent = torch.tensor(parent)
        new_val = torch.tensor(new_val, dtype=vals.dtype)
        return seqs[parent], vals.flatten()[parent], torch.tensor(new_ind), new_val, parent


def _trace_step(trace, logits):
    if trace is not None:
        top = torch.topk(logits, 2, dim=-1)
        trace.append({"top2_idx": top.indices.clone(), "top2_val": top.values.clone()})


# --------------------------------------------------------------------------------------
# LSTM decoder (deephumor/models/rnn_models.py)
# --------------------------------------------------------------------------------------
def _lstm_layers(sd, p):
    n = 0
    while f"{p}.weight_ih_l{n}" in sd:
        n += 1
    return n


def lstm_run(sd, p, x, state=None):
    """``nn.LSTM(batch_first=True)`` in eval mode, written out: gates = W_ih x + b_ih + W_hh h + b_hh,
    order i,f,g,o; c' = sigmoid(f) c + sigmoid(i) tanh(g); h' = sigmoid(o) tanh(c')."""
    n_layers = _lstm_layers(sd, p)
    bs, steps, _ = x.shape
    hid = sd[f"{p}.weight_hh_l0"].shape[1]
    if state is None:
        h = [x.new_zeros(bs, hid) for _ in range(n_layers)]
        c = [x.new_zeros(bs, hid) for _ in range(n_layers)]
    else:
        h, c = list(state[0].unbind(0)), list(state[1].unbind(0))
    outs = []
    for t in range(steps):
        inp = x[:, t]
        for l in range(n_layers):
            g = (F.linear(inp, sd[f"{p}.weight_ih_l{l}"], sd[f"{p}.bias_ih_l{l}"])
                 + F.linear(h[l], sd[f"{p}.weight_hh_l{l}"], sd[f"{p}.bias_hh_l{l}"]))
            gi, gf, gg, go = g.chunk(4, dim=1)
            c[l] = torch.sigmoid(gf) * c[l] + torch.sigmoid(gi) * torch.tanh(gg)
            h[l] = torch.sigmoid(go) * torch.tanh(c[l])
            inp = h[l]
        outs.append(inp)
    return torch.stack(outs, 1), (torch.stack(h, 0), torch.stack(c, 0))


def lstm_decoder_forward(sd, p, image_emb, captions, lengths=None):
    """LSTMDecoder.forward (rnn_models.py:28-46).  pack/pad_packed_sequence semantics: rows
    are valid for ``lengths[i]`` steps, later outputs are zeros, and the time axis is cut to
    ``max(lengths)``; the classifier then turns zero rows into its bias."""
    x = torch.cat([image_emb.unsqueeze(1), sd[p + ".embedding.weight"][captions]], dim=1)
    if lengths is None:
        lengths = torch.full((x.size(0),), x.size(1), dtype=torch.long)
    lengths = torch.as_tensor(lengths)
    out, _ = lstm_run(sd, p + ".lstm", x)
    steps = int(lengths.max())
    out = out[:, :steps]
    valid = torch.arange(steps)[None, :] < lengths[:, None]
    out = out * valid[..., None]
    return F.linear(out, sd[p + ".classifier.weight"], sd[p + ".classifier.bias"])


def lstm_decoder_generate(sd, p, image_emb, caption=None, max_len=25, temperature=1.0,
                          beam_size=10, top_k=50, eos_index=EOS, trace=None):
    """LSTMDecoder.generate (rnn_models.py:48-143).  ``image_emb`` is ``[1, 1, E]``."""
    book = BeamBook(temperature, beam_size, top_k, eos_index=eos_index)
    emb_w, cls_w, cls_b = sd[p + ".embedding.weight"], sd[p + ".classifier.weight"], sd[p + ".classifier.bias"]
    inputs = image_emb if caption is None else torch.cat([image_emb, emb_w[caption]], dim=1)
    out, (h, c) = lstm_run(sd, p + ".lstm", inputs)
    logits = F.linear(out[:, -1], cls_w, cls_b)
    _trace_step(trace, logits)
    h, c = h.repeat(1, beam_size, 1), c.repeat(1, beam_size, 1)                     # :84
    logits = book.keep_top_k(logits)
    first = book.draw(logits, beam_size)                                            # [1, B]
    vals = torch.gather(logits, 1, first).log_softmax(-1).T                         # [B, 1]
    last = first.T
    seqs = last.clone()
    if caption is not None:
        seqs = torch.cat([caption.repeat(beam_size, 1), seqs], dim=1)
    book.ended = (last == eos_index).view(-1)                                       # :103
    for _ in range(seqs.size(1), max_len):
        out, (h, c) = lstm_run(sd, p + ".lstm", emb_w[last], (h, c))
        logits = F.linear(out[:, -1], cls_w, cls_b)
        _trace_step(trace, logits)
        prev_seqs, prev_vals, new_ind, new_val, _ = book.expand(logits, seqs, vals)
        cand_seq = torch.cat([prev_seqs, new_ind[:, None]], dim=-1)
        cand_val = prev_vals + new_val
        keep = book.draw(cand_val, beam_size)                                       # :120
        vals, seqs = cand_val[keep], cand_seq[keep]
        last = seqs[:, -1:]
        book.ended = book.ended[keep]
        if bool(book.ended.all()):
            break
        # :135-137 -- indexes a dense B*B layout with candidate positions; once a beam has
        # ended the candidate list is shorter than B*B and the hidden states are misaligned.
        # Kept on purpose: it is the reference's behaviour.
        h = torch.repeat_interleave(h, beam_size, dim=1)[:, keep]
        c = torch.repeat_interleave(c, beam_size, dim=1)[:, keep]
    final = book.draw(vals, 1)
    return seqs[final, :].squeeze()


# --------------------------------------------------------------------------------------
# Transformer decoders (deephumor/models/transformers.py)
# --------------------------------------------------------------------------------------
def mha(sd, p, query, key, value, mask, n_heads):
    """MultiHeadAttentionLayer.forward (transformers.py:82-129).  NB the key is viewed with the
    QUERY's sequence length (:102), so callers pad both to one length."""
    bs, seq = query.shape[:2]
    hid = query.shape[-1]
    dh = hid // n_heads
    q = F.linear(query, sd[p + ".fc_q.weight"], sd[p + ".fc_q.bias"]).view(bs, seq, n_heads, dh).permute(0, 2, 1, 3)
    k = F.linear(key, sd[p + ".fc_k.weight"], sd[p + ".fc_k.bias"]).view(bs, seq, n_heads, dh).permute(0, 2, 3, 1)
    v = F.linear(value, sd[p + ".fc_v.weight"], sd[p + ".fc_v.bias"]).view(bs, seq, n_heads, dh).permute(0, 2, 1, 3)
    energy = (q @ k) / sd[p + ".scale"]
    if mask is not None:
        energy = energy.masked_fill(mask.unsqueeze(1), -1e8)                        # :110-111
    att = torch.softmax(energy, dim=-1)
    x = (att @ v).permute(0, 2, 1, 3).reshape(bs, seq, hid)
    return F.linear(x, sd[p + ".fc_o.weight"], sd[p + ".fc_o.bias"])


def _ln(sd, p, x):
    return F.layer_norm(x, x.shape[-1:], sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def transformer_forward(sd, p, x, enc_out, start_emb, pad_index, n_heads):
    """TransformerDecoder.forward (transformers.py:432-490) when ``enc_out`` is given,
    SelfAttentionTransformerDecoder.forward (:694-738) when it is ``None``."""
    bs, dec_len = x.shape
    if start_emb is not None:
        dec_len += 1
    if enc_out is not None:
        enc_len, hid = enc_out.shape[1:3]
        seq = max(dec_len, enc_len)                                                 # :450
        x = torch.cat([x, torch.full((bs, seq - dec_len), pad_index, dtype=torch.long)], dim=1)
        enc_out = torch.cat([enc_out, torch.zeros(bs, seq - enc_len, hid)], dim=1)
    tok = sd[p + ".tok_embedding.weight"][x]
    if start_emb is not None:
        tok = torch.cat([start_emb.unsqueeze(1), tok], dim=1)
    tok = tok / sd[p + ".scale"]                                                     # image slot scaled too, :462
    seq = tok.size(1)
    h = tok + sd[p + ".pos_embedding.weight"][torch.arange(seq)][None]
    ids = x if start_emb is None else torch.cat([torch.ones(bs, 1, dtype=torch.long), x], dim=1)   # :474
    key_is_pad = (ids == pad_index)[:, None, :].expand(bs, seq, seq)
    causal = torch.triu(torch.ones(seq, seq), 1).bool()[None]
    self_mask = key_is_pad | causal
    enc_mask = None
    if enc_out is not None:
        row_nonzero = (enc_out != 0.).all(dim=-1)                                    # :480
        enc_mask = (row_nonzero.long() == pad_index)[:, None, :].expand(bs, seq, seq)   # :481 get_pad_mask(x, enc_inp_mask, pad_index)
    n = 0
    while f"{p}.layers.{n}.self_attn.fc_q.weight" in sd:
        lp = f"{p}.layers.{n}"
        h = _ln(sd, lp + ".self_attn_ln", h + mha(sd, lp + ".self_attn", h, h, h, self_mask, n_heads))
        if enc_out is not None:
            h = _ln(sd, lp + ".enc_attn_ln", h + mha(sd, lp + ".enc_attn", h, enc_out, enc_out, enc_mask, n_heads))
        ff = F.linear(torch.relu(F.linear(h, sd[lp + ".pf.fc_1.weight"], sd[lp + ".pf.fc_1.bias"])),
                      sd[lp + ".pf.fc_2.weight"], sd[lp + ".pf.fc_2.bias"])
        h = _ln(sd, lp + ".pf_ln", h + ff)
        n += 1
    return F.linear(h, sd[p + ".classifier.weight"], sd[p + ".classifier.bias"])


def transformer_generate(sd, p, start_emb, enc_out, pad_index, n_heads, caption=None, max_len=25,
                         temperature=1.0, beam_size=10, top_k=50, eos_index=EOS, trace=None):
    """TransformerDecoder.generate (transformers.py:492-579) /
    SelfAttentionTransformerDecoder.generate (:740-825, ``enc_out=None``)."""
    book = BeamBook(temperature, beam_size, top_k, eos_index=eos_index)
    seqs = torch.full((1, max_len), pad_index, dtype=torch.long)
    pos = 0
    if caption is not None:
        pos = caption.size(1)
        seqs[:, :pos] = caption
    logits = transformer_forward(sd, p, seqs, enc_out, start_emb, pad_index, n_heads)[:, pos, :]
    _trace_step(trace, logits)
    logits = book.keep_top_k(logits)
    first = book.draw(logits, beam_size)
    vals = torch.gather(logits, 1, first).log_softmax(-1).T
    seqs = seqs.repeat(beam_size, 1)
    seqs[:, pos:pos + 1] = first.T
    if enc_out is not None:
        enc_out = enc_out.repeat(beam_size, 1, 1)
    start_emb = start_emb.repeat(beam_size, 1)
    # NB: ``book.ended`` is NOT updated from the first draw (unlike the LSTM, :540-545).
    i = pos
    for i in range(pos + 1, max_len + 1):
        logits = transformer_forward(sd, p, seqs, enc_out, start_emb, pad_index, n_heads)[:, i, :]
        _trace_step(trace, logits)
        prev_seqs, prev_vals, new_ind, new_val, _ = book.expand(logits, seqs, vals)
        if i < max_len:
            prev_seqs[:, i] = new_ind          # at i == max_len the reference's slice is empty (:557)
        cand_val = prev_vals + new_val
        keep = book.draw(cand_val, beam_size)
        vals, seqs = cand_val[keep], prev_seqs[keep]
        book.ended = book.ended[keep]
        if bool(book.ended.all()):
            break
    final = book.draw(vals, 1)
    return seqs[final, :i].squeeze()


# --------------------------------------------------------------------------------------
# captioning models (deephumor/models/caption_models.py)
# --------------------------------------------------------------------------------------
KINDS = ("CaptioningLSTM", "CaptioningLSTMWithLabels", "CaptioningTransformerBase",
         "CaptioningTransformer", "CaptioningTransformerWithLabels")


def _encode(kind, sd, images, labels):
    """Returns ``(start_emb, enc_out)``; the last kind is the BASELINE config-5 composition
    assembled from reference sub-modules as SURVEY.md section 8(a) row A4 defines it."""
    if kind == "CaptioningLSTM":
        return image_encoder(sd, "encoder", images, False), None
    if kind == "CaptioningLSTMWithLabels":
        return image_label_encoder(sd, "encoder", images, labels), None
    if kind == "CaptioningTransformerBase":
        return image_encoder(sd, "encoder", images, False), None
    if kind == "CaptioningTransformer":
        return image_encoder(sd, "encoder", images, True)
    if kind == "CaptioningTransformerWithLabels":
        emb, spatial = image_encoder(sd, "encoder.image_encoder", images, True)
        both = torch.cat([emb, label_encoder(sd, "encoder.label_encoder", labels)], dim=1)
        return F.linear(both, sd["encoder.linear.weight"], sd["encoder.linear.bias"]), spatial
    raise ValueError(kind)


def model_forward(kind, sd, hp, images, captions, lengths=None, labels=None):
    """``Captioning*.forward`` (caption_models.py:42-46, 138-142, 259-272, 393-406), eval mode."""
    with torch.no_grad():
        start, enc_out = _encode(kind, sd, images, labels)
        if "LSTM" in kind:
            return lstm_decoder_forward(sd, "decoder", start, captions, lengths)
        return transformer_forward(sd, "decoder", captions, enc_out, start, hp["pad_index"], hp["n_heads"])


def model_generate(kind, sd, hp, image, label=None, caption=None, max_len=25, temperature=1.0,
                   beam_size=10, top_k=50, eos_index=EOS, trace=None):
    """``Captioning*.generate`` for ONE image ``[1, 3, H, W]`` (caption_models.py:48-74, 144-171,
    274-300, 408-434)."""
    with torch.no_grad():
        start, enc_out = _encode(kind, sd, image, label)
        kw = dict(caption=caption, max_len=max_len, temperature=temperature, beam_size=beam_size,
                  top_k=top_k, eos_index=eos_index, trace=trace)
        if "LSTM" in kind:
            return lstm_decoder_generate(sd, "decoder", start.unsqueeze(1), **kw)
        return transformer_generate(sd, "decoder", start, enc_out, hp["pad_index"], hp["n_heads"], **kw)


def perplexity(logits, targets, lengths, pad_index=PAD):
    """deephumor/experiments/metrics.py:4-9: mean over sequences of exp(-sum_t log p(target_t) / length)."""
    logp = logits.log_softmax(-1).gather(-1, targets.unsqueeze(-1)).squeeze(-1)
    logp = logp / lengths.unsqueeze(1)
    logp = logp.masked_fill(targets == pad_index, 0.)
    return (-logp.sum(dim=-1)).exp().mean()
