"""LSTM caption decoder on the gfx950 kernels.

Drop-in for ``deephumor.models.rnn_models.LSTMDecoder`` (reference rnn_models.py:8-143): same
constructor, ``forward(image_emb, captions, lengths)`` and ``generate(image_emb, caption, ...)``;
``generate_batch`` is the new batched entry point (the reference is strictly one image per call,
SURVEY.md section 2b).  One time step of one layer is ONE launch on the bf16 path (``dh_lstm_layer_fused``: the
[x | h_prev[beam parent]] operand gathered by the loader, gate GEMM on the matrix cores, cell update in the epilogue,
state ping-pong); on the fp32 parity path ``dh_lstm_prepare`` (state/embedding gather incl. beam reorder) + per layer
``dh_linear`` ([x|h] gate GEMM) + ``dh_lstm_cell``.  Both are sequenced by the native ``dh_lstm_decode_step``.
"""
import os

import torch
from torch import nn

from .. import hip
from ._f32x_guard import f32x_guarded
from .beam import BeamOverflow, BeamSearchHelper, call_logits_hook, check_ids, check_lengths, classifier_must_be_finite, make_noise_source, resolve_seed, run_interleaved, warn_overflow_retry
from .encoders import _Planned


class LSTMDecoder(_Planned, nn.Module):
    """LSTM-based decoder (reference rnn_models.py:8-26)."""

    def __init__(self, num_tokens, emb_dim=256, hidden_size=512,
                 num_layers=3, dropout=0.1, embedding=None):
        super().__init__()
        self.num_tokens = num_tokens
        self.embedding = embedding if embedding is not None else nn.Embedding(num_tokens, emb_dim)
        self.lstm = nn.LSTM(emb_dim, hidden_size, num_layers, batch_first=True,
                            dropout=(0 if num_layers == 1 else dropout))
        self.classifier = nn.Linear(hidden_size, num_tokens)

    # -- derived constants: [W_ih | W_hh] per layer and the summed bias -------------------------
    def _build_plan(self):
        layers = []
        # fp32 weights with option "f32_split": the gate products and the classifier as three fp16 MFMAs on split operands
        # (csrc/gemm_f32x.hip; dh_split_f32x planes of the weights, made here once per weight version)
        split = (self.classifier.weight.dtype == torch.float32 and self.classifier.weight.is_cuda and bool(hip.option("f32_split"))
                 and all(hip.f32_split_ok(p) for p in self.parameters() if p.dim() > 1))
        for l in range(self.lstm.num_layers):
            w = torch.cat([getattr(self.lstm, f"weight_ih_l{l}").detach(),
                           getattr(self.lstm, f"weight_hh_l{l}").detach()], dim=1).contiguous()
            b = (getattr(self.lstm, f"bias_ih_l{l}").detach().float()
                 + getattr(self.lstm, f"bias_hh_l{l}").detach().float()).contiguous()
            if w.dtype in hip.HALF_DTYPES:
                # gate-interleaved copy for the fused step kernel: row 4u+g = gate g (i, f, g, o) of hidden unit u
                hh = self.lstm.hidden_size
                w_il = w.view(4, hh, -1).permute(1, 0, 2).reshape(4 * hh, -1).contiguous()
                b_il = b.view(4, hh).t().reshape(-1).contiguous()
                # ... and that copy in MFMA fragment order for the register-stationary step (decode shapes)
                w_pk = hip.pack_mfma_fragments(w_il) if hip.lstm_layer_wreg_supported(w.shape[1] - hh, hh) else None
            else:
                w_il = b_il = w_pk = None
            w_x = hip.split_f32x(w) if split else None
            # ... and the split planes in MFMA fragment order: the gate product of a decode step with the weights stationary in registers
            w_xp = hip.pack_f32x_fragments(w_x) if (split and hip.option("decode_wreg")) else None
            layers.append((w, b, w_il, b_il, w_pk, w_x, w_xp))
        plan = dict(layers=layers, emb=self.embedding.weight.detach(),
                    cls_w=self.classifier.weight.detach(), cls_b=self.classifier.bias.detach().float().contiguous(),
                    dtype=self.classifier.weight.dtype)
        if split:
            plan["cls_w_x"] = hip.split_f32x(plan["cls_w"].contiguous())
            # the classifier on a split top-layer state (csrc/gemm_f32xp.hip) fills the beam sampler's group maxima like the 16-bit paths
            plan["f32_planes"] = bool(hip.option("f32_planes")) and self.lstm.hidden_size % 32 == 0
        if plan["dtype"] in hip.HALF_DTYPES and plan["cls_w"].is_cuda and plan["cls_w"].shape[1] == 512 and hip.option("vocab_wreg"):
            # the beam-search classifier with the weights streamed from L2 into registers (csrc/vocab_wreg.hip): padded, fragment-packed copy
            plan["cls_w_pk"], plan["cls_b_pad"] = hip.pack_vocab_weights(plan["cls_w"], plan["cls_b"])
        return plan

    def _check_mode(self):
        if self.training and self.lstm.dropout > 0:
            raise RuntimeError("deephumor_amd implements the inference path; call model.eval()")

    class _State:
        """Recurrent state of n_img*beam logical rows + per-step scratch (cached per row count), described to
        the native step driver (``dh_lstm_decode_step``) through plain C structs."""

        def __init__(self, dec, plan, n_img, beam, dev):
            self.nl, self.hh, self.e = dec.lstm.num_layers, dec.lstm.hidden_size, dec.lstm.input_size
            self.dev, self.dtype = dev, plan["dtype"]
            self.rows_total = r = n_img * beam
            # never read before it is written: the first step runs with started == 0 (zero state by flag)
            self.h = torch.empty((self.nl, r, self.hh), device=dev, dtype=self.dtype)
            self.c = torch.empty((self.nl, r, self.hh), device=dev)             # cell state always fp32
            self.started = 0                                                    # 0: zero state, then 1, 2, 1, 2, ...
            self._scratch = {}
            self.planes = bool(plan.get("f32_planes"))
            self.c_layers = (hip.LstmLayer * self.nl)()
            for i, (w, b, w_il, b_il, w_pk, w_x, w_xp) in enumerate(plan["layers"]):
                if w_x is not None:
                    self.c_layers[i].w_x = w_x.data_ptr()
                if w_xp is not None:
                    self.c_layers[i].w_xp = w_xp.data_ptr()
                self.c_layers[i].w, self.c_layers[i].b = w.data_ptr(), b.data_ptr()
                if w_il is not None:
                    self.c_layers[i].w_il, self.c_layers[i].b_il = w_il.data_ptr(), b_il.data_ptr()
                if w_pk is not None:
                    self.c_layers[i].w_pk = w_pk.data_ptr()
            m = self.c_model = hip.LstmModel()
            m.n_layers, m.E, m.Hh, m.V = self.nl, self.e, self.hh, dec.num_tokens
            m.dtype = {torch.float32: hip.F32, torch.bfloat16: hip.BF16, torch.float16: hip.F16}[self.dtype]
            m.layers = self.c_layers
            m.emb, m.cls_w, m.cls_b = plan["emb"].data_ptr(), plan["cls_w"].data_ptr(), plan["cls_b"].data_ptr()
            if "cls_w_pk" in plan:
                m.cls_w_pk, m.cls_b_pad = plan["cls_w_pk"].data_ptr(), plan["cls_b_pad"].data_ptr()
            if "cls_w_x" in plan:
                m.cls_w_x = plan["cls_w_x"].data_ptr()
            m.h, m.c = self.h.data_ptr(), self.c.data_ptr()
            if self.dtype in hip.HALF_DTYPES:     # fused step kernel: other workgroups still gather the old state rows
                self.h_alt, self.c_alt = torch.empty_like(self.h), torch.empty_like(self.c)
                m.h_alt, m.c_alt = self.h_alt.data_ptr(), self.c_alt.data_ptr()

        def scratch(self, rows):
            if rows not in self._scratch:
                nl, hh, e, dev, dt = self.nl, self.hh, self.e, self.dev, self.dtype
                bufs = dict(
                    xcat0=torch.empty((rows, e + hh), device=dev, dtype=dt),
                    xcatl=torch.empty((max(nl - 1, 1), rows, 2 * hh), device=dev, dtype=dt),
                    c_cur=torch.empty((nl, rows, hh), device=dev),
                    gates=torch.empty((rows, 4 * hh), device=dev),              # gate pre-activations fp32
                    hout=torch.empty((rows, hh), device=dev, dtype=dt))
                if self.planes:                        # fp32, split operands: the top layer's state as fp16 planes for the classifier
                    bufs["topp"] = torch.empty((2, rows, hh), device=dev, dtype=torch.float16)
                    bufs["xcat0p"] = torch.empty((2, rows, e + hh), device=dev, dtype=torch.float16)       # ... and of the gate GEMMs' operands
                    bufs["xcatlp"] = torch.empty((max(nl - 1, 1), 2, rows, 2 * hh), device=dev, dtype=torch.float16)
                c = hip.LstmScratch()
                for k, v in bufs.items():
                    setattr(c, k, v.data_ptr())
                bufs["c"] = c
                self._scratch[rows] = bufs
            return self._scratch[rows]

    def _step(self, plan, st, rows, rpi, mult, rows_total, img_emb=None, tokens=None, tok_pos=0, hparent=None,
              hout=None, logits=None, group_max=None):
        """One LSTM time step for ``rows`` compact rows (+ classifier if ``logits``): one native call."""
        sc = st.scratch(rows)
        hip.lstm_decode_step(st.c_model, sc["c"], img_emb, tokens, tok_pos, hparent, st.started, rows, rpi, mult,
                             rows_total, h_out=hout, logits=logits, group_max=group_max)
        st.started = 2 if st.started == 1 else 1
        return hout if hout is not None else sc["hout"]

    @f32x_guarded
    def forward(self, image_emb, captions, lengths=None):
        """Teacher-forced logits ``[bs, max(lengths), num_tokens]`` (reference rnn_models.py:28-46).
        Rows past ``lengths[i]`` are the packed-sequence zeros, i.e. the classifier bias."""
        hs, bs, steps_out = self.hidden_states(image_emb, captions, lengths)
        plan = self._get_plan()
        out = hip.linear(hs.view(bs * steps_out, -1), plan["cls_w"], plan["cls_b"], out_dtype=torch.float32, tag="vocab", w_x=plan.get("cls_w_x"))
        return out.view(bs, steps_out, -1)

    def hidden_states(self, image_emb, captions, lengths=None):
        """The top layer's hidden states ``[bs, max(lengths), hidden]`` in front of the classifier (zero past
        ``lengths[i]``, as ``pad_packed_sequence`` leaves them) -> ``(hs, bs, steps)``."""
        self._check_mode()
        plan = self._get_plan()
        bs, steps = captions.shape[0], captions.shape[1] + 1
        dev = image_emb.device
        if lengths is None:
            lengths = torch.full((bs,), steps, dtype=torch.long)
        lengths = torch.as_tensor(lengths).cpu()
        check_lengths(lengths, steps)                     # (pack_padded_sequence's errors)
        check_ids(captions, self.embedding.num_embeddings)
        steps_out = int(lengths.max())
        hh = self.lstm.hidden_size
        tokens = captions.to(torch.int32).contiguous()
        st = self._State(self, plan, bs, 1, dev)
        hs = torch.zeros((bs, steps_out, hh), device=dev, dtype=plan["dtype"])
        for t in range(steps_out):
            self._step(plan, st, bs, 1, 1, bs, img_emb=image_emb if t == 0 else None,
                       tokens=None if t == 0 else tokens, tok_pos=t - 1, hout=hs[:, t, :])
        valid = (torch.arange(steps_out)[None, :] < lengths[:, None]).to(dev)
        hs.mul_(valid[..., None])          # pad_packed_sequence zero rows (mask, not arithmetic on valid rows)
        return hs, bs, steps_out

    @f32x_guarded
    def generate_batch(self, image_emb, caption=None, max_len=25, temperature=1.0, beam_size=10, top_k=50,
                       eos_index=3, seed=None, img0=0, noise_source=None, logits_hook=None, streams=1, seed_tensor=None,
                       defer_check=False, early_stop_every=0, exact=False, rng=None):
        """Batched beam-search sampling for ``image_emb [N, 1, E]`` or ``[N, E]``.

        Returns ``(tokens int64 [N, max_len] zero-padded, lengths int64 [N])``; row ``i`` equals
        what the reference's ``generate`` returns for image ``i`` under the same random draws
        (rnn_models.py:48-143, incl. the hidden-state indexing at :135-137).  ``streams`` > 1 decodes
        that many image sub-batches concurrently on separate HIP streams (same captions).
        ``early_stop_every=k`` (> 0) checks every k steps whether every image has finished (the reference's
        ``all_ended()`` break, rnn_models.py:131) and stops decoding then -- one host sync per check, same captions.
        ``exact=True`` draws every row through the general sampler from the start (what a batch is repeated with automatically
        when flat logits overflow the pre-filtered samplers, see ``BeamOverflow``).
        ``rng="torch"``: the draws consume torch CPU generators in the reference's order (``beam.TorchRngNoise``): on the fp32 path
        ``torch.manual_seed(s); decoder.generate(emb, rng="torch")`` returns the reference's sampled caption; in a batch image ``i``
        replays ``torch.manual_seed(seed + img0 + i)``."""
        self._check_mode()
        plan = self._get_plan()
        classifier_must_be_finite(plan)
        check_ids(caption, self.embedding.num_embeddings)
        rng_seed = seed
        seed = 0 if rng == "torch" else resolve_seed(seed, noise_source)
        # rng="torch" with seed=None draws from torch's DEFAULT generator: its state is snapshotted once per call so that a repeated
        # session (BeamOverflow retry) replays the same draws (beam.TorchRngNoise)
        rng_state0 = torch.get_rng_state() if (rng == "torch" and rng_seed is None) else None
        image_emb = image_emb.reshape(image_emb.shape[0], -1).to(plan["dtype"]).contiguous()

        def session(lo, hi):
            n, b = hi - lo, beam_size
            r = n * b
            dev = image_emb.device
            # a prefix of max_len or more tokens: the reference still makes its first draw and returns prefix + 1 tokens -- it never
            # truncates to max_len (rnn_models.py:97-101: the loop simply does not run); the token buffers are that wide then
            eff_len = max(max_len, (0 if caption is None else caption.shape[1]) + 1)
            helper = BeamSearchHelper(temperature, beam_size, top_k, eos_index=eos_index, device=dev, n_img=n,
                                      max_len=eff_len, seed=seed, img0=img0 + lo,
                                      noise_source=make_noise_source(rng, rng_seed, noise_source, lo, hi, img0, rng_state0), seed_tensor=seed_tensor,
                                      exact=exact[0])
            pos = 0
            if caption is not None:
                pos = caption.shape[1]
                helper.set_prefix(caption[lo:hi])
            st = self._State(self, plan, n, b, dev)
            # logits always fp32; row stride padded to 64 floats so rows are 16-byte aligned (vector stores)
            logits = torch.empty((r, (self.num_tokens + 255) // 256 * 256), device=dev)[:, :self.num_tokens]   # whole 256-column chunks (vocab_wreg)
            gmax = (torch.empty((r, 4 * ((self.num_tokens + 255) // 256)), device=dev)[:, :hip.n_groups(self.num_tokens)]
                    if plan["dtype"] in hip.HALF_DTYPES or plan.get("f32_planes") else None)    # column-group maxima (16-bit paths, f32x planes)
            gm = None if gmax is None else gmax[:n]
            # image slot, then the teacher-forced prefix: one row per image living at logical row img*beam
            lg = logits[:n]
            self._step(plan, st, n, 1, b, r, img_emb=image_emb[lo:hi], logits=lg if pos == 0 else None,
                       group_max=gm if pos == 0 else None)
            for j in range(pos):
                last = j == pos - 1
                self._step(plan, st, n, 1, b, r, tokens=helper.tokens, tok_pos=j, logits=lg if last else None,
                           group_max=gm if last else None)
            if logits_hook is not None:
                call_logits_hook(logits_hook, pos, lg, helper)
            helper.step(lg, first=True, write_pos=pos, t=0, step_index=pos, first_sets_ended=True, group_max=gm)
            yield
            for i in range(pos + 1, max_len):
                self._step(plan, st, r, b, 1, r, tokens=helper.tokens, tok_pos=i - 1, hparent=helper.hparent,
                           logits=logits, group_max=gmax)
                if logits_hook is not None:
                    call_logits_hook(logits_hook, i, logits, helper)
                helper.step(logits, first=False, write_pos=i, t=0, step_index=i, group_max=gmax)
                yield
                if early_stop_every and (i - pos) % early_stop_every == 0 and bool(helper.done.all()):
                    break                                   # finished images are frozen by dh_beam_select: nothing left to do
            # (no decode step when the prefix fills max_len - 1: the reference then returns beam 0 -- see finalize)
            return helper.finalize(len_bias_done=1, full_len=eff_len, defer_check=defer_check, first_beam=pos + 1 >= max_len)

        exact = [bool(exact)]
        try:
            return run_interleaved(session, image_emb.shape[0], streams)
        except BeamOverflow:              # flat logits: more ties at a row's top-k threshold than the fast samplers hold -- once more,
            if exact[0]:
                raise
            exact[0] = True               # every row draw through the general sampler (same seed: same captions where nothing overflowed)
            warn_overflow_retry()
            return run_interleaved(session, image_emb.shape[0], streams)

    def generate(self, image_emb, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        """Single-image API of the reference (rnn_models.py:48-49): ``image_emb [1, 1, E]`` ->
        1-D int64 token tensor."""
        toks, lens = self.generate_batch(image_emb, caption=caption, max_len=max_len, temperature=temperature,
                                         beam_size=beam_size, top_k=top_k, eos_index=eos_index, **kw)
        return self.single_output(toks, lens, caption, max_len, beam_size)

    @staticmethod
    def single_output(toks, lens, caption, max_len, beam_size):
        """Row 0 of ``generate_batch``'s result in the shape the reference's single-image ``generate`` returns."""
        seq = toks[0, :int(lens[0])]
        if (0 if caption is None else caption.shape[1]) + 1 >= max_len:
            # no decode step ran: the reference indexes ``sample_seq`` with the [beam, 1] result of its final draw on the
            # [beam, 1] scores of the first step (rnn_models.py:140-141) and returns ``beam_size`` copies of beam 0's row
            seq = seq.unsqueeze(0).repeat(beam_size, 1)
        return seq.squeeze()
