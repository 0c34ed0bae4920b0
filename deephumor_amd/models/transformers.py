"""Transformer caption decoders on the gfx950 kernels.

Drop-in for the decoder half of ``deephumor.models.transformers`` (reference
transformers.py:43-165, 309-825): same class names, constructor arguments, ``forward`` /
``generate`` signatures and state-dict keys (including the ``scale`` parameters, which are part
of the checkpoint format, transformers.py:77-80,424-427).

Design difference (MI355X-first, results-equivalent -- SURVEY.md 8(a) G-TR.4): the reference
re-runs the whole padded sequence through all layers for every generated token and applies the
vocabulary classifier to all ``max(max_len+1, 49)`` positions; here every position is decoded
ONCE against a per-layer KV cache kept in HBM, for all images and beams of a batch at a time, and
beam reordering is an ancestor-index table, never a cache copy.  ``forward()`` (teacher forcing)
runs all positions at once in prefill form (batched GEMMs, one causal self-attention launch per layer,
``_forward_prefill``) when the sequence fits the register-resident attention kernels (<= 56 positions in bf16,
40 in fp32), and otherwise position by position on the incremental engine; both give the same logits (tested).
"""
import os

import torch
from torch import nn

from .. import hip
from ._f32x_guard import f32x_guarded
from .beam import BeamOverflow, BeamSearchHelper, call_logits_hook, check_ids, classifier_must_be_finite, make_noise_source, resolve_seed, run_interleaved, warn_overflow_retry
from .encoders import _Planned


def get_pad_mask(query, key, pad_index=0):
    """Padding mask from the Query and Key id sequences (reference transformers.py:12-26):
    bool ``[bs, query_len, key_len]``, True where ``key == pad_index``."""
    return hip.pad_mask(query, key, pad_index)


def get_autoregressive_mask(seq):
    """Autoregressive mask for the decoder inputs (reference transformers.py:29-40): bool ``[bs, L, L]``, True above
    the diagonal."""
    return hip.autoregressive_mask(seq)


def _no_train_dropout(module):
    if module.training and module.dropout.p > 0:
        raise RuntimeError("deephumor_amd implements the inference path; call model.eval()")


def _fp32(t):
    return t.detach().float().contiguous()


def _add_ln(x, y, ln):
    bs, l, d = x.shape
    out = hip.add_layernorm(x.reshape(bs * l, d).contiguous(), y.reshape(bs * l, d).contiguous(), _fp32(ln.weight),
                            _fp32(ln.bias), eps=ln.eps)
    return out.view(bs, l, d)


class MultiHeadAttentionLayer(nn.Module):
    """The reference layer (transformers.py:43-129): fc_q / fc_k / fc_v / fc_o + ``scale``.  The captioning decoders do
    not call this ``forward`` (they run the KV-cached / prefill kernels on the fused weights); it serves callers of the
    module API: four ``dh_linear`` launches around ``dh_attn_masked``."""

    def __init__(self, hid_dim=512, n_heads=8, dropout=0.):
        super().__init__()
        assert hid_dim % n_heads == 0, "hid_dim must be divisible by n_heads"
        self.hid_dim, self.n_heads, self.head_dim = hid_dim, n_heads, hid_dim // n_heads
        self.fc_q = nn.Linear(hid_dim, hid_dim)
        self.fc_k = nn.Linear(hid_dim, hid_dim)
        self.fc_v = nn.Linear(hid_dim, hid_dim)
        self.fc_o = nn.Linear(hid_dim, hid_dim)
        self.dropout = nn.Dropout(dropout)
        self.scale = nn.Parameter(torch.sqrt(torch.tensor(self.head_dim, dtype=torch.float32)), requires_grad=False)

    def forward(self, query, key, value, mask=None):
        """``query/key/value [bs, seq_len, hid_dim]``, ``mask`` bool ``[bs, seq_len, seq_len]`` (True = masked with
        -1e8) -> ``[bs, seq_len, hid_dim]`` (transformers.py:82-129).  As in the reference, K and V are viewed with the
        QUERY's sequence length (:94,:102-103), so all three must have the same length."""
        _no_train_dropout(self)
        bs, seq_len = query.shape[:2]
        d = self.hid_dim
        if key.shape[0] * key.shape[1] != bs * seq_len or value.shape[0] * value.shape[1] != bs * seq_len:
            raise RuntimeError(f"shape '[{bs}, {seq_len}, {self.n_heads}, {self.head_dim}]' is invalid for input of size "
                               f"{key.numel()}")                              # the reference's k.view(...) error
        lin = lambda x, fc: hip.linear(x.reshape(bs * seq_len, d).contiguous(), fc.weight.detach(), _fp32(fc.bias))
        q, k, v = lin(query, self.fc_q), lin(key, self.fc_k), lin(value, self.fc_v)
        x = hip.attn_masked(q, k, v, mask, bs, seq_len, d, self.n_heads, float(self.scale))
        return hip.linear(x, self.fc_o.weight.detach(), _fp32(self.fc_o.bias)).view(bs, seq_len, d)


class PositionwiseFeedforwardLayer(nn.Module):
    """fc_2(relu(fc_1(x))) (transformers.py:132-165)."""

    def __init__(self, hid_dim=512, pf_dim=2048, dropout=0.):
        super().__init__()
        self.fc_1 = nn.Linear(hid_dim, pf_dim)
        self.fc_2 = nn.Linear(pf_dim, hid_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        """``[bs, seq_len, hid_dim] -> [bs, seq_len, hid_dim]`` (transformers.py:151-165)."""
        _no_train_dropout(self)
        shape = x.shape
        h = hip.linear(x.reshape(-1, shape[-1]).contiguous(), self.fc_1.weight.detach(), _fp32(self.fc_1.bias), relu=True)
        return hip.linear(h, self.fc_2.weight.detach(), _fp32(self.fc_2.bias)).view(shape)


class DecoderLayer(nn.Module):
    """Post-LN decoder layer with encoder attention (transformers.py:309-377)."""

    def __init__(self, hid_dim=512, n_heads=8, pf_dim=2048, dropout=0.):
        super().__init__()
        self.self_attn = MultiHeadAttentionLayer(hid_dim, n_heads, dropout)
        self.self_attn_ln = nn.LayerNorm(hid_dim)
        self.enc_attn = MultiHeadAttentionLayer(hid_dim, n_heads, dropout)
        self.enc_attn_ln = nn.LayerNorm(hid_dim)
        self.pf = PositionwiseFeedforwardLayer(hid_dim, pf_dim, dropout)
        self.pf_ln = nn.LayerNorm(hid_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, enc_out, input_mask=None, enc_mask=None):
        """Module-API form of one layer (transformers.py:343-377): ``x, enc_out [bs, seq_len, hid_dim]``, masks bool
        ``[bs, seq_len, seq_len]``."""
        _no_train_dropout(self)
        x = _add_ln(x, self.self_attn(x, x, x, mask=input_mask), self.self_attn_ln)
        x = _add_ln(x, self.enc_attn(x, enc_out, enc_out, mask=enc_mask), self.enc_attn_ln)
        return _add_ln(x, self.pf(x), self.pf_ln)


class SelfAttentionDecoderLayer(nn.Module):
    """Decoder layer without encoder attention (transformers.py:582-636)."""

    def __init__(self, hid_dim=512, n_heads=8, pf_dim=2048, dropout=0.):
        super().__init__()
        self.self_attn = MultiHeadAttentionLayer(hid_dim, n_heads, dropout)
        self.self_attn_ln = nn.LayerNorm(hid_dim)
        self.pf = PositionwiseFeedforwardLayer(hid_dim, pf_dim, dropout)
        self.pf_ln = nn.LayerNorm(hid_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x, input_mask=None):
        """Module-API form of one layer (transformers.py:612-636)."""
        _no_train_dropout(self)
        x = _add_ln(x, self.self_attn(x, x, x, mask=input_mask), self.self_attn_ln)
        return _add_ln(x, self.pf(x), self.pf_ln)


class TransformerEncoder(nn.Module):
    """Exported by the reference (models/__init__.py:6-8) but dead and broken there
    (``self.padding_index`` AttributeError at transformers.py:298; never instantiated) -- out of the
    hot-path scope."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        raise NotImplementedError("TransformerEncoder is dead code in the reference and not on the caption path")


class _IncrementalDecoder(_Planned, nn.Module):
    """Shared engine of TransformerDecoder / SelfAttentionTransformerDecoder."""

    _layer_cls = None
    _cross = False

    def __init__(self, num_tokens, hid_dim=512, n_layers=6, n_heads=8,
                 pf_dim=2048, dropout=0., pad_index=None, max_len=128):
        super().__init__()
        self.pad_index = pad_index
        self.tok_embedding = nn.Embedding(num_tokens, hid_dim)
        self.pos_embedding = nn.Embedding(max_len, hid_dim)
        self.dropout = nn.Dropout(dropout)
        self.layers = nn.ModuleList([self._layer_cls(hid_dim, n_heads, pf_dim, dropout) for _ in range(n_layers)])
        self.scale = nn.Parameter(torch.sqrt(torch.tensor(hid_dim, dtype=torch.float32)), requires_grad=False)
        self.classifier = nn.Linear(hid_dim, num_tokens)
        self.num_tokens, self.hid_dim, self.n_heads = num_tokens, hid_dim, n_heads

    # ---- derived constants: fused QKV / KV weights, scalar scales --------------------------------
    def _build_plan(self):
        d = lambda t: t.detach()
        f = lambda t: t.detach().float().contiguous()          # bias / LayerNorm vectors stay fp32
        layers = []
        for lyr in self.layers:
            sa = lyr.self_attn
            ent = dict(
                wqkv=torch.cat([d(sa.fc_q.weight), d(sa.fc_k.weight), d(sa.fc_v.weight)], 0).contiguous(),
                bqkv=torch.cat([f(sa.fc_q.bias), f(sa.fc_k.bias), f(sa.fc_v.bias)], 0).contiguous(),
                wo=d(sa.fc_o.weight), bo=f(sa.fc_o.bias), sa_scale=float(sa.scale),
                ln1=(f(lyr.self_attn_ln.weight), f(lyr.self_attn_ln.bias), lyr.self_attn_ln.eps),
                w1=d(lyr.pf.fc_1.weight), b1=f(lyr.pf.fc_1.bias), w2=d(lyr.pf.fc_2.weight), b2=f(lyr.pf.fc_2.bias),
                ln3=(f(lyr.pf_ln.weight), f(lyr.pf_ln.bias), lyr.pf_ln.eps))
            if self._cross:
                ea = lyr.enc_attn
                ent.update(
                    wq=d(ea.fc_q.weight), bq=f(ea.fc_q.bias),
                    wkv=torch.cat([d(ea.fc_k.weight), d(ea.fc_v.weight)], 0).contiguous(),
                    bkv=torch.cat([f(ea.fc_k.bias), f(ea.fc_v.bias)], 0).contiguous(),
                    weo=d(ea.fc_o.weight), beo=f(ea.fc_o.bias), ea_scale=float(ea.scale),
                    ln2=(f(lyr.enc_attn_ln.weight), f(lyr.enc_attn_ln.bias), lyr.enc_attn_ln.eps))
            layers.append(ent)
        dt = self.classifier.weight.dtype
        if dt in hip.HALF_DTYPES and self.hid_dim % 128 == 0 and self.hid_dim <= 512 and hip.option("deferred_ln"):
            # deferred-LayerNorm chain (dh_linear_ln): the gamma of the LayerNorm in front of a projection folded into its
            # weight, beta into its bias, plus the row sums of the folded (rounded) weight
            def fold(w, b, ln):
                g, be = ln[0], ln[1]
                wf = (w.float() * g[None, :]).to(dt).contiguous()
                return wf, (b + (w.float() * be[None, :]).sum(1)).contiguous(), wf.float().sum(1).contiguous()
            for i, ent in enumerate(layers):
                if i > 0:
                    ent["wqkv_f"], ent["bqkv_f"], ent["cs_qkv"] = fold(ent["wqkv"], ent["bqkv"], layers[i - 1]["ln3"])
                if self._cross:
                    ent["wq_f"], ent["bq_f"], ent["cs_q"] = fold(ent["wq"], ent["bq"], ent["ln1"])
                ent["w1_f"], ent["b1_f"], ent["cs_1"] = fold(ent["w1"], ent["b1"], ent["ln2"] if self._cross else ent["ln1"])
            if torch.cuda.is_available() and self.classifier.weight.is_cuda and hip.option("decode_wreg"):
                # register-stationary decode GEMMs (dh_linear_ln_wreg): fragment-packed copies of the chain's weights, once per plan
                hd, pf = self.hid_dim, self.layers[0].pf.fc_1.out_features
                for i, ent in enumerate(layers):
                    for name, src, n, k, lnx in (("wqkv_pk", "wqkv_f" if i > 0 else "wqkv", 3 * hd, hd, False), ("wo_pk", "wo", hd, hd, True),
                                                 ("weo_pk", "weo", hd, hd, True), ("w1_pk", "w1_f", pf, hd, False), ("w2_pk", "w2", hd, pf, True),
                                                 ("wq_pk", "wq_f", hd, hd, False)):
                        if src in ent and hip.linear_ln_wreg_supported(n, k, lnx):
                            ent[name] = hip.pack_mfma_fragments(ent[src].contiguous())
        plan = dict(layers=layers, tok=d(self.tok_embedding.weight), pos=d(self.pos_embedding.weight),
                    scale=float(self.scale), cls_w=d(self.classifier.weight), cls_b=f(self.classifier.bias),
                    dtype=self.classifier.weight.dtype)
        if (dt == torch.float32 and self.classifier.weight.is_cuda and hip.option("f32_split")
                and all(hip.f32_split_ok(p) for p in self.parameters() if p.dim() > 1)):
            # fp32 weights with option "f32_split": every dense layer as three fp16 MFMAs on split operands (csrc/gemm_f32x.hip)
            for ent in layers:
                for name in ("wqkv", "wo", "w1", "w2", "wq", "weo", "wkv"):
                    if name in ent:
                        ent[name + "_x"] = hip.split_f32x(ent[name].contiguous())
                        if name != "wkv" and hip.option("decode_wreg"):
                            # the decode position's layers with the split weights stationary in registers (csrc/linear_f32x_wreg.hip)
                            pk = hip.pack_f32x_fragments(ent[name + "_x"])
                            if pk is not None:
                                ent[name + "_xp"] = pk
            plan["cls_w_x"] = hip.split_f32x(plan["cls_w"].contiguous())
            # ... and the decode chain's GEMM operands stored split by their producers (csrc/gemm_f32xp.hip, runtime.hip): the scratch then
            # carries plane buffers and the classifier fills the beam sampler's group maxima like the 16-bit paths
            plan["f32_planes"] = bool(hip.option("f32_planes")) and self.hid_dim % 32 == 0
        if plan["dtype"] in hip.HALF_DTYPES and plan["cls_w"].is_cuda and self.hid_dim == 512 and hip.option("vocab_wreg"):
            # the register-streamed classifier (csrc/vocab_wreg.hip; bit-identical).  The LSTM decoder uses it at every size; here it is
            # selected per batch by row count (_Run): at <= "vocab_wreg_transformer_max_rows" rows per position (small shards, single
            # images: the 128-row A-stationary kernel takes 54 us at 380 rows) or always with option "vocab_wreg_transformer" -- at 1,280
            # rows a Transformer position moves ~1 GB of KV cache between two classifier launches, the 37 MB of weights do not survive in
            # the Infinity Cache and the step takes the same time with either kernel (20.4-20.7 ms, three alternating runs)
            plan["cls_w_pk"], plan["cls_b_pad"] = hip.pack_vocab_weights(plan["cls_w"], plan["cls_b"])
        return plan

    def _check_mode(self):
        if self.training and self.dropout.p > 0:
            raise RuntimeError("deephumor_amd implements the inference path; call model.eval()")
        if self.pad_index is None:
            raise TypeError("pad_index=None is unusable in the reference too (transformers.py:451); pass an int")

    def _enc_for_cross(self, enc_out, seq):
        """Encoder rows and their key mask as the reference's forward sees them (transformers.py:450-452, 480-481).
        ``pad_index == 0``: rows with a zero element are masked -- the zero rows the reference pads ``enc_out`` with up to ``seq`` are
        masked too, so they need not exist.  ``pad_index >= 2``: the 0/1 row flags never equal it, NOTHING is masked, and once
        ``seq`` exceeds the encoder length the padded zero rows are real keys (K = V = the projection biases): they are appended."""
        n, s, d = enc_out.shape
        if self.pad_index == 0:
            flat = enc_out.contiguous().view(n * s, d)
            return flat, s, hip.enc_key_mask(flat)
        if seq > s:
            padded = torch.zeros((n, seq, d), dtype=enc_out.dtype, device=enc_out.device)
            padded[:, :s] = enc_out
            enc_out, s = padded, seq
        return enc_out.contiguous().view(n * s, d), s, torch.zeros((n * s,), dtype=torch.uint8, device=enc_out.device)

    class _Run:
        """KV cache + cross-attention operands + scratch for one batch, described to the native step
        driver (``dh_transformer_decode_position``) through plain C structs."""

        def __init__(self, dec, plan, n_img, beam, n_pos, enc_out, dev):
            d, nl = dec.hid_dim, len(dec.layers)
            self.n_img, self.beam, self.rows_total, self.n_pos = n_img, beam, n_img * beam, n_pos
            self.dtype = plan["dtype"]
            self.kc = torch.empty((nl, n_pos, self.rows_total, d), device=dev, dtype=self.dtype)
            self.vc = torch.empty((nl, n_pos, self.rows_total, d), device=dev, dtype=self.dtype)
            self.kv, self.keymask, self.s, self.packed, self.dperm = None, None, 0, None, False
            if enc_out is not None:
                n = enc_out.shape[0]
                flat, s, self.keymask = dec._enc_for_cross(enc_out.to(self.dtype), n_pos)   # transformers.py:450-452, 480-481
                self.s = s
                self.kv = [hip.linear(flat, L["wkv"], L["bkv"], tag="enc_kv", w_x=L.get("wkv_x")) for L in plan["layers"]]   # once per image
                if (self.dtype in hip.HALF_DTYPES and s <= 64 and d == 64 * dec.n_heads and beam <= 16
                        and hip.option("packed_cross")):
                    # matrix-core cross-attention: K | V re-laid out per (image, head) in MFMA operand order, once per batch; on the
                    # deferred-LayerNorm chain with K's head-dim slots in accumulator-tile order (the layout of rounds 2-5's fused
                    # fc_q + attention launch, kept: it fixes the summation order over the head dimension, i.e. every 16-bit token)
                    self.dperm = "wq_f" in plan["layers"][0] and d <= 512
                    self.packed = [hip.attn_cross_pack(kv, n, s, d, dec.n_heads, dperm=self.dperm) for kv in self.kv]
            self._scratch = {}
            self.planes = bool(plan.get("f32_planes"))
            self.pf = dec.layers[0].pf.fc_1.out_features
            self.d, self.dev = d, dev
            P = lambda t: 0 if t is None else t.data_ptr()
            self.c_layers = (hip.TrLayer * nl)()
            for i, L in enumerate(plan["layers"]):
                c = self.c_layers[i]
                c.wqkv, c.wo, c.w1, c.w2 = P(L["wqkv"]), P(L["wo"]), P(L["w1"]), P(L["w2"])
                c.bqkv, c.bo, c.b1, c.b2 = P(L["bqkv"]), P(L["bo"]), P(L["b1"]), P(L["b2"])
                c.ln1_g, c.ln1_b, c.ln1_eps = P(L["ln1"][0]), P(L["ln1"][1]), L["ln1"][2]
                c.ln3_g, c.ln3_b, c.ln3_eps = P(L["ln3"][0]), P(L["ln3"][1]), L["ln3"][2]
                c.sa_scale = L["sa_scale"]
                if dec._cross:
                    c.wq, c.weo, c.bq, c.beo = P(L["wq"]), P(L["weo"]), P(L["bq"]), P(L["beo"])
                    c.ln2_g, c.ln2_b, c.ln2_eps, c.ea_scale = P(L["ln2"][0]), P(L["ln2"][1]), L["ln2"][2], L["ea_scale"]
                    c.kv = P(self.kv[i])
                    if self.packed is not None:
                        c.kp, c.vt, c.kp_dperm = P(self.packed[i][0]), P(self.packed[i][1]), int(self.dperm)
                for name in ("wqkv_f", "wq_f", "w1_f", "bqkv_f", "bq_f", "b1_f", "cs_qkv", "cs_q", "cs_1",
                             "wqkv_pk", "wo_pk", "weo_pk", "w1_pk", "w2_pk", "wq_pk", "wqkv_x", "wo_x", "w1_x", "w2_x", "wq_x", "weo_x",
                             "wqkv_xp", "wo_xp", "w1_xp", "w2_xp", "wq_xp", "weo_xp"):
                    if name in L:
                        setattr(c, name, P(L[name]))
                c.kcache, c.vcache = self.kc[i].data_ptr(), self.vc[i].data_ptr()
            m = self.c_model = hip.TrModel()
            m.n_layers, m.D, m.n_heads, m.pf_dim, m.V = nl, d, dec.n_heads, self.pf, dec.num_tokens
            m.pad_index, m.cross, m.S = dec.pad_index, int(dec._cross), self.s
            m.dtype = {torch.float32: hip.F32, torch.bfloat16: hip.BF16, torch.float16: hip.F16}[self.dtype]
            m.emb_scale = plan["scale"]
            m.layers = self.c_layers
            m.tok_emb, m.pos_emb, m.cls_w, m.cls_b = P(plan["tok"]), P(plan["pos"]), P(plan["cls_w"]), P(plan["cls_b"])
            m.keymask = P(self.keymask)
            if "cls_w_pk" in plan and (hip.option("vocab_wreg_transformer") or self.rows_total <= hip.option("vocab_wreg_transformer_max_rows")):
                m.cls_w_pk, m.cls_b_pad = P(plan["cls_w_pk"]), P(plan["cls_b_pad"])
            if "cls_w_x" in plan:
                m.cls_w_x = P(plan["cls_w_x"])
            # the decoder layers of a position as ONE persistent launch (csrc/decode_layers.hip): a device-resident table of this run's
            # per-layer pointers + the clusters' hand-over words (zero once; the kernel keeps them consistent from launch to launch)
            self.layers_table = self.layers_sync = None
            if hip.option("decode_layers") and self.dtype in hip.HALF_DTYPES and hip.decode_layers_supported(m, 1, 0):
                self.layers_sync = torch.zeros((512,), device=dev, dtype=torch.int32)
                self.layers_table = hip.decode_layers_table(m, dev)
                m.layers_table, m.layers_sync = self.layers_table.data_ptr(), self.layers_sync.data_ptr()

        def scratch(self, rows):
            if rows not in self._scratch:
                e = lambda *shape: torch.empty(shape, device=self.dev, dtype=self.dtype)
                bufs = dict(x=e(rows, self.d), qkv=e(rows, 3 * self.d), att=e(rows, self.d),
                            o=e(rows, self.d), q=e(rows, self.d), ff=e(rows, self.pf))
                if self.dtype in hip.HALF_DTYPES:      # deferred-LayerNorm chain: one more row buffer + partial statistics
                    bufs["y2"] = e(rows, self.d)
                    for k in ("st0", "st1", "st2"):
                        bufs[k] = torch.empty((rows, self.d // 64, 2), device=self.dev, dtype=torch.float32)
                if self.planes:                        # fp32, split operands: x / att / ff as fp16 planes (hi, lo)
                    h = lambda w: torch.empty((2, rows, w), device=self.dev, dtype=torch.float16)
                    bufs.update(xp=h(self.d), attp=h(self.d), ffp=h(self.pf))
                c = hip.TrScratch()
                for k, v in bufs.items():
                    setattr(c, k, v.data_ptr())
                bufs["c"] = c
                self._scratch[rows] = bufs
            return self._scratch[rows]

    def _decode_position(self, plan, run, t, rows, rpi, mult, tokens, src, start_emb, x_out=None, logits=None,
                         group_max=None):
        """Hidden state of position ``t`` for ``rows`` compact rows (all layers) [rows, D]; with ``logits``
        (fp32 [rows, V]) also the classifier -- one native call (``dh_transformer_decode_position``)."""
        sc = run.scratch(rows)
        hip.transformer_decode_position(run.c_model, sc["c"], start_emb, tokens, src, run.n_img, rpi, mult,
                                        run.rows_total, t, x_out=x_out, logits=logits, group_max=group_max)
        return x_out if x_out is not None else sc["x"]

    def _forward(self, x, enc_out, start_emb, num_positions=None, return_hidden=False):
        check_ids(x, self.tok_embedding.num_embeddings)                               # (nn.Embedding's IndexError)
        if start_emb is None or self.pad_index == 1:
            return self._forward_modules(x, enc_out, start_emb, return_hidden=return_hidden)
        self._check_mode()
        plan = self._get_plan()
        bs, dec_len = x.shape
        dev = start_emb.device
        dec_len += 1
        seq = dec_len if enc_out is None else max(dec_len, enc_out.shape[1])          # transformers.py:450
        if seq > self.pos_embedding.num_embeddings:
            raise IndexError("index out of range in self")                            # pos-embedding lookup
        if num_positions is not None and self.pad_index == 0:      # causal: positions >= num_positions cannot influence the rest
            seq = max(1, min(seq, int(num_positions)))             # (pad_index >= 2: the padded encoder rows depend on the full length)
        tokens = torch.full((bs, max(seq - 1, 1)), self.pad_index, dtype=torch.int32, device=dev)
        ncopy = min(x.shape[1], tokens.shape[1])
        tokens[:, :ncopy] = x[:, :ncopy].to(torch.int32)
        if self._prefill_ok(plan, seq):
            return self._forward_prefill(plan, tokens, enc_out, start_emb.to(plan["dtype"]).contiguous(), bs, seq, return_hidden)
        helper_src = (torch.arange(bs, dtype=torch.int32, device=dev))[:, None].expand(bs, seq).contiguous()
        run = self._Run(self, plan, bs, 1, seq, enc_out, dev)
        hs = torch.empty((bs, seq, self.hid_dim), device=dev, dtype=plan["dtype"])
        xt = torch.empty((bs, self.hid_dim), device=dev, dtype=plan["dtype"])
        for t in range(seq):
            self._decode_position(plan, run, t, bs, 1, 1, tokens, helper_src, start_emb.to(plan["dtype"]).contiguous(), x_out=xt)
            hs[:, t, :].copy_(xt)
        if return_hidden:
            return hs
        out = hip.linear(hs.view(bs * seq, self.hid_dim), plan["cls_w"], plan["cls_b"], out_dtype=torch.float32, tag="vocab", w_x=plan.get("cls_w_x"))
        return out.view(bs, seq, -1)

    def _forward_modules(self, x, enc_out, start_emb=None, return_hidden=False):
        """The reference's own formulation of ``forward`` on the module-API layers: pad ``x`` / ``enc_out`` to a common length
        (transformers.py:450-452), embeddings (:455-469), pad | causal input mask over ``[1 | x]`` (:473-477), encoder-row mask
        (:480-481), layers, classifier.  Two callers: ``forward`` WITHOUT a start embedding (:432 default ``start_emb=None``; never
        used by the captioning models), and ``pad_index == 1``.  With pad_index 1 the image slot's stand-in id 1 (:474) counts as
        padding: position 0 then has EVERY key masked and its softmax spreads uniformly over all ``seq_len`` positions, future
        <pad> rows included (:110-114), and in the encoder attention the real patch rows (flag 1) are the masked ones -- a position
        no longer depends only on positions <= it, so the KV-cached engine does not apply and every decode step re-runs the whole
        sequence, exactly as the reference does (``_generate_reforward``)."""
        self._check_mode()
        bs, dec_len = x.shape
        dt = self.classifier.weight.dtype
        has_start = start_emb is not None
        dec_len += int(has_start)
        seq = dec_len if enc_out is None else max(dec_len, enc_out.shape[1])
        if seq > self.pos_embedding.num_embeddings:
            raise IndexError("index out of range in self")
        toks = torch.full((bs, seq - int(has_start)), self.pad_index, dtype=torch.int64, device=x.device)
        toks[:, :x.shape[1]] = x
        ids = torch.cat([torch.ones((bs, 1), dtype=torch.int64, device=x.device), toks], 1) if has_start else toks   # :474
        emb = hip.embed_prefill(self.tok_embedding.weight.detach(), self.pos_embedding.weight.detach(),
                                start_emb.to(dt).contiguous() if has_start else None, toks.to(torch.int32).contiguous(), bs, seq,
                                float(self.scale)).view(bs, seq, self.hid_dim)
        input_mask = hip.mask_or(get_pad_mask(ids, ids, pad_index=self.pad_index), get_autoregressive_mask(ids))
        h = emb
        if enc_out is not None:
            enc = torch.zeros((bs, seq, self.hid_dim), dtype=dt, device=x.device)
            enc[:, :enc_out.shape[1]] = enc_out.to(dt)
            enc_mask = get_pad_mask(ids, hip.enc_nonzero_rows(enc), pad_index=self.pad_index)
            for layer in self.layers:
                h = layer(h, enc, input_mask=input_mask, enc_mask=enc_mask)
        else:
            for layer in self.layers:
                h = layer(h, input_mask=input_mask)
        if return_hidden:
            return h
        out = hip.linear(h.reshape(bs * seq, self.hid_dim), self.classifier.weight.detach(), _fp32(self.classifier.bias),
                         out_dtype=torch.float32, tag="vocab")
        return out.view(bs, seq, -1)

    def _generate_reforward(self, start_emb, enc_out, caption, max_len, temperature, beam_size, top_k, eos_index, seed, img0,
                            noise_source, logits_hook, rng, rng_seed, exact):
        """``generate`` by the reference's own algorithm (transformers.py:521-577): the whole padded sequence of every beam row is
        re-run for each token on the module-API layers.  Only ``pad_index == 1`` needs it (see ``_forward_modules``); the beam
        bookkeeping is the batched engine's."""
        n, b, dev = start_emb.shape[0], beam_size, start_emb.device
        helper = BeamSearchHelper(temperature, beam_size, top_k, eos_index=eos_index, device=dev, n_img=n, max_len=max_len,
                                  src_len=max_len + 1, seed=seed, img0=img0,
                                  noise_source=make_noise_source(rng, rng_seed, noise_source, 0, n, img0), exact=exact)
        helper.tokens.fill_(self.pad_index)
        pos = 0
        if caption is not None:
            pos = caption.shape[1]
            helper.set_prefix(caption)
        cls_w, cls_b = self.classifier.weight.detach(), _fp32(self.classifier.bias)

        def logits_at(tokens, semb, enc, t):
            h = self._forward_modules(tokens.long(), enc, semb, return_hidden=True)
            return hip.linear(h[:, t].contiguous(), cls_w, cls_b, out_dtype=torch.float32, tag="vocab")

        lg = logits_at(helper.tokens[::b], start_emb, enc_out, pos)
        if logits_hook is not None:
            call_logits_hook(logits_hook, pos, lg, helper)
        helper.step(lg, first=True, write_pos=pos, t=pos, step_index=pos, first_sets_ended=False)
        semb = start_emb.repeat_interleave(b, 0)
        enc = None if enc_out is None else enc_out.repeat_interleave(b, 0)
        for i in range(pos + 1, max_len + 1):
            lg = logits_at(helper.tokens, semb, enc, i)
            if logits_hook is not None:
                call_logits_hook(logits_hook, i, lg, helper)
            helper.step(lg, first=False, write_pos=i, t=i, step_index=i)
        return helper.finalize(len_bias_done=0, full_len=max_len, pad_index=self.pad_index)

    def _prefill_ok(self, plan, seq):
        """All positions at once (batched GEMMs, one causal-attention launch per layer) when the attention kernels'
        register-resident history covers the sequence; otherwise position by position on the decode engine."""
        limit = 56 if plan["dtype"] in hip.HALF_DTYPES else 40
        return self.hid_dim == 64 * self.n_heads and seq <= limit

    def _forward_prefill(self, plan, tokens, enc_out, start_emb, bs, seq, return_hidden=False):
        """Teacher-forced forward in prefill form: rows are sequence-major (row n*seq + t); per layer one QKV GEMM
        over all bs*seq rows, one causal self-attention launch, projection + residual LayerNorm, (cross-attention in
        chunks of positions), FFN -- ~12 launches per layer instead of ~11 per layer AND position."""
        d, nh, dt = self.hid_dim, self.n_heads, plan["dtype"]
        x = hip.embed_prefill(plan["tok"], plan["pos"], start_emb, tokens, bs, seq, plan["scale"])
        kv = keymask = None
        s_enc = 0
        if enc_out is not None:
            flat, s_enc, keymask = self._enc_for_cross(enc_out.to(dt), seq)            # transformers.py:450-452, 480-481
        packed_ok = dt in hip.HALF_DTYPES and 0 < s_enc <= 64

        def cross(q, L):
            kv = hip.linear(flat, L["wkv"], L["bkv"], tag="enc_kv", w_x=L.get("wkv_x"))
            if packed_ok:                                  # matrix-core cross-attention, 16 positions per launch
                # (dperm: the head-dim slot order of the decode chain's fused fc_q + attention launch, so that teacher-forced
                # logits and incremental decoding sum in the same order)
                dperm = "wq_f" in L
                kp, vt = hip.attn_cross_pack(kv, bs, s_enc, d, nh, dperm=dperm)
                return hip.attn_cross_prefill_packed(q, kp, vt, keymask, bs, seq, s_enc, d, nh, L["ea_scale"], dperm=dperm)
            return hip.attn_cross_prefill(q, kv, keymask, bs, seq, s_enc, d, nh, L["ea_scale"])

        if "w1_f" in plan["layers"][0]:
            # 16-bit dtypes: the deferred-LayerNorm chain of dh_transformer_decode_position (csrc/runtime.hip), all positions
            # at once -- the same arithmetic per row, so prefill and incremental decoding agree
            st = pend = None                               # statistics of x / (gamma, beta, eps) of the LayerNorm pending on x
            for L in plan["layers"]:
                if pend is None:
                    qkv = hip.linear_ln(x, L["wqkv"], L["bqkv"], tag="qkv")
                else:
                    qkv = hip.linear_ln(x, L["wqkv_f"], L["bqkv_f"], a_ln=(st, pend[2], L["cs_qkv"]), tag="qkv")
                att = hip.attn_self_prefill(qkv, tokens, bs, seq, d, nh, L["sa_scale"], self.pad_index)
                y, sy = hip.linear_ln(att, L["wo"], L["bo"], residual=x, want_stats=True, tag="proj",
                                      r_ln=None if pend is None else (st, pend[2], pend[0], pend[1]))
                ln_in = L["ln1"]
                if self._cross:
                    q = hip.linear_ln(y, L["wq_f"], L["bq_f"], a_ln=(sy, L["ln1"][2], L["cs_q"]), tag="proj")
                    att = cross(q, L)
                    y, sy = hip.linear_ln(att, L["weo"], L["beo"], residual=y, want_stats=True, tag="proj",
                                          r_ln=(sy, L["ln1"][2], L["ln1"][0], L["ln1"][1]))
                    ln_in = L["ln2"]
                ff = hip.linear_ln(y, L["w1_f"], L["b1_f"], relu=True, a_ln=(sy, ln_in[2], L["cs_1"]), tag="ffn")
                x, st = hip.linear_ln(ff, L["w2"], L["b2"], residual=y, want_stats=True, tag="ffn",
                                      r_ln=(sy, ln_in[2], ln_in[0], ln_in[1]))
                pend = L["ln3"]
            x = hip.add_layernorm(x, None, pend[0], pend[1], eps=pend[2])
            if return_hidden:
                return x.view(bs, seq, d)
            out = hip.linear(x, plan["cls_w"], plan["cls_b"], out_dtype=torch.float32, tag="vocab", w_x=plan.get("cls_w_x"))
            return out.view(bs, seq, -1)
        for L in plan["layers"]:
            qkv = hip.linear(x, L["wqkv"], L["bqkv"], tag="qkv", w_x=L.get("wqkv_x"))
            att = hip.attn_self_prefill(qkv, tokens, bs, seq, d, nh, L["sa_scale"], self.pad_index)
            o = hip.linear(att, L["wo"], L["bo"], tag="proj", w_x=L.get("wo_x"))
            x = hip.add_layernorm(x, o, L["ln1"][0], L["ln1"][1], eps=L["ln1"][2])
            if self._cross:
                q = hip.linear(x, L["wq"], L["bq"], tag="proj", w_x=L.get("wq_x"))
                att = cross(q, L)
                o = hip.linear(att, L["weo"], L["beo"], tag="proj", w_x=L.get("weo_x"))
                x = hip.add_layernorm(x, o, L["ln2"][0], L["ln2"][1], eps=L["ln2"][2])
            ff = hip.linear(x, L["w1"], L["b1"], relu=True, tag="ffn", w_x=L.get("w1_x"))
            o = hip.linear(ff, L["w2"], L["b2"], tag="ffn", w_x=L.get("w2_x"))
            x = hip.add_layernorm(x, o, L["ln3"][0], L["ln3"][1], eps=L["ln3"][2])
        if return_hidden:
            return x.view(bs, seq, d)
        out = hip.linear(x, plan["cls_w"], plan["cls_b"], out_dtype=torch.float32, tag="vocab", w_x=plan.get("cls_w_x"))
        return out.view(bs, seq, -1)

    def _generate_batch(self, start_emb, enc_out, caption, max_len, temperature, beam_size, top_k, eos_index,
                        seed=None, img0=0, noise_source=None, logits_hook=None, streams=1, seed_tensor=None,
                        defer_check=False, early_stop_every=0, exact=False, rng=None):
        self._check_mode()
        plan = self._get_plan()
        classifier_must_be_finite(plan)
        check_ids(caption, self.tok_embedding.num_embeddings)
        rng_seed = seed                   # rng="torch": the draws replay torch CPU generators (beam.TorchRngNoise)
        seed = 0 if rng == "torch" else resolve_seed(seed, noise_source)
        # rng="torch" with seed=None draws from torch's DEFAULT generator: its state is snapshotted once per call so that a repeated
        # session (BeamOverflow retry) replays the same draws (beam.TorchRngNoise)
        rng_state0 = torch.get_rng_state() if (rng == "torch" and rng_seed is None) else None
        if max_len + 1 > self.pos_embedding.num_embeddings:
            raise IndexError("index out of range in self")    # reference: pos_embedding lookup, SURVEY.md section 5
        start_emb = start_emb.to(plan["dtype"]).contiguous()
        if self.pad_index == 1:
            if seed_tensor is not None or defer_check:
                raise NotImplementedError("pad_index == 1 decodes by full re-forward on the host-driven module path: no hipGraph capture")
            try:
                return self._generate_reforward(start_emb, enc_out, caption, max_len, temperature, beam_size, top_k, eos_index, seed,
                                                img0, noise_source, logits_hook, rng, rng_seed, bool(exact))
            except BeamOverflow:
                if exact:
                    raise
                warn_overflow_retry()
                if rng_state0 is not None:
                    torch.set_rng_state(rng_state0)
                return self._generate_reforward(start_emb, enc_out, caption, max_len, temperature, beam_size, top_k, eos_index, seed,
                                                img0, noise_source, logits_hook, rng, rng_seed, True)

        def session(lo, hi):
            """Decodes images [lo, hi); yields after every position (see ``run_interleaved``)."""
            n, b = hi - lo, beam_size
            r = n * b
            dev = start_emb.device
            helper = BeamSearchHelper(temperature, beam_size, top_k, eos_index=eos_index, device=dev, n_img=n,
                                      max_len=max_len, src_len=max_len + 1, seed=seed, img0=img0 + lo,
                                      noise_source=make_noise_source(rng, rng_seed, noise_source, lo, hi, img0, rng_state0),
                                      seed_tensor=seed_tensor, exact=exact[0])
            if self.pad_index != 0:
                helper.tokens.fill_(self.pad_index)
            pos = 0
            if caption is not None:
                pos = caption.shape[1]
                helper.set_prefix(caption[lo:hi])
            run = self._Run(self, plan, n, b, max_len + 1, None if enc_out is None else enc_out[lo:hi], dev)
            semb = start_emb[lo:hi]
            # logits always fp32; row stride padded to 64 floats so rows are 16-byte aligned (vector stores)
            logits = torch.empty((r, (self.num_tokens + 255) // 256 * 256), device=dev)[:, :self.num_tokens]   # whole 256-column chunks (vocab_wreg)
            gmax = (torch.empty((r, 4 * ((self.num_tokens + 255) // 256)), device=dev)[:, :hip.n_groups(self.num_tokens)]
                    if plan["dtype"] in hip.HALF_DTYPES or plan.get("f32_planes") else None)    # column-group maxima (16-bit paths, f32x planes)
            gm = None if gmax is None else gmax[:n]
            # positions 0..pos with ONE row per image (logical row img*beam), sampling at `pos`
            lg = logits[:n]
            for t in range(pos + 1):
                self._decode_position(plan, run, t, n, 1, b, helper.tokens, helper.src, semb,
                                      logits=lg if t == pos else None, group_max=gm if t == pos else None)
                yield
            if logits_hook is not None:
                call_logits_hook(logits_hook, pos, lg, helper)
            helper.step(lg, first=True, write_pos=pos, t=pos, step_index=pos, first_sets_ended=False, group_max=gm)
            for i in range(pos + 1, max_len + 1):
                self._decode_position(plan, run, i, r, b, 1, helper.tokens, helper.src, semb, logits=logits,
                                      group_max=gmax)
                if logits_hook is not None:
                    call_logits_hook(logits_hook, i, logits, helper)
                # at i == max_len nothing is written (transformers.py:557) but beams are still re-drawn
                helper.step(logits, first=False, write_pos=i, t=i, step_index=i, group_max=gmax)
                yield
                if early_stop_every and (i - pos) % early_stop_every == 0 and bool(helper.done.all()):
                    break                                   # all_ended() break of the reference (transformers.py:585)
            out = helper.finalize(len_bias_done=0, full_len=max_len, pad_index=self.pad_index, defer_check=defer_check)
            if run.layers_sync is not None and not defer_check and int(run.layers_sync[320]) != 0:
                # a hand-over of the persistent layer kernel timed out (fewer than 256 resident workgroups?): its results are undefined
                hip.set_option("decode_layers", 0)
                raise RuntimeError("deephumor_amd: the persistent decoder-layer kernel (option decode_layers) timed out waiting for its "
                                   "workgroups; the option has been switched off for this process -- repeat the call")
            return out

        exact = [bool(exact)]
        try:
            return run_interleaved(session, start_emb.shape[0], streams)
        except BeamOverflow:              # flat logits (see LSTMDecoder.generate_batch): once more through the general sampler
            if exact[0]:
                raise
            exact[0] = True
            warn_overflow_retry()
            return run_interleaved(session, start_emb.shape[0], streams)


class TransformerDecoder(_IncrementalDecoder):
    """Multi-layer Transformer decoder with encoder attention (reference transformers.py:380-579)."""

    _layer_cls = DecoderLayer
    _cross = True

    @f32x_guarded
    def forward(self, x, enc_out, start_emb=None, *, num_positions=None):
        """Teacher-forced logits ``[bs, max(len(x)+1, S), num_tokens]`` (transformers.py:432-490).  The reference pads
        the decoder input up to the number of image patches (:450), so a 32-token caption costs 49 positions;
        ``num_positions`` (not in the reference) returns only the first ``num_positions`` of them -- identical values,
        the mask is causal -- for callers such as the perplexity scorer that never look further."""
        return self._forward(x, enc_out, start_emb, num_positions)

    @f32x_guarded
    def generate_batch(self, start_emb, enc_out, caption=None, max_len=25, temperature=1.0, beam_size=10,
                       top_k=50, eos_index=3, **kw):
        """``start_emb [N, D]``, ``enc_out [N, S, D]`` -> ``(tokens [N, max_len], lengths [N])``."""
        return self._generate_batch(start_emb, enc_out, caption, max_len, temperature, beam_size, top_k,
                                    eos_index, **kw)

    def generate(self, start_emb, enc_out, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        """Reference single-image API (transformers.py:492-493) -> 1-D (0-D for one token) int64."""
        toks, lens = self._generate_batch(start_emb, enc_out, caption, max_len, temperature, beam_size, top_k,
                                          eos_index, **kw)
        return toks[0, :int(lens[0])].squeeze()


class SelfAttentionTransformerDecoder(_IncrementalDecoder):
    """Transformer decoder without encoder attention (reference transformers.py:639-825)."""

    _layer_cls = SelfAttentionDecoderLayer
    _cross = False

    @f32x_guarded
    def forward(self, x, start_emb):
        """Teacher-forced logits ``[bs, len(x)+1, num_tokens]`` (transformers.py:694-738)."""
        return self._forward(x, None, start_emb)

    @f32x_guarded
    def generate_batch(self, start_emb, caption=None, max_len=25, temperature=1.0, beam_size=10,
                       top_k=50, eos_index=3, **kw):
        return self._generate_batch(start_emb, None, caption, max_len, temperature, beam_size, top_k,
                                    eos_index, **kw)

    def generate(self, start_emb, caption=None, max_len=25,
                 temperature=1.0, beam_size=10, top_k=50, eos_index=3, **kw):
        """Reference single-image API (transformers.py:740-741)."""
        toks, lens = self._generate_batch(start_emb, None, caption, max_len, temperature, beam_size, top_k,
                                          eos_index, **kw)
        return toks[0, :int(lens[0])].squeeze()
