// Native per-position drivers of the two decoders: ONE C call launches every kernel of a decode
// position (embedding -> L x [QKV GEMM, self-attention, projection, add+LayerNorm, (cross-attention
// block), FFN, add+LayerNorm] -> vocabulary GEMM), or of an LSTM time step, on the caller's stream.
// They only sequence the C-ABI entry points above them -- no allocation, no synchronisation -- so the
// host cost of a 6-layer decode position is one FFI crossing instead of ~70.
#include <stdlib.h>
#include "common.h"
#include "prof.h"
#include "options.h"

// Kernel selection here goes through the option table (options.h; dh_set_option): every use reads the current value.

static int cross_attention(const dh_tr_model_t* m, const dh_tr_layer_t& L, const void* q, void* att, int n_img, int rows_per_img,
                           int dt, void* stream) {
    if (L.kp && L.vt && DH_IS_16BIT(dt) && m->S <= 64 && m->D == 64 * m->n_heads && rows_per_img <= 16)
        return dh_attn_cross_decode_packed(q, m->D, L.kp, L.vt, m->keymask, att, n_img, rows_per_img, m->S, m->D, m->n_heads,
                                           L.ea_scale, L.kp_dperm, dt, stream);
    return dh_attn_cross_decode(q, m->D, L.kv, m->keymask, att, n_img, rows_per_img, m->S, m->D, m->n_heads, L.ea_scale, dt, stream);
}

// The classifier of a beam-search step with group maxima: the register-streamed kernel (csrc/vocab_wreg.hip; bit-identical) when the model
// carries its operands and it takes the shape, else dh_vocab_logits.  Option "vocab_wreg" = 0 switches it off (A/B runs).
static int classifier_groups(const void* A, int lda, const void* W, const float* bias, const void* W_pk, const float* bias_pad, float* logits,
                             int ldl, float* group_max, int gm_ld, int rows, int V, int K, int dt, void* stream) {
    if (dh_opt(DH_OPT_VOCAB_WREG) && W_pk && bias_pad && logits && dh_vocab_logits_wreg_supported(rows, V, K, ldl, gm_ld))
        return dh_vocab_logits_wreg(A, lda, W_pk, bias_pad, logits, ldl, group_max, gm_ld, rows, V, K, dt, stream);
    if (logits && K == 512 && rows >= 768 && (rows % 256) != 0 && dh_opt(DH_OPT_VOCAB_AREG)) {
        // a row count that is no multiple of 256 (C5: 300 templates x beam 10 = 3,000 rows) would take the 128 x 128 tile kernel for
        // ALL rows (211 us at 3,000 rows): the whole 256-row tiles go through the A-stationary 256-row kernel, the remainder through
        // the 128-row one -- rows are independent and every classifier kernel is bit-identical to dh_linear, so are the results
        const int head = rows / 256 * 256;
        DH_TRY(dh_vocab_logits(A, lda, W, K, bias, logits, ldl, group_max, gm_ld, head, V, K, dt, stream));
        return dh_vocab_logits((const char*)A + (size_t)head * lda * 2, lda, W, K, bias, logits + (size_t)head * ldl, ldl,
                               group_max + (size_t)head * gm_ld, gm_ld, rows - head, V, K, dt, stream);
    }
    return dh_vocab_logits(A, lda, W, K, bias, logits, ldl, group_max, gm_ld, rows, V, K, dt, stream);
}

// One GEMM of the deferred-LayerNorm chain: the register-stationary kernel (csrc/linear_wreg.hip; bit-identical results) when the
// layer carries fragment-packed weights, the position has enough rows to fill the chip and the shape is one it takes; else the
// tile kernels.  Option "decode_wreg" = 0 switches it off (A/B runs), "decode_wreg_min_rows" moves the threshold.
static int chain_linear(const void* A, int lda, const void* W, const void* W_pk, const float* bias, const void* res, int ldres, void* C,
                        int ldc, int rows, int N, int K, int relu, const dh_ln_fold_t* f, int dt, void* stream) {
    if (dh_opt(DH_OPT_DECODE_WREG) && W_pk && rows >= dh_opt(DH_OPT_DECODE_WREG_MIN_ROWS) && (res == nullptr) == (f->o_stats == nullptr) &&
        dh_linear_ln_wreg_occupancy(rows, N, K, res != nullptr) >= 0.85)
        return dh_linear_ln_wreg(A, lda, W_pk, bias, res, ldres, C, ldc, rows, N, K, relu, f, dt, stream);
    return dh_linear_ln(A, lda, W, K, bias, res, ldres, C, ldc, rows, N, K, relu, f, dt, stream);
}

// One dense layer of the plain (non-deferred) chains: fp32 models with split planes (option "f32_split") run it as three fp16
// MFMAs on split operands (gemm_f32x.hip), everything else through dh_linear.
static int plain_linear(const void* A, int lda, const void* W, const void* W_x, const void* W_xp, int ldw, const float* bias, void* C, int ldc,
                        int rows, int N, int K, int relu, int dt, void* stream) {
    if (dt == DH_F32 && W_x && dh_opt(DH_OPT_F32_SPLIT) && (lda % 4) == 0 && (K % 4) == 0 && ((uintptr_t)A % 16) == 0) {
        // a decode position's rows: the weights stationary in registers (linear_f32x_wreg.hip; bit-identical to the tile kernels)
        if (W_xp && bias && dh_opt(DH_OPT_DECODE_WREG) && (ldc % 4) == 0 && ((uintptr_t)C % 16) == 0 && ((uintptr_t)bias % 16) == 0 &&
            dh_linear_f32x_wreg_supported(rows, N, K))
            return dh_linear_f32x_wreg((const float*)A, lda, W_xp, bias, nullptr, 0, (float*)C, ldc, rows, N, K, relu, stream);
        return dh_linear_f32x((const float*)A, lda, W_x, (K + 31) / 32 * 32, bias, nullptr, nullptr, nullptr, 0, (float*)C, ldc, rows, N, K, relu, stream);
    }
    return dh_linear(A, lda, W, ldw, bias, nullptr, nullptr, nullptr, 0, C, ldc, rows, N, K, relu, dt, stream);
}

// The decode position on the deferred-LayerNorm chain (16-bit dtypes): 8 launches per layer instead of 11.  The residual
// stream is kept PRE-LayerNorm (buffers X = sc->x, Y1 = sc->o, Y2 = sc->y2 with partial statistics st0 / st1 / st2); every
// LayerNorm is applied where its output is consumed: folded into the next projection (gamma in the weight, beta in the bias,
// mean / rstd on the accumulators) and onto the residual operand of the next output projection (transformers.py:356-375).
static int decode_position_deferred(const dh_tr_model_t* m, const dh_tr_scratch_t* sc, const int32_t* tokens, int tok_ld,
                                    const int32_t* src, int src_ld, int n_img, int rows_per_img, int row_mult, int rows_total,
                                    int t, void* x_final, void* stream) {
    const int rows = n_img * rows_per_img, D = m->D, PF = m->pf_dim, dt = m->dtype, nt = D / 64;
    // Option "decode_layers": every layer of the position as ONE persistent launch (csrc/decode_layers.hip; clusters of 8 workgroups own
    // 40 rows through all layers; bit-identical to the launches below)
    const bool layers = dh_opt(DH_OPT_DECODE_LAYERS) && m->layers_table && m->layers_sync && dh_opt(DH_OPT_DECODE_WREG) &&
                        dh_decode_layers_supported(m, rows_per_img, t);
    if (layers) {
        DH_TRY(dh_decode_layers(m, sc, m->layers_table, tokens, tok_ld, src, src_ld, n_img, rows_per_img, row_mult, rows_total, t, m->layers_sync, stream));
    }
    for (int l = 0; l < m->n_layers && !layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        const dh_tr_layer_t* P = l > 0 ? &m->layers[l - 1] : nullptr;       // its LN3 is pending on X
        dh_ln_fold_t f{};
        // 1. qkv = LN3_prev(X) Wqkv^T + b  (layer 0: X is the embedding, no LayerNorm in front), then the self-attention over the
        //    row's history
        if (P) { f.a_stats = sc->st0; f.a_tiles = nt; f.a_eps = P->ln3_eps; f.a_colsum = L.cs_qkv; }
        dh_prof_set_tag("qkv");
        DH_TRY(chain_linear(sc->x, D, P ? L.wqkv_f : L.wqkv, L.wqkv_pk, P ? L.bqkv_f : L.bqkv, nullptr, 0, sc->qkv, 3 * D, rows, 3 * D, D, 0, &f, dt, stream));
        DH_TRY(dh_attn_self_decode(sc->qkv, L.kcache, L.vcache, src, src_ld, tokens, tok_ld, sc->att, n_img, rows_per_img,
                                   row_mult, rows_total, t, D, m->n_heads, L.sa_scale, m->pad_index, dt, stream));
        // 2. Y1 = LN3_prev(X) + att Wo^T + bo, statistics of Y1 -> st1
        f = dh_ln_fold_t{};
        if (P) { f.r_stats = sc->st0; f.r_tiles = nt; f.r_eps = P->ln3_eps; f.r_gamma = P->ln3_g; f.r_beta = P->ln3_b; }
        f.o_stats = sc->st1;
        dh_prof_set_tag("proj");
        DH_TRY(chain_linear(sc->att, D, L.wo, L.wo_pk, L.bo, sc->x, D, sc->o, D, rows, D, D, 0, &f, dt, stream));
        const void* yin = sc->o; const float* st_in = sc->st1;               // rows entering the FFN block, LayerNorm pending
        const float *g_in = L.ln1_g, *b_in = L.ln1_b; float eps_in = L.ln1_eps;
        if (m->cross) {
            // 3. q = LN1(Y1) Wq^T + bq (its own register-stationary GEMM), then the attention over the image's patches on the packed
            //    K / V^T tiles (the fused fc_q + attention launch of rounds 2-5 took the same time per step and was removed in round 6)
            f = dh_ln_fold_t{};
            f.a_stats = sc->st1; f.a_tiles = nt; f.a_eps = L.ln1_eps; f.a_colsum = L.cs_q;
            dh_prof_set_tag("proj");
            DH_TRY(chain_linear(sc->o, D, L.wq_f, L.wq_pk, L.bq_f, nullptr, 0, sc->q, D, rows, D, D, 0, &f, dt, stream));
            DH_TRY(cross_attention(m, L, sc->q, sc->att, n_img, rows_per_img, dt, stream));
            // 4. Y2 = LN1(Y1) + att Weo^T + beo, statistics -> st2
            f = dh_ln_fold_t{};
            f.r_stats = sc->st1; f.r_tiles = nt; f.r_eps = L.ln1_eps; f.r_gamma = L.ln1_g; f.r_beta = L.ln1_b; f.o_stats = sc->st2;
            dh_prof_set_tag("proj");
            DH_TRY(chain_linear(sc->att, D, L.weo, L.weo_pk, L.beo, sc->o, D, sc->y2, D, rows, D, D, 0, &f, dt, stream));
            yin = sc->y2; st_in = sc->st2; g_in = L.ln2_g; b_in = L.ln2_b; eps_in = L.ln2_eps;
        }
        // 5. ff = relu(LN(Yin) W1^T + b1)
        f = dh_ln_fold_t{};
        f.a_stats = st_in; f.a_tiles = nt; f.a_eps = eps_in; f.a_colsum = L.cs_1;
        dh_prof_set_tag("ffn");
        DH_TRY(chain_linear(yin, D, L.w1_f, L.w1_pk, L.b1_f, nullptr, 0, sc->ff, PF, rows, PF, D, 1, &f, dt, stream));
        // 6. X = LN(Yin) + ff W2^T + b2, statistics -> st0  (LN3 of this layer now pending on X)
        f = dh_ln_fold_t{};
        f.r_stats = st_in; f.r_tiles = nt; f.r_eps = eps_in; f.r_gamma = g_in; f.r_beta = b_in; f.o_stats = sc->st0;
        dh_prof_set_tag("ffn");
        DH_TRY(chain_linear(sc->ff, PF, L.w2, L.w2_pk, L.b2, yin, D, sc->x, D, rows, D, PF, 0, &f, dt, stream));
    }
    const dh_tr_layer_t& Z = m->layers[m->n_layers - 1];
    return dh_add_layernorm(sc->x, nullptr, Z.ln3_g, Z.ln3_b, x_final, rows, D, Z.ln3_eps, dt, stream);
}

// The decode position of an fp32 model on the split-operand path with every GEMM operand STORED split (options "f32_split" +
// "f32_planes"; round 6).  The producers -- LayerNorm, the two attentions, relu(fc_1) -- write the fp16 planes their consumer would make
// of an fp32 tensor (the same numbers), so dh_linear_f32xp_wreg starts its MFMAs straight after the LDS-DMA: no split pass, and the
// classifier (dh_linear_f32xp) hands the beam sampler its 64-column group maxima like the 16-bit paths.  Same arithmetic, same
// results as the fp32-activation launches below it (tests/test_f32x_gpu.py).
static bool f32xp_ready(const dh_tr_model_t* m, const dh_tr_scratch_t* sc, int rows) {
    if (m->dtype != DH_F32 || !dh_opt(DH_OPT_F32_SPLIT) || !dh_opt(DH_OPT_F32_PLANES) || !dh_opt(DH_OPT_DECODE_WREG)) return false;
    if (!sc->xp || !sc->attp || !sc->ffp || !m->cls_w_x || (m->D % 32) != 0) return false;
    const int D = m->D, PF = m->pf_dim;
    if (!dh_linear_f32x_wreg_supported(rows, 3 * D, D) || !dh_linear_f32x_wreg_supported(rows, D, D) ||
        !dh_linear_f32x_wreg_supported(rows, PF, D) || !dh_linear_f32x_wreg_supported(rows, D, PF)) return false;
    for (int l = 0; l < m->n_layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        if (!L.wqkv_xp || !L.wo_xp || !L.w1_xp || !L.w2_xp || (m->cross && (!L.wq_xp || !L.weo_xp))) return false;
    }
    return true;
}

static int decode_position_f32xp(const dh_tr_model_t* m, const dh_tr_scratch_t* sc, const int32_t* tokens, int tok_ld, const int32_t* src,
                                 int src_ld, int n_img, int rows_per_img, int row_mult, int rows_total, int t, void* x_out, float* logits,
                                 int ldl, float* group_max, int gm_ld, void* stream) {
    const int rows = n_img * rows_per_img, D = m->D, PF = m->pf_dim;
    float *x = (float*)sc->x, *o = (float*)sc->o;
    for (int l = 0; l < m->n_layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        dh_prof_set_tag("qkv");
        if (l == 0) {      // the embedded rows are fp32 (dh_embed_rows): the one GEMM of the position that splits its operand itself
            DH_TRY(dh_linear_f32x_wreg(x, D, L.wqkv_xp, L.bqkv, nullptr, 0, (float*)sc->qkv, 3 * D, rows, 3 * D, D, 0, stream));
        } else {
            DH_TRY(dh_linear_f32xp_wreg(sc->xp, L.wqkv_xp, L.bqkv, nullptr, 0, (float*)sc->qkv, 3 * D, nullptr, rows, 3 * D, D, 0, stream));
        }
        DH_TRY(dh_attn_self_decode(sc->qkv, L.kcache, L.vcache, src, src_ld, tokens, tok_ld, sc->attp, n_img, rows_per_img,
                                   row_mult, rows_total, t, D, m->n_heads, L.sa_scale, m->pad_index, DH_F32_OUT_PLANES, stream));
        dh_prof_set_tag("proj");
        DH_TRY(dh_linear_f32xp_wreg(sc->attp, L.wo_xp, L.bo, nullptr, 0, o, D, nullptr, rows, D, D, 0, stream));
        DH_TRY(dh_add_layernorm_f32x(x, o, L.ln1_g, L.ln1_b, x, sc->xp, rows, D, L.ln1_eps, stream));
        if (m->cross) {
            dh_prof_set_tag("proj");
            DH_TRY(dh_linear_f32xp_wreg(sc->xp, L.wq_xp, L.bq, nullptr, 0, (float*)sc->q, D, nullptr, rows, D, D, 0, stream));
            DH_TRY(dh_attn_cross_decode(sc->q, D, L.kv, m->keymask, sc->attp, n_img, rows_per_img, m->S, D, m->n_heads, L.ea_scale,
                                        DH_F32_OUT_PLANES, stream));
            dh_prof_set_tag("proj");
            DH_TRY(dh_linear_f32xp_wreg(sc->attp, L.weo_xp, L.beo, nullptr, 0, o, D, nullptr, rows, D, D, 0, stream));
            DH_TRY(dh_add_layernorm_f32x(x, o, L.ln2_g, L.ln2_b, x, sc->xp, rows, D, L.ln2_eps, stream));
        }
        dh_prof_set_tag("ffn");
        DH_TRY(dh_linear_f32xp_wreg(sc->xp, L.w1_xp, L.b1, nullptr, 0, nullptr, 0, sc->ffp, rows, PF, D, 1, stream));   // relu(fc_1): planes only
        dh_prof_set_tag("ffn");
        DH_TRY(dh_linear_f32xp_wreg(sc->ffp, L.w2_xp, L.b2, nullptr, 0, o, D, nullptr, rows, D, PF, 0, stream));
        float* dst = (l == m->n_layers - 1 && x_out) ? (float*)x_out : x;
        DH_TRY(dh_add_layernorm_f32x(x, o, L.ln3_g, L.ln3_b, dst, sc->xp, rows, D, L.ln3_eps, stream));
    }
    if (logits) {
        dh_prof_set_tag("vocab");
        DH_TRY(dh_linear_f32xp(sc->xp, m->cls_w_x, D, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, nullptr, group_max, gm_ld, rows, m->V, 0,
                               stream));
    }
    return DH_OK;
}

extern "C" int dh_transformer_decode_position(const dh_tr_model_t* m, const dh_tr_scratch_t* sc,
                                              const void* start_emb, const int32_t* tokens, int tok_ld,
                                              const int32_t* src, int src_ld, int n_img, int rows_per_img,
                                              int row_mult, int rows_total, int t, void* x_out, float* logits,
                                              int ldl, float* group_max, int gm_ld, void* stream) {
    DH_REQUIRE(m && sc && m->layers && n_img > 0 && rows_per_img > 0 && t >= 0 && (!logits || ldl >= m->V));
    const int rows = n_img * rows_per_img, D = m->D, dt = m->dtype;
    const size_t esz = dt == DH_F32 ? 4 : 2;
    DH_TRY(dh_embed_rows(m->tok_emb, m->pos_emb, start_emb, tokens, tok_ld, sc->x, rows, rows_per_img, row_mult, t,
                         D, m->emb_scale, dt, stream));
    bool deferred = DH_IS_16BIT(dt) && (D % 128) == 0 && D <= 512 && sc->y2 && sc->st0 && sc->st1 && sc->st2;
    for (int l = 0; l < m->n_layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        deferred = deferred && L.w1_f && L.b1_f && L.cs_1 && (l == 0 || (L.wqkv_f && L.bqkv_f && L.cs_qkv)) &&
                   (!m->cross || (L.wq_f && L.bq_f && L.cs_q));
    }
    if (deferred) {
        void* xf = x_out ? x_out : sc->att;          // the final LayerNorm's output feeds the classifier (att is free by then)
        DH_TRY(decode_position_deferred(m, sc, tokens, tok_ld, src, src_ld, n_img, rows_per_img, row_mult, rows_total, t, xf, stream));
        if (logits && group_max) {
            DH_TRY(classifier_groups(xf, D, m->cls_w, m->cls_b, m->cls_w_pk, m->cls_b_pad, logits, ldl, group_max, gm_ld, rows, m->V, D, dt, stream));
        } else if (logits) {
            dh_prof_set_tag("vocab");
            DH_TRY(dh_linear(xf, D, m->cls_w, D, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, rows, m->V, D, 0,
                             dt == DH_F16 ? DH_F16_OUT_F32 : DH_BF16_OUT_F32, stream));
        }
        return DH_OK;
    }
    if (f32xp_ready(m, sc, rows))
        return decode_position_f32xp(m, sc, tokens, tok_ld, src, src_ld, n_img, rows_per_img, row_mult, rows_total, t, x_out, logits, ldl,
                                     group_max, gm_ld, stream);
    for (int l = 0; l < m->n_layers; ++l) {
        const dh_tr_layer_t& L = m->layers[l];
        dh_prof_set_tag("qkv");
        DH_TRY(plain_linear(sc->x, D, L.wqkv, L.wqkv_x, L.wqkv_xp, D, L.bqkv, sc->qkv, 3 * D, rows, 3 * D, D, 0, dt, stream));
        DH_TRY(dh_attn_self_decode(sc->qkv, L.kcache, L.vcache, src, src_ld, tokens, tok_ld, sc->att, n_img, rows_per_img,
                                   row_mult, rows_total, t, D, m->n_heads, L.sa_scale, m->pad_index, dt, stream));
        dh_prof_set_tag("proj");
        DH_TRY(plain_linear(sc->att, D, L.wo, L.wo_x, L.wo_xp, D, L.bo, sc->o, D, rows, D, D, 0, dt, stream));
        DH_TRY(dh_add_layernorm(sc->x, sc->o, L.ln1_g, L.ln1_b, sc->x, rows, D, L.ln1_eps, dt, stream));
        if (m->cross) {
            dh_prof_set_tag("proj");
            DH_TRY(plain_linear(sc->x, D, L.wq, L.wq_x, L.wq_xp, D, L.bq, sc->q, D, rows, D, D, 0, dt, stream));
            DH_TRY(cross_attention(m, L, sc->q, sc->att, n_img, rows_per_img, dt, stream));
            dh_prof_set_tag("proj");
            DH_TRY(plain_linear(sc->att, D, L.weo, L.weo_x, L.weo_xp, D, L.beo, sc->o, D, rows, D, D, 0, dt, stream));
            DH_TRY(dh_add_layernorm(sc->x, sc->o, L.ln2_g, L.ln2_b, sc->x, rows, D, L.ln2_eps, dt, stream));
        }
        dh_prof_set_tag("ffn");
        DH_TRY(plain_linear(sc->x, D, L.w1, L.w1_x, L.w1_xp, D, L.b1, sc->ff, m->pf_dim, rows, m->pf_dim, D, 1, dt, stream));
        dh_prof_set_tag("ffn");
        DH_TRY(plain_linear(sc->ff, m->pf_dim, L.w2, L.w2_x, L.w2_xp, m->pf_dim, L.b2, sc->o, D, rows, D, m->pf_dim, 0, dt, stream));
        void* dst = (l == m->n_layers - 1 && x_out) ? x_out : sc->x;
        DH_TRY(dh_add_layernorm(sc->x, sc->o, L.ln3_g, L.ln3_b, dst, rows, D, L.ln3_eps, dt, stream));
    }
    (void)esz;
    if (logits && group_max && DH_IS_16BIT(dt)) {
        DH_TRY(classifier_groups(x_out ? x_out : sc->x, D, m->cls_w, m->cls_b, m->cls_w_pk, m->cls_b_pad, logits, ldl, group_max, gm_ld, rows,
                                 m->V, D, dt, stream));
    } else if (logits) {
        dh_prof_set_tag("vocab");
        if (dt == DH_F32 && group_max) {
            // (a caller that wants group maxima from an fp32 model gets them from the planes classifier or an error, never stale memory)
            if (!(m->cls_w_x && sc->xp && (D % 32) == 0)) return DH_ERR_UNSUPPORTED;
            DH_TRY(dh_split_act_f32x((const float*)(x_out ? x_out : sc->x), D, sc->xp, rows, D, D, stream));
            DH_TRY(dh_linear_f32xp(sc->xp, m->cls_w_x, D, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, nullptr, group_max, gm_ld, rows, m->V,
                                   0, stream));
        } else if (dt == DH_F32) {
            DH_TRY(plain_linear(x_out ? x_out : sc->x, D, m->cls_w, m->cls_w_x, nullptr, D, m->cls_b, logits, ldl, rows, m->V, D, 0, dt, stream));
        } else {
            DH_TRY(dh_linear(x_out ? x_out : sc->x, D, m->cls_w, D, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, rows,
                             m->V, D, 0, dt == DH_F16 ? DH_F16_OUT_F32 : DH_BF16_OUT_F32, stream));
        }
    }
    return DH_OK;
}

extern "C" int dh_lstm_decode_step(const dh_lstm_model_t* m, const dh_lstm_scratch_t* sc, const void* img_emb,
                                   const int32_t* tokens, int tok_ld, int tok_pos, const int32_t* hparent,
                                   int started, int rows, int rows_per_img, int row_mult, int rows_total,
                                   void* h_out, int ld_out, float* logits, int ldl, float* group_max, int gm_ld, void* stream) {
    DH_REQUIRE(m && sc && m->layers && rows > 0 && rows_per_img > 0 && row_mult > 0 && (!logits || ldl >= m->V));
    const int E = m->E, Hh = m->Hh, dt = m->dtype, nl = m->n_layers;
    const size_t esz = dt == DH_F32 ? 4 : 2;
    bool fused = DH_IS_16BIT(dt) && m->h_alt && m->c_alt;
    for (int l = 0; l < nl; ++l) fused = fused && m->layers[l].w_il && m->layers[l].b_il;
    if (fused) {
        DH_REQUIRE(started >= 0 && started <= 2);
        void* top = h_out ? h_out : sc->hout;
        const int top_ld = h_out ? ld_out : Hh;
        const char* hr = started == 0 ? nullptr : (const char*)(started == 1 ? m->h : m->h_alt);
        const float* cr = started == 0 ? nullptr : (started == 1 ? m->c : m->c_alt);
        char* hw = (char*)(started == 1 ? m->h_alt : m->h);
        float* cw = started == 1 ? m->c_alt : m->c;
        for (int l = 0; l < nl; ++l) {
            const size_t so = (size_t)l * rows_total * Hh;
            const void* x = l == 0 ? img_emb : (const char*)sc->xcatl + (size_t)(l - 1) * rows * 2 * Hh * esz;
            void* dst = l + 1 < nl ? (void*)((char*)sc->xcatl + (size_t)l * rows * 2 * Hh * esz) : top;
            const int El = l == 0 ? E : Hh;
            // decode shapes: the step with the gate weights stationary in registers (option "lstm_wreg" = 0: the tile kernel)
            if (dh_opt(DH_OPT_LSTM_WREG) && m->layers[l].w_pk && dh_lstm_layer_wreg_supported(El, Hh) && rows >= dh_opt(DH_OPT_LSTM_WREG_MIN_ROWS)) {
                DH_TRY(dh_lstm_layer_wreg(x, l == 0 ? E : 2 * Hh, l == 0 ? rows_per_img : 1, l == 0 ? m->emb : nullptr,
                                          l == 0 ? tokens : nullptr, tok_ld, tok_pos, hr ? hr + so * esz : nullptr,
                                          cr ? cr + so : nullptr, hparent, hw + so * esz, cw + so, dst,
                                          l + 1 < nl ? 2 * Hh : top_ld, m->layers[l].w_pk, m->layers[l].b_il, rows, row_mult, El, Hh,
                                          dt, stream));
                continue;
            }
            DH_TRY(dh_lstm_layer_fused(x, l == 0 ? E : 2 * Hh, l == 0 ? rows_per_img : 1, l == 0 ? m->emb : nullptr,
                                       l == 0 ? tokens : nullptr, tok_ld, tok_pos, hr ? hr + so * esz : nullptr,
                                       cr ? cr + so : nullptr, hparent, hw + so * esz, cw + so, dst,
                                       l + 1 < nl ? 2 * Hh : top_ld, m->layers[l].w_il, m->layers[l].b_il, rows, row_mult,
                                       l == 0 ? E : Hh, Hh, dt, stream));
        }
        if (logits && group_max) {
            DH_TRY(classifier_groups(top, top_ld, m->cls_w, m->cls_b, m->cls_w_pk, m->cls_b_pad, logits, ldl, group_max, gm_ld, rows, m->V, Hh, dt, stream));
        } else if (logits) {
            dh_prof_set_tag("vocab");
            DH_TRY(dh_linear(top, top_ld, m->cls_w, Hh, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, rows, m->V, Hh, 0,
                             dt == DH_F16 ? DH_F16_OUT_F32 : DH_BF16_OUT_F32, stream));
        }
        return DH_OK;
    }
    void* top = h_out ? h_out : sc->hout;
    const int top_ld = h_out ? ld_out : Hh;
    // fp32 on the split-operand path with the GEMM operands stored split by the row kernels (options "f32_split" + "f32_planes"): the
    // gate products run on dh_linear_f32xp_wreg (no split pass), the classifier on dh_linear_f32xp with group maxima
    bool planes = dt == DH_F32 && dh_opt(DH_OPT_F32_SPLIT) && dh_opt(DH_OPT_F32_PLANES) && dh_opt(DH_OPT_DECODE_WREG) && sc->xcat0p && sc->topp &&
                  (nl == 1 || sc->xcatlp) && m->cls_w_x && (Hh % 32) == 0 && (E % 32) == 0;
    for (int l = 0; l < nl && planes; ++l)
        planes = m->layers[l].w_xp && dh_linear_f32x_wreg_supported(rows, 4 * Hh, l == 0 ? E + Hh : 2 * Hh);
    if (planes) {
        DH_TRY(dh_lstm_prepare_f32x((const float*)m->emb, (const float*)img_emb, tokens, tok_ld, tok_pos, hparent, started ? (const float*)m->h : nullptr,
                                    started ? m->c : nullptr, (float*)sc->xcat0, (float*)sc->xcatl, sc->c_cur, sc->xcat0p, sc->xcatlp, rows,
                                    rows_per_img, row_mult, rows_total, nl, E, Hh, stream));
        for (int l = 0; l < nl; ++l) {
            const int k = l == 0 ? E + Hh : 2 * Hh;
            const uint16_t* ap = l == 0 ? (const uint16_t*)sc->xcat0p : (const uint16_t*)sc->xcatlp + (size_t)(l - 1) * 2 * rows * 2 * Hh;
            dh_prof_set_tag("gates");
            DH_TRY(dh_linear_f32xp_wreg(ap, m->layers[l].w_xp, m->layers[l].b, nullptr, 0, sc->gates, 4 * Hh, nullptr, rows, 4 * Hh, k, 0, stream));
            const bool last = l + 1 == nl;
            float* dst = last ? (float*)top : (float*)sc->xcatl + (size_t)l * rows * 2 * Hh;
            uint16_t* dp = last ? (uint16_t*)sc->topp : (uint16_t*)sc->xcatlp + (size_t)l * 2 * rows * 2 * Hh;
            DH_TRY(dh_lstm_cell_f32x(sc->gates, sc->c_cur + (size_t)l * rows * Hh, (float*)m->h + (size_t)l * rows_total * Hh,
                                     m->c + (size_t)l * rows_total * Hh, dst, last ? top_ld : 2 * Hh, dp,
                                     (long long)rows * (last ? Hh : 2 * Hh), last ? Hh : 2 * Hh, rows, row_mult, Hh, stream));
        }
        if (logits) {
            dh_prof_set_tag("vocab");
            DH_TRY(dh_linear_f32xp(sc->topp, m->cls_w_x, Hh, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, nullptr, group_max, gm_ld, rows,
                                   m->V, 0, stream));
        }
        return DH_OK;
    }
    DH_TRY(dh_lstm_prepare(m->emb, img_emb, tokens, tok_ld, tok_pos, hparent, started ? m->h : nullptr,
                           started ? m->c : nullptr, sc->xcat0, sc->xcatl, sc->c_cur, rows, rows_per_img, row_mult,
                           rows_total, nl, E, Hh, dt, stream));
    const int gate_dt = dt == DH_F32 ? DH_F32 : (dt == DH_F16 ? DH_F16_OUT_F32 : DH_BF16_OUT_F32);
    for (int l = 0; l < nl; ++l) {
        const void* a = l == 0 ? sc->xcat0 : (const char*)sc->xcatl + (size_t)(l - 1) * rows * 2 * Hh * esz;
        const int k = l == 0 ? E + Hh : 2 * Hh;
        dh_prof_set_tag("gates");
        if (dt == DH_F32) {
            DH_TRY(plain_linear(a, k, m->layers[l].w, m->layers[l].w_x, m->layers[l].w_xp, k, m->layers[l].b, sc->gates, 4 * Hh, rows, 4 * Hh, k, 0, dt, stream));
        } else {
            DH_TRY(dh_linear(a, k, m->layers[l].w, k, m->layers[l].b, nullptr, nullptr, nullptr, 0, sc->gates, 4 * Hh, rows,
                             4 * Hh, k, 0, gate_dt, stream));
        }
        void* dst = l + 1 < nl ? (void*)((char*)sc->xcatl + (size_t)l * rows * 2 * Hh * esz) : top;
        const int ld = l + 1 < nl ? 2 * Hh : top_ld;
        DH_TRY(dh_lstm_cell(sc->gates, sc->c_cur + (size_t)l * rows * Hh, (char*)m->h + (size_t)l * rows_total * Hh * esz,
                            m->c + (size_t)l * rows_total * Hh, dst, ld, rows, row_mult, Hh, dt, stream));
    }
    if (logits && group_max && DH_IS_16BIT(dt)) {
        DH_TRY(classifier_groups(top, top_ld, m->cls_w, m->cls_b, m->cls_w_pk, m->cls_b_pad, logits, ldl, group_max, gm_ld, rows, m->V, Hh, dt, stream));
    } else if (logits) {
        dh_prof_set_tag("vocab");
        if (dt == DH_F32 && group_max && !(m->cls_w_x && sc->topp && (Hh % 32) == 0 && (top_ld % 4) == 0)) return DH_ERR_UNSUPPORTED;
        if (dt == DH_F32 && m->cls_w_x && sc->topp && (group_max || (dh_opt(DH_OPT_F32_SPLIT) && dh_opt(DH_OPT_F32_PLANES))) && (Hh % 32) == 0 &&
            (top_ld % 4) == 0) {
            // the classifier on a split top-layer state (csrc/gemm_f32xp.hip: both operands by LDS-DMA, three slabs deep) + the group
            // maxima the beam sampler takes on the 16-bit paths; bit-identical logits
            DH_TRY(dh_split_act_f32x((const float*)top, top_ld, sc->topp, rows, Hh, Hh, stream));
            DH_TRY(dh_linear_f32xp(sc->topp, m->cls_w_x, Hh, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, nullptr, group_max, gm_ld, rows,
                                   m->V, 0, stream));
        } else if (dt == DH_F32) {
            DH_TRY(plain_linear(top, top_ld, m->cls_w, m->cls_w_x, nullptr, Hh, m->cls_b, logits, ldl, rows, m->V, Hh, 0, dt, stream));
        } else {
            DH_TRY(dh_linear(top, top_ld, m->cls_w, Hh, m->cls_b, nullptr, nullptr, nullptr, 0, logits, ldl, rows, m->V, Hh, 0,
                             gate_dt, stream));
        }
    }
    return DH_OK;
}
