// Run-time options of the library: ONE table (name, environment variable that supplies the default, built-in default), read
// through dh_opt() at every use -- nothing is latched in function-local statics -- and settable through the C-ABI
// (dh_set_option / dh_get_option, include/deephumor_hip.h).  The Python layer keeps its own kernel-selection switches in the
// same table (hip.option), so `dh_option_name` enumerates every switch the product has.
#pragma once

enum DhOption {
    // ---- dispatch inside the native step drivers (runtime.hip) ----
    DH_OPT_VOCAB_WREG = 0,           // register-streamed classifier of the LSTM chain (vocab_wreg.hip)
    DH_OPT_DECODE_WREG,              // register-stationary decode-chain GEMMs (linear_wreg.hip)
    DH_OPT_DECODE_WREG_MIN_ROWS,     // ... from this many rows per position
    DH_OPT_QKV_FUSION_MAX_ROWS,      // dh_attn_self_qkv_decode instead of GEMM + attention up to this many rows per position
    DH_OPT_CROSS_QPROJ,              // 1: fc_q inside the cross-attention launch (dh_attn_cross_qproj_decode); 0 (default): its own GEMM
    DH_OPT_CROSS_KV_PREFETCH,        // fc_q as its own GEMM: this many extra workgroups of it pull the attention's K / V tiles into L2 (0 = none)
    DH_OPT_DECODE_CHAIN_FUSION,      // (enc_)fc_o -> fc_1 -> fc_2 -> next fc_qkv of a decode position as ONE launch (dh_decode_gemm_chain)
    DH_OPT_LSTM_WREG,                // register-stationary LSTM step (lstm_wreg.hip)
    DH_OPT_LSTM_WREG_MIN_ROWS,       // ... from this many rows
    DH_OPT_VOCAB_SPLIT_ROWS,         // classifier of a row count that is no multiple of 256: whole 256-row tiles + remainder as two launches
    // ---- tile choices of single kernels ----
    DH_OPT_GEMM64_NS, DH_OPT_VOCAB_TILE, DH_OPT_VOCAB_GMAX_TILE, DH_OPT_VOCAB_AREG, DH_OPT_LOGPROB_TILE,
    DH_OPT_LSTM_BM, DH_OPT_LSTM_NS, DH_OPT_VOCAB_WREG_NT, DH_OPT_VOCAB_WREG_PREFETCH,
    // ---- fp32 models: arithmetic of the dense layers ----
    DH_OPT_F32_SPLIT,                // 1: fp32 GEMMs / convolutions as three fp16 MFMAs on split operands (gemm_f32x.hip)
    // ---- switches read by the Python layer (kernel selection in the plans) ----
    DH_OPT_CONV1X1_WREG, DH_OPT_CONV_S4, DH_OPT_DIRECT_3X3, DH_OPT_DIRECT_STEM, DH_OPT_STEM_POOL, DH_OPT_FUSED_TAIL,
    DH_OPT_S1_CONV1_FUSION, DH_OPT_S2_CONV1_FUSION, DH_OPT_S3_TAIL, DH_OPT_S2_TAIL, DH_OPT_VOCAB_WREG_PLAN,
    DH_OPT_VOCAB_WREG_TRANSFORMER, DH_OPT_VOCAB_WREG_TRANSFORMER_MAX_ROWS, DH_OPT_DEFERRED_LN, DH_OPT_DECODE_WREG_PLAN, DH_OPT_PACKED_CROSS, DH_OPT_QPROJ_FUSION,
    DH_OPT_FUSED_BEAM_STEP, DH_OPT_FUSED_BEAM_STEP_MAX_ROWS, DH_OPT_PIPE_PRIO, DH_OPT_DIST_ALWAYS, DH_OPT_DECODE_STREAMS,
    DH_OPT_COUNT
};

int dh_opt(int which);      // current value (first use reads the environment)
