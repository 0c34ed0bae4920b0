// Run-time options of the library: ONE table (name, environment variable that supplies the default, built-in default), read
// through dh_opt() at every use -- nothing is latched in function-local statics -- and settable through the C-ABI
// (dh_set_option / dh_get_option, include/deephumor_hip.h).  The Python layer keeps its own kernel-selection switches in the
// same table (hip.option), so `dh_option_name` enumerates every switch the product has.
#pragma once

enum DhOption {
    // ---- dispatch inside the native step drivers (runtime.hip) and weight plans of the Python layer ----
    DH_OPT_VOCAB_WREG = 0,           // register-streamed classifier (vocab_wreg.hip): the LSTM decoder's, and the plans that pack its weights
    DH_OPT_VOCAB_AREG,               // A-stationary classifier kernels (vocab_areg.h); 0 = the 128 x 128 tile kernel, 128 = never the 256-row form
    DH_OPT_VOCAB_WREG_TRANSFORMER,   // 1: the Transformer decoders use the register-streamed classifier at every row count
    DH_OPT_VOCAB_WREG_TRANSFORMER_MAX_ROWS,   // ... else up to this many rows per position
    DH_OPT_DECODE_WREG,              // register-stationary decode-chain GEMMs (linear_wreg.hip) and the plans that pack their weights
    DH_OPT_DECODE_WREG_MIN_ROWS,     // ... from this many rows per position
    DH_OPT_DECODE_LAYERS,            // the decoder layers of a position as ONE persistent launch (decode_layers.hip)
    DH_OPT_LSTM_WREG,                // register-stationary LSTM step (lstm_wreg.hip)
    DH_OPT_LSTM_WREG_MIN_ROWS,       // ... from this many rows
    // ---- fp32 models: arithmetic of the dense layers ----
    DH_OPT_F32_SPLIT,                // 1: fp32 GEMMs / convolutions as three fp16 MFMAs on split operands (gemm_f32x.hip)
    DH_OPT_F32_PLANES,               // ... with the decode chain's GEMM operands stored split by their producers (gemm_f32xp.hip; 0: A/B)
    // ---- switches read by the Python layer (kernel selection in the plans) ----
    DH_OPT_DEFERRED_LN,              // deferred-LayerNorm decode chain (0: LayerNorm launches between the GEMMs)
    DH_OPT_PACKED_CROSS,             // matrix-core cross-attention on packed K / V^T tiles (0: the LDS kernel)
    DH_OPT_ENCODER_GENERIC,          // 0: every specialised encoder kernel; 1: round-3 set (no streaming 1x1 / stage-1,2,4 tails / conv1 fusions);
                                     // 2: everything through the implicit-GEMM tile kernel (the reference point of the bit-identity A/B tests)
    DH_OPT_DIST_ALWAYS,              // a one-rank process group still runs its collectives (bench.py --rccl-single, tests)
    DH_OPT_DECODE_STREAMS,           // HIP streams a batch's decode is interleaved over (1)
    DH_OPT_COUNT
};

int dh_opt(int which);      // current value (first use reads the environment)
