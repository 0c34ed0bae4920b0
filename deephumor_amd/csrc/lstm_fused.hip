// One LSTM layer time step as ONE kernel on the bf16 path (nn.LSTM of rnn_models.py:23-24, one step of :80 / :108):
//   gates[m, :] = [x_m | h_prev[parent(m)]] * [W_ih | W_hh]^T + (b_ih + b_hh)      (matrix cores, fp32 accumulation)
//   c' = sigmoid(f) * c_prev[parent(m)] + sigmoid(i) * tanh(g);   h' = sigmoid(o) * tanh(c')
// What the unfused path does in three launches (dh_lstm_prepare gather, dh_linear, dh_lstm_cell) happens here in the
// operand loader and the register epilogue:
//   * the A operand is never materialised: per output row the loader reads the x part from the token embedding /
//     the image embedding / the layer below and the h part from the PARENT beam's state row (beam reorder = index
//     gather), both straight into LDS by LDS-DMA;
//   * the weight rows are stored gate-interleaved (row 4u+g = gate g of hidden unit u; repacked once when the model
//     is planned), so an MFMA accumulator quad is exactly (i, f, g, o) of one unit for one row and the cell update
//     runs in registers -- the 4*Hh-wide fp32 gate matrix never goes to memory;
//   * the recurrent state is double-buffered (read h_prev/c_prev, write h_next/c_next at the logical row), because
//     other workgroups still gather the old rows while this one writes.
// Same ring / swizzle / wave layout as gemm_bf16_kernel<64, 64, 2, false, 4, 4>.
#include "common.h"
#include "prof.h"
#include "options.h"

typedef dh_f32x4 f32x4;
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

__device__ uint4 lf_zero_page[4];          // zero-initialised: source of padding chunks / zero initial state

struct LstmFusedParams {
    const uint16_t* x_rows; int ldx, x_div;      // compact x rows: row m reads x_rows[(m / x_div) * ldx + ...]
    const uint16_t* emb; const int32_t* tokens; int tok_ld, tok_pos;   // or: x row = emb[tokens[rl * tok_ld + tok_pos]]
    const uint16_t* h_prev; const float* c_prev; const int32_t* hparent;
    uint16_t* h_next; float* c_next;             // state out, logical rows
    uint16_t* h_out; int ld_out;                 // compact output rows (next layer's x / classifier input)
    const uint16_t* W; const float* bias;        // gate-interleaved [4*Hh, E+Hh], [4*Hh]
    int rows, row_mult, E, Hh, tiles_m, tiles_n;
};

__device__ __forceinline__ void lf_wait_vmcnt(int n) {
    switch (n) {
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 21: asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); break;
        case 28: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// Gate non-linearities on the hardware exponential (v_exp_f32) and reciprocal: ~6 instructions each instead of the
// ~25-40 of expf / tanhf; relative error ~1e-6, far below the bf16 rounding of h (the fp32 path keeps expf / tanhf).
__device__ __forceinline__ float lf_sigmoid(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float lf_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));             // in (0, 1]: no overflow
    const float t = (1.0f - e) * __frcp_rn(1.0f + e);
    return copysignf(t, x);
}

// BM = 64 (default): 64 x 64 tiles (640 workgroups at 1280 rows x 2048 gate columns, 2-3 per CU).  BM = 160 (opt-in, see the
// entry point): 160 x 64 tiles, 80 x 32 per wave -- at 1280 rows exactly 256 workgroups, one per CU.
template <typename OT, int NS, int BM = 64>
__global__ __launch_bounds__(256) void lstm_layer_fused_kernel(LstmFusedParams p) {
    constexpr int BN = 64, BK = 64, NW = 4, WAVES_M = 2;
    constexpr int A_BYTES = BM * 128, SLAB = A_BYTES + BN * 128;
    constexpr int WM = BM / WAVES_M, WN = 32, TM = WM / 16, TN = 2;
    static_assert(BM % (8 * NW) == 0 && WM % 16 == 0, "tile shape");
    constexpr int IA = BM / (8 * NW), IB = BN / (8 * NW), G = IA + IB;      // 2 + 2 LDS-DMA instructions per wave per slab
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLAB];

    const int nblk = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {   // XCD-aware bijective remap (consecutive tile ids on one XCD's L2)
        const int q = nblk / 8, r = nblk % 8, xcd = bid % 8, idx = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % p.tiles_m, tn = bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm0 = (wave % WAVES_M) * WM, wn0 = (wave / WAVES_M) * WN;
    const int lr = lane >> 3, lpos = lane & 7, swz = lpos ^ lr;
    const int l15 = lane & 15, lq = lane >> 4;
    const unsigned char* zero = reinterpret_cast<const unsigned char*>(lf_zero_page);
    const int K = p.E + p.Hh, N = 4 * p.Hh;

    // ---- per-lane source rows: x part and (gathered) h part of the virtual [x | h] operand ------------------------
    const uint16_t* ax[IA]; const uint16_t* ah[IA];
    bool a_ok[IA];
    // every index load of the kernel first (token ids and beam parents of the loader's rows, beam parents of the
    // epilogue's rows), at clamped rows and without branches: one memory round trip for all of them
    constexpr int TM_ = TM;
    int tok_i[IA], hp_i[IA], hp_e[TM_];
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int m = m0 + (wave * IA + i) * 8 + lr;
        a_ok[i] = m < p.rows;
        const int rl = (a_ok[i] ? m : 0) * p.row_mult;
        tok_i[i] = p.tokens ? p.tokens[(size_t)rl * p.tok_ld + p.tok_pos] : 0;
        hp_i[i] = p.hparent ? p.hparent[rl] : rl;
    }
#pragma unroll
    for (int i = 0; i < TM_; ++i) {
        const int m = m0 + wm0 + 16 * i + l15;
        const int rl = (m < p.rows ? m : 0) * p.row_mult;
        hp_e[i] = p.hparent ? p.hparent[rl] : rl;
    }
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int m = m0 + (wave * IA + i) * 8 + lr;
        const int mm = a_ok[i] ? m : 0;
        ax[i] = p.tokens ? p.emb + (size_t)tok_i[i] * p.E : p.x_rows + (size_t)(mm / p.x_div) * p.ldx;
        ah[i] = p.h_prev ? p.h_prev + (size_t)hp_i[i] * p.Hh : nullptr;
    }
    const uint16_t* b_src[IB];
    bool b_ok[IB];
#pragma unroll
    for (int i = 0; i < IB; ++i) {
        const int n = n0 + (wave * IB + i) * 8 + lr;
        b_ok[i] = n < N;
        b_src[i] = p.W + (size_t)(b_ok[i] ? n : 0) * K + swz * 8;
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* slab = lds + __builtin_amdgcn_readfirstlane(buf) * SLAB;
        const int k = k0 + swz * 8;                       // E % 8 == 0: a 16-byte chunk lies in the x part or in the h part
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const void* src = zero;
            if (a_ok[i] && k < K) {
                if (k < p.E) src = ax[i] + k;
                else if (ah[i]) src = ah[i] + (k - p.E);
            }
            dh_lds_dma16(src, slab + (wave * IA + i) * 1024);
        }
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const void* src = (b_ok[i] && k < K) ? (const void*)(b_src[i] + k0) : (const void*)zero;
            dh_lds_dma16(src, slab + A_BYTES + (wave * IB + i) * 1024);
        }
    };

    // epilogue operands requested before the reduction: bias quads and the parent's cell state
    float4 b4[TN];
    float c0[TM][TN];
    int rl_e[TM];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + 16 * j + 4 * lq;
        b4[j] = n < N ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + 16 * i + l15;
        const bool ok = m < p.rows;
        rl_e[i] = (ok ? m : 0) * p.row_mult;
        const int hp = hp_e[i];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int u = (n0 + wn0 + 16 * j) / 4 + lq;
            const bool live = ok && p.c_prev && u < p.Hh;
            const float t = p.c_prev ? p.c_prev[(size_t)hp * p.Hh + (u < p.Hh ? u : 0)] : 0.f;     // unconditional, clamped
            c0[i][j] = live ? t : 0.f;
        }
    }

    f32x4 acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nslab = (K + BK - 1) / BK;
#pragma unroll
    for (int u = 0; u < NS - 1; ++u)
        if (u < nslab) stage(u, u * BK);
    for (int t = 0; t < nslab; ++t) {
        {   // (the b4 / c0 loads are older than every slab)  steady state: an immediate; the run-time switch only in the K tail
            // (a switch on a run-time count is a tree of taken scalar branches: gemm_bf16.hip, wait_vmcnt_hot)
            const int newer = min(NS - 2, nslab - 1 - t);
            if (__builtin_expect(newer == NS - 2, 1)) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * G) : "memory");
            else lf_wait_vmcnt(newer * G);
        }
        __builtin_amdgcn_s_barrier();
        if (t + NS - 1 < nslab) stage((t + NS - 1) % NS, (t + NS - 1) * BK);
        const unsigned char* sa = lds + (t % NS) * SLAB;
        const unsigned char* sb = sa + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int c = kk * 4 + lq;
            uint4 fa[TM], fw[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wm0 + i * 16 + l15;
                fa[i] = *reinterpret_cast<const uint4*>(sa + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + j * 16 + l15;
                fw[j] = *reinterpret_cast<const uint4*>(sb + r * 128 + ((c ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    acc[j][i] = Op16<OT>::mfma(fw[j], fa[i], acc[j][i]);
        }
    }
    // ---- cell update in registers: acc[j][i] = (i, f, g, o) pre-activations of unit u for row m -------------------
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + wm0 + 16 * i + l15;
        if (m >= p.rows) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int u = (n0 + wn0 + 16 * j) / 4 + lq;
            if (u >= p.Hh) continue;
            const float gi = acc[j][i][0] + b4[j].x, gf = acc[j][i][1] + b4[j].y;
            const float gg = acc[j][i][2] + b4[j].z, go = acc[j][i][3] + b4[j].w;
            const float c1 = lf_sigmoid(gf) * c0[i][j] + lf_sigmoid(gi) * lf_tanh(gg);
            const float h1 = lf_sigmoid(go) * lf_tanh(c1);
            const uint16_t hb = Op16<OT>::from_f32(h1);
            p.c_next[(size_t)rl_e[i] * p.Hh + u] = c1;
            p.h_next[(size_t)rl_e[i] * p.Hh + u] = hb;
            p.h_out[(size_t)m * p.ld_out + u] = hb;
        }
    }
}

extern "C" int dh_lstm_layer_fused(const void* x_rows, int ldx, int x_div, const void* emb, const int32_t* tokens,
                                   int tok_ld, int tok_pos, const void* h_prev, const float* c_prev,
                                   const int32_t* hparent, void* h_next, float* c_next, void* h_out, int ld_out,
                                   const void* w_il, const float* b_il, int rows, int row_mult, int E, int Hh,
                                   int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE((x_rows || (emb && tokens)) && h_next && c_next && h_out && w_il && b_il);
    DH_REQUIRE(rows > 0 && row_mult > 0 && x_div > 0 && (E % 8) == 0 && (Hh % 8) == 0 && (ldx % 8) == 0 && ld_out >= Hh);
    DH_REQUIRE((h_prev == nullptr) == (c_prev == nullptr) && h_prev != h_next && c_prev != c_next);
    DH_REQUIRE(((uintptr_t)w_il % 16) == 0 && ((uintptr_t)b_il % 16) == 0 && ((uintptr_t)x_rows % 16) == 0 &&
               ((uintptr_t)emb % 16) == 0 && ((uintptr_t)h_prev % 16) == 0);
    LstmFusedParams p{};
    p.x_rows = (const uint16_t*)x_rows; p.ldx = ldx; p.x_div = x_div;
    p.emb = (const uint16_t*)emb; p.tokens = tokens; p.tok_ld = tok_ld; p.tok_pos = tok_pos;
    p.h_prev = (const uint16_t*)h_prev; p.c_prev = c_prev; p.hparent = hparent;
    p.h_next = (uint16_t*)h_next; p.c_next = c_next; p.h_out = (uint16_t*)h_out; p.ld_out = ld_out;
    p.W = (const uint16_t*)w_il; p.bias = b_il; p.rows = rows; p.row_mult = row_mult; p.E = E; p.Hh = Hh;
    p.tiles_m = dh_cdiv(rows, 64); p.tiles_n = dh_cdiv(4 * Hh, 64);
    const double K = E + Hh;
    DhProfScope prof("dh_lstm_layer_fused", 2.0 * rows * 4 * Hh * K, 2.0 * (rows * K + 4.0 * Hh * K) + 12.0 * rows * Hh, stream);
    // ring depth by workgroup count so that all tiles are co-resident in ONE round where possible (16 KB per slab):
    // measured at 1280 rows x 2048 gate columns (640 workgroups): 4 slabs (2 per CU, 1.25 rounds) 20.7 / 24.3 us,
    // 3 slabs (3 per CU) 16.6 / 19.0 us, 2 slabs (5 per CU) 16.4 / 18.6 us  (E = 256 / 512)
    const int blocks = p.tiles_m * p.tiles_n;
    DH_DISPATCH_16(dtype, {
        if (blocks > 768 && blocks <= 1280)
            hipLaunchKernelGGL((lstm_layer_fused_kernel<T, 2>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        else if (blocks > 512 && blocks <= 768)
            hipLaunchKernelGGL((lstm_layer_fused_kernel<T, 3>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
        else
            hipLaunchKernelGGL((lstm_layer_fused_kernel<T, 4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
    });
    DH_LAUNCH_CHECK();
}
