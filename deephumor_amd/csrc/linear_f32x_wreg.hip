// dh_linear_f32x for the DECODE shapes of an fp32 model on the split-operand path (option "f32_split"), with the WEIGHTS STATIONARY IN
// REGISTERS -- linear_wreg.hip's partition applied to gemm_f32x.hip's arithmetic (round 6).
// (transformers.py:97 fc_q/k/v, :127 fc_o, :162-163 fc_1 / fc_2 and rnn_models.py:80 the LSTM gate products, applied to the rows of ONE
// decode position: 1,280 rows at 256 images x beam 5.)
//
// The tile kernels of gemm_f32x.hip run these launches as 64 x 64 tiles with one memory round trip per 32-k slab: 16 - 25 us per launch,
// 1,188 of them per 256-image Transformer step (30 ms of a 59 ms step).  Here:
//   * a workgroup owns 16 x NW output columns x RL activation rows; a wave keeps the hi AND lo fp16 planes of its 16 weight rows x KC
//     k values as MFMA fragments in registers (KC / 32 x 8 VGPRs: 128 for KC = 512), loaded straight from L2 out of fragment-packed
//     planes (dh_pack_mfma_fragments of each plane of dh_split_f32x: coalesced 1 KB loads);
//   * the [RL x KC] fp32 activation block comes into LDS ONCE by LDS-DMA (every piece requested up front, no staging registers) and
//     is split IN PLACE (x = hi + lo * 2^-11, exactly as gemm_f32x.hip): the 32 bytes of 8 fp32 values become 16 bytes of hi and 16
//     bytes of lo fp16 values in the slab layout of linear_wreg.hip (conflict-free ds_read_b128 fragments) -- no per-slab round
//     trip, one wait + two barriers per K chunk;
//   * per (k step, row tile) two LDS fragment reads feed three MFMAs: acc += w_hi a_hi; cor += w_hi a_lo; cor += w_lo a_hi -- the
//     same products in the same order as the tile kernels, so the results are BIT-IDENTICAL to dh_linear_f32x;
//   * K = 2,048 (fc_2) runs as four K chunks of 512 through the same registers and LDS (accumulators live across the chunks).
// Range guard of the activations as in gemm_f32x.hip (sticky per-stream word).
#include "common.h"
#include "prof.h"

unsigned* dh_f32x_range_flag_of(hipStream_t s);          // gemm_f32x.hip

namespace {
constexpr float kLoScale = 2048.0f, kLoInv = 1.0f / 2048.0f, kF16Max = 65504.0f;

struct FwParams {
    const float* A; int lda;
    const uint16_t* Ap; unsigned a_plane_b;              // PIN: the activation as planes [2][M][lda] fp16 (hi, lo); bytes from hi to lo
    uint16_t* Cp; size_t c_plane; int ldcp;              // optional: the result as planes [2][M][ldcp] (as well as / instead of C)
    const uint4* wh; const uint4* wl;                    // fragment-packed planes [K / 32][N / 16][64] x 16 bytes
    const float* bias; const float* res; int ldres;
    float* C; int ldc;
    int M, N, K, relu, tiles_m, tiles_n, xn;
    unsigned* range_flag;
};

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    const f16_t ha = (f16_t)a, hb = (f16_t)b;
    const f16_t la = (f16_t)((a - (float)ha) * kLoScale), lb = (f16_t)((b - (float)hb) * kLoScale);
    hi = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
    lo = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
}

// global -> LDS, 16 bytes per lane: wave-uniform base (SGPR pair) + per-lane byte offset
__device__ __forceinline__ void fw_dma16(const void* base, unsigned off, void* lds_dst) {
    const unsigned m = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(dh_lptr_t)lds_dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" :: "s"(m), "v"(off), "s"(base) : "memory");
}

// NW waves (16 output columns each), RL activation rows, K chunks of KC = 32 NS k values (K = a multiple of KC).
// LDS: slab s = the 32 k values 32 s .. + 31 of all RL rows, 128 bytes per row (fp32), 16-byte slots XOR-swizzled by the row (as
// linear_wreg.hip).  The fp32 block comes in by LDS-DMA (no staging registers, every piece requested up front), then every 32-byte
// piece (8 k values of one row: the two adjacent slots 2q ^ m, (2q + 1) ^ m) is split IN PLACE into its fp16 hi plane (first
// slot) and lo plane (second slot): an MFMA fragment is then one ds_read_b128 per plane.
// PIN: the activation arrives SPLIT (planes written by its producer: LayerNorm, attention, the ReLU layer in front): the hi / lo pieces go
// by LDS-DMA straight into the slots the in-place split would fill -- no split pass, one barrier less, no VALU work before the MFMAs.
template <int NW, int RL, int NS, bool PIN>
__global__ __launch_bounds__(64 * NW, 1) void linear_f32x_wreg_kernel(FwParams p) {
    constexpr int NT = 64 * NW, KC = 32 * NS, KF = NS, TM = (RL + 15) / 16, RG = RL / 8, SLABB = RL * 128;
    constexpr int PIECES = RL * NS * 4;                  // 32-byte pieces of a chunk
    constexpr int C_IT = (PIECES + NT - 1) / NT;
    constexpr int PF = NW == 8 ? 1 : 2;                  // LDS fragment pairs read this many MFMA groups ahead (two waves per SIMD hide more)
    constexpr bool PARTIAL = (RL % 16) != 0;
    static_assert(RL % 8 == 0 && NS * SLABB <= 163840 && (NS * RG) % NW == 0, "LDS budget / pieces per wave");
    __shared__ __attribute__((aligned(16))) unsigned char lds[NS * SLABB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lr = lane >> 3, lpos = lane & 7;
    int cb, rb;
    if (p.xn) {                                          // XCD x = blockIdx % 8 owns column group x % xn, row group x / xn
        const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
        const int cpg = p.tiles_n / p.xn, rpg = p.tiles_m / (8 / p.xn);
        cb = (xcd % p.xn) * cpg + idx % cpg; rb = (xcd / p.xn) * rpg + idx / cpg;
    } else {
        cb = blockIdx.x % p.tiles_n; rb = blockIdx.x / p.tiles_n;
    }
    const int m0 = rb * RL, n0 = cb * 16 * NW;

    dh_f32x4 acc[TM], cor[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { acc[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; cor[i] = dh_f32x4{0.f, 0.f, 0.f, 0.f}; }
    // LDS read bases of the MFMA fragments: row 16 i + l15 (row & 7 == l15 & 7), piece lq of the slab: hi slot, lo slot = hi slot ^ 1
    // (with RL % 16 == 8 the upper half of the last tile does not exist: those lanes re-read the lower half's rows, outputs dropped)
    const unsigned rd_base = (unsigned)(l15 * 128 + (((2 * lq) ^ (l15 & 7)) << 4));
    const unsigned rd_last = PARTIAL ? (unsigned)((l15 & 7) * 128 + (((2 * lq) ^ (l15 & 7)) << 4)) : rd_base;
    // source byte offset of this lane's 16 bytes in row (8 g + lr) of the block: slot lpos of a row with (row & 7) == lr
    const unsigned swz = (unsigned)((lpos ^ lr) << 4), ldb = (unsigned)p.lda * 4u, ldb2 = (unsigned)p.lda * 2u;
    const unsigned pin_off = (unsigned)(((lpos ^ lr) >> 1) << 4) + (((lpos ^ lr) & 1) ? p.a_plane_b : 0u);
    float amax = 0.f;
    const size_t fstep = (size_t)(p.N / 16) * 64;
    const int nchunk = p.K / KC;
    for (int kc = 0; kc < nchunk; ++kc) {
        if (kc > 0) __syncthreads();                     // every wave is done reading the previous chunk
        // ---- the activation chunk by LDS-DMA: piece = 8 rows x 128 bytes; wave w stages slabs w, w + NW, ... ---------------------------
        {
            constexpr int SPW = (NS + NW - 1) / NW;
#pragma unroll
            for (int sl = 0; sl < SPW; ++sl) {
                const int s = wave + NW * sl;
                if (s < NS) {
#pragma unroll
                    for (int g = 0; g < RG; ++g) {
                        if (PIN)     // slot lpos of row (8 g + lr) holds logical slot lpos ^ lr = 2 q + plane: 8 k values of one plane
                            fw_dma16(p.Ap, (unsigned)min(m0 + g * 8 + lr, p.M - 1) * ldb2 + pin_off + (unsigned)(kc * KC + 32 * s) * 2u, lds + s * SLABB + g * 1024);
                        else
                            fw_dma16(p.A, (unsigned)min(m0 + g * 8 + lr, p.M - 1) * ldb + swz + (unsigned)(kc * KC + 32 * s) * 4u, lds + s * SLABB + g * 1024);
                    }
                }
            }
        }
        // ---- this wave's 16 weight rows x KC, both planes: 2 KF fragments of 1 KB straight into registers ------------------------------
        uint4 wfh[KF], wfl[KF];
        {
            const size_t base = ((size_t)kc * KF * (p.N / 16) + (size_t)(n0 / 16 + wave)) * 64 + lane;
#pragma unroll
            for (int f = 0; f < KF; ++f) { wfh[f] = p.wh[base + (size_t)f * fstep]; wfl[f] = p.wl[base + (size_t)f * fstep]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                 // the block is in LDS
        // ---- split in place: piece pc = (slab, row, q): x0 = slot (2q) ^ m, x1 = its neighbour -> hi, lo -----------------------------------
#pragma unroll 2
        for (int it = 0; it < (PIN ? 0 : C_IT); ++it) {
            const int pc = tid + it * NT;
            if (C_IT * NT == PIECES || pc < PIECES) {
                const int q = pc & 3, row = (pc >> 2) % RL, s = (pc >> 2) / RL;
                unsigned char* a0 = lds + s * SLABB + row * 128 + (((2 * q) ^ (row & 7)) << 4);
                unsigned char* a1 = lds + s * SLABB + row * 128 + (((2 * q + 1) ^ (row & 7)) << 4);
                const float4 x0 = *reinterpret_cast<const float4*>(a0), x1 = *reinterpret_cast<const float4*>(a1);
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))));
                amax = fmaxf(amax, fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w))));
                uint4 hi, lo;
                split2(x0.x, x0.y, hi.x, lo.x); split2(x0.z, x0.w, hi.y, lo.y);
                split2(x1.x, x1.y, hi.z, lo.z); split2(x1.z, x1.w, hi.w, lo.w);
                *reinterpret_cast<uint4*>(a0) = hi;
                *reinterpret_cast<uint4*>(a1) = lo;
            }
        }
        if (!PIN) __syncthreads();
        // ---- TM row tiles x KF k-steps: two fragment reads (PF groups ahead), three MFMAs in the tile kernels' order ---------------------
        uint4 fh[PF + 1], fl[PF + 1];
        auto rd = [&](int t) {                           // t = TM f + i
            const int f = t / TM, i = t - f * TM;
            const unsigned base = ((PARTIAL && i == TM - 1) ? rd_last : rd_base) + f * SLABB + i * 2048;
            fh[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + base);
            fl[t % (PF + 1)] = *reinterpret_cast<const uint4*>(lds + (base ^ 16u));
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) rd(t);
#pragma unroll
        for (int t = 0; t < KF * TM; ++t) {
            const int f = t / TM, i = t - f * TM;
            if (t + PF < KF * TM) rd(t + PF);
            acc[i] = Op16<f16_t>::mfma(wfh[f], fh[t % (PF + 1)], acc[i]);      // hi * hi
            cor[i] = Op16<f16_t>::mfma(wfh[f], fl[t % (PF + 1)], cor[i]);      // hi * lo
            cor[i] = Op16<f16_t>::mfma(wfl[f], fh[t % (PF + 1)], cor[i]);      // lo * hi
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (amax >= kF16Max) atomicOr(p.range_flag, 1u);
    // ---- epilogue: acc[i][r] = C[m0 + 16 i + l15][n0 + 16 wave + 4 lq + r]: one 16-byte store per row tile.  The lane coordinates are
    // re-derived here (v_mbcnt): kept live across the MFMA loop they cost the 8-wave form two spilled registers ---------------------------
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int e15 = ln & 15, eq = ln >> 4;
    const int n = n0 + 16 * wave + 4 * eq;
    const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
    float omax = 0.f;                                    // range guard of a result stored as planes
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = 16 * i + e15, m = m0 + row;
        if ((PARTIAL && row >= RL) || m >= p.M) continue;
        float4 v;
        v.x = fmaf(cor[i][0], kLoInv, acc[i][0]) + b4.x; v.y = fmaf(cor[i][1], kLoInv, acc[i][1]) + b4.y;
        v.z = fmaf(cor[i][2], kLoInv, acc[i][2]) + b4.z; v.w = fmaf(cor[i][3], kLoInv, acc[i][3]) + b4.w;
        if (p.res) {
            const float4 rr = *reinterpret_cast<const float4*>(p.res + (size_t)m * p.ldres + n);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
        if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        if (p.C) *reinterpret_cast<float4*>(p.C + (size_t)m * p.ldc + n) = v;
        if (p.Cp) {
            uint2 hi, lo;
            split2(v.x, v.y, hi.x, lo.x); split2(v.z, v.w, hi.y, lo.y);
            omax = fmaxf(omax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
            *reinterpret_cast<uint2*>(p.Cp + (size_t)m * p.ldcp + n) = hi;
            *reinterpret_cast<uint2*>(p.Cp + p.c_plane + (size_t)m * p.ldcp + n) = lo;
        }
    }
    if (omax >= kF16Max) atomicOr(p.range_flag, 1u);
}

int pick_xn(int tiles_m, int tiles_n) {
    for (int xn = 8; xn >= 1; xn /= 2)                   // as many column groups as divide: an XCD's L2 then fetches the fewest weight blocks
        if (tiles_n % xn == 0 && tiles_m % (8 / xn) == 0) return xn;
    return 0;
}
}  // namespace

namespace {
// 64-column x 40-row blocks (4 waves, 80 KB of LDS: two workgroups per CU) while they fit one residency round of 512; wider outputs
// (fc_q|k|v, fc_1, the LSTM gates at 1,280 rows) as 128-column x 80-row blocks (8 waves, the whole 160 KB) up to two rounds of 256;
// beyond that the tile kernels of gemm_f32x.hip are faster (measured, tools/f32x_kbench.py: 3,000 x 2,048 x 512 47 against 36 us)
int fw_form(int M, int N) {
    const long long narrow = (long long)dh_cdiv(M, 40) * (N / 64), wide = (long long)dh_cdiv(M, 80) * (N / 128);
    if (narrow <= 512) return 1;
    if ((N % 128) == 0 && wide <= 512) return 2;
    return 0;
}
}  // namespace

// 1 when dh_linear_f32x_wreg takes the shape: N a multiple of 64, K a multiple of 512 (the Transformer decoder's projections and
// feed-forward layers) or of 384 (the LSTM gate product of the released models: E + Hh = 256 + 512), and few enough rows that the
// launch stays within two residency rounds (a decode position; teacher-forced batches keep the tile kernels)
extern "C" int dh_linear_f32x_wreg_supported(int M, int N, int K) {
    return M > 0 && N > 0 && (N % 64) == 0 && K > 0 && ((K % 512) == 0 || (K % 384) == 0) && K <= 4096 && fw_form(M, N) != 0;
}

namespace {
template <bool PIN>
int fw_launch(FwParams& p, hipStream_t s) {
    p.range_flag = dh_f32x_range_flag_of(s);
    if (!p.range_flag) return DH_ERR_LAUNCH;
    const bool k384 = (p.K % 512) != 0;
    const bool wide = fw_form(p.M, p.N) == 2;
    p.tiles_n = p.N / (wide ? 128 : 64); p.tiles_m = dh_cdiv(p.M, wide ? 80 : 40);
    p.xn = pick_xn(p.tiles_m, p.tiles_n);
    const dim3 grid((unsigned)(p.tiles_m * p.tiles_n));
    if (wide) {
        if (k384) hipLaunchKernelGGL((linear_f32x_wreg_kernel<8, 80, 12, PIN>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((linear_f32x_wreg_kernel<8, 80, 16, PIN>), grid, dim3(512), 0, s, p);
    } else {
        if (k384) hipLaunchKernelGGL((linear_f32x_wreg_kernel<4, 40, 12, PIN>), grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL((linear_f32x_wreg_kernel<4, 40, 16, PIN>), grid, dim3(256), 0, s, p);
    }
    return hipGetLastError() == hipSuccess ? DH_OK : DH_ERR_LAUNCH;
}
}  // namespace

// C [M, ldc] fp32 = act(A W^T + bias (+ residual)) as dh_linear_f32x computes it (bit-identical), with w_packed = the two planes of
// dh_split_f32x(W [N, K]) each through dh_pack_mfma_fragments: [2][K / 32][N / 16][64] x 16 bytes.
extern "C" int dh_linear_f32x_wreg(const float* A, int lda, const void* w_packed, const float* bias, const float* residual, int ldres,
                                   float* C, int ldc, int M, int N, int K, int relu, void* stream) {
    DH_REQUIRE(A && w_packed && bias && C && dh_linear_f32x_wreg_supported(M, N, K) && lda >= K && ldc >= N);
    DH_REQUIRE((lda % 4) == 0 && (ldc % 4) == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 && ((uintptr_t)C % 16) == 0 &&
               ((uintptr_t)bias % 16) == 0 && (!residual || (ldres >= N && (ldres % 4) == 0 && ((uintptr_t)residual % 16) == 0)));
    FwParams p{};
    p.A = A; p.lda = lda; p.wh = (const uint4*)w_packed; p.wl = p.wh + (size_t)(K / 32) * (N / 16) * 64;
    p.bias = bias; p.res = residual; p.ldres = ldres; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.relu = relu;
    dh_prof_set_dims(M, N, K);
    DhProfScope prof("dh_linear_f32x", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), stream);
    return fw_launch<false>(p, (hipStream_t)stream);
}

// The same launch for an activation stored as planes (a_planes [2][M][K] fp16: hi, lo * 2^11 -- written by the producing kernel or
// dh_split_act_f32x): no split pass in front of the MFMAs.  The result goes to C (fp32) and / or c_planes [2][M][N] (either may be NULL:
// a layer whose only consumer is the next GEMM -- relu(fc_1) -- writes planes only).  Bit-identical to dh_linear_f32x on the same values.
extern "C" int dh_linear_f32xp_wreg(const void* a_planes, const void* w_packed, const float* bias, const float* residual, int ldres,
                                    float* C, int ldc, void* c_planes, int M, int N, int K, int relu, void* stream) {
    DH_REQUIRE(a_planes && w_packed && bias && (C || c_planes) && dh_linear_f32x_wreg_supported(M, N, K) && (!C || ldc >= N));
    DH_REQUIRE((!C || ((ldc % 4) == 0 && ((uintptr_t)C % 16) == 0)) && ((uintptr_t)a_planes % 16) == 0 && ((uintptr_t)w_packed % 16) == 0 &&
               (!c_planes || ((uintptr_t)c_planes % 16) == 0) && ((uintptr_t)bias % 16) == 0 &&
               (!residual || (ldres >= N && (ldres % 4) == 0 && ((uintptr_t)residual % 16) == 0)));
    DH_REQUIRE((size_t)M * K * 2 < (1ull << 31));
    FwParams p{};
    p.Ap = (const uint16_t*)a_planes; p.a_plane_b = (unsigned)((size_t)M * K * 2); p.lda = K;
    p.wh = (const uint4*)w_packed; p.wl = p.wh + (size_t)(K / 32) * (N / 16) * 64;
    p.bias = bias; p.res = residual; p.ldres = ldres; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.relu = relu;
    p.Cp = (uint16_t*)c_planes; p.c_plane = (size_t)M * N; p.ldcp = N;
    dh_prof_set_dims(M, N, K);
    DhProfScope prof("dh_linear_f32x", 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), stream);
    return fw_launch<true>(p, (hipStream_t)stream);
}
