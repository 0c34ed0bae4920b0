// Single-position multi-head attention for incremental decoding (HBM-bound: every key/value row is
// read once per query; arithmetic intensity ~1 flop/byte, so no matrix cores here yet -- the fp32
// path keeps the reference's exact softmax formulation).
//
// Workgroup = (image, head); one wave per beam row of that image, so the <= beam distinct ancestor
// rows the beams share (self-attention) or the image's S patch keys (cross-attention) are fetched
// from HBM once and re-served by the CU's L1 / the XCD's L2 to the sibling waves.
// Phase 1: lane = key position j: 16-byte loads of its own contiguous head slice (full 128-B lines
//          per lane), dot with q (broadcast from LDS), energy = dot/scale, masked keys get -1e8
//          exactly as masked_fill does (transformers.py:110-111) -> wave-shuffle max and sum.
// Phase 2: lane = head dim d: out[d] = sum_j p_j * v_j[d], coalesced 256-B value rows.
#include "common.h"
#include "prof.h"
#include "attn_items.h"

unsigned* dh_f32x_range_flag_of(hipStream_t s);      // gemm_f32x.hip

template <typename T>
struct AttnParams {
    const T* q; int ldq;                  // query rows (compact)
    const T* knew; const T* vnew; int ldnew;   // self: this position's k/v (compact rows)
    T* kc; T* vc;                         // self: cache of one layer [pos][rows_total][D]
    const int32_t* src; int src_ld;       // self: ancestor row per position
    const int32_t* tokens; int tok_ld;    // self: pad masking
    const T* kv; const uint8_t* keymask;  // cross: [n_img*S][2D], [n_img*S]
    T* out;
    int rows_per_img, row_mult, rows_total, L, D, dh, lcap, pad_index;
    float scale;
    int row_si;                           // compact row of (image, w) = img * row_si + w (decode: rows_per_img; prefill chunks: n_pos)
    // fp32 rows, dtype DH_F32_OUT_PLANES: the result is stored as the fp16 planes [2][rows][D] (hi, lo * 2^11) of the split-operand
    // GEMM that consumes it (fc_o: csrc/linear_f32x_wreg.hip) instead of fp32 -- the same numbers that GEMM would split it into
    uint16_t* outp; size_t out_plane; unsigned* range_flag;
};

template <typename T>
__device__ __forceinline__ void out_store1(const AttnParams<T>& p, size_t off, float v) {
    if constexpr (sizeof(T) == 4) {
        if (p.outp) {
            const f16_t h = (f16_t)v;
            const f16_t l = (f16_t)((v - (float)h) * 2048.0f);
            p.outp[off] = __builtin_bit_cast(uint16_t, h);
            p.outp[p.out_plane + off] = __builtin_bit_cast(uint16_t, l);
            if (fabsf(v) >= 65504.0f) atomicOr(p.range_flag, 1u);
            return;
        }
    }
    stf(p.out + off, v);
}
template <typename T>
__device__ __forceinline__ void out_store8(const AttnParams<T>& p, size_t off, const float (&o)[8]) {
    if constexpr (sizeof(T) == 4) {
        if (p.outp) {
            uint32_t h[4], l[4];
            float amax = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const f16_t ha = (f16_t)o[j], hb = (f16_t)o[j + 1];
                const f16_t la = (f16_t)((o[j] - (float)ha) * 2048.0f), lb = (f16_t)((o[j + 1] - (float)hb) * 2048.0f);
                h[j / 2] = (uint32_t)__builtin_bit_cast(uint16_t, ha) | ((uint32_t)__builtin_bit_cast(uint16_t, hb) << 16);
                l[j / 2] = (uint32_t)__builtin_bit_cast(uint16_t, la) | ((uint32_t)__builtin_bit_cast(uint16_t, lb) << 16);
                amax = fmaxf(amax, fmaxf(fabsf(o[j]), fabsf(o[j + 1])));
            }
            *reinterpret_cast<uint4*>(p.outp + off) = make_uint4(h[0], h[1], h[2], h[3]);
            *reinterpret_cast<uint4*>(p.outp + p.out_plane + off) = make_uint4(l[0], l[1], l[2], l[3]);
            if (amax >= 65504.0f) atomicOr(p.range_flag, 1u);
            return;
        }
    }
    store8(p.out + off, o);
}

// Beam row of this wave: a workgroup holds DH_ATTN_RPB (= 16) waves = 16 rows of its image; images with more rows (beam_size > 16,
// beam.py:7-9 allows any beam_size <= top_k) take blockIdx.z row blocks.  Waves past the image's last row compute a copy of
// that row (they still join the block barriers and own an LDS strip) and store nothing.
#define DH_ATTN_RPB 16
#define DH_ATTN_ROW(w, ok)                                                                   \
    const int w##_raw = (int)blockIdx.z * DH_ATTN_RPB + (int)(threadIdx.x >> 6);             \
    const bool ok = w##_raw < p.rows_per_img;                                                \
    const int w = ok ? w##_raw : p.rows_per_img - 1

template <typename T, bool CROSS>
__global__ __launch_bounds__(1024) void attn_decode_kernel(AttnParams<T> p) {
    constexpr int VN = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int img = blockIdx.x, h = blockIdx.y, wl = threadIdx.x >> 6, lane = threadIdx.x & 63;
    DH_ATTN_ROW(w, w_ok);
    const int rc = img * p.row_si + w, rl = rc * p.row_mult;
    const int dh = p.dh, D = p.D, L = p.L, t = L - 1;
    float* qs = smem + (size_t)wl * (dh + 2 * p.lcap);
    float* sc = qs + dh;
    int* ph = reinterpret_cast<int*>(sc + p.lcap);

    for (int d = lane; d < dh; d += 64) qs[d] = ldf(p.q + (size_t)rc * p.ldq + h * dh + d);
    __syncthreads();

    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
        const T* kp;
        bool masked;
        if (CROSS) {
            kp = p.kv + (size_t)(img * L + j) * (2 * D) + h * dh;
            masked = p.keymask[img * L + j] != 0;
        } else {
            int phys = rl;
            if (j < t) {
                phys = p.src[(size_t)rl * p.src_ld + j];
                kp = p.kc + ((size_t)j * p.rows_total + phys) * D + h * dh;
            } else {
                kp = p.knew + (size_t)rc * p.ldnew + h * dh;
            }
            ph[j] = phys;
            masked = (j >= 1) && (p.tokens[(size_t)rl * p.tok_ld + j - 1] == p.pad_index);
        }
        float e = -1e8f;
        if (!masked) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            for (int d = 0; d < dh; d += VN) {
                float kk[VN];
                load16(kp + d, kk);
#pragma unroll
                for (int u = 0; u < VN; u += 4) {
                    const float4 qq = *reinterpret_cast<const float4*>(qs + d + u);
                    a0 = fmaf(kk[u], qq.x, a0); a1 = fmaf(kk[u + 1], qq.y, a1);
                    a2 = fmaf(kk[u + 2], qq.z, a2); a3 = fmaf(kk[u + 3], qq.w, a3);
                }
            }
            e = SmMath<T>::div((a0 + a1) + (a2 + a3), p.scale);
        }
        sc[j] = e;
        mx = fmaxf(mx, e);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float e = SmMath<T>::exp(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < L; j += 64) sc[j] = SmMath<T>::div(sc[j], sum);     // attention weights, as torch.softmax
    __syncthreads();

    for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) {
            const float pj = sc[j];
            if (pj == 0.f) continue;              // masked keys underflow to exactly 0 in fp32
            const T* vp;
            if (CROSS) vp = p.kv + (size_t)(img * L + j) * (2 * D) + D + h * dh;
            else if (j < t) vp = p.vc + ((size_t)j * p.rows_total + ph[j]) * D + h * dh;
            else vp = p.vnew + (size_t)rc * p.ldnew + h * dh;
            acc = fmaf(pj, ldf(vp + d), acc);
        }
        if (!w_ok) continue;
        out_store1(p, (size_t)rc * D + h * dh + d, acc);
        if (!CROSS) {   // append this position to the cache at the row's own logical slot
            p.kc[((size_t)t * p.rows_total + rl) * D + h * dh + d] = p.knew[(size_t)rc * p.ldnew + h * dh + d];
            p.vc[((size_t)t * p.rows_total + rl) * D + h * dh + d] = p.vnew[(size_t)rc * p.ldnew + h * dh + d];
        }
    }
}


// ---- bandwidth-shaped variant (head_dim 32/64/128) ------------------------------------------------------
// A wave still owns one (row, head), but its 64 lanes are arranged as (key slot, 8-dim chunk):
// LPK = DH/8 lanes cover one key's head slice with ONE 16-byte load each (bf16; two for fp32), so a
// wave instruction reads 64/LPK whole key slices (1 KiB for bf16) instead of 64 different lines, and the
// key loop has L/(64/LPK) iterations of independent loads instead of L dependent ones.  Scores are
// reduced across the LPK lanes of a key (xor-shuffles), softmax is the same exp/sum/divide as the
// reference, the weighted value sum is accumulated per (key slot, chunk) and reduced across key slots.
template <typename T, bool CROSS, int DH>
__global__ __launch_bounds__(1024) void attn_decode_fast_kernel(AttnParams<T> p) {
    constexpr int LPK = DH / 8, KPI = 64 / LPK;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int img = blockIdx.x, h = blockIdx.y, wl = threadIdx.x >> 6, lane = threadIdx.x & 63;
    DH_ATTN_ROW(w, w_ok);
    const int rc = img * p.row_si + w, rl = rc * p.row_mult;
    const int D = p.D, L = p.L, t = L - 1;
    const int kg = lane / LPK, dc = lane % LPK;
    float* sc = smem + (size_t)wl * 2 * p.lcap;
    int* ph = reinterpret_cast<int*>(sc + p.lcap);

    float qv[8];
    load8(p.q + (size_t)rc * p.ldq + h * DH + dc * 8, qv);

    float mx = -INFINITY;
    for (int j0 = 0; j0 < L; j0 += KPI) {
        const int j = j0 + kg;
        float e = -INFINITY;
        if (j < L) {
            const T* kp;
            bool masked;
            int phys = rl;
            if (CROSS) {
                kp = p.kv + (size_t)(img * L + j) * (2 * D) + h * DH;
                masked = p.keymask[img * L + j] != 0;
            } else {
                if (j < t) {
                    phys = p.src[(size_t)rl * p.src_ld + j];
                    kp = p.kc + ((size_t)j * p.rows_total + phys) * D + h * DH;
                } else {
                    kp = p.knew + (size_t)rc * p.ldnew + h * DH;
                }
                masked = (j >= 1) && (p.tokens[(size_t)rl * p.tok_ld + j - 1] == p.pad_index);
            }
            float kk[8];
            load8(kp + dc * 8, kk);
            float a = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fmaf(kk[u], qv[u], a);
#pragma unroll
            for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o, 64);
            e = masked ? -1e8f : SmMath<T>::div(a, p.scale);
            if (dc == 0) { sc[j] = e; if (!CROSS) ph[j] = phys; }
        }
        mx = fmaxf(mx, e);
    }
    mx = wave_max(mx);
    __syncthreads();
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float e = SmMath<T>::exp(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wave_sum(sum);
    for (int j = lane; j < L; j += 64) sc[j] = SmMath<T>::div(sc[j], sum);     // attention weights, as torch.softmax
    __syncthreads();

    float o[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) o[u] = 0.f;
#pragma unroll 4
    for (int j0 = 0; j0 < L; j0 += KPI) {
        const int j = j0 + kg;
        if (j < L) {
            const T* vp;
            if (CROSS) vp = p.kv + (size_t)(img * L + j) * (2 * D) + D + h * DH;
            else if (j < t) vp = p.vc + ((size_t)j * p.rows_total + ph[j]) * D + h * DH;
            else vp = p.vnew + (size_t)rc * p.ldnew + h * DH;
            float vv[8];
            load8(vp + dc * 8, vv);
            const float pj = sc[j];
#pragma unroll
            for (int u = 0; u < 8; ++u) o[u] = fmaf(pj, vv[u], o[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        {
            if constexpr (LPK == 8) o[u] = key_slots_sum8(o[u]);
            else {
#pragma unroll
                for (int s2 = LPK; s2 < 64; s2 <<= 1) o[u] += __shfl_xor(o[u], s2, 64);
            }
        }
    if (kg == 0 && w_ok) {
        out_store8(p, (size_t)rc * D + h * DH + dc * 8, o);
        if (!CROSS) {   // append this position to the cache at the row's own logical slot
            copy8(p.kc + ((size_t)t * p.rows_total + rl) * D + h * DH + dc * 8, p.knew + (size_t)rc * p.ldnew + h * DH + dc * 8);
            copy8(p.vc + ((size_t)t * p.rows_total + rl) * D + h * DH + dc * 8, p.vnew + (size_t)rc * p.ldnew + h * DH + dc * 8);
        }
    }
}


// ---- register-resident variant: every K and V slice of the row's history is requested up front --------
// Same (key slot, chunk) lane layout, but the key loop is fully unrolled (NIT iterations cover
// L <= NIT * 64/LPK keys): all K and V loads are issued before the first use, scores stay in
// registers, softmax is reduced with shuffles only -- no LDS, no block barrier, two dependent memory
// round trips per wave (ancestor index -> K/V) instead of four.
template <typename T, bool CROSS, int DH, int NIT>
__global__ __launch_bounds__(1024) void attn_decode_reg_kernel(AttnParams<T> p) {
    constexpr int LPK = DH / 8, KPI = 64 / LPK;
    const int img = blockIdx.x, h = blockIdx.y, lane = threadIdx.x & 63;
    DH_ATTN_ROW(w, w_ok);
    if (!w_ok) return;                                   // (no LDS, no block barrier in this kernel)
    const int rc = img * p.row_si + w, rl = rc * p.row_mult;
    const int D = p.D, L = p.L, t = L - 1;
    const int kg = lane / LPK, dc = lane % LPK;

    const T* kptr[NIT];
    const T* vptr[NIT];
    bool live[NIT], masked[NIT];
    // Index loads first, all of them, with clamped (always valid) indices and no per-slot branches: the ancestor row
    // of every key slot, the token that decides its pad mask / the cross-attention key mask.  Issued back to back they
    // cost ONE memory round trip; loaded inside the per-slot conditionals they cost one round trip EACH (the compiler
    // has to wait before the dependent address arithmetic of that slot): 2.3 us per 8 keys of history.
    int phys[NIT], aux[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) { phys[it] = 0; aux[it] = 0; }
    if (CROSS) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) aux[it] = p.keymask[img * L + min(it * KPI + kg, L - 1)];
    } else if (t > 0) {                                  // wave-uniform
#pragma unroll
        for (int it = 0; it < NIT; ++it) phys[it] = p.src[(size_t)rl * p.src_ld + min(it * KPI + kg, t - 1)];
        if (p.tokens) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) aux[it] = p.tokens[(size_t)rl * p.tok_ld + min(max(it * KPI + kg - 1, 0), t - 1)];
        }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = it * KPI + kg;
        live[it] = j < L;
        masked[it] = false;
        kptr[it] = p.q; vptr[it] = p.q;              // any valid address for dead slots (never loaded)
        if (live[it]) {
            if (CROSS) {
                kptr[it] = p.kv + (size_t)(img * L + j) * (2 * D) + h * DH + dc * 8;
                vptr[it] = kptr[it] + D;
                masked[it] = aux[it] != 0;
            } else if (j < t) {
                const size_t off = ((size_t)j * p.rows_total + phys[it]) * D + h * DH + dc * 8;
                kptr[it] = p.kc + off; vptr[it] = p.vc + off;
            } else {
                kptr[it] = p.knew + (size_t)rc * p.ldnew + h * DH + dc * 8;
                vptr[it] = p.vnew + (size_t)rc * p.ldnew + h * DH + dc * 8;
            }
            if (!CROSS) masked[it] = (j >= 1) && p.tokens && (aux[it] == p.pad_index);
        }
    }
    Raw8<T> kr[NIT], vr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) if (live[it]) raw_load(kptr[it], kr[it]);
#pragma unroll
    for (int it = 0; it < NIT; ++it) if (live[it]) raw_load(vptr[it], vr[it]);
    float qv[8];
    load8(p.q + (size_t)rc * p.ldq + h * DH + dc * 8, qv);
    __builtin_amdgcn_sched_barrier(0);                 // every K, V and q load is issued before any of the arithmetic below

    float e[NIT];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        e[it] = -INFINITY;
        if (live[it]) {
            float kk[8];
            raw_unpack(kr[it], kk);
            float a = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) a = fmaf(kk[u], qv[u], a);
            if (LPK == 8) a = sum8(a);                // DPP fold over the row's 8 chunk lanes
            else {
#pragma unroll
                for (int o = 1; o < LPK; o <<= 1) a += __shfl_xor(a, o, 64);
            }
            e[it] = masked[it] ? -1e8f : SmMath<T>::div(a, p.scale);
        }
        mx = fmaxf(mx, e[it]);
    }
    mx = wave_max(mx);                                                              // across key slots
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        e[it] = live[it] ? SmMath<T>::exp(e[it] - mx) : 0.f;
        sum += e[it];
    }
    // every chunk lane holds its key's e: the wave sum counts each key LPK times
    sum = wave_sum(sum) / (float)LPK;
    float o8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) o8[u] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        if (live[it]) {
            float vv[8];
            raw_unpack(vr[it], vv);
            const float pj = SmMath<T>::div(e[it], sum);                                             // attention weight, as torch.softmax
#pragma unroll
            for (int u = 0; u < 8; ++u) o8[u] = fmaf(pj, vv[u], o8[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        {
            if constexpr (LPK == 8) o8[u] = key_slots_sum8(o8[u]);
            else {
#pragma unroll
                for (int s2 = LPK; s2 < 64; s2 <<= 1) o8[u] += __shfl_xor(o8[u], s2, 64);
            }
        }
    if (kg == 0) {
        out_store8(p, (size_t)rc * D + h * DH + dc * 8, o8);
        if (!CROSS) {
            copy8(p.kc + ((size_t)t * p.rows_total + rl) * D + h * DH + dc * 8, p.knew + (size_t)rc * p.ldnew + h * DH + dc * 8);
            copy8(p.vc + ((size_t)t * p.rows_total + rl) * D + h * DH + dc * 8, p.vnew + (size_t)rc * p.ldnew + h * DH + dc * 8);
        }
    }
}


// ---- cross-attention through LDS ---------------------------------------------------------------------------
// The S patch keys/values of an (image, head) are shared by all its beams (transformers.py:544).  The
// workgroup (one wave per beam) stages that K|V head slice ONCE in LDS with coalesced 16-byte loads; then
// lane = key: every lane owns one key row (row stride padded by 16 B: conflict-free 16-byte LDS reads) and
// computes its full dot product against the broadcast query -- no cross-lane traffic for QK^T; softmax is two
// wave reductions; for PV the two half-waves take even / odd keys and each lane owns a pair of head dims.
template <typename T, int HPB>          // HPB heads per workgroup: fewer, fatter waves (wave dispatch is the floor)
__global__ __launch_bounds__(1024) void attn_cross_lds_kernel(AttnParams<T> p) {
    constexpr int DH = 64, VN = Vec16<T>::N, CPR = DH / VN;          // 16-byte chunks per row
    constexpr int ROW = DH + VN;                                       // padded row (elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int img = blockIdx.x, h0 = blockIdx.y * HPB, wl = threadIdx.x >> 6, lane = threadIdx.x & 63;
    DH_ATTN_ROW(w, w_ok);
    const int nthreads = blockDim.x, S = p.L, D = p.D;
    T* ks = reinterpret_cast<T*>(smem_raw);                            // [HPB][S][ROW]
    T* vs = ks + (size_t)HPB * S * ROW;                                // [HPB][S][ROW]
    float* fbase = reinterpret_cast<float*>(vs + (size_t)HPB * S * ROW);   // per wave: q[HPB*64] then p[64]
    float* qs = fbase + wl * (HPB * 64 + 64);
    float* ps = qs + HPB * 64;
    const int per_head = S * CPR;
    for (int c = threadIdx.x; c < HPB * per_head * 2; c += nthreads) {
        const int isv = c >= HPB * per_head, cc = isv ? c - HPB * per_head : c;
        const int hh = cc / per_head, r = cc - hh * per_head, j = r / CPR, ch = r - j * CPR;
        const uint4 val = *reinterpret_cast<const uint4*>(p.kv + (size_t)(img * S + j) * (2 * D) + (isv ? D : 0) + (h0 + hh) * DH + ch * VN);
        *reinterpret_cast<uint4*>((isv ? vs : ks) + ((size_t)hh * S + j) * ROW + ch * VN) = val;
    }
    const int rc = img * p.row_si + w;
#pragma unroll
    for (int hh = 0; hh < HPB; ++hh) qs[hh * 64 + lane] = ldf(p.q + (size_t)rc * p.ldq + (h0 + hh) * DH + lane);
    const bool masked = lane < S && p.keymask[img * S + lane] != 0;
    __syncthreads();
    const int d2 = (lane & 31) * 2, par = lane >> 5;
#pragma unroll 1
    for (int hh = 0; hh < HPB; ++hh) {
        const T* kh = ks + (size_t)hh * S * ROW;
        const T* vh = vs + (size_t)hh * S * ROW;
        float e = -INFINITY;
        if (lane < S) {
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int ch = 0; ch < CPR; ++ch) {
                float kk[VN];
                load16(kh + (size_t)lane * ROW + ch * VN, kk);
#pragma unroll
                for (int u = 0; u < VN; u += 4) {
                    const float4 qq = *reinterpret_cast<const float4*>(qs + hh * 64 + ch * VN + u);
                    a0 = fmaf(kk[u], qq.x, a0); a1 = fmaf(kk[u + 1], qq.y, a1);
                    a2 = fmaf(kk[u + 2], qq.z, a2); a3 = fmaf(kk[u + 3], qq.w, a3);
                }
            }
            e = masked ? -1e8f : SmMath<T>::div((a0 + a1) + (a2 + a3), p.scale);
        }
        const float mx = wave_max(e);
        const float ex = lane < S ? SmMath<T>::exp(e - mx) : 0.f;
        const float sum = wave_sum(ex);
        ps[lane] = SmMath<T>::div(ex, sum);                                           // attention weights, as torch.softmax
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");         // ps is wave-private: order write -> reads
        // PV: lane l -> dims 2*(l&31), +1 ; half-wave (l>>5) takes keys of its parity
        float o0 = 0.f, o1 = 0.f;
#pragma unroll 5
        for (int j = par; j < S; j += 2) {
            const float pj = ps[j];
            o0 = fmaf(pj, ldf(vh + (size_t)j * ROW + d2), o0);
            o1 = fmaf(pj, ldf(vh + (size_t)j * ROW + d2 + 1), o1);
        }
        o0 += __shfl_xor(o0, 32, 64);
        o1 += __shfl_xor(o1, 32, 64);
        if (par == 0 && w_ok) {
            out_store1(p, (size_t)rc * D + (h0 + hh) * DH + d2, o0);
            out_store1(p, (size_t)rc * D + (h0 + hh) * DH + d2 + 1, o1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
}

template <typename T, bool CROSS>
static bool launch_fast(AttnParams<T>& p, int n_img, int n_heads, int rows_per_img, hipStream_t s) {
    const int rpb = rows_per_img < DH_ATTN_RPB ? rows_per_img : DH_ATTN_RPB;
    const size_t lds = (size_t)rpb * 2 * p.lcap * sizeof(float);
    const dim3 grid(n_img, n_heads, dh_cdiv(rows_per_img, DH_ATTN_RPB)), block(64 * rpb);
    if (p.dh == 64 && p.L <= (sizeof(T) == 2 ? 56 : 40)) {    // the caption models' shape: whole history in registers
        if (p.L <= 16) hipLaunchKernelGGL((attn_decode_reg_kernel<T, CROSS, 64, 2>), grid, block, 0, s, p);
        else if (p.L <= 40) hipLaunchKernelGGL((attn_decode_reg_kernel<T, CROSS, 64, 5>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn_decode_reg_kernel<T, CROSS, 64, 7>), grid, block, 0, s, p);
        return true;
    }
    if (p.dh == 64) hipLaunchKernelGGL((attn_decode_fast_kernel<T, CROSS, 64>), grid, block, lds, s, p);
    else if (p.dh == 128) hipLaunchKernelGGL((attn_decode_fast_kernel<T, CROSS, 128>), grid, block, lds, s, p);
    else if (p.dh == 32) hipLaunchKernelGGL((attn_decode_fast_kernel<T, CROSS, 32>), grid, block, lds, s, p);
    else return false;
    return true;
}

template <typename T>
static void launch_self(const void* qkv, void* kcache, void* vcache, const int32_t* src, int src_ld,
                        const int32_t* tokens, int tok_ld, void* out, int n_img, int rows_per_img, int row_mult,
                        int rows_total, int t, int D, int n_heads, float scale, int pad_index, hipStream_t s, bool planes = false) {
    AttnParams<T> p{};
    if (planes) { p.outp = (uint16_t*)out; p.out_plane = (size_t)n_img * rows_per_img * D; p.range_flag = dh_f32x_range_flag_of(s); }
    p.q = (const T*)qkv; p.ldq = 3 * D;
    p.knew = (const T*)qkv + D; p.vnew = (const T*)qkv + 2 * D; p.ldnew = 3 * D;
    p.kc = (T*)kcache; p.vc = (T*)vcache; p.src = src; p.src_ld = src_ld;
    p.tokens = tokens; p.tok_ld = tok_ld; p.out = (T*)out;
    p.rows_per_img = rows_per_img; p.row_mult = row_mult; p.rows_total = rows_total; p.row_si = rows_per_img;
    p.L = t + 1; p.D = D; p.dh = D / n_heads; p.lcap = (t + 1 + 3) & ~3; p.pad_index = pad_index; p.scale = scale;
    if (launch_fast<T, false>(p, n_img, n_heads, rows_per_img, s)) return;
    const int rpb = rows_per_img < DH_ATTN_RPB ? rows_per_img : DH_ATTN_RPB;
    const size_t lds = (size_t)rpb * (p.dh + 2 * p.lcap) * sizeof(float);
    hipLaunchKernelGGL((attn_decode_kernel<T, false>), dim3(n_img, n_heads, dh_cdiv(rows_per_img, DH_ATTN_RPB)), dim3(64 * rpb), lds, s, p);
}

extern "C" int dh_attn_self_decode(const void* qkv, void* kcache, void* vcache, const int32_t* src, int src_ld,
                                   const int32_t* tokens, int tok_ld, void* out, int n_img, int rows_per_img,
                                   int row_mult, int rows_total, int t, int D, int n_heads, float scale,
                                   int pad_index, int dtype, void* stream) {
    DH_REQUIRE(qkv && kcache && vcache && out && n_img > 0 && rows_per_img > 0 && rows_per_img <= DH_BEAM_MAX_BEAMS);
    DH_REQUIRE(t >= 0 && n_heads > 0 && D % n_heads == 0 && ((D / n_heads) % 8) == 0);
    DH_REQUIRE((t == 0 || src) && (t == 0 || pad_index < 0 || tokens));
    const double esz_ = dtype == DH_F32 ? 4.0 : 2.0;
    DhProfScope prof("dh_attn_self_decode", 4.0 * n_img * rows_per_img * (t + 1) * D,
                     esz_ * n_img * rows_per_img * ((t + 1) * 2.0 * D + 2.0 * D), stream);
    if (dtype == DH_F32_OUT_PLANES) {     // fp32 rows, the result as split planes [2][rows][D] (see AttnParams)
        DH_REQUIRE(((uintptr_t)out % 16) == 0 && dh_f32x_range_flag_of((hipStream_t)stream));
        launch_self<float>(qkv, kcache, vcache, src, src_ld, tokens, tok_ld, out, n_img, rows_per_img, row_mult, rows_total, t, D, n_heads, scale,
                           pad_index, (hipStream_t)stream, true);
        DH_LAUNCH_CHECK();
    }
    DH_DISPATCH_T(dtype, launch_self<T>(qkv, kcache, vcache, src, src_ld, tokens, tok_ld, out, n_img, rows_per_img,
                                        row_mult, rows_total, t, D, n_heads, scale, pad_index, (hipStream_t)stream));
    DH_LAUNCH_CHECK();
}

template <typename T>
static void launch_cross(const void* q, int ldq, const void* kv, const uint8_t* keymask, void* out, int n_img,
                         int rows_per_img, int S, int D, int n_heads, float scale, hipStream_t s, int row_si = 0, bool planes = false) {
    AttnParams<T> p{};
    if (planes) { p.outp = (uint16_t*)out; p.out_plane = (size_t)n_img * rows_per_img * D; p.range_flag = dh_f32x_range_flag_of(s); }
    p.q = (const T*)q; p.ldq = ldq; p.kv = (const T*)kv; p.keymask = keymask; p.out = (T*)out;
    p.rows_per_img = rows_per_img; p.row_mult = 1; p.rows_total = 0; p.row_si = row_si > 0 ? row_si : rows_per_img;
    p.L = S; p.D = D; p.dh = D / n_heads; p.lcap = (S + 3) & ~3; p.pad_index = -1; p.scale = scale;
    if (p.dh == 64 && S <= 64) {                         // the caption models' shape: K|V staged once per (image, head group)
        constexpr int VN = Vec16<T>::N;
        const int hpb = 1;     // measured: 1 head per workgroup 20 us, 4 heads 24 us per launch (per-wave latency chain dominates)
        const int rpb = rows_per_img < DH_ATTN_RPB ? rows_per_img : DH_ATTN_RPB;
        const size_t lds = (size_t)2 * hpb * S * (64 + VN) * sizeof(T) + (size_t)rpb * (hpb * 64 + 64) * sizeof(float);
        const dim3 grid(n_img, n_heads / hpb, dh_cdiv(rows_per_img, DH_ATTN_RPB)), block(64 * rpb);
        if (hpb == 4) hipLaunchKernelGGL((attn_cross_lds_kernel<T, 4>), grid, block, lds, s, p);
        else if (hpb == 2) hipLaunchKernelGGL((attn_cross_lds_kernel<T, 2>), grid, block, lds, s, p);
        else hipLaunchKernelGGL((attn_cross_lds_kernel<T, 1>), grid, block, lds, s, p);
        return;
    }
    if (launch_fast<T, true>(p, n_img, n_heads, rows_per_img, s)) return;
    const int rpb = rows_per_img < DH_ATTN_RPB ? rows_per_img : DH_ATTN_RPB;
    const size_t lds = (size_t)rpb * (p.dh + 2 * p.lcap) * sizeof(float);
    hipLaunchKernelGGL((attn_decode_kernel<T, true>), dim3(n_img, n_heads, dh_cdiv(rows_per_img, DH_ATTN_RPB)), dim3(64 * rpb), lds, s, p);
}

extern "C" int dh_attn_cross_decode(const void* q, int ldq, const void* kv, const uint8_t* keymask, void* out,
                                    int n_img, int rows_per_img, int S, int D, int n_heads, float scale,
                                    int dtype, void* stream) {
    DH_REQUIRE(q && kv && keymask && out && n_img > 0 && rows_per_img > 0 && rows_per_img <= DH_BEAM_MAX_BEAMS);
    DH_REQUIRE(S > 0 && n_heads > 0 && D % n_heads == 0 && ((D / n_heads) % 8) == 0 && ldq >= D);
    const double esz_ = dtype == DH_F32 ? 4.0 : 2.0;
    DhProfScope prof("dh_attn_cross_decode", 4.0 * n_img * rows_per_img * S * D,
                     esz_ * n_img * (S * 2.0 * D + rows_per_img * 2.0 * D), stream);
    if (dtype == DH_F32_OUT_PLANES) {
        DH_REQUIRE(((uintptr_t)out % 16) == 0 && dh_f32x_range_flag_of((hipStream_t)stream));
        launch_cross<float>(q, ldq, kv, keymask, out, n_img, rows_per_img, S, D, n_heads, scale, (hipStream_t)stream, 0, true);
        DH_LAUNCH_CHECK();
    }
    DH_DISPATCH_T(dtype, launch_cross<T>(q, ldq, kv, keymask, out, n_img, rows_per_img, S, D, n_heads, scale,
                                         (hipStream_t)stream));
    DH_LAUNCH_CHECK();
}


// ---- teacher-forced (prefill) attention: every position of every sequence in one launch ---------------------------
// Rows are sequence-major: row n * n_pos + t holds position t of sequence n, qkv [rows, 3D] comes from ONE batched
// projection GEMM.  A wave owns (row, head) as in attn_decode_reg_kernel; the history of position t is simply rows
// n * n_pos + 0 .. t of the same matrix (causal mask = the key count), so there is no KV cache and no ancestor table.
template <typename T, int NIT>
__global__ __launch_bounds__(256) void attn_self_prefill_kernel(const T* __restrict__ qkv, const int32_t* __restrict__ tokens,
                                                                 int tok_ld, T* __restrict__ out, int rows, int n_pos, int D,
                                                                 float scale, int pad_index) {
    constexpr int DH = 64, LPK = DH / 8, KPI = 64 / LPK;
    const int rc = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, lane = threadIdx.x & 63;
    if (rc >= rows) return;
    const int n = rc / n_pos, t = rc - n * n_pos, L = t + 1;
    const int kg = lane / LPK, dc = lane % LPK;
    const size_t ld = 3 * (size_t)D;
    int aux[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) aux[it] = 0;
    if (tokens && t > 0) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) aux[it] = tokens[(size_t)n * tok_ld + min(max(it * KPI + kg - 1, 0), t - 1)];
    }
    bool live[NIT], masked[NIT];
    Raw8<T> kr[NIT], vr[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int j = it * KPI + kg;
        live[it] = j < L;
        masked[it] = live[it] && j >= 1 && tokens && aux[it] == pad_index;
        const T* kp = qkv + ((size_t)n * n_pos + min(j, t)) * ld + D + h * DH + dc * 8;
        raw_load(kp, kr[it]);
        raw_load(kp + D, vr[it]);
    }
    float qv[8];
    load8(qkv + (size_t)rc * ld + h * DH + dc * 8, qv);
    __builtin_amdgcn_sched_barrier(0);
    float e[NIT];
    float mx = -INFINITY;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float kk[8];
        raw_unpack(kr[it], kk);
        float a = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) a = fmaf(kk[u], qv[u], a);
        a = sum8(a);
        e[it] = !live[it] ? -INFINITY : (masked[it] ? -1e8f : SmMath<T>::div(a, scale));
        mx = fmaxf(mx, e[it]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        e[it] = live[it] ? SmMath<T>::exp(e[it] - mx) : 0.f;
        sum += e[it];
    }
    sum = wave_sum(sum) / (float)LPK;
    float o8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) o8[u] = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        float vv[8];
        raw_unpack(vr[it], vv);
        const float pj = SmMath<T>::div(e[it], sum);
#pragma unroll
        for (int u = 0; u < 8; ++u) o8[u] = fmaf(pj, vv[u], o8[u]);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
        {
            if constexpr (LPK == 8) o8[u] = key_slots_sum8(o8[u]);
            else {
#pragma unroll
                for (int s2 = LPK; s2 < 64; s2 <<= 1) o8[u] += __shfl_xor(o8[u], s2, 64);
            }
        }
    if (kg == 0) store8(out + (size_t)rc * D + h * DH + dc * 8, o8);
}

extern "C" int dh_attn_self_prefill(const void* qkv, const int32_t* tokens, int tok_ld, void* out, int n_seq, int n_pos,
                                    int D, int n_heads, float scale, int pad_index, int dtype, void* stream) {
    DH_REQUIRE(qkv && out && n_seq > 0 && n_pos > 0 && n_heads > 0 && D == n_heads * 64);
    DH_REQUIRE(n_pos <= (dtype == DH_F32 ? 40 : 56) && (pad_index < 0 || n_pos == 1 || tokens));
    const int rows = n_seq * n_pos;
    const double esz_ = dtype == DH_F32 ? 4.0 : 2.0;
    DhProfScope prof("dh_attn_self_prefill", 2.0 * rows * (n_pos + 1) * D, esz_ * rows * 4.0 * D, stream);
    const dim3 grid(dh_cdiv(rows, 4), n_heads), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == DH_F32)
        hipLaunchKernelGGL((attn_self_prefill_kernel<float, 5>), grid, block, 0, s, (const float*)qkv, tokens, tok_ld, (float*)out,
                           rows, n_pos, D, scale, pad_index);
    else if (dtype == DH_BF16)
        hipLaunchKernelGGL((attn_self_prefill_kernel<bf16_t, 7>), grid, block, 0, s, (const bf16_t*)qkv, tokens, tok_ld,
                           (bf16_t*)out, rows, n_pos, D, scale, pad_index);
    else if (dtype == DH_F16)
        hipLaunchKernelGGL((attn_self_prefill_kernel<f16_t, 7>), grid, block, 0, s, (const f16_t*)qkv, tokens, tok_ld,
                           (f16_t*)out, rows, n_pos, D, scale, pad_index);
    else return DH_ERR_UNSUPPORTED;
    DH_LAUNCH_CHECK();
}

// Cross-attention for sequence-major prefill rows (row n * n_pos + t): the decode kernels take at most
// DH_BEAM_MAX_BEAMS query rows per image and launch, so the positions go in chunks of that many.
extern "C" int dh_attn_cross_prefill(const void* q, int ldq, const void* kv, const uint8_t* keymask, void* out, int n_img,
                                     int n_pos, int S, int D, int n_heads, float scale, int dtype, void* stream) {
    DH_REQUIRE(q && kv && keymask && out && n_img > 0 && n_pos > 0 && S > 0 && n_heads > 0 && D % n_heads == 0);
    DH_REQUIRE(((D / n_heads) % 8) == 0 && ldq >= D);
    const size_t esz = dtype == DH_F32 ? 4 : 2;
    DhProfScope prof("dh_attn_cross_prefill", 4.0 * n_img * n_pos * S * D, (double)esz * n_img * (S * 2.0 * D + n_pos * 2.0 * D), stream);
    for (int t0 = 0; t0 < n_pos; t0 += DH_ATTN_RPB) {
        const int cnt = n_pos - t0 < DH_ATTN_RPB ? n_pos - t0 : DH_ATTN_RPB;
        const char* qc = (const char*)q + (size_t)t0 * ldq * esz;
        char* oc = (char*)out + (size_t)t0 * D * esz;
        DH_DISPATCH_T(dtype, launch_cross<T>(qc, ldq, kv, keymask, oc, n_img, cnt, S, D, n_heads, scale, (hipStream_t)stream, n_pos));
    }
    DH_LAUNCH_CHECK();
}


// ---- MultiHeadAttentionLayer.forward with an explicit boolean mask (transformers.py:82-127) -----------------------
// The module-level API of the reference: q, k, v are the projected [bs*L, D] matrices (fc_q / fc_k / fc_v outputs, row
// b*L + t), mask is the caller's [bs, L, L] boolean tensor (1 = masked_fill(-1e8)) or NULL; out = softmax(q k^T / scale) v
// with heads merged back, [bs*L, D].  One wave per (query row, head): lane = key for the scores (LDS), lane = head
// dim for the weighted value sum.  Any head dim, L <= 4096.  Not a hot-path kernel (generate / forward use the
// cached / prefill kernels above); exact softmax formulation of the reference.
template <typename T>
__global__ __launch_bounds__(256) void attn_masked_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ k, int ldk,
                                                           const T* __restrict__ v, int ldv, const uint8_t* __restrict__ mask,
                                                           T* __restrict__ out, int rows, int L, int D, int dh, float scale) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + w, h = blockIdx.y;
    float* qs = smem + (size_t)w * (dh + L);
    float* sc = qs + dh;
    const bool live = r < rows;
    const int b = live ? r / L : 0, t = live ? r - b * L : 0;
    if (live)
        for (int d = lane; d < dh; d += 64) qs[d] = ldf(q + (size_t)r * ldq + h * dh + d);
    __syncthreads();
    if (!live) return;
    float mx = -INFINITY;
    for (int j = lane; j < L; j += 64) {
        const T* kp = k + (size_t)(b * L + j) * ldk + h * dh;
        float a = 0.f;
        for (int d = 0; d < dh; ++d) a = fmaf(ldf(kp + d), qs[d], a);
        float e = a / scale;
        if (mask && mask[((size_t)b * L + t) * L + j]) e = -1e8f;
        sc[j] = e;
        mx = fmaxf(mx, e);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
    sum = wave_sum(sum);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // sc is wave-private: order the writes before the reads below
    for (int d = lane; d < dh; d += 64) {
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(sc[j] / sum, ldf(v + (size_t)(b * L + j) * ldv + h * dh + d), acc);
        stf(out + (size_t)r * D + h * dh + d, acc);
    }
}

extern "C" int dh_attn_masked(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const uint8_t* mask,
                              void* out, int bs, int L, int D, int n_heads, float scale, int dtype, void* stream) {
    DH_REQUIRE(q && k && v && out && bs > 0 && L > 0 && L <= 4096 && n_heads > 0 && D % n_heads == 0);
    DH_REQUIRE(ldq >= D && ldk >= D && ldv >= D);
    const int rows = bs * L, dh = D / n_heads;
    DhProfScope prof("dh_attn_masked", 4.0 * rows * L * D, 0.0, stream);
    const size_t lds = (size_t)4 * (dh + L) * sizeof(float);
    if (lds > 64 * 1024) return DH_ERR_UNSUPPORTED;       // the default dynamic-LDS limit: L <= 4096 - dh (head dim 64: 4032 keys)
    DH_DISPATCH_T(dtype, hipLaunchKernelGGL(attn_masked_kernel<T>, dim3(dh_cdiv(rows, 4), n_heads), dim3(256), lds, (hipStream_t)stream,
                                            (const T*)q, ldq, (const T*)k, ldk, (const T*)v, ldv, mask, (T*)out, rows, L, D, dh, scale));
    DH_LAUNCH_CHECK();
}


// ---- cross-attention on the matrix cores, operands straight from HBM into MFMA fragments (16-bit paths) -----------------
// The S <= 64 patch keys / values of an (image, head) are shared by all its beams (transformers.py:544).  Once per batch
// and layer dh_attn_cross_pack re-lays the encoder K|V rows out per (image, head) in exactly the order an MFMA fragment
// load wants them:
//   Kp [img][head][64 keys][64 d]            key-major (keys >= S are zero rows)
//   Vt [img][head][64 d][64 key slots]       d-major (V transposed), key slot ks = 32*kk + 8*lq + e holds key
//                                            16*(2*kk + (e >> 2)) + 4*lq + (e & 3)
// With that, ONE wave handles one (image, head) with no LDS and no barrier: every lane's 16-byte loads ARE its MFMA
// operands (8 K fragments, 8 V^T fragments, 2 q fragments: all requested up front, one memory round trip),
//   S[m][key]  = sum_d q[m][d] K[key][d]      4 key tiles x 2 k-steps of v_mfma_f32_16x16x32 (rows m = beams, <= 16)
//   softmax over the keys of a row: 16 values in the lane's own accumulators + the 3 other lanes of the row's column quad
//   out[m][d]  = sum_ks P[m][ks] V^T[d][ks]   the lane's OWN probabilities, rounded to the 16-bit type, are its B operand:
//                                            the key-slot permutation above is chosen so that no cross-lane move is needed
// Masked keys get -1e8 exactly as masked_fill does (transformers.py:110-111); keys >= S do not exist (weight 0).
template <typename T>
__global__ __launch_bounds__(256) void attn_cross_pack_kernel(const T* __restrict__ kv, T* __restrict__ kp, T* __restrict__ vt,
                                                               int S, int D, int H, int dperm) {
    // one workgroup per (image, head): stage the V head slice [S][64] in LDS, write K rows straight through and V transposed
    __shared__ uint16_t vs[64][64 + 2];
    const int img = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    const uint16_t* src = reinterpret_cast<const uint16_t*>(kv);
    uint16_t* kd = reinterpret_cast<uint16_t*>(kp) + ((size_t)img * H + h) * 4096;
    uint16_t* vd = reinterpret_cast<uint16_t*>(vt) + ((size_t)img * H + h) * 4096;
    for (int c = tid; c < 64 * 8; c += 256) {                   // 16-byte chunks of the [64 keys][64 d] K tile
        const int key = c >> 3, ch = c & 7;
        uint4 val = make_uint4(0u, 0u, 0u, 0u), vv = val;
        if (key < S) {
            const uint16_t* row = src + (size_t)(img * S + key) * (2 * D) + h * 64 + ch * 8;
            val = *reinterpret_cast<const uint4*>(row);
            vv = *reinterpret_cast<const uint4*>(row + D);
            if (dperm) {
                // head-dim slots in the order a q row leaves a 16 x 16 MFMA accumulator tile (the layout of rounds 2-5's fused fc_q + attention
                // launch, kept so that the summation order over the head dimension -- and every 16-bit token -- stays what it was):
                // slot 32 kk + 8 lq + e holds dim 16 (2 kk + (e >> 2)) + 4 lq + (e & 3) -- two runs of 4 consecutive dims
                const int kk = ch >> 2, lq4 = ch & 3;
                const uint16_t* r0 = src + (size_t)(img * S + key) * (2 * D) + h * 64;
                const uint2 lo = *reinterpret_cast<const uint2*>(r0 + 32 * kk + 4 * lq4);
                const uint2 hi = *reinterpret_cast<const uint2*>(r0 + 32 * kk + 16 + 4 * lq4);
                val = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }
        }
        *reinterpret_cast<uint4*>(kd + key * 64 + ch * 8) = val;
        const uint16_t* pv = reinterpret_cast<const uint16_t*>(&vv);
#pragma unroll
        for (int u = 0; u < 8; ++u) vs[key][ch * 8 + u] = pv[u];
    }
    __syncthreads();
    for (int c = tid; c < 64 * 64; c += 256) {                  // Vt[d][ks]
        const int d = c >> 6, ks = c & 63;
        const int kk = ks >> 5, lq = (ks >> 3) & 3, e = ks & 7;
        const int key = 16 * (2 * kk + (e >> 2)) + 4 * lq + (e & 3);
        vd[d * 64 + ks] = vs[key][d];
    }
}

extern "C" int dh_attn_cross_pack(const void* kv, void* kp, void* vt, int n_img, int S, int D, int n_heads, int dperm, int dtype,
                                  void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(kv && kp && vt && n_img > 0 && S > 0 && S <= 64 && n_heads > 0 && D == 64 * n_heads);
    DH_REQUIRE(((uintptr_t)kv % 16) == 0 && ((uintptr_t)kp % 16) == 0 && ((uintptr_t)vt % 16) == 0);
    DhProfScope prof("dh_attn_cross_pack", 0.0, 2.0 * n_img * (2.0 * S * D + 2.0 * 64 * D), stream);
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL(attn_cross_pack_kernel<T>, dim3(n_img, n_heads), dim3(256), 0, (hipStream_t)stream,
                                             (const T*)kv, (T*)kp, (T*)vt, S, D, n_heads, dperm));
    DH_LAUNCH_CHECK();
}

template <typename T>
__global__ __launch_bounds__(256) void attn_cross_mfma_kernel(const T* __restrict__ q, int ldq, const T* __restrict__ kp,
                                                               const T* __restrict__ vt, const uint8_t* __restrict__ keymask,
                                                               T* __restrict__ out, int n_img, int rows_per_img, int row_si, int S,
                                                               int D, int H, float scale, int dperm) {
    const int wid = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (wid >= n_img * H) return;                                  // wave-uniform
    const int img = wid / H, h = wid - img * H;
    const int l15 = lane & 15, lq = lane >> 4;
    const uint16_t* kb = reinterpret_cast<const uint16_t*>(kp) + ((size_t)img * H + h) * 4096;
    const uint16_t* vb = reinterpret_cast<const uint16_t*>(vt) + ((size_t)img * H + h) * 4096;
    // every load of the kernel up front: K / V^T fragments, the row's q fragments, the key-mask bytes of the lane's 16 keys
    uint4 kf[4][2], vf[4][2], qf[2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // key rows >= S are padding whose scores are replaced by -inf: those lanes re-read row S - 1 (a line the wave fetches
            // anyway) instead of pulling the 64 - S zero rows of every (image, head) from HBM (15 of 64 rows at S = 49)
            kf[j][kk] = *reinterpret_cast<const uint4*>(kb + min(16 * j + l15, S - 1) * 64 + 32 * kk + 8 * lq);
            vf[j][kk] = *reinterpret_cast<const uint4*>(vb + (16 * j + l15) * 64 + 32 * kk + 8 * lq);
        }
    const bool live = l15 < rows_per_img;
    const size_t qrow = (size_t)img * row_si + (live ? l15 : 0);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const uint16_t* qp = reinterpret_cast<const uint16_t*>(q) + qrow * ldq + h * 64;
        uint4 t;
        if (dperm) {            // K was packed with permuted head-dim slots: read q in the same slot order (two runs of 4 dims)
            const uint2 lo = *reinterpret_cast<const uint2*>(qp + 32 * kk + 4 * lq), hi = *reinterpret_cast<const uint2*>(qp + 32 * kk + 16 + 4 * lq);
            t = make_uint4(lo.x, lo.y, hi.x, hi.y);
        } else {
            t = *reinterpret_cast<const uint4*>(qp + 32 * kk + 8 * lq);
        }
        qf[kk] = live ? t : make_uint4(0u, 0u, 0u, 0u);
    }
    const uint64_t kbits = __ballot(keymask[img * S + min(lane, S - 1)] != 0);     // bit key = that key is masked (one byte per lane)
    uint16_t* orow = reinterpret_cast<uint16_t*>(out) + qrow * D + h * 64;
    cross_core<T>(kf, vf, qf, kbits, S, scale, live, orow, lq);
}

// q [n_img * row_si rows, ldq] (image i's rows start at row i * row_si; rows_per_img <= 16 of them are used), out likewise
// [.., D]; kp / vt from dh_attn_cross_pack.
static int launch_cross_packed(const void* q, int ldq, const void* kp, const void* vt, const uint8_t* keymask, void* out, int n_img,
                               int rows_per_img, int row_si, int S, int D, int n_heads, float scale, int dperm, int dtype, hipStream_t s) {
    const int waves = n_img * n_heads;
    DH_DISPATCH_16(dtype, hipLaunchKernelGGL(attn_cross_mfma_kernel<T>, dim3(dh_cdiv(waves, 4)), dim3(256), 0, s, (const T*)q, ldq,
                                             (const T*)kp, (const T*)vt, keymask, (T*)out, n_img, rows_per_img, row_si, S, D, n_heads, scale, dperm));
    return DH_OK;
}

extern "C" int dh_attn_cross_decode_packed(const void* q, int ldq, const void* kp, const void* vt, const uint8_t* keymask, void* out,
                                           int n_img, int rows_per_img, int S, int D, int n_heads, float scale, int dperm,
                                           int dtype, void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(q && kp && vt && keymask && out && n_img > 0 && rows_per_img > 0 && rows_per_img <= 16);
    DH_REQUIRE(S > 0 && S <= 64 && n_heads > 0 && D == 64 * n_heads && ldq >= D && (ldq % 8) == 0);
    DH_REQUIRE(((uintptr_t)q % 16) == 0 && ((uintptr_t)kp % 16) == 0 && ((uintptr_t)vt % 16) == 0 && ((uintptr_t)out % 8) == 0);
    DhProfScope prof("dh_attn_cross_decode", 4.0 * n_img * rows_per_img * S * D, 2.0 * n_img * (S * 2.0 * D + rows_per_img * 2.0 * D), stream);
    const int rc = launch_cross_packed(q, ldq, kp, vt, keymask, out, n_img, rows_per_img, rows_per_img, S, D, n_heads, scale, dperm,
                                       dtype, (hipStream_t)stream);
    if (rc != DH_OK) return rc;
    DH_LAUNCH_CHECK();
}

extern "C" int dh_attn_cross_prefill_packed(const void* q, int ldq, const void* kp, const void* vt, const uint8_t* keymask, void* out,
                                            int n_img, int n_pos, int S, int D, int n_heads, float scale, int dperm, int dtype,
                                            void* stream) {
    if (!DH_IS_16BIT(dtype)) return DH_ERR_UNSUPPORTED;
    DH_REQUIRE(q && kp && vt && keymask && out && n_img > 0 && n_pos > 0 && S > 0 && S <= 64 && n_heads > 0 && D == 64 * n_heads);
    DH_REQUIRE(ldq >= D && (ldq % 8) == 0 && ((uintptr_t)q % 16) == 0 && ((uintptr_t)out % 8) == 0);
    DhProfScope prof("dh_attn_cross_prefill", 4.0 * n_img * n_pos * S * D, 2.0 * n_img * (S * 2.0 * D + n_pos * 2.0 * D), stream);
    for (int t0 = 0; t0 < n_pos; t0 += 16) {           // 16 query rows per MFMA tile: the positions go in chunks of 16
        const int cnt = n_pos - t0 < 16 ? n_pos - t0 : 16;
        const char* qc = (const char*)q + (size_t)t0 * ldq * 2;
        char* oc = (char*)out + (size_t)t0 * D * 2;
        const int rc = launch_cross_packed(qc, ldq, kp, vt, keymask, oc, n_img, cnt, n_pos, S, D, n_heads, scale, dperm, dtype, (hipStream_t)stream);
        if (rc != DH_OK) return rc;
    }
    DH_LAUNCH_CHECK();
}
