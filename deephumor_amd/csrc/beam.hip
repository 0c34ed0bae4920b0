// Device-side beam-search step (deephumor/models/beam.py), batched over images: no host sync and no
// torch.multinomial in the token loop.
//   beam_row_sample : per (image, beam) row -- k-th-largest threshold by 4-pass radix select over the
//                     logits row (L2-resident after the vocabulary GEMM wrote it), survivor
//                     compaction into LDS, temperature softmax, Exp(1)-race draw of `beam` tokens
//                     (== CPU torch.multinomial without replacement), log-softmax over the picks.
//   beam_select     : per image -- candidate list with ended-beam dedup, second race, in-place
//                     rewrite of tokens / scores / ended flags / KV-ancestor table / parent rows.
//   beam_finalize   : per image -- final single draw and output copy.
#include "common.h"
#include "prof.h"

#define CAP DH_BEAM_MAX_SURVIVORS

// order-preserving float -> uint key (larger float <=> larger key; -0.0 and +0.0 share ONE key, as they compare equal in the
// reference's `logits < threshold` (beam.py:34): with distinct keys a row whose k-th largest logit is a zero dropped the zeros of the
// other sign that torch keeps as ties -- found by tools/fuzz_sampler.py on quantised logits)
__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}


// Picks the radix digit holding the k-th largest key: the digit d with (#keys in digits > d) < k <= (... >= d).
// Parallel suffix sum over the 256 bins (wave shuffles + 4 wave totals) instead of a serial scan whose
// dependent LDS reads cost ~10 us per pass.  Must be called by all threads of the block; threads
// 0..255 own one bin each.  `bins` may be 1 or 4 histograms of 256 (summed).
__device__ __forceinline__ void radix_pick_digit(const int* hist, int n_hist, int shift, uint32_t prefix,
                                                 uint32_t* s_prefix, int* s_k, int* wtot) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = *s_k;
    int c = 0, suf = 0;
    if (tid < 256) {
        for (int h = 0; h < n_hist; ++h) c += hist[h * 256 + tid];
        // inclusive suffix sum inside the wave (towards higher lanes): DPP row_shl scan inside the 16-lane rows (a lane
        // shifted in from beyond the row end reads 0), then the totals of the higher rows through v_readlane
        suf = c;
        suf += __builtin_amdgcn_update_dpp(0, suf, 0x101, 0xF, 0xF, true);
        suf += __builtin_amdgcn_update_dpp(0, suf, 0x102, 0xF, 0xF, true);
        suf += __builtin_amdgcn_update_dpp(0, suf, 0x104, 0xF, 0xF, true);
        suf += __builtin_amdgcn_update_dpp(0, suf, 0x108, 0xF, 0xF, true);
        {
            const int t1 = __builtin_amdgcn_readlane(suf, 16), t2 = __builtin_amdgcn_readlane(suf, 32), t3 = __builtin_amdgcn_readlane(suf, 48);
            const int row = lane >> 4;
            suf += row == 0 ? t1 + t2 + t3 : row == 1 ? t2 + t3 : row == 2 ? t3 : 0;
        }
        if (lane == 0) wtot[wave] = suf;
    }
    __syncthreads();
    if (tid < 256) {
        int above = suf - c;                       // bins above me inside my wave
        for (int w = wave + 1; w < 4; ++w) above += wtot[w];
        if (above < k && k <= above + c) { *s_prefix = prefix | ((uint32_t)tid << shift); *s_k = k - above; }
    }
    __syncthreads();
}

// lanes of one wave exchanging data through LDS without a workgroup barrier: keep the compiler from moving LDS accesses
// across the hand-over (the hardware executes a wave's LDS operations in order)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float block_reduce_256(float v, float* red, bool is_max) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = is_max ? fmaxf(red[tid], red[tid + s]) : red[tid] + red[tid + s];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

template <int NT>
__device__ __attribute__((noinline)) void row_overflow(const float* __restrict__ row, int V, uint32_t thr, int rc, int ldl, int rows_per_img, int beam,
                             float temperature, int unk, const float* __restrict__ noise, uint64_t seed,
                             const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* __restrict__ pick_idx,
                             float* __restrict__ pick_val, float* sq, int* si, int* picks, int32_t* __restrict__ err);
__device__ __forceinline__ void pick_store(int32_t* pi, float* pv, size_t at, int32_t idx, float val);

// softmax over a row's survivors: max m, sum s of exp(x - m).  Finite logits give a finite m and s >= 1 (the maximum contributes exp(0));
// a NaN among them makes s NaN, +inf makes m = +inf and s NaN, only-NaN survivors leave m = -inf: the reference's torch.multinomial raises
// on such a row ("probability tensor contains either `inf`, `nan` or element < 0", beam.py:46) -- the samplers flag DH_BEAM_ERR_NONFINITE
// and hand on a finite dummy pick, so that nothing downstream indexes with garbage
__device__ __forceinline__ bool dh_softmax_nonfinite(float m, float s) { return !(s >= 1.0f) || !(fabsf(m) < INFINITY); }

__global__ __launch_bounds__(256) void beam_row_sample_kernel(
    const float* __restrict__ logits, int ldl, int V, int rows_per_img, int beam, int top_k,
    float temperature, int unk, const float* __restrict__ noise, uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0, int step,
    int32_t* __restrict__ pick_idx, float* __restrict__ pick_val, int32_t* __restrict__ err) {
    __shared__ int hist[4][256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_k, s_cnt, wtot[4];
    __shared__ __attribute__((aligned(16))) int idx_a[CAP], idx_b[CAP];
    __shared__ __attribute__((aligned(16))) float val_a[CAP], val_b[CAP], qv[CAP];
    __shared__ float red[256];
    __shared__ int picks[DH_BEAM_MAX_BEAMS];

    const int rc = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const float* row = logits + (size_t)rc * ldl;

    // ---- k-th largest by MSB-first radix select on the order-preserving key --------------------
    if (tid == 0) { s_prefix = 0u; s_k = top_k; s_cnt = 0; }
    uint32_t mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 1024; i += 256) (&hist[0][0])[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        for (int i = tid; i < V; i += 256) {
            const uint32_t key = f2key(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[wave][(key >> shift) & 255u], 1);
        }
        __syncthreads();
        radix_pick_digit(&hist[0][0], 4, shift, prefix, &s_prefix, &s_k, wtot);
        mask |= 0xFFu << shift;
    }
    const uint32_t thr = s_prefix;

    // ---- survivors: logit >= threshold (ties kept, beam.py:34), unk always dropped (:35) -------
    for (int i = tid; i < V; i += 256) {
        const float v = row[i];
        if (f2key(v) >= thr && i != unk) {
            const int p = atomicAdd(&s_cnt, 1);
            if (p < CAP) { idx_a[p] = i; val_a[p] = v; }
        }
    }
    __syncthreads();
    int n = s_cnt;
    if (n > CAP) {                   // more ties at the threshold than the candidate buffers hold: the draw over the row itself
        row_overflow<256>(row, V, thr, rc, ldl, rows_per_img, beam, temperature, unk, noise, seed, seed_ptr, img0, step, pick_idx, pick_val,
                          qv, idx_b, picks, err);
        return;
    }
    if (n == 0) {
        if (tid == 0) atomicOr(err, DH_BEAM_ERR_ALL_FILTERED);
        if (tid < beam) { pick_idx[(size_t)rc * beam + tid] = 0; pick_val[(size_t)rc * beam + tid] = 0.f; }
        return;
    }
    // deterministic order: sort by token index (rank sort, n is ~top_k)
    for (int i = tid; i < n; i += 256) {
        const int me = idx_a[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (idx_a[j] < me);
        idx_b[r] = me; val_b[r] = val_a[i];
    }
    __syncthreads();

    // ---- p = softmax(filtered / T);  q = p / Exp(1) noise -------------------------------------
    float m = -INFINITY;
    for (int i = tid; i < n; i += 256) m = fmaxf(m, val_b[i] / temperature);
    m = block_reduce_256(m, red, true);
    float s = 0.f;
    for (int i = tid; i < n; i += 256) { const float e = expf(val_b[i] / temperature - m); qv[i] = e; s += e; }
    s = block_reduce_256(s, red, false);
    if (dh_softmax_nonfinite(m, s)) {                  // (block-uniform)
        if (tid == 0) atomicOr(err, DH_BEAM_ERR_NONFINITE);
        if (tid < beam) { pick_idx[(size_t)rc * beam + tid] = 0; pick_val[(size_t)rc * beam + tid] = 0.f; }
        return;
    }
    const int img = rc / rows_per_img, rin = rc % rows_per_img;
    for (int i = tid; i < n; i += 256) {
        const float nz = noise ? noise[(size_t)rc * ldl + idx_b[i]]
                               : philox_exp1(seed ^ (seed_ptr ? *seed_ptr : 0ull), (uint32_t)(img0 + img), (uint32_t)step, 0u, (uint32_t)rin, (uint32_t)idx_b[i]);
        qv[i] = (qv[i] / s) / nz;
    }
    if (tid < DH_BEAM_MAX_BEAMS) picks[tid] = -1;
    __syncthreads();
    // top-`beam` of q, descending; ties -> lower token index first (argmax semantics for beam == 1)
    for (int i = tid; i < n; i += 256) {
        const float me = qv[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (qv[j] > me) || (qv[j] == me && j < i);
        if (r < beam) picks[r] = i;
    }
    __syncthreads();
    if (tid == 0) {
        if (n < beam) atomicOr(err, DH_BEAM_ERR_TOO_FEW);
        // log_softmax over the gathered (un-tempered) logits of the picks (beam.py:79)
        float mx = -INFINITY;
        for (int b = 0; b < beam; ++b) if (picks[b] >= 0) mx = fmaxf(mx, val_b[picks[b]]);
        float se = 0.f;
        for (int b = 0; b < beam; ++b) if (picks[b] >= 0) se += expf(val_b[picks[b]] - mx);
        const float lse = logf(se);
        for (int b = 0; b < beam; ++b) {
            const int pi = picks[b];
            pick_idx[(size_t)rc * beam + b] = pi >= 0 ? idx_b[pi] : 0;
            pick_val[(size_t)rc * beam + b] = pi >= 0 ? (val_b[pi] - mx) - lse : -INFINITY;
        }
    }
}


// Exact candidate set of a row for the pre-filtered kernels' overflow case (their bound let more than CAP values
// through -- flat or heavily tied logits): k-th largest key by the 4-pass MSB-first radix select over the whole row
// (as beam_row_sample_kernel), then every value >= it is compacted into idx_a / val_a (*s_cnt = their number; more
// than CAP only if that many logits tie at the threshold).  hist: 4 x 256 ints.  Block-uniform call; synchronised on return.
template <int NT>
__device__ void radix_row_candidates(const float* __restrict__ row, int V, int top_k, int* hist, uint32_t* s_prefix, int* s_k,
                                     int* s_cnt, int* wtot, int* idx_a, float* val_a) {
    const int tid = threadIdx.x, hw = (tid >> 6) & 3;
    __syncthreads();
    if (tid == 0) { *s_prefix = 0u; *s_k = top_k; *s_cnt = 0; }
    uint32_t mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 1024; i += NT) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = *s_prefix;
        for (int i = tid; i < V; i += NT) {
            const uint32_t key = f2key(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[hw * 256 + ((key >> shift) & 255u)], 1);
        }
        __syncthreads();
        radix_pick_digit(hist, 4, shift, prefix, s_prefix, s_k, wtot);
        mask |= 0xFFu << shift;
    }
    const uint32_t thr = *s_prefix;
    for (int i = tid; i < V; i += NT) {
        const float v = row[i];
        if (f2key(v) >= thr) {
            const int p = atomicAdd(s_cnt, 1);
            if (p < CAP) { idx_a[p] = i; val_a[p] = v; }
        }
    }
    __syncthreads();
}

// ---- single-pass variant --------------------------------------------------------------------------
// The logits row is read from HBM exactly ONCE, straight into registers (512 threads x EPT values,
// every load issued up front).  A valid lower bound of the k-th largest value is the k-th largest of
// the 512 per-thread maxima (k <= 512): at least k elements are >= it.  Everything >= that bound
// (about top_k..2*top_k values on real logits) is compacted into LDS and the exact threshold,
// the survivors and the draws are computed there.  If more than CAP values pass the bound (flat / tied
// logits) the kernel re-derives the exact candidate set in place with radix_row_candidates.
__device__ __forceinline__ void pick_store(int32_t* pi, float* pv, size_t at, int32_t idx, float val) { pi[at] = idx; pv[at] = val; }

// More than CAP logits of a row tie at (or exceed) its top-k threshold -- flat or constant logits -- so the survivors do not fit the
// LDS candidate buffers: the draw runs over the ROW in global memory instead (beam.py:32-48 unchanged: every logit >= the threshold
// survives, <unk> dropped, softmax / T, `beam` winners of p / Exp(1), ties to the lower index).  `beam` block-wide arg-max rounds over
// V elements: slow and exact, for a case that trained models do not produce.  sq / si: NT floats / ints of LDS scratch.
// Only the general kernel (beam_row_sample_kernel) carries it: inside the pre-filtered kernels of the hot path it cost 1-2 % of the C2 step
// even as a never-taken call (registers 44 -> 88, scratch set-up), so those still flag DH_BEAM_ERR_OVERFLOW and the host repeats the batch
// through dh_beam_row_sample_exact (deephumor_amd/models/beam.py).
template <int NT>
__device__ __attribute__((noinline)) void row_overflow(const float* __restrict__ row, int V, uint32_t thr, int rc, int ldl, int rows_per_img, int beam,
                             float temperature, int unk, const float* __restrict__ noise, uint64_t seed,
                             const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* __restrict__ pick_idx,
                             float* __restrict__ pick_val, float* sq, int* si, int* picks, int32_t* __restrict__ err) {
    const int tid = threadIdx.x;
    // (every loop rolled: this path must not raise the register count of the kernels that call it -- at 116 VGPRs instead of 44 the
    //  group-guided sampler lost a wave of occupancy, its 1,280 workgroups no longer fitted the chip in one round: +9 us per launch)
    float m = -INFINITY;
#pragma unroll 1
    for (int i = tid; i < V; i += NT) { const float v = row[i]; if (f2key(v) >= thr && i != unk) m = fmaxf(m, v / temperature); }
    __syncthreads();
    sq[tid] = m;
    __syncthreads();
#pragma unroll 1
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) sq[tid] = fmaxf(sq[tid], sq[tid + st]); __syncthreads(); }
    m = sq[0];
    __syncthreads();
    float part = 0.f;
#pragma unroll 1
    for (int i = tid; i < V; i += NT) { const float v = row[i]; if (f2key(v) >= thr && i != unk) part += expf(v / temperature - m); }
    sq[tid] = part;
    __syncthreads();
#pragma unroll 1
    for (int st = NT / 2; st > 0; st >>= 1) { if (tid < st) sq[tid] += sq[tid + st]; __syncthreads(); }
    const float s = sq[0];
    if (dh_softmax_nonfinite(m, s)) {                  // (block-uniform)
        if (tid == 0) atomicOr(err, DH_BEAM_ERR_NONFINITE);
        if (tid < beam) pick_store(pick_idx, pick_val, (size_t)rc * beam + tid, 0, 0.f);
        return;
    }
    const int img = rc / rows_per_img, rin = rc % rows_per_img;
    const uint64_t sd = seed ^ (seed_ptr ? *seed_ptr : 0ull);
#pragma unroll 1
    for (int round = 0; round < beam; ++round) {
        float bq = -1.f; int bi = 0x7FFFFFFF;
#pragma unroll 1
        for (int i = tid; i < V; i += NT) {
            const float v = row[i];
            if (!(f2key(v) >= thr && i != unk)) continue;
            bool taken = false;
#pragma unroll 1
            for (int j = 0; j < round; ++j) taken |= picks[j] == i;
            if (taken) continue;
            const float nz = noise ? noise[(size_t)rc * ldl + i] : philox_exp1(sd, (uint32_t)(img0 + img), (uint32_t)step, 0u, (uint32_t)rin, (uint32_t)i);
            const float q = (expf(v / temperature - m) / s) / nz;
            if (q > bq) { bq = q; bi = i; }             // ascending i: the first (lowest) index wins a tie
        }
        __syncthreads();
        sq[tid] = bq; si[tid] = bi;
        __syncthreads();
#pragma unroll 1
        for (int st = NT / 2; st > 0; st >>= 1) {
            if (tid < st) {
                const float oq = sq[tid + st]; const int oi = si[tid + st];
                if (oq > sq[tid] || (oq == sq[tid] && oi < si[tid])) { sq[tid] = oq; si[tid] = oi; }
            }
            __syncthreads();
        }
        if (tid == 0) picks[round] = si[0];
        __syncthreads();
    }
    if (tid == 0) {                                    // log_softmax over the gathered (un-tempered) logits of the picks (beam.py:79)
        float mx = -INFINITY;
#pragma unroll 1
        for (int b = 0; b < beam; ++b) mx = fmaxf(mx, row[picks[b]]);
        float se = 0.f;
#pragma unroll 1
        for (int b = 0; b < beam; ++b) se += expf(row[picks[b]] - mx);
        const float lse = logf(se);
#pragma unroll 1
        for (int b = 0; b < beam; ++b) pick_store(pick_idx, pick_val, (size_t)rc * beam + b, picks[b], (row[picks[b]] - mx) - lse);
    }
}

struct RowLds {
    int* idx_a; int* idx_b; float* val_a; float* val_b; float* qv; float* red; int* picks; int* s_cnt; uint32_t* s_thr;
};

// Common tail of the row kernels.  On entry idx_a/val_a hold *s_cnt candidates that include every value >= the
// row's k-th largest (block-synchronised).  Exact threshold -> survivors (ties kept, unk dropped) -> index
// order -> softmax(filtered / T) -> Exp(1) race for `beam` picks -> log_softmax over the picks.
template <int NT>
__device__ __forceinline__ void row_tail(const RowLds& L, int rc, int ldl, int rows_per_img, int beam, int top_k,
                                         float temperature, int unk, const float* __restrict__ noise, uint64_t seed,
                                         const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* __restrict__ pick_idx,
                                         float* __restrict__ pick_val, int32_t* __restrict__ err) {
    const int tid = threadIdx.x;
    int* idx_a = L.idx_a; int* idx_b = L.idx_b; float* val_a = L.val_a; float* val_b = L.val_b;
    float* qv = L.qv; float* red = L.red; int* picks = L.picks;
#define s_cnt (*L.s_cnt)
#define s_thr (*L.s_thr)
    int n0 = s_cnt;
    __syncthreads();                // every wave has read the candidate count before thread 0 resets it below
    if (n0 > CAP) { if (tid == 0) atomicOr(err, DH_BEAM_ERR_OVERFLOW); n0 = CAP; }
    // exact k-th largest among the candidates: the value whose "strictly greater" count is < k <= "greater or equal"
    for (int i = tid; i < n0; i += NT) {
        const uint32_t me = f2key(val_a[i]);
        int gt = 0, ge = 0;
        for (int j = 0; j < n0; ++j) { const uint32_t o = f2key(val_a[j]); gt += (o > me); ge += (o >= me); }
        if (gt < top_k && top_k <= ge) s_thr = me;      // all writers agree on the value
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const uint32_t thr = s_thr;
    // survivors: >= threshold (ties kept), unk dropped
    for (int i = tid; i < n0; i += NT) {
        if (f2key(val_a[i]) >= thr && idx_a[i] != unk) {
            const int p = atomicAdd(&s_cnt, 1);
            idx_b[p] = idx_a[i]; val_b[p] = val_a[i];
        }
    }
    __syncthreads();
    const int n = s_cnt;
    if (n == 0) {
        if (tid == 0) atomicOr(err, DH_BEAM_ERR_ALL_FILTERED);
        if (tid < beam) pick_store(pick_idx, pick_val, (size_t)rc * beam + tid, 0, 0.f);
        return;
    }
    // deterministic order: sort survivors by token index
    for (int i = tid; i < n; i += NT) {
        const int me = idx_b[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (idx_b[j] < me);
        idx_a[r] = me; val_a[r] = val_b[i];
    }
    __syncthreads();
    // block max / sum: wave shuffles, then the NT/64 wave partials through LDS (2 barriers each instead of a log2(NT) tree)
    constexpr int NWV = NT / 64;
    float m = -INFINITY;
    for (int i = tid; i < n; i += NT) m = fmaxf(m, val_a[i] / temperature);
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) m = fmaxf(m, red[w]);
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < n; i += NT) { const float e = expf(val_a[i] / temperature - m); qv[i] = e; s += e; }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = red[0];
#pragma unroll
    for (int w = 1; w < NWV; ++w) s += red[w];      // same order in every thread: one value for the whole block
    if (dh_softmax_nonfinite(m, s)) {                  // (block-uniform)
        if (tid == 0) atomicOr(err, DH_BEAM_ERR_NONFINITE);
        if (tid < beam) pick_store(pick_idx, pick_val, (size_t)rc * beam + tid, 0, 0.f);
        return;
    }
    const int img = rc / rows_per_img, rin = rc % rows_per_img;
    for (int i = tid; i < n; i += NT) {
        const float nz = noise ? noise[(size_t)rc * ldl + idx_a[i]]
                               : philox_exp1(seed ^ (seed_ptr ? *seed_ptr : 0ull), (uint32_t)(img0 + img), (uint32_t)step, 0u, (uint32_t)rin, (uint32_t)idx_a[i]);
        qv[i] = (qv[i] / s) / nz;
    }
    if (tid < DH_BEAM_MAX_BEAMS) picks[tid] = -1;
    __syncthreads();
    for (int i = tid; i < n; i += NT) {
        const float me = qv[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += (qv[j] > me) || (qv[j] == me && j < i);
        if (r < beam) picks[r] = i;
    }
    __syncthreads();
    if (tid < 64) {
        // log_softmax over the gathered (un-tempered) logits of the picks (beam.py:79): lane b owns pick b
        if (tid == 0 && n < beam) atomicOr(err, DH_BEAM_ERR_TOO_FEW);
        const int pi = tid < beam ? picks[tid] : -1;
        const float lv = pi >= 0 ? val_a[pi] : -INFINITY;
        const float mx = wave_max(lv);
        const float se = wave_sum(pi >= 0 ? expf(lv - mx) : 0.f);
        const float lse = logf(se);
        if (tid < beam) {
            pick_store(pick_idx, pick_val, (size_t)rc * beam + tid, pi >= 0 ? idx_a[pi] : 0, pi >= 0 ? (lv - mx) - lse : -INFINITY);
        }
    }
#undef s_cnt
#undef s_thr
}

template <int EPT, int NT, int WPE>
__global__ __launch_bounds__(NT, WPE) void beam_row_sample_fast_kernel(
    const float* __restrict__ logits, int ldl, int V, int rows_per_img, int beam, int top_k,
    float temperature, int unk, const float* __restrict__ noise, uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0, int step,
    int32_t* __restrict__ pick_idx, float* __restrict__ pick_val, int32_t* __restrict__ err) {
    __shared__ uint32_t lmax[NT];
    __shared__ int hist[4 * 256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_k, s_cnt, wtot[4];
    __shared__ __attribute__((aligned(16))) int idx_a[CAP], idx_b[CAP];
    __shared__ __attribute__((aligned(16))) float val_a[CAP], val_b[CAP], qv[CAP];
    __shared__ float red[NT];
    __shared__ int picks[DH_BEAM_MAX_BEAMS];
    __shared__ uint32_t s_thr;

    const int rc = blockIdx.x, tid = threadIdx.x;
    const float* row = logits + (size_t)rc * ldl;
    float v[EPT];
    uint32_t best = 0u;                          // key 0 < key of every real float
    // buffer loads: one shared per-lane offset + scalar strides, hardware bounds check (reads past V
    // return 0 and are ignored below) -> ~1 VGPR per element instead of an address pair + predicate
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, V * 4, 0x00020000);
#pragma unroll
    for (int e = 0; e < EPT; ++e)
        v[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, tid * 4, e * NT * 4, 0));
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * NT;
        if (i < V) best = max(best, f2key(v[e]));
    }
    lmax[tid] = best;
    if (tid == 0) { s_prefix = 0u; s_k = top_k; s_cnt = 0; s_thr = 0u; }
    // k-th largest of the 512 thread maxima (one key per thread: cheap LDS radix select)
    uint32_t mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        if ((best & mask) == prefix) atomicAdd(&hist[(best >> shift) & 255u], 1);
        __syncthreads();
        radix_pick_digit(hist, 1, shift, prefix, &s_prefix, &s_k, wtot);
        mask |= 0xFFu << shift;
    }
    const uint32_t bound = s_prefix;             // <= key of the row's k-th largest value
    // compact every value >= bound
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid + e * NT;
        if (i < V && f2key(v[e]) >= bound) {
            const int p = atomicAdd(&s_cnt, 1);
            if (p < CAP) { idx_a[p] = i; val_a[p] = v[e]; }
        }
    }
    __syncthreads();
    if (s_cnt > CAP) radix_row_candidates<NT>(row, V, top_k, hist, &s_prefix, &s_k, &s_cnt, wtot, idx_a, val_a);
    const RowLds L{idx_a, idx_b, val_a, val_b, qv, red, picks, &s_cnt, &s_thr};
    row_tail<NT>(L, rc, ldl, rows_per_img, beam, top_k, temperature, unk, noise, seed, seed_ptr, img0, step, pick_idx, pick_val, err);
}

// ---- candidate draw of one image (beam_select_kernel) -----------------------------------------------------------------------
struct SelectParams {
    const int32_t* pick_idx; const float* pick_val;
    int32_t* tokens; int tok_ld; float* vals; uint8_t* ended; int32_t* src; int src_ld;
    int32_t* parent; int32_t* hparent; uint8_t* done; int32_t* end_step;
    int beam, first, first_sets_ended, write_pos, t, step_index, eos, img0;
    float temperature; const float* noise; uint64_t seed; const uint64_t* seed_ptr;
};

// LDS scratch of one image's candidate draw, carved by the caller (beam_select_kernel's own arrays)
struct SelLds {
    int32_t* stage;                                     // [beam][tok_ld] tokens, then [beam][t] ancestors
    int* ctok; int* cpar; int* keep; float* cval; float* q; uint8_t* cend; int* s_n;
    int32_t* pki = nullptr; float* pkv = nullptr;       // optional [beam * beam]: the image's picks, brought in with the first round trip
};

// The candidate draw + in-place rewrite of one image's beam state, executed by ONE wave (lane = 0..63); LDS hand-overs are
// wave-local (wave_lds_sync).  MB = the largest beam
// count the instantiation takes (register arrays of the beams' flags / scores): 16 for the usual settings, 64 for beam_size > 16.
template <int MB>
__device__ __forceinline__ void beam_select_image(const SelectParams& p, const int img, const int lane, const SelLds& L) {
    int32_t* stage = L.stage;
    int* ctok = L.ctok; int* cpar = L.cpar; int* keep = L.keep; float* cval = L.cval; float* q = L.q; uint8_t* cend = L.cend;
#define s_n (*L.s_n)
    const int B = p.beam, base = img * B;
    // Stand-alone kernel (L.pki set): EVERYTHING the image needs -- its token rows, ancestor rows and all its picks -- is requested
    // up front by LDS-DMA, next to the loads of done / ended / vals: ONE memory round trip instead of four dependent ones (done ->
    // ended -> picks -> token rows), which were most of this 8 us kernel.
    const bool pre = L.pki != nullptr;
    int32_t* const tokbuf0 = stage;
    int32_t* const srcbuf0 = stage + (size_t)B * p.tok_ld;
    if (pre) {
        for (int i = lane; i < B * p.tok_ld; i += 64) dh_lds_dma4(p.tokens + (size_t)base * p.tok_ld + i, tokbuf0 + (i - lane));
        if (p.src)
            for (int b = 0; b < B; ++b)
                for (int j = lane; j < p.t; j += 64) dh_lds_dma4(p.src + (size_t)(base + b) * p.src_ld + j, srcbuf0 + b * p.t + (j - lane));
        const int npk = p.first ? B : B * B;
        const size_t pk0 = p.first ? (size_t)img * B : (size_t)base * B;
        for (int i = lane; i < npk; i += 64) {
            dh_lds_dma4(p.pick_idx + pk0 + i, L.pki + (i - lane));
            dh_lds_dma4(p.pick_val + pk0 + i, L.pkv + (i - lane));
        }
    }
    const uint8_t is_done = p.done[img];
    uint8_t was_ended[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) was_ended[b] = p.ended[base + min(b, B - 1)];   // one round trip for all
    float val_b[MB];
#pragma unroll
    for (int b = 0; b < MB; ++b) val_b[b] = p.vals[base + min(b, B - 1)];
    if (pre) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); wave_lds_sync(); }
    if (is_done) return;

    // candidate list in the reference's order: beam b contributes 1 candidate if it has ended, else B.
    // Every lane derives the (short) offset table itself; candidates are then filled in parallel.
    if (p.first) {
        for (int j = lane; j < B; j += 64) {
            const int tok = pre ? L.pki[j] : p.pick_idx[(size_t)img * B + j];
            ctok[j] = tok; cval[j] = pre ? L.pkv[j] : p.pick_val[(size_t)img * B + j]; cpar[j] = 0;
            cend[j] = (uint8_t)(p.first_sets_ended && tok == p.eos);
            keep[j] = j;
        }
        if (lane == 0) s_n = B;
    } else {
        int off[MB + 1];
        off[0] = 0;
#pragma unroll
        for (int b = 0; b < MB; ++b)
            off[b + 1] = off[b] + (b < B ? (was_ended[b] ? 1 : B) : 0);
        const int total = off[B];
        for (int c = lane; c < total; c += 64) {
            int b = 0;
#pragma unroll
            for (int k = 1; k < MB; ++k) b += (k < B && c >= off[k]);
            const int j = c - off[b];
            bool was = false; float vb = 0.f;             // (static register indexing: b is a run-time value)
#pragma unroll
            for (int k = 0; k < MB; ++k) if (k == b) { was = was_ended[k] != 0; vb = val_b[k]; }
            const int tok = was ? 0 : (pre ? L.pki[b * B + j] : p.pick_idx[(size_t)(base + b) * B + j]);
            ctok[c] = tok;
            cval[c] = vb + (was ? 0.f : (pre ? L.pkv[b * B + j] : p.pick_val[(size_t)(base + b) * B + j]));
            cpar[c] = b;
            cend[c] = (uint8_t)(was || tok == p.eos);
        }
        if (lane == 0) s_n = total;
    }
    wave_lds_sync();
    const int n = s_n;
    if (!p.first) {
        // draw `beam` candidates without replacement from softmax(cand_val / T)
        float m = -INFINITY;
        for (int c = lane; c < n; c += 64) m = fmaxf(m, cval[c] / p.temperature);
        m = wave_max(m);
        float s = 0.f;
        for (int c = lane; c < n; c += 64) { const float e = expf(cval[c] / p.temperature - m); q[c] = e; s += e; }
        s = wave_sum(s);
        for (int c = lane; c < n; c += 64) {
            const float nz = p.noise ? p.noise[(size_t)img * B * B + c]
                                     : philox_exp1(p.seed ^ (p.seed_ptr ? *p.seed_ptr : 0ull), (uint32_t)(p.img0 + img), (uint32_t)p.step_index, 1u, 0u, (uint32_t)c);
            q[c] = (q[c] / s) / nz;
        }
        // (every slot valid whatever the scores are: with a NaN among them no rank below comes out right, and the rewrite loop must
        //  not index with what the LDS held before)
        for (int j = lane; j < B; j += 64) keep[j] = min(j, n - 1);
        wave_lds_sync();
        for (int c = lane; c < n; c += 64) {
            const float me = q[c];
            int r = 0;
            for (int j = 0; j < n; ++j) r += (q[j] > me) || (q[j] == me && j < c);
            if (r < B) keep[r] = c;
        }
    }
    // stage the image's token rows and ancestor rows, then rewrite them in place
    int32_t* tokbuf = stage;
    int32_t* srcbuf = stage + (size_t)B * p.tok_ld;
    if (!pre) {
        for (int i = lane; i < B * p.tok_ld; i += 64) tokbuf[i] = p.tokens[(size_t)base * p.tok_ld + i];
        if (p.src)
            for (int b = 0; b < B; ++b)
                for (int j = lane; j < p.t; j += 64) srcbuf[b * p.t + j] = p.src[(size_t)(base + b) * p.src_ld + j];
    }
    wave_lds_sync();
    int all_ended = 1;
    for (int b = 0; b < B; ++b) {
        const int c = keep[b], par = cpar[c];
        for (int i = lane; i < p.tok_ld; i += 64)
            p.tokens[(size_t)(base + b) * p.tok_ld + i] = (i == p.write_pos) ? ctok[c] : tokbuf[par * p.tok_ld + i];
        if (p.src) {
            for (int j = lane; j < p.t; j += 64) p.src[(size_t)(base + b) * p.src_ld + j] = srcbuf[par * p.t + j];
            if (lane == 0) p.src[(size_t)(base + b) * p.src_ld + p.t] = base + par;
        }
        if (lane == 0) {
            p.vals[base + b] = cval[c];
            p.ended[base + b] = cend[c];
            p.parent[base + b] = base + par;
            p.hparent[base + b] = base + c / B;        // rnn_models.py:135-137: dense B*B layout index
        }
        all_ended &= cend[c];
    }
    // the reference only tests all_ended() inside the token loop (rnn_models.py:131), never after the first draw
    if (lane == 0 && all_ended && !p.first) { p.done[img] = 1; p.end_step[img] = p.step_index; }
#undef s_n
}

// ---- group-max guided variant -----------------------------------------------------------------------------
// The vocabulary GEMM (dh_vocab_logits) leaves, next to the logits, the maximum of every group of `gcols`
// consecutive columns of each row.  The k-th largest GROUP maximum is a lower bound of the row's k-th largest
// value (k groups hold a value >= it), and only groups whose maximum reaches that bound can contain one of the
// top-k values: about top_k of the ~570 groups.  So this kernel reads ~9 % of the row instead of all of it.
template <int NT>
__global__ __launch_bounds__(NT) void beam_row_sample_groups_kernel(
    const float* __restrict__ logits, int ldl, int V, const float* __restrict__ gmax, int gm_ld, int n_groups,
    int gcols, int rows_per_img, int beam, int top_k, float temperature, int unk, const float* __restrict__ noise,
    uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* __restrict__ pick_idx, float* __restrict__ pick_val,
    int32_t* __restrict__ err) {
    constexpr int MAXG = 1024, GPT = MAXG / NT;       // group keys per thread, kept in registers
    __shared__ int glist[MAXG];
    __shared__ int hist[4][256];
    __shared__ uint32_t s_prefix, s_thr;
    __shared__ int s_k, s_cnt, s_ng, wtot[4];
    __shared__ __attribute__((aligned(16))) int32_t pool[5 * CAP];       // the five candidate buffers
    int* const idx_a = pool; int* const idx_b = pool + CAP;
    float* const val_a = reinterpret_cast<float*>(pool + 2 * CAP); float* const val_b = reinterpret_cast<float*>(pool + 3 * CAP);
    float* const qv = reinterpret_cast<float*>(pool + 4 * CAP);
    __shared__ float red[NT];
    __shared__ int picks[DH_BEAM_MAX_BEAMS];
    const int rc = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (size_t)rc * ldl;
    uint32_t gk[GPT];
#pragma unroll
    for (int e = 0; e < GPT; ++e) {
        // unconditional loads at a clamped index (one memory round trip for all of them, not one per conditional load)
        const int g = tid + e * NT;
        const uint32_t k = f2key(gmax[(size_t)rc * gm_ld + min(g, n_groups - 1)]);
        gk[e] = g < n_groups ? k : 0u;                                        // key 0 < key of every real float
    }
    for (int i = tid; i < 512; i += NT) (&hist[0][0])[i] = 0;
    if (tid == 0) { s_prefix = 0u; s_k = min(top_k, n_groups); s_cnt = 0; s_ng = 0; s_thr = 0u; }
    __syncthreads();
    // Lower edge of the 16-bit key bucket (sign, exponent, 7 mantissa bits) that holds the k-th largest group maximum:
    // at least k groups -- hence at least k logits -- are >= it, and only a handful of extra groups share the bucket.
    // Two radix passes instead of the four an exact k-th largest needs.
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int shift = 24 - 8 * pass;
        const uint32_t prefix = s_prefix, mask = pass == 0 ? 0u : 0xFF000000u;
#pragma unroll
        for (int e = 0; e < GPT; ++e)
            if (tid + e * NT < n_groups && (gk[e] & mask) == prefix) atomicAdd(&hist[pass][(gk[e] >> shift) & 255u], 1);
        __syncthreads();
        radix_pick_digit(hist[pass], 1, shift, prefix, &s_prefix, &s_k, wtot);
    }
    const uint32_t bound = s_prefix;                 // <= the row's k-th largest logit (as a key)
#pragma unroll
    for (int e = 0; e < GPT; ++e)
        if (tid + e * NT < n_groups && gk[e] >= bound) glist[atomicAdd(&s_ng, 1)] = tid + e * NT;
    __syncthreads();
    const int ng = s_ng;
    // one wave per selected group (coalesced 256-B reads), all of a wave's groups requested before the first is used
    constexpr int NWV = NT / 64, UN = 16;     // 64 groups in one round of loads (a row selects ~55)
    for (int q0 = wave; q0 < ng; q0 += NWV * UN) {
        float v[UN]; int ci[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int q = q0 + u * NWV;
            ci[u] = q < ng ? glist[min(q, ng - 1)] * gcols + lane : V;
            v[u] = row[min(ci[u], V - 1)];                  // unconditional (clamped) loads: all in flight together
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
            if (lane < gcols && ci[u] < V && f2key(v[u]) >= bound) {
                const int pp = atomicAdd(&s_cnt, 1);
                if (pp < CAP) { idx_a[pp] = ci[u]; val_a[pp] = v[u]; }
            }
    }
    __syncthreads();
    if (s_cnt > CAP) radix_row_candidates<NT>(row, V, top_k, &hist[0][0], &s_prefix, &s_k, &s_cnt, wtot, idx_a, val_a);
    const RowLds L{idx_a, idx_b, val_a, val_b, qv, red, picks, &s_cnt, &s_thr};
    row_tail<NT>(L, rc, ldl, rows_per_img, beam, top_k, temperature, unk, noise, seed, seed_ptr, img0, step, pick_idx, pick_val, err);
}

extern "C" int dh_beam_row_sample_groups(const float* logits, int ldl, int V, const float* group_max, int gm_ld,
                                         int n_groups, int group_cols, int rows, int rows_per_img, int beam,
                                         int top_k, float temperature, int unk_index, const float* noise,
                                         uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* pick_idx, float* pick_val,
                                         int32_t* err, void* stream) {
    DH_REQUIRE(logits && group_max && pick_idx && pick_val && err && rows > 0 && rows_per_img > 0 && V > 0 && ldl >= V);
    DH_REQUIRE(beam >= 1 && beam <= DH_BEAM_MAX_BEAMS && beam <= top_k && top_k <= V && temperature > 0.f);
    DH_REQUIRE(n_groups > 0 && n_groups <= 1024 && top_k <= n_groups && gm_ld >= n_groups && group_cols > 0 && group_cols <= 64 &&
               (long long)n_groups * group_cols >= V);
    DhProfScope prof("dh_beam_row_sample", 0.0, 0.0, stream);
    hipLaunchKernelGGL((beam_row_sample_groups_kernel<256>), dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V,
                       group_max, gm_ld, n_groups, group_cols, rows_per_img, beam, top_k, temperature, unk_index, noise,
                       seed, seed_ptr, img0, step, pick_idx, pick_val, err);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_beam_row_sample(const float* logits, int ldl, int V, int rows, int rows_per_img, int beam,
                                  int top_k, float temperature, int unk_index, const float* noise,
                                  uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* pick_idx, float* pick_val,
                                  int32_t* err, void* stream) {
    DH_REQUIRE(logits && pick_idx && pick_val && err && rows > 0 && rows_per_img > 0 && V > 0 && ldl >= V);
    DH_REQUIRE(beam >= 1 && beam <= DH_BEAM_MAX_BEAMS && beam <= top_k && top_k <= V && temperature > 0.f);
    DhProfScope prof("dh_beam_row_sample", 0.0, 0.0, stream);
#define DH_FAST(EPT, NT, WPE) hipLaunchKernelGGL((beam_row_sample_fast_kernel<EPT, NT, WPE>), dim3(rows), dim3(NT), 0, \
        (hipStream_t)stream, logits, ldl, V, rows_per_img, beam, top_k, temperature, unk_index, noise, seed, seed_ptr, img0, \
        step, pick_idx, pick_val, err)
    if (top_k <= 256 && V <= 512 * 8) DH_FAST(8, 512, 4);
    else if (top_k <= 256 && V <= 1024 * 16) DH_FAST(16, 1024, 8);
    else if (top_k <= 256 && V <= 1024 * 36) DH_FAST(36, 1024, 8);
    else if (top_k <= 256 && V <= 1024 * 64) DH_FAST(64, 1024, 4);
    else
        hipLaunchKernelGGL(beam_row_sample_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V,
                           rows_per_img, beam, top_k, temperature, unk_index, noise, seed, seed_ptr, img0, step, pick_idx,
                           pick_val, err);
#undef DH_FAST
    DH_LAUNCH_CHECK();
}

// dh_beam_row_sample on the general kernel only (4-pass radix select over the whole row; any top_k, any V; a row with more survivors than
// DH_BEAM_MAX_SURVIVORS is drawn over the row itself instead of flagging DH_BEAM_ERR_OVERFLOW): the fall-back the host takes when a batch
// flagged that overflow in the pre-filtered kernels.
extern "C" int dh_beam_row_sample_exact(const float* logits, int ldl, int V, int rows, int rows_per_img, int beam,
                                        int top_k, float temperature, int unk_index, const float* noise,
                                        uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0, int step, int32_t* pick_idx, float* pick_val,
                                        int32_t* err, void* stream) {
    DH_REQUIRE(logits && pick_idx && pick_val && err && rows > 0 && rows_per_img > 0 && V > 0 && ldl >= V);
    DH_REQUIRE(beam >= 1 && beam <= DH_BEAM_MAX_BEAMS && beam <= top_k && top_k <= V && temperature > 0.f);
    DhProfScope prof("dh_beam_row_sample", 0.0, 0.0, stream);
    hipLaunchKernelGGL(beam_row_sample_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V,
                       rows_per_img, beam, top_k, temperature, unk_index, noise, seed, seed_ptr, img0, step, pick_idx,
                       pick_val, err);
    DH_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------

template <int MB>
__global__ __launch_bounds__(64) void beam_select_kernel(SelectParams p) {
    extern __shared__ int32_t stage[];
    __shared__ int ctok[MB * MB], cpar[MB * MB], keep[MB];
    __shared__ float cval[MB * MB], q[MB * MB];
    __shared__ uint8_t cend[MB * MB];
    __shared__ int s_n;
    __shared__ int32_t pki[MB * MB];
    __shared__ float pkv[MB * MB];
    const SelLds L{stage, ctok, cpar, keep, cval, q, cend, &s_n, pki, pkv};
    beam_select_image<MB>(p, blockIdx.x, threadIdx.x, L);
}

extern "C" int dh_beam_select(const int32_t* pick_idx, const float* pick_val, int32_t* tokens, int tok_ld,
                              float* vals, uint8_t* ended, int32_t* src, int src_ld, int32_t* parent,
                              int32_t* hparent, uint8_t* done, int32_t* end_step, int n_img, int beam,
                              int first, int first_sets_ended, int write_pos, int t, int step_index,
                              float temperature, int eos_index, const float* noise, uint64_t seed,
                              const uint64_t* seed_ptr, int img0, void* stream) {
    DH_REQUIRE(pick_idx && pick_val && tokens && vals && ended && parent && hparent && done && end_step);
    DH_REQUIRE(n_img > 0 && beam >= 1 && beam <= DH_BEAM_MAX_BEAMS && tok_ld > 0 && t >= 0 && temperature > 0.f);
    DH_REQUIRE(!src || src_ld > t);
    DhProfScope prof("dh_beam_select", 0.0, 0.0, stream);
    SelectParams p{pick_idx, pick_val, tokens, tok_ld, vals, ended, src, src_ld, parent, hparent, done, end_step,
                   beam, first, first_sets_ended, write_pos, t, step_index, eos_index, img0, temperature, noise, seed, seed_ptr};
    const size_t lds = (size_t)beam * (tok_ld + (src ? t : 0)) * sizeof(int32_t);
    DH_REQUIRE(lds <= 56 * 1024);                         // beam * (tok_ld + t) ints of staging next to the candidate arrays
    if (beam <= 16) hipLaunchKernelGGL(beam_select_kernel<16>, dim3(n_img), dim3(64), lds, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(beam_select_kernel<DH_BEAM_MAX_BEAMS>, dim3(n_img), dim3(64), lds, (hipStream_t)stream, p);   // beam.py:7-9: any beam_size <= top_k
    DH_LAUNCH_CHECK();
}

// ---- the reference's METHOD surface (deephumor/models/beam.py:32-108), one kernel per method ------------------------------------
// BeamSearchHelper.filter_top_k / sample_k_indices / filter_by_indices / process_logits as the reference's own generate() loops
// call them (rnn_models.py:87-128, transformers.py:532-569): host-driven, one image at a time, tensors of the reference's shapes
// and dtypes (int64 indices, fp32 values, in-place -inf filter).  The batched decode engine above does not use them.

__device__ __forceinline__ float key2f(uint32_t k) {           // inverse of f2key
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// beam.py:34-36: logits[logits < kth_largest] = -inf (strict: ties at the threshold stay), logits[:, unk] = -inf
__global__ __launch_bounds__(256) void beam_filter_topk_kernel(float* __restrict__ logits, int ldl, int V, int top_k, int unk) {
    __shared__ int hist[4 * 256];
    __shared__ uint32_t s_prefix;
    __shared__ int s_k, wtot[4];
    const int tid = threadIdx.x, hw = tid >> 6;
    float* row = logits + (size_t)blockIdx.x * ldl;
    if (tid == 0) { s_prefix = 0u; s_k = top_k; }
    uint32_t mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 1024; i += 256) hist[i] = 0;
        __syncthreads();
        const uint32_t prefix = s_prefix;
        for (int i = tid; i < V; i += 256) {
            const uint32_t key = f2key(row[i]);
            if ((key & mask) == prefix) atomicAdd(&hist[hw * 256 + ((key >> shift) & 255u)], 1);
        }
        __syncthreads();
        radix_pick_digit(hist, 4, shift, prefix, &s_prefix, &s_k, wtot);
        mask |= 0xFFu << shift;
    }
    const float thr = key2f(s_prefix);                          // the k-th largest value of the row
    for (int i = tid; i < V; i += 256)
        if (row[i] < thr || i == unk) row[i] = -INFINITY;
}

// beam.py:39-48: p = softmax(x / T); torch.multinomial(p, k) without replacement == the k largest of p / Exp(1) noise, in
// descending order (CPU torch).  k rounds of a block arg-max (ties -> lower index) over the whole row.
__global__ __launch_bounds__(256) void beam_sample_k_kernel(const float* __restrict__ x, int ld, int V, int k, float temperature,
                                                            const float* __restrict__ noise, int noise_ld, uint64_t seed,
                                                            const uint64_t* __restrict__ seed_ptr, int stream_id, int draw,
                                                            int64_t* __restrict__ out, int32_t* __restrict__ err) {
    __shared__ float red[256];
    __shared__ float rq[256];
    __shared__ int ri[256];
    __shared__ int picked[64];
    const int tid = threadIdx.x, r = blockIdx.x;
    const float* row = x + (size_t)r * ld;
    float m = -INFINITY;
    for (int i = tid; i < V; i += 256) m = fmaxf(m, row[i] / temperature);
    m = block_reduce_256(m, red, true);
    if (m == -INFINITY) {                                       // softmax of an all -inf row is NaN: torch raises
        if (tid == 0) atomicOr(err, DH_BEAM_ERR_ALL_FILTERED);
        for (int j = tid; j < k; j += 256) out[(size_t)r * k + j] = 0;
        return;
    }
    float s = 0.f, cnt = 0.f;
    for (int i = tid; i < V; i += 256) { const float e = expf(row[i] / temperature - m); s += e; cnt += e > 0.f ? 1.f : 0.f; }
    s = block_reduce_256(s, red, false);
    cnt = block_reduce_256(cnt, red, false);
    if (tid == 0 && cnt < (float)k) atomicOr(err, DH_BEAM_ERR_TOO_FEW);      // "not enough non-negative category to sample"
    const uint64_t sd = seed ^ (seed_ptr ? *seed_ptr : 0ull);
    for (int round = 0; round < k; ++round) {
        float bq = -1.f; int bi = 0x7FFFFFFF;
        for (int i = tid; i < V; i += 256) {
            bool taken = false;
            for (int j = 0; j < round; ++j) taken |= picked[j] == i;
            if (taken) continue;
            const float nz = noise ? noise[(size_t)r * noise_ld + i] : philox_exp1(sd, (uint32_t)stream_id, (uint32_t)draw, 3u, (uint32_t)r, (uint32_t)i);
            const float q = (expf(row[i] / temperature - m) / s) / nz;
            if (q > bq) { bq = q; bi = i; }                     // ascending i: the first (lowest) index wins a tie
        }
        rq[tid] = bq; ri[tid] = bi;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                const float oq = rq[tid + st]; const int oi = ri[tid + st];
                if (oq > rq[tid] || (oq == rq[tid] && oi < ri[tid])) { rq[tid] = oq; ri[tid] = oi; }
            }
            __syncthreads();
        }
        if (tid == 0) { picked[round] = ri[0]; out[(size_t)r * k + round] = ri[0] == 0x7FFFFFFF ? 0 : ri[0]; }
        __syncthreads();
    }
}

// beam.py:50-53: torch.gather(values, 1, indices)
__global__ void beam_gather_kernel(const float* __restrict__ values, int ld, int V, const int64_t* __restrict__ indices, int k,
                                   float* __restrict__ out, int total, int32_t* __restrict__ err) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int64_t i = indices[t];
    if (i < 0 || i >= V) { if (err) atomicOr(err, DH_BEAM_ERR_OVERFLOW); out[t] = 0.f; return; }
    out[t] = values[(size_t)(t / k) * ld + i];
}

// beam.py:78-106: log_softmax over each row's gathered picks, then the candidate expansion -- a live beam contributes `beam`
// candidates, an ended one a single candidate with token 0 / score 0 -- with the new has_ended flags and the repeated sequences.
__global__ __launch_bounds__(64) void beam_expand_kernel(const int64_t* __restrict__ ind, const float* __restrict__ gathered,
                                                         const uint8_t* __restrict__ ended, const int64_t* __restrict__ seqs, int seq_len,
                                                         const float* __restrict__ vals, int val_w, int n, int beam, int eos,
                                                         int64_t* __restrict__ prev_seqs, float* __restrict__ prev_vals,
                                                         int64_t* __restrict__ new_ind, float* __restrict__ new_val,
                                                         uint8_t* __restrict__ new_ended) {
    const int lane = threadIdx.x;
    int total = 0;
    for (int b = 0; b < n; ++b) total += ended[b] ? 1 : beam;
    for (int c = lane; c < total; c += 64) {
        int b = 0, off = 0;
        for (;;) { const int w = ended[b] ? 1 : beam; if (c < off + w) break; off += w; ++b; }
        const int j = c - off;
        const bool was = ended[b] != 0;
        float mx = -INFINITY;
        for (int u = 0; u < beam; ++u) mx = fmaxf(mx, gathered[(size_t)b * beam + u]);
        float se = 0.f;
        for (int u = 0; u < beam; ++u) se += expf(gathered[(size_t)b * beam + u] - mx);
        const int64_t tok = was ? 0 : ind[(size_t)b * beam + j];
        new_ind[c] = tok;
        new_val[c] = was ? 0.f : (gathered[(size_t)b * beam + j] - mx) - logf(se);
        new_ended[c] = (uint8_t)(was || tok == eos);
        for (int i = 0; i < seq_len; ++i) prev_seqs[(size_t)c * seq_len + i] = seqs[(size_t)b * seq_len + i];
        for (int i = 0; i < val_w; ++i) prev_vals[(size_t)c * val_w + i] = vals[(size_t)b * val_w + i];
    }
}

extern "C" int dh_beam_filter_top_k(float* logits, int ldl, int V, int rows, int top_k, int unk_index, void* stream) {
    DH_REQUIRE(logits && rows > 0 && V > 0 && ldl >= V && top_k >= 1 && top_k <= V);
    DhProfScope prof("dh_beam_filter_top_k", 0.0, 0.0, stream);
    hipLaunchKernelGGL(beam_filter_topk_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, ldl, V, top_k, unk_index);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_beam_sample_k(const float* x, int ld, int V, int rows, int k, float temperature, const float* noise, int noise_ld,
                                uint64_t seed, const uint64_t* seed_ptr, int stream_id, int draw, int64_t* out, int32_t* err,
                                void* stream) {
    DH_REQUIRE(x && out && err && rows > 0 && V > 0 && ld >= V && k >= 1 && k <= 64 && k <= V && temperature > 0.f);
    DH_REQUIRE(!noise || noise_ld >= V);
    DhProfScope prof("dh_beam_sample_k", 0.0, 0.0, stream);
    hipLaunchKernelGGL(beam_sample_k_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, V, k, temperature, noise, noise_ld,
                       seed, seed_ptr, stream_id, draw, out, err);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_beam_gather(const float* values, int ld, int V, const int64_t* indices, int k, float* out, int rows, int32_t* err,
                              void* stream) {
    DH_REQUIRE(values && indices && out && rows > 0 && k > 0 && V > 0 && ld >= V);
    DhProfScope prof("dh_beam_gather", 0.0, 0.0, stream);
    const int total = rows * k;
    hipLaunchKernelGGL(beam_gather_kernel, dim3(dh_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, values, ld, V, indices, k, out,
                       total, err);
    DH_LAUNCH_CHECK();
}

extern "C" int dh_beam_expand(const int64_t* new_ind, const float* gathered, const uint8_t* ended, const int64_t* seqs, int seq_len,
                              const float* vals, int val_width, int n_rows, int beam, int eos_index, int64_t* prev_seqs,
                              float* prev_vals, int64_t* out_ind, float* out_val, uint8_t* out_ended, void* stream) {
    DH_REQUIRE(new_ind && gathered && ended && seqs && vals && prev_seqs && prev_vals && out_ind && out_val && out_ended);
    DH_REQUIRE(n_rows > 0 && n_rows <= 1024 && beam >= 1 && beam <= 64 && seq_len >= 0 && val_width >= 1);
    DhProfScope prof("dh_beam_expand", 0.0, 0.0, stream);
    hipLaunchKernelGGL(beam_expand_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, new_ind, gathered, ended, seqs, seq_len, vals,
                       val_width, n_rows, beam, eos_index, prev_seqs, prev_vals, out_ind, out_val, out_ended);
    DH_LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void beam_finalize_kernel(
    const int32_t* __restrict__ tokens, int tok_ld, const float* __restrict__ vals, const uint8_t* __restrict__ done,
    const int32_t* __restrict__ end_step, int32_t* __restrict__ out, int out_ld, int32_t* __restrict__ out_len,
    int beam, int len_bias_done, int full_len, int pad_index, float temperature, const float* __restrict__ noise,
    uint64_t seed, const uint64_t* __restrict__ seed_ptr, int img0) {
    const int img = blockIdx.x, lane = threadIdx.x, base = img * beam;
    // ind = multinomial(softmax(vals / T), 1) == arg-max of p / Exp(1) (first index on ties)
    float x = lane < beam ? vals[base + lane] / temperature : -INFINITY;
    const float m = wave_max(x);
    float e = lane < beam ? expf(x - m) : 0.f;
    const float s = wave_sum(e);
    float qq = -1.f;
    if (lane < beam) {
        const float nz = noise ? noise[(size_t)img * beam + lane]
                               : philox_exp1(seed ^ (seed_ptr ? *seed_ptr : 0ull), (uint32_t)(img0 + img), 0xFFFFFFFFu, 2u, 0u, (uint32_t)lane);
        qq = (e / s) / nz;
    }
    const float best = wave_max(qq);
    const unsigned long long bal = __ballot(qq == best && lane < beam);
    const int ind = max(__ffsll((long long)bal) - 1, 0);          // (NaN scores: no lane equals the maximum -- never index row -1)
    int len = done[img] ? end_step[img] + len_bias_done : full_len;
    len = min(len, min(out_ld, tok_ld));
    for (int i = lane; i < out_ld; i += 64)
        out[(size_t)img * out_ld + i] = i < len ? tokens[(size_t)(base + ind) * tok_ld + i] : pad_index;
    if (lane == 0) out_len[img] = len;
}

extern "C" int dh_beam_finalize(const int32_t* tokens, int tok_ld, const float* vals, const uint8_t* done,
                                const int32_t* end_step, int32_t* out, int out_ld, int32_t* out_len,
                                int n_img, int beam, int len_bias_done, int full_len, int pad_index,
                                float temperature, const float* noise, uint64_t seed, const uint64_t* seed_ptr, int img0,
                                void* stream) {
    DH_REQUIRE(tokens && vals && done && end_step && out && out_len && n_img > 0);
    DH_REQUIRE(beam >= 1 && beam <= DH_BEAM_MAX_BEAMS && temperature > 0.f);
    DhProfScope prof("dh_beam_finalize", 0.0, 0.0, stream);
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(n_img), dim3(64), 0, (hipStream_t)stream, tokens, tok_ld, vals,
                       done, end_step, out, out_ld, out_len, beam, len_bias_done, full_len, pad_index,
                       temperature, noise, seed, seed_ptr, img0);
    DH_LAUNCH_CHECK();
}
