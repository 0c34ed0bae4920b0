// What a DEPENDENT kernel launch costs on MI355X and which part of it can be removed.
//   hipcc --offload-arch=gfx950 -O3 -o launch_floor_probe launch_floor_probe.hip && ./launch_floor_probe
// The Transformer decode chain runs 47 dependent launches per position at 8.3 us each when nearly empty (one image) and 10.7 us at
// 256 images.  This probe separates that figure into
//   gap   : end of kernel i (last wave's last instruction) -> start of kernel i+1 (first wave's first instruction), from
//           s_memrealtime stamps (100 MHz constant clock) taken INSIDE the kernels: command-processor work, the release /
//           acquire cache maintenance of the boundary, wave dispatch;
//   body  : start -> end of a kernel that does what the chain's kernels do first and last: read rows the previous kernel wrote
//           (cold: written by other XCDs), read weights nobody wrote (is L2 kept across a boundary?), write rows;
// for chains launched eagerly, replayed from a hipGraph, and in one persistent launch with an XCD-local / grid-wide barrier in
// place of each boundary.  Environment variables that change the runtime's boundary (AMD_OPT_FLUSH, DEBUG_CLR_GRAPH_PACKET_CAPTURE)
// are swept by the caller (tools/r5_floor.sh): the probe prints which ones it sees.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t now100() { return __builtin_amdgcn_s_memrealtime(); }

// stamps[2 i] = earliest start over the workgroups of launch i, stamps[2 i + 1] = latest end
__device__ __forceinline__ void stamp_start(unsigned long long* st, int i) { if (threadIdx.x == 0) atomicMin(&st[2 * i], (unsigned long long)now100()); }
__device__ __forceinline__ void stamp_end(unsigned long long* st, int i) {
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(&st[2 * i + 1], (unsigned long long)now100());
}

__global__ void k_empty(unsigned long long* st, int i, int stamps) {
    if (stamps) { stamp_start(st, i); stamp_end(st, i); }
}

// body of a chain kernel: workgroup b reads 16 B per thread of the rows the PREVIOUS launch wrote (its own slice and, with
// `cross`, the slice of workgroup b + 1: written on another XCD), `wbytes` of weights per workgroup (never written: read-only
// across launches; `wshared`: every workgroup the same bytes), writes its slice.  lat[] (optional) = cycles of the FIRST row load
// and the FIRST weight load of workgroup 0 (s_memtime around a dependent use).
__global__ void k_body(const float4* __restrict__ in, float4* __restrict__ out, const float4* __restrict__ w, int wvec_per_thread, int wshared,
                       int cross, unsigned long long* st, int i, int stamps, unsigned* lat) {
    if (stamps) stamp_start(st, i);
    const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x, nt = blockDim.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    float4 v = in[(size_t)(cross ? (b + 1) % nb : b) * nt + t];
    float acc = v.x + v.y + v.z + v.w;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    const float4* wp = w + (size_t)(wshared ? 0 : b) * nt * wvec_per_thread;
    float4 first = make_float4(0.f, 0.f, 0.f, 0.f);
    if (wvec_per_thread > 0) first = wp[t];
    acc += first.x;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint64_t t2 = __builtin_amdgcn_s_memtime();
    for (int j = 1; j < wvec_per_thread; ++j) { const float4 x = wp[(size_t)j * nt + t]; acc += x.x + x.y + x.z + x.w; }
    out[(size_t)b * nt + t] = make_float4(acc, v.y, v.z, v.w);
    if (lat && b == 0 && t == 0) { lat[2 * i] = (unsigned)(t1 - t0); lat[2 * i + 1] = (unsigned)(t2 - t1); }
    if (stamps) stamp_end(st, i);
}

// the same bodies inside ONE launch, a barrier in place of each kernel boundary.  scope 0: barrier among the workgroups of one XCD
// (blockIdx % 8; rows stay XCD-local, plain stores + workgroup-visible L2), scope 1: grid-wide with agent-scope release / acquire.
__global__ void k_persistent(float4* bufa, float4* bufb, const float4* __restrict__ w, int wvec_per_thread, int wshared, int n_steps, unsigned* counters,
                             int scope, unsigned long long* st) {
    const int b = blockIdx.x, nb = gridDim.x, t = threadIdx.x, nt = blockDim.x;
    const int xcd = b % 8;
    unsigned* ctr = scope == 0 ? counters + 32 * xcd : counters + 32 * 8;
    const unsigned members = scope == 0 ? (unsigned)((nb - xcd + 7) / 8) : (unsigned)nb;
    if (t == 0) atomicMin(&st[0], (unsigned long long)now100());
    for (int i = 0; i < n_steps; ++i) {
        const float4* in = (i & 1) ? bufb : bufa;
        float4* out = (i & 1) ? bufa : bufb;
        // neighbour slice within the same barrier group (XCD-local: b + 8)
        const int src = scope == 0 ? ((b + 8 < nb) ? b + 8 : xcd) : (b + 1) % nb;
        float4 v = in[(size_t)src * nt + t];                                   // (this CU's L1 was invalidated behind the barrier)
        float acc = v.x + v.y + v.z + v.w;
        const float4* wp = w + (size_t)(wshared ? 0 : b) * nt * wvec_per_thread;
        for (int j = 0; j < wvec_per_thread; ++j) { const float4 x = wp[(size_t)j * nt + t]; acc += x.x + x.y + x.z + x.w; }
        out[(size_t)b * nt + t] = make_float4(acc, v.y, v.z, v.w);
        __syncthreads();
        if (t == 0) {
            if (scope == 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // stores have left the CU (L1 is write-through)
            const unsigned target = (unsigned)(i + 1) * members;
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int spin = 0; spin < (1 << 16) && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spin) __builtin_amdgcn_s_sleep(1);
            if (scope == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
        if (scope == 0) asm volatile("buffer_inv sc0" ::: "memory");           // this CU's L1 only
        else asm volatile("buffer_inv sc1" ::: "memory");
    }
    if (t == 0) atomicMax(&st[1], (unsigned long long)now100());
}

struct Cfg { const char* name; int wg, threads, wvec, wshared, cross; };

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main() {
    const int N = 600;
    int dev = 0; CK(hipSetDevice(dev));
    hipStream_t s; CK(hipStreamCreate(&s));
    float4 *bufa, *bufb, *w; unsigned long long* st; unsigned* lat; unsigned* counters;
    const size_t rows_bytes = 1024 * 1024 * 16, w_bytes = 64u << 20;
    CK(hipMalloc(&bufa, rows_bytes)); CK(hipMalloc(&bufb, rows_bytes)); CK(hipMalloc(&w, w_bytes));
    CK(hipMalloc(&st, 2 * N * 8)); CK(hipMalloc(&lat, 2 * N * 4)); CK(hipMalloc(&counters, 4096));
    CK(hipMemset(bufa, 0, rows_bytes)); CK(hipMemset(bufb, 0, rows_bytes)); CK(hipMemset(w, 0, w_bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("{\"env\": {\"AMD_OPT_FLUSH\": \"%s\", \"DEBUG_CLR_GRAPH_PACKET_CAPTURE\": \"%s\", \"HIP_FORCE_DEV_KERNARG\": \"%s\", \"GPU_MAX_HW_QUEUES\": \"%s\"}}\n",
           getenv("AMD_OPT_FLUSH") ? getenv("AMD_OPT_FLUSH") : "", getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE") ? getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE") : "",
           getenv("HIP_FORCE_DEV_KERNARG") ? getenv("HIP_FORCE_DEV_KERNARG") : "", getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "");

    const Cfg cfgs[] = {
        {"empty", 256, 256, -1, 0, 0},
        {"rows_only_own_slice", 256, 256, 0, 0, 0},
        {"rows_only_neighbour_slice", 256, 256, 0, 0, 1},
        {"rows+8KB_weights_per_wg", 256, 256, 2, 0, 1},
        {"rows+64KB_weights_per_wg", 256, 256, 16, 0, 1},
        {"rows+64KB_shared_weights", 256, 256, 16, 1, 1},
        {"rows+256KB_weights_per_wg(64MB)", 256, 256, 64, 0, 1},
        {"38wg_rows+64KB_weights", 38, 256, 16, 0, 1},
        {"1024wg_rows+8KB_weights", 1024, 256, 2, 0, 1},
    };
    for (const Cfg& c : cfgs) {
        auto launch = [&](int i, int stamps) {
            const float4* in = (i & 1) ? bufb : bufa; float4* out = (i & 1) ? bufa : bufb;
            if (c.wvec < 0) hipLaunchKernelGGL(k_empty, dim3(c.wg), dim3(c.threads), 0, s, st, i, stamps);
            else hipLaunchKernelGGL(k_body, dim3(c.wg), dim3(c.threads), 0, s, in, out, w, c.wvec, c.wshared, c.cross, st, i, stamps, stamps ? lat : nullptr);
        };
        // (1) eager, no stamps: us per launch by events
        std::vector<double> eager, graphed;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) launch(i, 0);
            CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); eager.push_back(ms * 1000.0 / N);
        }
        // (2) the same chain replayed from a captured hipGraph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) launch(i, 0);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); graphed.push_back(ms * 1000.0 / N);
        }
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
        // (3) eager with in-kernel stamps: gap (end i -> start i+1) and body (start i -> end i), latency of the first loads
        std::vector<unsigned long long> h(2 * N); std::vector<unsigned> hl(2 * N, 0);
        for (int i = 0; i < N; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
        CK(hipMemcpy(st, h.data(), 2 * N * 8, hipMemcpyHostToDevice)); CK(hipMemset(lat, 0, 2 * N * 4));
        for (int i = 0; i < N; ++i) launch(i, 1);
        CK(hipStreamSynchronize(s));
        CK(hipMemcpy(h.data(), st, 2 * N * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hl.data(), lat, 2 * N * 4, hipMemcpyDeviceToHost));
        std::vector<double> gap, body, l_row, l_w;
        for (int i = 50; i + 1 < N; ++i) {
            gap.push_back((double)(h[2 * (i + 1)] - h[2 * i + 1]) * 0.01);
            body.push_back((double)(h[2 * i + 1] - h[2 * i]) * 0.01);
            l_row.push_back(hl[2 * i]); l_w.push_back(hl[2 * i + 1]);
        }
        printf("{\"chain\": \"%s\", \"workgroups\": %d, \"weights_bytes_per_wg\": %d, \"us_per_launch_eager\": %.2f, \"us_per_launch_graph\": %.2f, "
               "\"stamped_gap_us\": %.2f, \"stamped_body_us\": %.2f, \"first_row_load_cycles\": %.0f, \"first_weight_load_cycles\": %.0f}\n",
               c.name, c.wg, c.wvec > 0 ? c.wvec * c.threads * 16 : 0, median(eager), median(graphed), median(gap), median(body), median(l_row), median(l_w));
        fflush(stdout);
    }
    // (4) one persistent launch, barriers in place of boundaries
    for (int scope = 0; scope < 2; ++scope)
        for (int wvec : {0, 16}) {
            std::vector<double> per;
            for (int rep = 0; rep < 4; ++rep) {
                unsigned long long h2[2] = {~0ull, 0};
                CK(hipMemcpy(st, h2, 16, hipMemcpyHostToDevice)); CK(hipMemset(counters, 0, 4096));
                hipLaunchKernelGGL(k_persistent, dim3(256), dim3(256), 0, s, bufa, bufb, w, wvec, 0, N, counters, scope, st);
                CK(hipStreamSynchronize(s));
                CK(hipMemcpy(h2, st, 16, hipMemcpyDeviceToHost));
                per.push_back((double)(h2[1] - h2[0]) * 0.01 / N);
            }
            printf("{\"persistent\": \"%s\", \"workgroups\": 256, \"weights_bytes_per_wg\": %d, \"us_per_step\": %.2f}\n",
                   scope == 0 ? "xcd-local barrier (32 workgroups per group, no release / acquire)" : "grid-wide barrier (agent-scope release + acquire)",
                   wvec * 256 * 16, median(per));
            fflush(stdout);
        }
    return 0;
}
