// In-library launch profiler: HIP events on the launch stream around every C-ABI kernel launch while
// enabled (dh_prof_begin .. dh_prof_end), aggregated per "entry[tag]" with the algorithmic flops/bytes
// of each launch.  Off by default (one branch per launch).
#pragma once
#include <hip/hip_runtime.h>

struct DhProfScope {
    hipStream_t s;
    int rec;
    DhProfScope(const char* name, double flops, double bytes, void* stream);
    ~DhProfScope();
};
void dh_prof_set_tag(const char* tag);       // role of the next launch (qkv / proj / ffn / vocab / gates ...)
void dh_prof_set_dims(int m, int n, int k);  // GEMM shape of the next launch: the profiler keys it "entry[tag]{MxNxK}"
