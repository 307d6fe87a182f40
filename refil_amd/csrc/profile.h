// Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
// roofline fraction; off by default: zero overhead beyond one branch per launch).
#pragma once
#include <hip/hip_runtime.h>

namespace refil {

bool prof_enabled();
void prof_begin(const char* kernel, double flops, double bytes, hipStream_t st);
void prof_end(hipStream_t st);

struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(const char* kernel, double flops, double bytes, hipStream_t s) : st(s), on(prof_enabled()) {
        if (on) prof_begin(kernel, flops, bytes, st);
    }
    ~ProfScope() { if (on) prof_end(st); }
};

}  // namespace refil
