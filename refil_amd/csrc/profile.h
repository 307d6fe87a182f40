// Optional per-kernel timing (used by bench.py for the roofline fraction; off by default: zero overhead
// beyond one branch per launch). Two timers:
//  * HIP events around the launch (mode 1). Exact when the stream has the GPU to itself; with the two-stream
//    overlap they also count the time a kernel waits behind the other stream's kernels.
//  * a device-side span {first workgroup start, last workgroup end} on the 100 MHz wall clock, written by the
//    kernel itself (modes 1 and 2; GEMM, attention, GRU kernels). This is the kernel's own begin->end, the
//    quantity rocprofv3 --kernel-trace reports, and is cheap enough to stay on during the timed region.
#pragma once
#include <hip/hip_runtime.h>

namespace refil {

bool prof_enabled();
void prof_begin(const char* kernel, double flops, double bytes);
bool prof_take_events(hipEvent_t* e0, hipEvent_t* e1);   // true once per ProfScope while profiling
void prof_end();
unsigned long long* prof_clock_slot();                    // device {min start, max end} slot of the open scope, or nullptr

struct ProfScope {
    bool on;
    ProfScope(const char* kernel, double flops, double bytes, hipStream_t) : on(prof_enabled()) {
        if (on) prof_begin(kernel, flops, bytes);
    }
    ~ProfScope() { if (on) prof_end(); }
};

}  // namespace refil

#define REFIL_LAUNCH(kernel, grid, block, smem, st, ...)                                                       \
    do {                                                                                                       \
        hipEvent_t e0_, e1_;                                                                                   \
        const bool ev_ = refil::prof_take_events(&e0_, &e1_);                                                  \
        if (ev_) hipEventRecord(e0_, st);                                                                      \
        hipLaunchKernelGGL(kernel, grid, block, smem, st, __VA_ARGS__);                                        \
        if (ev_) hipEventRecord(e1_, st);                                                                      \
    } while (0)
