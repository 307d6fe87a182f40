// Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
// roofline fraction; off by default: zero overhead beyond one branch per launch).
#pragma once
#include <hip/hip_runtime.h>

namespace refil {

bool prof_enabled();
// rows_dev / rows_max (optional): the launch runs over a device-side row list; flops / bytes are given for rows_max
// rows and are scaled by *rows_dev / rows_max when the profile is collected (the count never visits the host earlier)
// flops_split: the part of `flops` that runs as bf16 x 6 products (refil_profile_entry.flops_bf16x6)
void prof_begin(const char* kernel, double flops, double bytes, hipStream_t st, const int* rows_dev = nullptr, double rows_max = 0.0, double flops_split = 0.0);
void prof_end(hipStream_t st);

struct ProfScope {
    hipStream_t st; bool on;
    ProfScope(const char* kernel, double flops, double bytes, hipStream_t s, const int* rows_dev = nullptr, double rows_max = 0.0, double flops_split = 0.0)
        : st(s), on(prof_enabled()) {
        if (on) prof_begin(kernel, flops, bytes, st, rows_dev, rows_max, flops_split);
    }
    ~ProfScope() { if (on) prof_end(st); }
};

}  // namespace refil
