// Ground-truth probe: fp32 MFMA issue rate on gfx950 for (a) the dependent-chain order used by the GEMM
// (4 back-to-back MFMAs on one accumulator) and (b) a round-robin order over 4 accumulators.
// build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[16];
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < 16; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, F launch, double flops_per_block_iter, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch(blocks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops_per_block_iter * blocks * iters / (ms * 1e-3) / 1e12);
}

int main() {
    float* out;
    hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    for (int bpc = 1; bpc <= 4; ++bpc) {
        int blocks = 256 * bpc;
        run("32x32x2 chain-of-4 (GEMM order)", [&](int b, int it) { hipLaunchKernelGGL(k32<0>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 16 * 4096, blocks, iters);
        run("32x32x2 round-robin over 4 acc", [&](int b, int it) { hipLaunchKernelGGL(k32<1>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 16 * 4096, blocks, iters);
        run("16x16x4 chain-of-2", [&](int b, int it) { hipLaunchKernelGGL(k16<0>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 32 * 2048, blocks, iters);
        run("16x16x4 round-robin over 16 acc", [&](int b, int it) { hipLaunchKernelGGL(k16<1>, dim3(b), dim3(256), 0, 0, out, it, 1.f, 2.f); }, 4.0 * 32 * 2048, blocks, iters);
    }
    return 0;
}
