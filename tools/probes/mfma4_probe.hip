// Ground truth for v_mfma_f32_4x4x1_16b_f32 on gfx950 (used by the 4-row recurrence tiles in gru.hip):
//  (a) operand layout: D[lane l][reg i] = A[lane 4*(l/4) + i] * B[lane l]  (16 independent 4x4 outer products, block = l/4)
//  (b) issue rate of back-to-back instructions on 6 independent accumulators
// build: hipcc --offload-arch=gfx950 -O3 mfma4_probe.hip -o mfma4_probe ; run: ./mfma4_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void layout(float* out) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int l = threadIdx.x;
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(l + 1), (float)(1000 + l), acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
__global__ __launch_bounds__(256) void rate(float* out, int iters, float a, float b) {
    f32x4 acc[6];
    for (int t = 0; t < 6; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    a += threadIdx.x;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[t], 0, 0, 0);
    float s = 0;
    for (int t = 0; t < 6; ++t) for (int r = 0; r < 4; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 1 << 22);
    hipLaunchKernelGGL(layout, dim3(1), dim3(64), 0, 0, out);
    float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int i = 0; i < 4; ++i) {
        const float want = (float)(4 * (l / 4) + i + 1) * (float)(1000 + l);
        if (h[l * 4 + i] != want) { if (bad < 8) printf("lane %d reg %d: got %.0f want %.0f\n", l, i, h[l * 4 + i], want); ++bad; }
    }
    printf("layout D[l][i] = A[4*(l/4)+i] * B[l]: %s (%d mismatches)\n", bad ? "NO" : "OK", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000, blocks = 256 * 2;
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * 48 * 2;      // instructions per SIMD (2 waves per SIMD at 512 blocks on 256 CUs)
    printf("4x4x1_16b: %.3f ms, %.1f ns per instruction per SIMD, %.1f TFLOP/s\n", ms, ms * 1e6 / n, 512.0 * iters * 48 * blocks * 4 / (ms * 1e-3) / 1e12);
    return 0;
}
