// Companion of mfma_dvfs_probe.hip: a bare v_mfma_f32_32x32x16_bf16 stream (no memory traffic) on constant / random / zero operands,
// 1 and 2 waves per SIMD. Question: what does the bf16 matrix pipe sustain under the same power budget that holds the fp32 stream at
// ~135 TFLOP/s on random data -- i.e. what would an fp32 product emulated as 6 or 9 bf16 products (3-way split of both operands,
// fp32 accumulate) run at?  build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_probe.hip -o mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int NV = 16;
__global__ __launch_bounds__(256) void k(const bf16x8* __restrict__ av, const bf16x8* __restrict__ bv, float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    bf16x8 a[NV], b[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) { a[i] = av[(blockIdx.x % 64) * 256 * NV + i * 256 + threadIdx.x]; b[i] = bv[(blockIdx.x % 64) * 256 * NV + i * 256 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + t) % NV], acc[t], 0, 0, 0);
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double gauss() { double u = (rand() + 1.0) / (RAND_MAX + 2.0), v = (rand() + 1.0) / (RAND_MAX + 2.0); return sqrt(-2 * log(u)) * cos(6.283185307179586 * v); }
static unsigned short tobf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main() {
    const size_t n = 64 * 256 * NV * 8;
    unsigned short *ha = (unsigned short*)malloc(n * 2), *hb = (unsigned short*)malloc(n * 2);
    bf16x8 *da, *db; float* out;
    hipMalloc(&da, n * 2); hipMalloc(&db, n * 2); hipMalloc(&out, 4096 * 256 * 4);
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        const char* name = mode == 0 ? "constant" : (mode == 1 ? "random N(0,1)" : (mode == 2 ? "zeros" : "random N(0,1), second pass"));
        for (size_t i = 0; i < n; ++i) {
            ha[i] = tobf(mode == 0 ? 1.f + (i % 7) : (mode == 2 ? 0.f : (float)gauss()));
            hb[i] = tobf(mode == 0 ? 2.f : (mode == 2 ? 0.f : (float)gauss()));
        }
        hipMemcpy(da, ha, n * 2, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 2, hipMemcpyHostToDevice);
        for (int bpc = 1; bpc <= 2; ++bpc) {
            const int blocks = 256 * bpc;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, da, db, out, iters);
            hipDeviceSynchronize();
            float best = 1e30f, sum = 0;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, da, db, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best; sum += ms;
            }
            const double fl = 4.0 * NV * 4 * 32768.0 * blocks * iters;
            const double tf = fl / (sum / 5 * 1e-3) / 1e12;
            printf("%-28s blocks=%4d  mean %7.3f ms  %7.1f TFLOP/s bf16 = %.3f of 2516  -> fp32-equivalent: x9 %6.1f, x6 %6.1f TFLOP/s\n", name, blocks, sum / 5,
                   tf, tf / 2516.6, tf / 9, tf / 6);
        }
    }
    return 0;
}
