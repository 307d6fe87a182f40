// Probe for DESIGN.md section 10 item 1 (written in round 5 after the GPU pool closed: compiled, NOT yet executed -- run it before building on it).
//
// attention_qkv.hip leaves K^T, Q^T and V in the accumulator layout of v_mfma_f32_16x16x32_bf16 (D[row = 4 (lane / 16) + reg][col = lane % 16])
// and feeds the attention core's fp32 16x16x4 products from those registers. Claim to check: the SAME registers are valid operands of the
// bf16 instruction for the core's two products once split -- the eight values a lane holds for (key tile, channel tiles 0 / 1) are its
// eight reduction indices if BOTH operands use the virtual order  k-group q = channels {4 q .. 4 q + 3} u {16 + 4 q .. 16 + 4 q + 3}
// (S^T = K Q^T, reduction over the 32 channels of a head) resp. keys {4 q .. 4 q + 3} u {16 + 4 q .. } (O^T = V^T P^T, reduction over 32 keys).
// Then S^T costs 6 x 16 cycles per key tile instead of 8 x 32, O^T 6 x 16 per channel tile instead of 8 x 32: the core 1024 -> 384 cycles.
//
// One wave, one head: K, V [32 entities][32 channels], Q [16 agents][32 channels]; P = S^T / sqrt(32) (no softmax: a layout check).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/attn_core_bf16_probe.hip -o /tmp/attn_core_bf16_probe ; run: /tmp/attn_core_bf16_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ inline unsigned pk(float x, float y) { f32x2 v = {x, y}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ inline void split2(float x, float y, unsigned& h, unsigned& m, unsigned& l) {
    h = pk(x, y);
    x -= __uint_as_float(h << 16); y -= __uint_as_float(h & 0xFFFF0000u);
    m = pk(x, y);
    x -= __uint_as_float(m << 16); y -= __uint_as_float(m & 0xFFFF0000u);
    l = pk(x, y);
}
// two accumulator quads (the lane's eight reduction indices) -> the three planes of an operand fragment
__device__ inline void split_acc(const f32x4& a, const f32x4& b, u32x4 (&o)[3]) {
    unsigned h[4], m[4], l[4];
    split2(a[0], a[1], h[0], m[0], l[0]); split2(a[2], a[3], h[1], m[1], l[1]);
    split2(b[0], b[1], h[2], m[2], l[2]); split2(b[2], b[3], h[3], m[3], l[3]);
    o[0] = u32x4{h[0], h[1], h[2], h[3]}; o[1] = u32x4{m[0], m[1], m[2], m[3]}; o[2] = u32x4{l[0], l[1], l[2], l[3]};
}
__device__ inline f32x4 mfma6(const u32x4 (&a)[3], const u32x4 (&b)[3]) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};       // (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
#pragma unroll
    for (int p = 0; p < 6; ++p)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[PA[p]]), __builtin_bit_cast(bf16x8, b[PB[p]]), acc, 0, 0, 0);
    return acc;
}

__global__ __launch_bounds__(64) void core(const float* K, const float* V, const float* Q, float* St_out /*[32 e][16 a]*/, float* Ot_out /*[32 ch][16 a]*/) {
    const int lane = threadIdx.x, c16 = lane & 15, q = lane >> 4;
    // what attention_qkv.hip's projections leave in the accumulators
    f32x4 Kt[2][2], Vv[2][2], Qt[2];
    for (int jt = 0; jt < 2; ++jt)
        for (int ct = 0; ct < 2; ++ct)
            for (int r = 0; r < 4; ++r) {
                Kt[jt][ct][r] = K[(16 * jt + c16) * 32 + 16 * ct + 4 * q + r];         // K^T tile: row = channel, col = entity
                Vv[jt][ct][r] = V[(16 * jt + 4 * q + r) * 32 + 16 * ct + c16];         // V tile:   row = entity,  col = channel
            }
    for (int ct = 0; ct < 2; ++ct)
        for (int r = 0; r < 4; ++r) Qt[ct][r] = Q[c16 * 32 + 16 * ct + 4 * q + r];     // Q^T tile: row = channel, col = agent
    // S^T[jt] = K[jt] Q^T: A index = entity (lane % 16), B index = agent (lane % 16), k-group q = channels {4q..} u {16 + 4q..}
    u32x4 qf[3];
    split_acc(Qt[0], Qt[1], qf);
    f32x4 St[2];
    for (int jt = 0; jt < 2; ++jt) {
        u32x4 kf[3];
        split_acc(Kt[jt][0], Kt[jt][1], kf);
        St[jt] = mfma6(kf, qf);                       // D[row = entity 16 jt + 4 q + reg][col = agent c16]
        for (int r = 0; r < 4; ++r) St_out[(16 * jt + 4 * q + r) * 16 + c16] = St[jt][r];
    }
    const float scale = 1.0f / sqrtf(32.f);
    for (int jt = 0; jt < 2; ++jt) St[jt] = St[jt] * scale;
    // O^T[ct] = V^T P^T: A index = channel (lane % 16), B index = agent, k-group q = keys {4q..} u {16 + 4q..}
    u32x4 pf[3];
    split_acc(St[0], St[1], pf);
    for (int ct = 0; ct < 2; ++ct) {
        u32x4 vf[3];
        split_acc(Vv[0][ct], Vv[1][ct], vf);
        const f32x4 Ot = mfma6(vf, pf);               // D[row = channel 16 ct + 4 q + reg][col = agent c16]
        for (int r = 0; r < 4; ++r) Ot_out[(16 * ct + 4 * q + r) * 16 + c16] = Ot[r];
    }
}

int main() {
    float K[32 * 32], V[32 * 32], Q[16 * 32], St[32 * 16], Ot[32 * 16];
    srand(3);
    for (auto& v : K) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : V) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : Q) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *dK, *dV, *dQ, *dS, *dO;
    (void)hipMalloc(&dK, sizeof(K)); (void)hipMalloc(&dV, sizeof(V)); (void)hipMalloc(&dQ, sizeof(Q)); (void)hipMalloc(&dS, sizeof(St)); (void)hipMalloc(&dO, sizeof(Ot));
    (void)hipMemcpy(dK, K, sizeof(K), hipMemcpyHostToDevice); (void)hipMemcpy(dV, V, sizeof(V), hipMemcpyHostToDevice); (void)hipMemcpy(dQ, Q, sizeof(Q), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(core, dim3(1), dim3(64), 0, 0, dK, dV, dQ, dS, dO);
    (void)hipMemcpy(St, dS, sizeof(St), hipMemcpyDeviceToHost); (void)hipMemcpy(Ot, dO, sizeof(Ot), hipMemcpyDeviceToHost);
    double es = 0, eo = 0;
    double S[32][16];
    for (int e = 0; e < 32; ++e)
        for (int a = 0; a < 16; ++a) {
            double s = 0;
            for (int c = 0; c < 32; ++c) s += (double)K[e * 32 + c] * Q[a * 32 + c];
            S[e][a] = s;
            es = fmax(es, fabs(s - St[e * 16 + a]));
        }
    for (int c = 0; c < 32; ++c)
        for (int a = 0; a < 16; ++a) {
            double o = 0;
            for (int e = 0; e < 32; ++e) o += (double)V[e * 32 + c] * S[e][a] / sqrt(32.0);
            eo = fmax(eo, fabs(o - Ot[c * 16 + a]));
        }
    printf("S^T = K Q^T from the accumulator registers on 16x16x32 bf16 x 6: max error %.3e (values ~ 3): %s\n", es, es < 1e-5 ? "layout OK" : "LAYOUT WRONG");
    printf("O^T = V^T P^T from the accumulator registers:                    max error %.3e: %s\n", eo, eo < 1e-5 ? "layout OK" : "LAYOUT WRONG");
    return 0;
}
