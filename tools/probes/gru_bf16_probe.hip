// Probe for DESIGN.md section 10 item 4 (written in round 5 after the cycle stamps of profiles/r05_gru_timing.txt; NOT part of the library,
// never executed when it was committed -- the GPU pool was closed: run it first thing next round).
//
// Question: how long is a dependent GRU step when the recurrent product h W_hh^T runs as v_mfma_f32_16x16x32_bf16 x 6 (exact 3-way
// split, fp32 accumulate) instead of 48 x v_mfma_f32_4x4x1_16b_f32 (one every ~20 cycles from one wave: 55 % of today's 1.15 us step)?
//
// Layout idea: a workgroup keeps the 4-row tile of gru_fwd4_kernel, but the four live rows sit at rows 0, 4, 8, 12 of a 16-row MFMA
// tile. D[row = 4 (lane / 16) + reg][col = lane % 16]: register 0 of lane group q is (live row q, column lane % 16) -- every lane owns
// exactly ONE live element, no redistribution, no k-slice sums. The other 12 rows of the A operand are whatever the lanes read
// (row i of D depends on row i of A only). Wave w owns hidden columns 16 w .. 16 w + 15 of all three gates: 3 gates x 2 k-steps x 6
// products = 36 MFMAs of 16 cycles per step; W_hh's planes stay in 72 registers; h is split on the dependent path (16 values per lane).
//
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/gru_bf16_probe.hip -o /tmp/gru_bf16_probe ; run: /tmp/gru_bf16_probe
// prints the max error against an fp64 host recurrence (expect ~1e-6: fp32-accurate products) and ns per dependent step.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int H = 64, HP = H + 4, ROWS = 4;

__device__ inline unsigned pk(float x, float y) { f32x2 v = {x, y}; return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
__device__ inline float sub1(float a, float b) { float r; asm("v_sub_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// (x, y) -> packed bf16 pairs hi / mid / lo with x = hi + mid + lo exactly (to 2^-25 |x|)
__device__ inline void split2(float x, float y, unsigned& h, unsigned& m, unsigned& l) {
    h = pk(x, y);
    x = sub1(x, __uint_as_float(h << 16)); y = sub1(y, __uint_as_float(h & 0xFFFF0000u));
    m = pk(x, y);
    x = sub1(x, __uint_as_float(m << 16)); y = sub1(y, __uint_as_float(m & 0xFFFF0000u));
    l = pk(x, y);
}
// 8 consecutive floats -> the three planes of an MFMA operand fragment
__device__ inline void split8(const float4& a, const float4& b, u32x4 (&o)[3]) {
    unsigned h[4], m[4], l[4];
    split2(a.x, a.y, h[0], m[0], l[0]); split2(a.z, a.w, h[1], m[1], l[1]);
    split2(b.x, b.y, h[2], m[2], l[2]); split2(b.z, b.w, h[3], m[3], l[3]);
    o[0] = u32x4{h[0], h[1], h[2], h[3]}; o[1] = u32x4{m[0], m[1], m[2], m[3]}; o[2] = u32x4{l[0], l[1], l[2], l[3]};
}
__device__ inline float sigm(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ inline float tanh_(float x) { const float e = __expf(-2.0f * fabsf(x)); const float t = (1.0f - e) / (1.0f + e); return x < 0.f ? -t : t; }

// gi [tiles][T][ROWS][3 H] (the input-gate pre-activations incl. b_ih), hs [tiles][T + 1][ROWS][H] (slot 0 = h_{-1})
__global__ __launch_bounds__(256) void gru_bf16_steps(const float* __restrict__ gi, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                      float* __restrict__ hs, int T) {
    __shared__ __attribute__((aligned(16))) float hbuf[2][ROWS * HP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int col = 16 * wave + c16;                // hidden column this lane owns; the row it owns is q
    // B operand (W_hh) fragments: index j = lane % 16 -> output column col of gate g, k = 32 ks + 8 q .. + 7
    u32x4 bw[3][2][3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* wr = w_hh + (long)(g * H + col) * H + 32 * ks + 8 * q;
            split8(*reinterpret_cast<const float4*>(wr), *reinterpret_cast<const float4*>(wr + 4), bw[g][ks]);
        }
    const float bhr = b_hh[col], bhz = b_hh[H + col], bhn = b_hh[2 * H + col];
    const long tile = blockIdx.x;
    const float* gt = gi + tile * T * ROWS * 3 * H + (long)q * 3 * H + col;
    float* ht = hs + tile * (T + 1) * ROWS * H + (long)q * H + col;
    float hold = ht[0];
    hbuf[0][q * HP + col] = hold;
    __syncthreads();
    float g0 = gt[0], g1 = gt[H], g2 = gt[2 * H];
    for (int t = 0; t < T; ++t) {
        // next step's inputs (one step ahead is enough for a probe)
        const int tn = t + 1 < T ? t + 1 : t;
        const float n0 = gt[(long)tn * ROWS * 3 * H], n1 = gt[(long)tn * ROWS * 3 * H + H], n2 = gt[(long)tn * ROWS * 3 * H + 2 * H];
        // A operand: tile row i = lane % 16 carries live row i / 4 (rows with i % 4 != 0 are copies: their D rows are never read)
        const float* hb = hbuf[t & 1] + (c16 >> 2) * HP + 8 * q;
        u32x4 ah[2][3];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) split8(*reinterpret_cast<const float4*>(hb + 32 * ks), *reinterpret_cast<const float4*>(hb + 32 * ks + 4), ah[ks]);
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // products smallest first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi); planes: 0 hi, 1 mid, 2 lo
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[ks][PA[p]]), __builtin_bit_cast(bf16x8, bw[g][ks][PB[p]]), acc[g], 0, 0, 0);
        // register 0 = (tile row 4 q = live row q, column c16): this lane's element
        const float rg = sigm(g0 + acc[0][0] + bhr);
        const float zg = sigm(g1 + acc[1][0] + bhz);
        const float ng = tanh_(g2 + rg * (acc[2][0] + bhn));
        hold = (1.0f - zg) * ng + zg * hold;
        hbuf[(t + 1) & 1][q * HP + col] = hold;
        ht[(long)(t + 1) * ROWS * H] = hold;
        g0 = n0; g1 = n1; g2 = n2;
        __syncthreads();
    }
}

// Variant 2: the PRODUCER splits. The lane that computes h_t(row q, col) splits that one value and writes its three bf16 pieces into
// LDS planes [plane][row][H] (2-byte stores); the A fragments of the next step are then three 16-byte LDS reads per k-step with no
// arithmetic between the barrier and the first MFMA (variant 1 splits 16 values per lane on the dependent path: ~90 operations).
constexpr int HPB = H + 8;                          // bf16 elements per plane row (144 bytes: the four rows start in different banks)
__global__ __launch_bounds__(256) void gru_bf16_steps_planes(const float* __restrict__ gi, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
                                                             float* __restrict__ hs, int T) {
    __shared__ __attribute__((aligned(16))) unsigned short hpl[2][3][ROWS * HPB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int col = 16 * wave + c16;
    u32x4 bw[3][2][3];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const float* wr = w_hh + (long)(g * H + col) * H + 32 * ks + 8 * q;
            split8(*reinterpret_cast<const float4*>(wr), *reinterpret_cast<const float4*>(wr + 4), bw[g][ks]);
        }
    const float bhr = b_hh[col], bhz = b_hh[H + col], bhn = b_hh[2 * H + col];
    const long tile = blockIdx.x;
    const float* gt = gi + tile * T * ROWS * 3 * H + (long)q * 3 * H + col;
    float* ht = hs + tile * (T + 1) * ROWS * H + (long)q * H + col;
    float hold = ht[0];
    auto publish = [&](int buf, float x) {
        unsigned h, m, l;
        split2(x, 0.f, h, m, l);                    // (the low half of each packed pair is x's piece)
        hpl[buf][0][q * HPB + col] = (unsigned short)h; hpl[buf][1][q * HPB + col] = (unsigned short)m; hpl[buf][2][q * HPB + col] = (unsigned short)l;
    };
    publish(0, hold);
    __syncthreads();
    float g0 = gt[0], g1 = gt[H], g2 = gt[2 * H];
    for (int t = 0; t < T; ++t) {
        const int tn = t + 1 < T ? t + 1 : t;
        const float n0 = gt[(long)tn * ROWS * 3 * H], n1 = gt[(long)tn * ROWS * 3 * H + H], n2 = gt[(long)tn * ROWS * 3 * H + 2 * H];
        u32x4 ah[2][3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                ah[ks][pl] = *reinterpret_cast<const u32x4*>(&hpl[t & 1][pl][(c16 >> 2) * HPB + 32 * ks + 8 * q]);
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah[ks][PA[p]]), __builtin_bit_cast(bf16x8, bw[g][ks][PB[p]]), acc[g], 0, 0, 0);
        const float rg = sigm(g0 + acc[0][0] + bhr);
        const float zg = sigm(g1 + acc[1][0] + bhz);
        const float ng = tanh_(g2 + rg * (acc[2][0] + bhn));
        hold = (1.0f - zg) * ng + zg * hold;
        publish((t + 1) & 1, hold);
        ht[(long)(t + 1) * ROWS * H] = hold;
        g0 = n0; g1 = n1; g2 = n2;
        __syncthreads();
    }
}

typedef void (*kern_t)(const float*, const float*, const float*, float*, int);

int main() {
    const int T = 81, tiles = 384, check_tiles = 4;
    std::vector<float> gi((size_t)tiles * T * ROWS * 3 * H), whh(3 * H * H), bhh(3 * H), hs((size_t)tiles * (T + 1) * ROWS * H, 0.f);
    srand(1);
    auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
    for (auto& v : gi) v = rnd();
    for (auto& v : whh) v = rnd() / 8.f;
    for (auto& v : bhh) v = rnd() / 8.f;
    for (int tl = 0; tl < tiles; ++tl)
        for (int i = 0; i < ROWS * H; ++i) hs[(size_t)tl * (T + 1) * ROWS * H + i] = rnd() / 2.f;
    float *dgi, *dw, *db, *dhs;
    hipMalloc(&dgi, gi.size() * 4); hipMalloc(&dw, whh.size() * 4); hipMalloc(&db, bhh.size() * 4); hipMalloc(&dhs, hs.size() * 4);
    hipMemcpy(dgi, gi.data(), gi.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dw, whh.data(), whh.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, bhh.data(), bhh.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dhs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    const kern_t kerns[2] = {gru_bf16_steps, gru_bf16_steps_planes};
    const char* names[2] = {"consumer splits h (16 values per lane and step)", "producer splits h (bf16 planes in LDS)"};
    for (int kv = 0; kv < 2; ++kv) {
    printf("-- %s\n", names[kv]);
    hipMemcpy(dhs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kerns[kv], dim3(tiles), dim3(256), 0, 0, dgi, dw, db, dhs, T);
    hipDeviceSynchronize();
    std::vector<float> out(hs.size());
    hipMemcpy(out.data(), dhs, hs.size() * 4, hipMemcpyDeviceToHost);
    // fp64 host recurrence of the first tiles
    double maxerr = 0.0, maxabs = 0.0;
    for (int tl = 0; tl < check_tiles; ++tl)
        for (int r = 0; r < ROWS; ++r) {
            std::vector<double> h(H), hn(H);
            for (int c = 0; c < H; ++c) h[c] = hs[((size_t)tl * (T + 1) * ROWS + r) * H + c];
            for (int t = 0; t < T; ++t) {
                const float* g = &gi[(((size_t)tl * T + t) * ROWS + r) * 3 * H];
                for (int c = 0; c < H; ++c) {
                    double pr = bhh[c], pz = bhh[H + c], pn = bhh[2 * H + c];
                    for (int k = 0; k < H; ++k) { pr += h[k] * whh[(size_t)c * H + k]; pz += h[k] * whh[(size_t)(H + c) * H + k]; pn += h[k] * whh[(size_t)(2 * H + c) * H + k]; }
                    const double rg = 1.0 / (1.0 + exp(-(g[c] + pr))), zg = 1.0 / (1.0 + exp(-(g[H + c] + pz)));
                    const double ng = tanh(g[2 * H + c] + rg * pn);
                    hn[c] = (1.0 - zg) * ng + zg * h[c];
                }
                h = hn;
                for (int c = 0; c < H; ++c) {
                    const double d = fabs(h[c] - (double)out[(((size_t)tl * (T + 1) + t + 1) * ROWS + r) * H + c]);
                    if (d > maxerr) maxerr = d;
                    if (fabs(h[c]) > maxabs) maxabs = fabs(h[c]);
                }
            }
        }
    printf("max |h - fp64 reference| over %d tiles x %d steps: %.3e (max |h| %.3f)\n", check_tiles, T, maxerr, maxabs);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemcpy(dhs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kerns[kv], dim3(tiles), dim3(256), 0, 0, dgi, dw, db, dhs, T);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%d tiles x %d steps: %.1f us per launch = %.0f ns per dependent step (gru_fwd4_kernel at this shape: 92.9 us = 1146 ns)\n", tiles, T, ms * 100.0, ms * 1e5 / T);
    }
    return 0;
}
