// TEST INFRASTRUCTURE: self-test of the CPU wavefront emulator (tests/emu). Small kernels written against the DOCUMENTED behaviour of the
// gfx950 instructions the product kernels rely on (cdna_hip_programming.md section 3 for the matrix-instruction layouts, the gfx9 ISA for
// DPP / permute / buffer range checks), checked against plain host arithmetic. tests/test_emu_selftest.py builds and runs it.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

static unsigned short to_bf16(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float from_bf16(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- matrix instructions: D = A B + C with A [M][K], B [K][N] row-major in global memory, one wave ----
__global__ void k_mfma_32x32x2(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 1.0f;                                    // C = 1
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[(l & 31) * 2 + (l >> 5)], B[(l >> 5) * 32 + (l & 31)], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void k_mfma_16x16x4(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f32x4 acc = {1.f, 1.f, 1.f, 1.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_mfma_4x4x1(const float* A, const float* B, float* D) {           // 16 blocks: A [16][4], B [16][4], D [16][4][4]
    const int l = threadIdx.x;
    f32x4 acc = {1.f, 1.f, 1.f, 1.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[((l >> 2) * 4 + i) * 4 + (l & 3)] = acc[i];
}
__global__ void k_mfma_16x16x32_bf16(const unsigned short* A, const unsigned short* B, float* D) {     // A [16][32], B [32][16]
    const int l = threadIdx.x;
    unsigned short a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[(l & 15) * 32 + 8 * (l >> 4) + e]; b[e] = B[(8 * (l >> 4) + e) * 16 + (l & 15)]; }
    bf16x8 av, bv;
    memcpy(&av, a, 16); memcpy(&bv, b, 16);
    f32x4 acc = {1.f, 1.f, 1.f, 1.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
}
__global__ void k_mfma_32x32x16_bf16(const unsigned short* A, const unsigned short* B, float* D) {     // A [32][16], B [16][32]
    const int l = threadIdx.x;
    unsigned short a[8], b[8];
    for (int e = 0; e < 8; ++e) { a[e] = A[(l & 31) * 16 + 8 * (l >> 5) + e]; b[e] = B[(8 * (l >> 5) + e) * 32 + (l & 31)]; }
    bf16x8 av, bv;
    memcpy(&av, a, 16); memcpy(&bv, b, 16);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 1.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}

// ---- cross-lane ----
template <int CTRL>
__device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(-7, v, CTRL, 0xF, 0xF, false); }      // old = -7 where there is no source
__global__ void k_cross(int* out) {
    const int l = threadIdx.x, v = 1000 + l;
    int* o = out + l * 16;
    o[0] = dpp<0xB1>(v);      // quad_perm [1,0,3,2]
    o[1] = dpp<0x4E>(v);      // quad_perm [2,3,0,1]
    o[2] = dpp<0x141>(v);     // row_half_mirror
    o[3] = dpp<0x140>(v);     // row_mirror
    o[4] = dpp<0x124>(v);     // row_ror:4
    o[5] = dpp<0x128>(v);     // row_ror:8
    o[6] = dpp<0x111>(v);     // row_shr:1
    o[7] = dpp<0x101>(v);     // row_shl:1
    o[8] = __builtin_amdgcn_ds_bpermute(((l * 7 + 3) & 63) << 2, v);             // read lane (7 l + 3) % 64
    o[9] = __builtin_amdgcn_ds_permute(((l * 5 + 1) & 63) << 2, v);              // send to lane (5 l + 1) % 64 (a permutation)
    o[10] = __shfl_xor(v, 16, 64);
    o[11] = __shfl_up(v, 3, 64);
    o[12] = __builtin_amdgcn_readlane(v, 37);
    o[13] = __builtin_amdgcn_readfirstlane(v);
    const unsigned long long b = __ballot(l % 3 == 0);
    o[14] = (int)(b & 0xffffffffu);
    o[15] = (int)(b >> 32);
}
// a collective in divergent control flow: only the lanes that are there take part
__global__ void k_divergent(int* out) {
    const int l = threadIdx.x;
    int v = 100 + l;
    if (l < 20) v = __shfl_xor(v, 1, 64);
    else if (l >= 40) v = __shfl_xor(v, 2, 64);
    const unsigned long long b = l & 1 ? __ballot(1) : 0ull;
    out[l] = v;
    out[64 + l] = (int)(b >> 32) ^ (int)(b & 0xffffffffu);
}

// ---- buffer range check, LDS, barriers, early exits, atomics ----
__global__ void k_buffer(const float* src, float* dst, int n_bytes) {
    const int l = threadIdx.x;
    auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, n_bytes, 0x00020000);
    auto rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, n_bytes, 0x00020000);
    auto v = __builtin_amdgcn_raw_buffer_load_b128(rs, l * 16, 0, 0);             // 64 lanes x 16 bytes; n_bytes cuts through a lane's vector
    unsigned u[4];
    memcpy(u, &v, 16);
    for (int d = 0; d < 4; ++d) u[d] += 1;                                        // (integer +1 on the bits: zeros from dropped dwords become 1)
    memcpy(&v, u, 16);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, l * 16, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0xdeadu, rd, 0x7ffffff0, 0, 0);         // far out of range: dropped
}
__global__ void k_block(float* out, int* counter) {
    __shared__ float red[4];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t >= 200) return;                                                         // lanes that leave before the barriers
    float s = (float)(t + 1);
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);                  // wave 3 runs with 8 lanes
    if (lane == 0) red[wave] = s;
    dyn[t] = (float)t;
    __syncthreads();
    if (t == 0) out[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    if (t < 64) out[gridDim.x + blockIdx.x * 64 + t] = dyn[199 - t];
    atomicAdd(counter, 1);
}
// an inter-workgroup spin barrier (the one-launch row lists' pattern): every workgroup publishes, then waits for all
__global__ void k_grid_spin(int* flags, int* out) {
    const int b = blockIdx.x, n = gridDim.x;
    if (threadIdx.x == 0) {
        __hip_atomic_store(flags + b, b + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int sum = 0;
        for (int i = 0; i < n; ++i) {
            int v;
            while ((v = __hip_atomic_load(flags + i, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) == 0) __builtin_amdgcn_s_sleep(2);
            sum += v;
        }
        out[b] = sum;
    }
}

static std::string g_msg;
static int g_fail;
static void fail(const char* what, int i, double got, double want) {
    if (++g_fail <= 12) { char b[256]; snprintf(b, sizeof b, "%s[%d]: got %g want %g\n", what, i, got, want); g_msg += b; }
}
static void check_mm(const char* what, const std::vector<float>& A, const std::vector<float>& B, const float* D, int M, int N, int K, double tol) {
    for (int i = 0; i < M; ++i)
        for (int j = 0; j < N; ++j) {
            double s = 1.0;
            for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[k * N + j];
            if (fabs(D[i * N + j] - s) > tol * (1.0 + fabs(s))) fail(what, i * N + j, D[i * N + j], s);
        }
}

extern "C" int emu_selftest(char* msg, int n) {
    g_msg.clear();
    g_fail = 0;
    srand(7);
    auto rnd = [] { return (float)(rand() % 2001 - 1000) / 500.0f; };
    float* D;
    hipMalloc(&D, 4096 * sizeof(float));
    {   // fp32 forms
        std::vector<float> A(64), B(64);
        for (auto& x : A) x = rnd();
        for (auto& x : B) x = rnd();
        hipLaunchKernelGGL(k_mfma_32x32x2, dim3(1), dim3(64), 0, 0, A.data(), B.data(), D);
        check_mm("mfma_32x32x2_f32", A, B, D, 32, 32, 2, 1e-6);
        hipLaunchKernelGGL(k_mfma_16x16x4, dim3(1), dim3(64), 0, 0, A.data(), B.data(), D);
        check_mm("mfma_16x16x4_f32", A, B, D, 16, 16, 4, 1e-6);
        hipLaunchKernelGGL(k_mfma_4x4x1, dim3(1), dim3(64), 0, 0, A.data(), B.data(), D);
        for (int b = 0; b < 16; ++b)
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const double want = 1.0 + (double)A[b * 4 + i] * B[b * 4 + j];
                    if (fabs(D[(b * 4 + i) * 4 + j] - want) > 1e-6 * (1 + fabs(want))) fail("mfma_4x4x1_f32", (b * 4 + i) * 4 + j, D[(b * 4 + i) * 4 + j], want);
                }
    }
    {   // bf16 forms
        std::vector<unsigned short> A(512), B(512);
        std::vector<float> Af(512), Bf(512);
        for (int i = 0; i < 512; ++i) { A[i] = to_bf16(rnd()); B[i] = to_bf16(rnd()); Af[i] = from_bf16(A[i]); Bf[i] = from_bf16(B[i]); }
        hipLaunchKernelGGL(k_mfma_16x16x32_bf16, dim3(1), dim3(64), 0, 0, A.data(), B.data(), D);
        check_mm("mfma_16x16x32_bf16", Af, Bf, D, 16, 16, 32, 1e-6);
        hipLaunchKernelGGL(k_mfma_32x32x16_bf16, dim3(1), dim3(64), 0, 0, A.data(), B.data(), D);
        check_mm("mfma_32x32x16_bf16", Af, Bf, D, 32, 32, 16, 1e-6);
    }
    {   // cross-lane
        int* o;
        hipMalloc(&o, 64 * 16 * sizeof(int));
        hipLaunchKernelGGL(k_cross, dim3(1), dim3(64), 0, 0, o);
        int inv5[64];
        for (int l = 0; l < 64; ++l) inv5[(l * 5 + 1) & 63] = l;
        unsigned long long bal = 0;
        for (int l = 0; l < 64; ++l) if (l % 3 == 0) bal |= 1ull << l;
        for (int l = 0; l < 64; ++l) {
            const int row = l & ~15, i = l & 15, qb = l & ~3, qi = l & 3;
            const int q1[4] = {1, 0, 3, 2}, q2[4] = {2, 3, 0, 1};
            const int want[16] = {1000 + qb + q1[qi], 1000 + qb + q2[qi], 1000 + row + (i & 8) + 7 - (i & 7), 1000 + row + 15 - i,
                                  1000 + row + ((i - 4) & 15), 1000 + row + ((i - 8) & 15), i >= 1 ? 1000 + l - 1 : -7, i <= 14 ? 1000 + l + 1 : -7,
                                  1000 + ((l * 7 + 3) & 63), 1000 + inv5[l], 1000 + (l ^ 16), l >= 3 ? 1000 + l - 3 : 1000 + l, 1037, 1000,
                                  (int)(bal & 0xffffffffu), (int)(bal >> 32)};
            for (int c = 0; c < 16; ++c)
                if (o[l * 16 + c] != want[c]) fail("cross-lane", l * 16 + c, o[l * 16 + c], want[c]);
        }
        hipLaunchKernelGGL(k_divergent, dim3(1), dim3(64), 0, 0, o);
        unsigned long long odd = 0;
        for (int l = 1; l < 64; l += 2) odd |= 1ull << l;
        for (int l = 0; l < 64; ++l) {
            const int want = l < 20 ? 100 + (l ^ 1) : (l >= 40 ? 100 + (l ^ 2) : 100 + l);
            if (o[l] != want) fail("divergent shuffle", l, o[l], want);
            const int wb = l & 1 ? (int)(odd >> 32) ^ (int)(odd & 0xffffffffu) : 0;
            if (o[64 + l] != wb) fail("divergent ballot", l, o[64 + l], wb);
        }
        hipFree(o);
    }
    {   // buffer range check: 1000 bytes = 250 dwords: lane 62's vector is cut after 2 dwords, lane 63's is gone
        std::vector<float> src(256);
        std::vector<unsigned> dst(256, 0xabcdu);
        for (int i = 0; i < 256; ++i) { unsigned u = 5u + i; memcpy(&src[i], &u, 4); }
        hipLaunchKernelGGL(k_buffer, dim3(1), dim3(64), 0, 0, src.data(), reinterpret_cast<float*>(dst.data()), 1000);
        for (int i = 0; i < 256; ++i) {
            const unsigned want = i < 250 ? 6u + i : 0xabcdu;            // in range: loaded value + 1; past num_records: the store is dropped
            if (dst[i] != want) fail("buffer range", i, dst[i], want);
        }
    }
    {   // block: partial last wave, early exits, static + dynamic LDS, atomics
        const int nb = 5;
        std::vector<float> out(nb + nb * 64, -1.f);
        int counter = 0;
        hipLaunchKernelGGL(k_block, dim3(nb), dim3(256), 200 * sizeof(float), 0, out.data(), &counter);
        for (int b = 0; b < nb; ++b) {
            if (out[b] != 200.f * 201.f / 2.f) fail("block reduce", b, out[b], 200.0 * 201.0 / 2.0);
            for (int t = 0; t < 64; ++t)
                if (out[nb + b * 64 + t] != (float)(199 - t)) fail("dynamic LDS", b * 64 + t, out[nb + b * 64 + t], 199 - t);
        }
        if (counter != nb * 200) fail("atomicAdd", 0, counter, nb * 200);
    }
    {   // co-resident workgroups
        const int nb = 12;
        std::vector<int> flags(nb, 0), out(nb, -1);
        hipLaunchKernelGGL(k_grid_spin, dim3(nb), dim3(64), 0, 0, flags.data(), out.data());
        for (int b = 0; b < nb; ++b)
            if (out[b] != nb * (nb + 1) / 2) fail("grid spin barrier", b, out[b], nb * (nb + 1) / 2);
    }
    hipFree(D);
    if (msg && n > 0) { strncpy(msg, g_msg.c_str(), (size_t)n - 1); msg[n - 1] = 0; }
    return g_fail;
}
