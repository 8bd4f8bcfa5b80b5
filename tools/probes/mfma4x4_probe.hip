// mfma4x4_probe.hip -- r03: would K1's colour layer 3 (64 -> 3, today 192 v_fmac + 64 v_max on the VALU) be cheaper on the matrix cores as
// 64 x v_mfma_f32_4x4x1_16b_f32?  That instruction takes fp32 operands (no fp16 hi / lo split), multiplies one hidden unit into 4 output rows
// for 4 samples per 4-lane block, 16 blocks per wave -- and the accumulators of colour layer 2 already hold "one register = 2 hidden units x
// 32 samples".  r02 found the fp32-INPUT 32x32x2 MFMA to run at the vector rate and to exclude the VALU; is the 4x4x1 form the same?
//   A. one wave per SIMD: cycles per {4x4x1 MFMA + NF independent v_fma_f32}
//   B. 3 waves per SIMD, K1-shaped iteration: {NV v_fma ; 120 f16 32x32x16 MFMAs ; N4 4x4x1 MFMAs}: today's mix (1428, 120, 0) against the
//      candidate (1240, 120, 64)
// hipcc --offload-arch=gfx950 -O3 mfma4x4_probe.hip -o mfma4x4_probe && ./mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__

DEV uint64_t memtime() {
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}

template <int NF>
__global__ __launch_bounds__(256) void same_wave(int n, uint64_t* cyc, float* out) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x[8], a = threadIdx.x * 1e-3f, b = 0.5f, c1 = 0.999f, c2 = 1e-3f;
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    const uint64_t t0 = memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * NF + f) & 7]) : "v"(c1), "v"(c2));
        }
    }
    const uint64_t t1 = memtime();
    asm volatile("s_nop 15\n s_nop 15");
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int m = 0; m < 4; ++m) r += acc[m][0] + acc[m][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (r == 123.456f) out[0] = r;
}

template <int NV, int NB, int N4, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void phases(int n, float* out) {
    extern __shared__ float pad[];
    f32x16 big[4];
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) big[i][r] = 0.f;
    f16x8 ha, hb;
    for (int e = 0; e < 8; ++e) {
        ha[e] = (_Float16)(threadIdx.x * 1e-5f);
        hb[e] = (_Float16)0.5f;
    }
    float x[8], a = threadIdx.x * 1e-3f, b = 0.5f, c1 = 0.999f, c2 = 1e-3f;
    for (int i = 0; i < 8; ++i) x[i] = a + i;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int f = 0; f < NV; ++f) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 7]) : "v"(c1), "v"(c2));
#pragma unroll
        for (int m = 0; m < NB; ++m) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(big[m & 3]) : "v"(ha), "v"(hb));
#pragma unroll
        for (int m = 0; m < N4; ++m) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(x[m & 7]), "v"(b));
    }
    asm volatile("s_nop 15\n s_nop 15");
    float r = 0.f;
    for (int i = 0; i < 8; ++i) r += x[i];
    for (int m = 0; m < 4; ++m) r += acc[m][0] + big[m][0];
    if (r == 123.456f) out[0] = r + pad[0];
}

template <int NV, int NB, int N4, int WAVES>
static float run_phases(int n, float* out) {
    auto k = phases<NV, NB, N4, WAVES>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 100 * 1024 / (WAVES / 4 > 2 ? 2 : 1), 0, n, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        best = std::min(best, t);
    }
    printf("B phases  %d waves/SIMD, per iteration %4d v_fma + %3d MFMA f16 32x32x16 + %2d MFMA f32 4x4x1: %.3f ms for %d iterations\n", WAVES / 4, NV, NB, N4, best, n);
    return best;
}

template <int NF>
static void run_same(int n, uint64_t* cyc, float* out) {
    uint64_t bestc = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(same_wave<NF>, dim3(256), dim3(256), 0, 0, n, cyc, out);
        hipDeviceSynchronize();
        uint64_t c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        bestc = std::min(bestc, c);
    }
    printf("A same-wave  v_mfma_f32_4x4x1_16b_f32 + %d v_fma_f32: %.1f cycles per group\n", NF, (double)bestc / (4.0 * n));
}

int main() {
    uint64_t* cyc;
    float* out;
    hipMalloc(&cyc, 64);
    hipMalloc(&out, 4);
    run_same<0>(20000, cyc, out);
    run_same<1>(20000, cyc, out);
    run_same<2>(20000, cyc, out);
    run_same<4>(20000, cyc, out);
    run_same<8>(20000, cyc, out);
    run_phases<1428, 120, 0, 12>(600, out);
    run_phases<1240, 120, 64, 12>(600, out);
    run_phases<1240, 120, 0, 12>(600, out);
    run_phases<1428, 0, 0, 12>(600, out);
    run_phases<1240, 0, 64, 12>(600, out);
    run_phases<0, 0, 64, 12>(6000, out);
    return 0;
}
