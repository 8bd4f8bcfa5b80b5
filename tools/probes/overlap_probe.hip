// overlap_probe.hip -- do the matrix pipe and the VALU of one SIMD run concurrently when the instructions come from DIFFERENT
// waves?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run a chain-free MFMA loop, waves 4-7 (the second wave of
// each SIMD) run a VALU FMA loop.  Timed: MFMA waves alone, VALU waves alone, both.  both ~ max(...) => concurrent;
// both ~ sum => the pipes serialise.  Also a gather variant (waves 4-7 issue L2-resident 64-lane gathers instead of FMAs).
// Standalone: hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND /*0 f32 32x32x2, 1 f16 32x32x16*/>
__device__ float mfma_loop(int n, float seed) {
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x8 ha, hb;
    for (int e = 0; e < 8; ++e) {
        ha[e] = (_Float16)seed;
        hb[e] = (_Float16)(seed * 0.5f);
    }
    for (int i = 0; i < n; ++i) {
        if (KIND == 0) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a3, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a3, 0, 0, 0);
        }
    }
    return a0[0] + a1[1] + a2[2] + a3[3];
}

// 16 independent full-rate integer chains (v_xad_u32: not packable, 4 cycles per wave64 instruction): issue-bound VALU work
__device__ float valu_loop(int n, float seed) {
    unsigned x[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) x[u] = (unsigned)seed * 977u + u;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int u = 0; u < 16; ++u) x[u] = (x[u] ^ 0x9e3779b9u) + (x[(u + 1) & 15] | 1u);
    }
    unsigned t = 0;
#pragma unroll
    for (int u = 0; u < 16; ++u) t ^= x[u];
    return (float)t;
}

__device__ float gather_loop(const float* table, unsigned table_bytes, int n, unsigned wave) {
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)table_bytes, 0x00020000);
    unsigned state = wave * 2654435761u + (threadIdx.x & 63) * 805459861u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < n; ++i) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        u2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            state = state * 1664525u + 1013904223u;
            v[k] = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)((state >> 8) & (table_bytes - 8)), 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += __uint_as_float(v[k].x);
    }
    return acc;
}

// mode bit 0: MFMA waves run; bit 1: second-wave workload runs.  other: 0 = VALU FMAs, 1 = gathers
template <int KIND, int OTHER>
__global__ __launch_bounds__(512) void probe(int mode, int n_mfma, int n_other, const float* table, unsigned table_bytes, float* out) {
    extern __shared__ float pad[];  // sized so that ONE workgroup fits per CU
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (mode & 1) r = mfma_loop<KIND>(n_mfma, (float)threadIdx.x * 1e-3f);
    } else {
        if (mode & 2) r = OTHER == 0 ? valu_loop(n_other, (float)threadIdx.x) : gather_loop(table, table_bytes, n_other, blockIdx.x * 8 + wave);
    }
    if (r == 123.456f) out[0] = r + pad[0];
}

template <int KIND, int OTHER>
static void run(const char* name, int n_mfma, int n_other, const float* table, unsigned table_bytes, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((probe<KIND, OTHER>), dim3(256), dim3(512), 100 * 1024, 0, mode, n_mfma, n_other, table, table_bytes, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
    printf("%-34s mfma alone %7.3f ms   other alone %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n", name, ms[1], ms[2], ms[3],
           ms[1] + ms[2], ms[1] > ms[2] ? ms[1] : ms[2]);
}


// Same-wave variant: every wave runs 4 MFMAs and NV independent FMAs per iteration, interleaved by the compiler's choice or
// (SEP) separated by scheduling fences.  One wave per SIMD (256-thread workgroup, one per CU).
template <int KIND, int NV, bool DO_MFMA, bool DO_VALU>
__global__ __launch_bounds__(256) void same_wave(int n, float* out) {
    extern __shared__ float pad[];
    const float seed = (float)threadIdx.x * 1e-3f;
    f32x16 a0 = {}, a1 = {}, a2 = {}, a3 = {};
    f16x8 ha, hb;
    for (int e = 0; e < 8; ++e) {
        ha[e] = (_Float16)seed;
        hb[e] = (_Float16)(seed * 0.5f);
    }
    unsigned x[8];
    for (int u = 0; u < 8; ++u) x[u] = (unsigned)threadIdx.x * 977u + u;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (DO_MFMA) {
                f32x16& a = m == 0 ? a0 : m == 1 ? a1 : m == 2 ? a2 : a3;
                if (KIND == 0) a = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a, 0, 0, 0);
                else a = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, a, 0, 0, 0);
            }
            if (DO_VALU) {
#pragma unroll
                for (int v = 0; v < NV / 4; ++v) x[v & 7] = (x[v & 7] ^ 0x9e3779b9u) + (x[(v + 1) & 7] | 1u);
            }
        }
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int u = 0; u < 8; ++u) r += (float)x[u];
    if (r == 123.456f) out[0] = r + pad[0];
}

template <int KIND, int NV>
static void run_same(const char* name, int n, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[3];
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL((same_wave<KIND, NV, true, false>), dim3(256), dim3(256), 100 * 1024, 0, n, out);
            if (mode == 1) hipLaunchKernelGGL((same_wave<KIND, NV, false, true>), dim3(256), dim3(256), 100 * 1024, 0, n, out);
            if (mode == 2) hipLaunchKernelGGL((same_wave<KIND, NV, true, true>), dim3(256), dim3(256), 100 * 1024, 0, n, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[mode], e0, e1);
        }
    printf("%-34s mfma only %7.3f ms   valu only %7.3f ms   interleaved %7.3f ms   (sum %7.3f, max %7.3f)\n", name, ms[0], ms[1], ms[2],
           ms[0] + ms[1], ms[0] > ms[1] ? ms[0] : ms[1]);
}

int main() {
    const unsigned table_bytes = 2u << 20;
    float *table, *out;
    hipMalloc(&table, table_bytes);
    hipMemset(table, 0, table_bytes);
    hipMalloc(&out, 4);
    hipFuncSetAttribute((const void*)probe<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)probe<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    // 4 MFMAs per iteration: f32 32x32x2 = 64 cycles each, f16 32x32x16 = 32 cycles each; VALU loop = 32 integer ops per iteration
    run<0, 0>("f32 32x32x2 MFMA | VALU fma", 20000, 40000, table, table_bytes, out);
    run<1, 0>("f16 32x32x16 MFMA | VALU fma", 40000, 40000, table, table_bytes, out);
    run<0, 1>("f32 32x32x2 MFMA | 64-lane gathers", 20000, 20000, table, table_bytes, out);
    run<1, 1>("f16 32x32x16 MFMA | 64-lane gathers", 40000, 20000, table, table_bytes, out);
    hipFuncSetAttribute((const void*)same_wave<0, 32, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<0, 32, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<0, 32, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<1, 16, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<1, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<1, 16, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<1, 32, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<1, 32, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipFuncSetAttribute((const void*)same_wave<1, 32, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    // same wave: 4 MFMAs + NV FMAs per iteration (f32: 256 MFMA cycles vs 32 FMAs = 128 cycles; f16: 128 vs 16 FMAs = 64 / 32 FMAs = 128)
    run_same<0, 32>("same wave: f32 MFMA + 32 FMA/iter", 20000, out);
    run_same<1, 16>("same wave: f16 MFMA + 16 FMA/iter", 40000, out);
    run_same<1, 32>("same wave: f16 MFMA + 32 FMA/iter", 40000, out);
    return 0;
}
