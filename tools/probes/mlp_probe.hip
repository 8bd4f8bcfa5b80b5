// mlp_probe.hip -- the MLP phase of K1 (sn_main_field_h: 5 layers, 120 v_mfma_f32_32x32x16_f16 + the hi/lo operand splits) on
// synthetic features, timed alone at 1 / 2 / 3 waves per SIMD, and with its two halves removed in turn:
//   -DSN_PROBE_NOMFMA   the MFMAs become empty asm statements that keep the data dependencies (VALU + LDS part alone)
//   -DSN_PROBE_NOSPLIT  the operand splits become empty asm statements (MFMA + LDS part alone)
// Question: are the two parts additive in the real instruction stream, and at which occupancy?
// hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -mllvm -disable-vector-combine [-D...] mlp_probe.hip -o mlp_probe[_x]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#ifdef SN_PROBE_NOMFMA
#define SN_MFMA_H(ACC, A, B) asm volatile("" : "+v"(ACC) : "v"(A), "v"(B))
#endif
#include "../../signerf_amd/csrc/sn_main.h"

__global__ __launch_bounds__(256, 3) void mlp_kernel(const float* wimg, int n, float* out, uint64_t* cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid * 4; i < SnMainImg::TOTAL; i += 256 * 4) *(f32x4*)(lds + i) = *(const f32x4*)(wimg + i);
    __syncthreads();
    const int lane = tid & 63;
    SnShOpsH shh;
    const float d[3] = {0.3f + 0.001f * lane, 0.5f, -0.8f};
    shh.build(d, 0);
    float accum = 0.f;
    uint64_t t0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
        asm volatile("" ::: "memory");
        float feat[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) feat[k] = accum * 1e-3f + 0.01f * (float)(k + lane) + (float)i * 1e-4f;
        float h0, rgb[3];
        sn_main_field_h((const char*)lds, feat, shh, lane, h0, rgb);
        accum += h0 + rgb[0] + rgb[1] + rgb[2];
    }
    uint64_t t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (accum == 123.456f) out[0] = accum;
}

int main() {
    std::vector<float> img(SnMainImg::TOTAL);
    uint16_t* h = (uint16_t*)img.data();
    for (int i = 0; i < SnMainImgH::FP32 / 2; ++i) h[i] = (uint16_t)(0x2c00 + (i * 37) % 1024);  // fp16 values ~0.06..0.12
    for (int i = SnMainImgH::FP32 / 4; i < SnMainImg::TOTAL; ++i) img[i] = 0.01f * (float)(i % 17) - 0.05f;
    float *wimg, *out;
    uint64_t* cyc;
    hipMalloc(&wimg, img.size() * 4);
    hipMemcpy(wimg, img.data(), img.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 4);
    hipMalloc(&cyc, 8);
    hipFuncSetAttribute((const void*)mlp_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int n = 2000;
    for (int waves = 1; waves <= 3; ++waves) {
        // LDS per workgroup sized so that exactly `waves` workgroups (= waves per SIMD) fit a CU
        const size_t lds = waves == 1 ? 120 * 1024 : (waves == 2 ? 70 * 1024 : 44 * 1024);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f;
        uint64_t c = 0;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mlp_kernel, dim3(256 * waves), dim3(256), lds, 0, wimg, n, out, cyc);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) {
                best = ms;
                hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            }
        }
        printf("%d wave(s)/SIMD: %8.3f ms for %d wave-steps per wave -> %8.1f ns per wave-step per SIMD; wave 0: %8.1f cycles per wave-step (own clock)\n",
               waves, best, n, best * 1e6 / (n * waves), (double)c / n);
    }
    return 0;
}
