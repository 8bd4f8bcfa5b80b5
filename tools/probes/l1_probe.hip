// l1_probe.hip -- how many bytes per clock does one gfx950 CU get out of its vector L1 (all hits), by load width and address pattern?
// (r02: the proposal kernel K2 moves 320 B per sample through the texture path; is that path its roof?)
//   pattern 0: every lane its own 16-byte slot of one 1-KB block (fully coalesced)      pattern 1: all 64 lanes the SAME address
//   pattern 2: 8 distinct 32-byte entries per wave (lanes of an 8x8 tile sharing voxels), pattern 3: random slots inside a 16-KB window
// hipcc --offload-arch=gfx950 -O3 l1_probe.hip -o l1_probe && ./l1_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int WIDTH>
__global__ __launch_bounds__(256) void probe(const float* base, int bytes, int pattern, int iters, uint64_t* cyc, uint32_t* sink) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    uint32_t off;
    if (pattern == 0) off = lane * 16;
    else if (pattern == 1) off = 0;
    else if (pattern == 2) off = (lane & 7) * 32;
    else off = ((lane * 2654435761u) >> 8) % 1024 * 16;
    off += (threadIdx.x >> 6) * 64;  // waves of a workgroup start on different lines
    uint32_t acc = 0;
    uint64_t t0;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("" : "+v"(off));  // opaque: the eight loads are neither hoisted nor merged
            const int so = ((i * 8 + k) & 7) * 1024;  // walk 8 KB so that consecutive loads are different lines, all L1-resident
            if (WIDTH == 4) {
                u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, so, 0);
                acc ^= r.x ^ r.w;
            } else if (WIDTH == 2) {
                u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, so, 0);
                acc ^= r.x ^ r.y;
            } else {
                acc ^= __builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)off, so, 0);
            }
        }
    }
    uint64_t t1;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int WIDTH>
static void run(const float* buf, int bytes, int pattern, int wg_per_cu, uint64_t* cyc, uint32_t* sink) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(probe<WIDTH>, dim3(256 * wg_per_cu), dim3(256), 0, 0, buf, bytes, pattern, iters, cyc, sink);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        best = best < ms ? best : ms;
    }
    uint64_t c;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // per CU: wg_per_cu workgroups x 4 waves x iters x 8 loads x 64 lanes x WIDTH*4 bytes
    const double bytes_cu = (double)wg_per_cu * 4 * iters * 8 * 64 * WIDTH * 4;
    const double instr_cu = (double)wg_per_cu * 4 * iters * 8;
    // s_memtime counts at a constant 100 MHz; use wall time and report per-microsecond rates as well
    printf("dwordx%d pattern %d  %2d waves/CU: %.3f ms  -> %.1f GB/s per CU, %.2f wave-loads per ns per CU (all 256 CUs: %.1f TB/s)\n", WIDTH, pattern,
           wg_per_cu * 4, best, bytes_cu / (best * 1e-3) / 1e9, instr_cu / (best * 1e6), bytes_cu * 256 / (best * 1e-3) / 1e12);
}

int main() {
    float* buf;
    const int bytes = 64 * 1024;
    (void)hipMalloc(&buf, bytes);
    (void)hipMemset(buf, 0, bytes);
    uint64_t* cyc;
    uint32_t* sink;
    (void)hipMalloc(&cyc, 8);
    (void)hipMalloc(&sink, 4);
    for (int pattern = 0; pattern < 4; ++pattern) {
        for (int wg : {2, 4}) {
            run<4>(buf, bytes, pattern, wg, cyc, sink);
            run<2>(buf, bytes, pattern, wg, cyc, sink);
            run<1>(buf, bytes, pattern, wg, cyc, sink);
        }
    }
    return 0;
}
