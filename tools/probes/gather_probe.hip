// gather_probe.hip -- what does one 64-lane gather cost in the L1 (TCP) as a function of the number of distinct 128-B lines it
// touches and of the access width?  Standalone (hipcc gather_probe.hip -o gather_probe); run under
//   rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d out -- ./gather_probe
// Each wave issues N_ITER gathers into a table that fits the L2 (hits; we are probing the L1 front end, not DRAM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int WIDTH /*dwords per lane: 2 or 4*/>
__global__ void probe(const float* __restrict__ table, unsigned table_bytes, int groups, int n_iter, float* out) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)table_bytes, 0x00020000);
    // lanes are split into `groups` groups; each group reads one (pseudo-random, per iteration) 128-B line; within a group the
    // lanes read consecutive WIDTH*4-byte chunks of that line (wrapping), i.e. the best case for that many lines.
    const int per = 64 / groups;
    const unsigned g = lane / per, within = (lane % per) * (WIDTH * 4) % 128;
    unsigned state = wave * 2654435761u + g * 805459861u + 12345u;
    const unsigned line_mask = table_bytes / 128 - 1;
    float acc = 0.f;
    for (int i = 0; i < n_iter; ++i) {
        state = state * 1664525u + 1013904223u;
        const unsigned off = ((state >> 8) & line_mask) * 128u + within;
        if (WIDTH == 2) {
            typedef unsigned u2 __attribute__((ext_vector_type(2)));
            u2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, 0, 0);
            acc += __uint_as_float(v.x) + __uint_as_float(v.y);
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0);
            acc += __uint_as_float(v.x) + __uint_as_float(v.w);
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const unsigned table_bytes = 2u << 20;  // 2 MiB: L2-resident per XCD
    float *table, *out;
    hipMalloc(&table, table_bytes);
    hipMemset(table, 0, table_bytes);
    hipMalloc(&out, 4);
    const int n_iter = 4096, blocks = 256 * 8, threads = 256;  // 8 workgroups of 4 waves per CU
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("width groups  us  cycles_per_gather_per_CU(at 2.3GHz)\n");
    for (int width : {2, 4})
        for (int groups : {1, 2, 4, 8, 16, 32, 64}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (width == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(threads), 0, 0, table, table_bytes, groups, n_iter, out);
                else hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(threads), 0, 0, table, table_bytes, groups, n_iter, out);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double gathers_per_cu = (double)blocks * (threads / 64) * n_iter / 256.0;
            printf("%5d %6d %8.1f %8.2f\n", width, groups, ms * 1e3, ms * 1e-3 * 2.3e9 / gathers_per_cu);
        }
    return 0;
}
