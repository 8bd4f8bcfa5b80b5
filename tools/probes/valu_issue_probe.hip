// valu_issue_probe.hip -- round 6: what does ONE plain VALU instruction of a 64-lane wave cost the vector issue port of a gfx950 SIMD?
// (VERDICT r05 "weak 5": bench.py prices the simd-issue roof at 4 cycles per wave-instruction; MI355X_MICROARCH.md "Per-instruction
// cycle constants" lists `v_fma_f32 (wave64) 2 cyc (SIMD-32)` and, three rows further down, "32 cyc/SIMD ~ 8 issue slots of ~4 cyc".)
//
// Method: W waves per SIMD (W = 1, 2, 3, 4, 6, 8), every wave a loop of UNROLL asm-volatile instructions of one kind over ILP independent
// registers (ILP = 1: a dependent chain; 8: eight independent chains -- no instruction waits for its own result).  No memory, no MFMA, no
// LDS.  One workgroup of 4 W waves per CU (W <= 4), two for W = 6, 8; LDS padding keeps further workgroups off the CU.  Reported:
//   cyc/inst/SIMD = shader cycles (s_memtime, max over the waves of a CU, median over CUs) / (instructions per wave x W)
// i.e. the port time one wave-instruction takes when the SIMD always has another wave ready -- the denominator of an issue roof.
// The fp32 vector peak of the part (157.3 TFLOP/s = 256 CUs x 4 SIMDs x 2.4 GHz x 64 flop/cycle) is reached by `v_pk_fma_f32`
// (4 flop per lane) at 4 cycles per wave-instruction, or by `v_fma_f32` (2 flop per lane) at 2: the v_pk_fma_f32 row says which.
//
//   hipcc --offload-arch=gfx950 -O3 valu_issue_probe.hip -o valu_issue_probe && ./valu_issue_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DEV __device__ __forceinline__

enum { K_FMA = 0, K_PKFMA, K_ADDU, K_MAD24, K_CVTPK, K_FMAMIX, K_XOR, K_RCP,
       // the rest of K1's opcode mix (tools/kernel_counts.py --opcodes)
       K_FMAC, K_SUB, K_MUL, K_PKMAXH, K_MAXI, K_CVTI, K_FRACT, K_BITOP3, K_LSHL, K_MUL24, K_SWAP, K_EXP, K_ADDF64, K_CVTF64, K_CNDMASK, K_MAX3, K_MOV,
       K_NKIND };
static const char* kind_name[K_NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_add_u32", "v_mad_u32_u24", "v_cvt_pkrtz_f16_f32", "v_fma_mix_f32", "v_xor_b32", "v_rcp_f32",
                                         "v_fmac_f32", "v_sub_f32", "v_mul_f32", "v_pk_max_f16", "v_max_i32", "v_cvt_i32_f32", "v_fract_f32", "v_bitop3_b32",
                                         "v_lshlrev_b32", "v_mul_u32_u24", "v_permlane32_swap", "v_exp_f32", "v_add_f64", "v_cvt_f64_f32", "v_cndmask_b32(vcc)",
                                         "v_max3_f32", "v_mov_b32"};

// One asm statement holds a whole unrolled body (hipcc puts an `s_nop 0` between two asm statements that touch the same VGPR; inside one
// statement the stream is exactly the written one).  BODY8(I) = the instruction I over the eight registers %0..%7 (ILP 8) ; BODY1(I) = eight
// times over %0 (ILP 1: a dependent chain).  %8 / %9 = the two constants; packed kinds use register PAIRS (%0..%7 are 64-bit operands).
#define REP8(X) X X X X X X X X
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %8\n"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_CVTPK(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
#define I_FMAMIX(n) "v_fma_mix_f32 %" #n ", %" #n ", -1.0, %9 op_sel_hi:[1,0,0]\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define I_SUB(n) "v_sub_f32 %" #n ", %" #n ", %9\n"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PKMAXH(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
#define I_MAXI(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_FRACT(n) "v_fract_f32 %" #n ", %" #n "\n"
#define I_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x28\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_SWAP(n) "v_permlane32_swap_b32 %" #n ", %8\n"
#define I_EXP(n) "v_exp_f32 %" #n ", %" #n "\n"
#define I_ADDF64(n) "v_add_f64 %" #n ", %" #n ", %8\n"
#define I_CVTF64(n) "v_cvt_f64_f32 %" #n ", %9\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_MAX3(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define BODY8(I) REP8(I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7))
#define BODY1(I) REP8(I(0) I(0) I(0) I(0) I(0) I(0) I(0) I(0))
#define EMIT(BODY, T, x, c1, c2) \
    asm volatile(BODY : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(c1), "v"(c2))

// 64 instructions of kind K over ILP (8 or 1) independent registers
template <int K, int ILP>
DEV void body(float* x, f32x2* xp, float c1, float c2, f32x2 cp) {
#define CASE(KK, I, X, C1, C2)                     \
    if (K == KK) {                                \
        if (ILP == 8) EMIT(BODY8(I), , X, C1, C2); \
        else EMIT(BODY1(I), , X, C1, C2);          \
    }
    CASE(K_FMA, I_FMA, x, c1, c2)
    CASE(K_PKFMA, I_PKFMA, xp, cp, cp)
    CASE(K_ADDU, I_ADDU, x, c1, c2)
    CASE(K_MAD24, I_MAD24, x, c1, c2)
    CASE(K_CVTPK, I_CVTPK, x, c1, c2)
    CASE(K_FMAMIX, I_FMAMIX, x, c1, c2)
    CASE(K_XOR, I_XOR, x, c1, c2)
    CASE(K_RCP, I_RCP, x, c1, c2)
    CASE(K_FMAC, I_FMAC, x, c1, c2)
    CASE(K_SUB, I_SUB, x, c1, c2)
    CASE(K_MUL, I_MUL, x, c1, c2)
    CASE(K_PKMAXH, I_PKMAXH, x, c1, c2)
    CASE(K_MAXI, I_MAXI, x, c1, c2)
    CASE(K_CVTI, I_CVTI, x, c1, c2)
    CASE(K_FRACT, I_FRACT, x, c1, c2)
    CASE(K_BITOP3, I_BITOP3, x, c1, c2)
    CASE(K_LSHL, I_LSHL, x, c1, c2)
    CASE(K_MUL24, I_MUL24, x, c1, c2)
    CASE(K_EXP, I_EXP, x, c1, c2)
    CASE(K_ADDF64, I_ADDF64, xp, cp, cp)
    CASE(K_CNDMASK, I_CNDMASK, x, c1, c2)
    CASE(K_MAX3, I_MAX3, x, c1, c2)
    CASE(K_MOV, I_MOV, x, c1, c2)
#undef CASE
    // (two-destination / mixed-width kinds spelled out)
    if (K == K_SWAP) {   // swaps the upper half of %n with the lower half of a second register: both are written
        float y = c1;
        if (ILP == 8) asm volatile(BODY8(I_SWAP) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(y) : "v"(c2));
        else asm volatile(BODY1(I_SWAP) : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(y) : "v"(c2));
        x[0] += y;
    }
    if (K == K_CVTF64) {  // 64-bit destinations, 32-bit source
        if (ILP == 8) asm volatile(BODY8(I_CVTF64) : "+v"(xp[0]), "+v"(xp[1]), "+v"(xp[2]), "+v"(xp[3]), "+v"(xp[4]), "+v"(xp[5]), "+v"(xp[6]), "+v"(xp[7]) : "v"(c1), "v"(c2));
        else asm volatile(BODY1(I_CVTF64) : "+v"(xp[0]), "+v"(xp[1]), "+v"(xp[2]), "+v"(xp[3]), "+v"(xp[4]), "+v"(xp[5]), "+v"(xp[6]), "+v"(xp[7]) : "v"(c1), "v"(c2));
    }
}

constexpr int UNROLL = 64;

template <int K, int ILP>
__global__ __launch_bounds__(1024) void probe(int iters, float c1, float c2, unsigned long long* cyc, float* sink) {
    extern __shared__ float pad[];
    float x[8];
    f32x2 xp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x[j] = 1.0f + 0.001f * (threadIdx.x + j);
        xp[j] = f32x2{x[j], x[j] + 0.5f};
    }
    const f32x2 cp = {c1, c2};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();  // s_memtime: shader cycles
    for (int it = 0; it < iters; ++it) {
        body<K, ILP>(x, xp, c1, c2, cp);
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j] + xp[j].x + xp[j].y;
    if (s == 12345.678f) sink[0] = s + pad[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));            \
            return 1;                                                          \
        }                                                                      \
    } while (0)

template <int K, int ILP>
static int run(int waves_per_simd, unsigned long long* d_cyc, float* d_sink, int n_cu) {
    const int iters = 2000;
    // one workgroup per CU up to 4 waves per SIMD (1024 threads); 6 / 8 waves per SIMD = two workgroups of 12 / 16 waves per CU
    const int wgs_per_cu = waves_per_simd > 4 ? 2 : 1;
    const int waves_per_wg = 4 * waves_per_simd / wgs_per_cu;
    const int lds = wgs_per_cu == 1 ? 96 * 1024 : 64 * 1024;  // 160 KB per CU: one (two) such workgroup(s) fit, no more
    CK(hipFuncSetAttribute((const void*)probe<K, ILP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int blocks = n_cu * wgs_per_cu;
    CK(hipMemset(d_cyc, 0, sizeof(unsigned long long) * 16 * blocks));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((probe<K, ILP>), dim3(blocks), dim3(64 * waves_per_wg), lds, 0, 10, 1.0001f, 0.5f, d_cyc, d_sink);  // warm-up
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((probe<K, ILP>), dim3(blocks), dim3(64 * waves_per_wg), lds, 0, iters, 1.0001f, 0.5f, d_cyc, d_sink);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(16 * blocks);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
    std::vector<double> per_block;
    for (int bk = 0; bk < blocks; ++bk) {
        unsigned long long mx = 0;
        for (int w = 0; w < waves_per_wg; ++w) mx = std::max(mx, h[bk * 16 + w]);
        per_block.push_back((double)mx);
    }
    std::sort(per_block.begin(), per_block.end());
    const double cyc = per_block[per_block.size() / 2];
    const double n_inst = (double)iters * UNROLL;
    const double per_simd = cyc / (n_inst * waves_per_simd);
    // wall-clock view: wave-instructions per second over the whole chip, and the cycles that is at the clock the cycle counter implies
    const double ghz = cyc / (ms * 1e-3) / 1e9;
    const double ginst = n_inst * waves_per_simd * 4.0 * n_cu / (ms * 1e-3) / 1e9;
    printf("%-20s ILP %d  waves/SIMD %d | cycles/inst/wave %6.2f | cycles/inst/SIMD %5.2f | wall %7.3f ms, clock %.2f GHz, %7.1f G wave-inst/s (chip), "
           "= %5.2f cycles/inst/SIMD at that clock\n",
           kind_name[K], ILP, waves_per_simd, cyc / n_inst, per_simd, ms, ghz, ginst, 4.0 * n_cu * ghz / ginst);
    if (ghz < 1.4)   // the cycle counter saw less than the wall clock did: the two workgroups of a CU did not run side by side
        printf("    ^ implausible clock: with two workgroups per CU (6 / 8 waves per SIMD) the second one started late -- ignore this row\n");
    return 0;
}

template <int K>
static int sweep(unsigned long long* d_cyc, float* d_sink, int n_cu, bool full) {
    if (run<K, 1>(1, d_cyc, d_sink, n_cu)) return 1;
    if (run<K, 8>(1, d_cyc, d_sink, n_cu)) return 1;
    const int ws[] = {2, 3, 4, 6, 8};
    for (int w : ws) {
        if (!full && w != 3 && w != 4 && w != 8) continue;
        if (run<K, 8>(w, d_cyc, d_sink, n_cu)) return 1;
    }
    if (full && run<K, 1>(3, d_cyc, d_sink, n_cu)) return 1;
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %.0f MHz\n", p.gcnArchName, n_cu, p.clockRate / 1000.0);
    unsigned long long* d_cyc;
    float* d_sink;
    CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * 16 * 2 * n_cu));
    CK(hipMalloc(&d_sink, 64));
    if (sweep<K_FMA>(d_cyc, d_sink, n_cu, true)) return 1;
    if (sweep<K_PKFMA>(d_cyc, d_sink, n_cu, true)) return 1;
    if (sweep<K_ADDU>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_MAD24>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_CVTPK>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_FMAMIX>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_XOR>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_RCP>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_FMAC>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_SUB>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_MUL>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_PKMAXH>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_MAXI>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_CVTI>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_FRACT>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_BITOP3>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_LSHL>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_MUL24>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_SWAP>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_EXP>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_ADDF64>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_CVTF64>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_CNDMASK>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_MAX3>(d_cyc, d_sink, n_cu, false)) return 1;
    if (sweep<K_MOV>(d_cyc, d_sink, n_cu, false)) return 1;
    printf("reading: 'cycles/inst/SIMD' at 3+ waves per SIMD with ILP 8 is the port time of one wave64 instruction.  fp32 vector peak of the part = "
           "256 CUs x 4 SIMDs x 2.4 GHz x 64 flop/cycle = 157.3 TFLOP/s: reached by v_pk_fma_f32 (256 flop per wave-instruction) at 4 cycles, "
           "by v_fma_f32 (128 flop) only if it took 2.\n");
    return 0;
}
