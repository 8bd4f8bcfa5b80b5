// overlap2_probe.hip -- round 2 re-measurement of "do the matrix pipe and the VALU of one gfx950 SIMD overlap?"
// (/opt/skills/guides/MI355X_MICROARCH.md says an MFMA hides <= 5 single-issue VALU per 32-cycle gap,
// r01's overlap_probe.hip said "sum, not max").  Every instruction is an `asm volatile`, so the emitted stream is exactly the
// written one (check with llvm-objdump); times are shader cycles from s_memtime of wave 0 AND wall time.
//
//   A. same wave:  loop { MFMA ; NF fillers } with 4 independent accumulators, one wave per SIMD; NF = 0..12, per filler kind
//   B. two waves of one SIMD: waves 0-3 MFMA only, waves 4-7 fillers only (alone / together, with s_setprio variants)
//   C. phases, 3 waves per SIMD all running {NA fillers ; NB MFMAs} like K1: A only, B only, both
//
// hipcc --offload-arch=gfx950 -O3 overlap2_probe.hip -o overlap2_probe && ./overlap2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define DEV __device__ __forceinline__

enum { F_FMA = 0, F_PKFMA, F_XAD, F_CVTPK, F_MUL24, F_EXP, F_PKADD, F_XOR, F_FMAMIX, F_MAXI, F_FRACT, F_MOV, F_MIXLO, F_MIXHI, F_PKMAXH, F_MIXCLAMP, F_CVTI, F_BITOP3, F_FMAC, F_SUB, F_ADD3, F_MAD24, F_LSHL, F_CNDMASK, F_ADDF64, F_CVTF64, F_SWAP, F_FLOOR, F_RCP, F_CVTU, F_CNDMASK64, F_BFI, F_CMP, F_CNDMASKC, F_NKIND };
static const char* kind_name[F_NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_xad_u32", "v_cvt_pkrtz", "v_mul_u32_u24", "v_exp_f32",
                                         "v_pk_add_f32", "v_xor_b32", "v_fma_mix_f32", "v_max_i32", "v_fract_f32", "v_mov_b32",
                                         "v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_pk_max_f16", "fma_mix clamp",
                                         "v_cvt_i32_f32", "v_bitop3_b32", "v_fmac_f32", "v_sub_f32", "v_add3_u32", "v_mad_u32_u24", "v_lshlrev_b32",
                                         "v_cndmask_b32", "v_add_f64", "v_cvt_f64_f32", "v_permlane32_swap", "v_floor_f32", "v_rcp_f32", "v_cvt_f32_u32",
                                         "v_cndmask_b32_e64(sgpr)", "v_bfi_b32", "v_cmp_lt_f32", "v_cndmask const"};

template <int K>
DEV void filler(float& x, f32x2& xp, float c1, float c2, f32x2 cp) {
    if (K == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(xp) : "v"(cp));
    if (K == F_XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_CVTPK) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (K == F_MUL24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (K == F_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    if (K == F_PKADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp) : "v"(cp));
    if (K == F_XOR) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (K == F_FMAMIX) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(c2));
    if (K == F_MAXI) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x) : "v"(c1));
    if (K == F_FRACT) asm volatile("v_fract_f32 %0, %0" : "+v"(x));
    if (K == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(c1));
    // 16-bit results written into one half of the destination (r02: tried for the low part of the operand split, slower in K1)
    if (K == F_MIXLO) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_MIXHI) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_PKMAXH) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(x) : "v"(c1));
    // the rest of K1's opcode mix (r02: which of them are not full rate?)
    if (K == F_CVTI) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x));
    if (K == F_BITOP3) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x28" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_FMAC) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_SUB) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(c2));
    if (K == F_ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_MAD24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_LSHL) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x));
    if (K == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c1));
    if (K == F_ADDF64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(xp) : "v"(cp));
    if (K == F_CVTF64) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(xp) : "v"(c1));
    if (K == F_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(xp.x));
    if (K == F_FLOOR) asm volatile("v_floor_f32 %0, %0" : "+v"(x));
    if (K == F_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));
    if (K == F_CVTU) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(x));
    if (K == F_CNDMASK64) {
        uint64_t m = 0x5555555555555555ull;
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "s"(m));
    }
    if (K == F_BFI) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(x) : "v"(c1), "v"(c2));
    if (K == F_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(c1) : "vcc");
    if (K == F_CNDMASKC) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(x) : "v"(c1));
    if (K == F_MIXCLAMP) asm volatile("v_fma_mix_f32 %0, %0, -1.0, %1 op_sel_hi:[1,0,0] clamp" : "+v"(x) : "v"(c2));
}

template <int MK>
DEV void mfma(f32x16& acc, const f16x8& a, const f16x8& b, float fa, float fb) {
    if (MK == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    if (MK == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
}

// r04: the 16x16x32 tile (K = 32 per instruction, a FOUR-register accumulator, 16 cycles of matrix pipe): the same MACs as one 32x32x16
// in two instructions, half of the accumulator read-modify-write traffic
DEV void mfma_small(f32x4& acc, const f16x8& a, const f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

DEV uint64_t memtime() {
    uint64_t t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t));
    return t;
}
DEV uint32_t hw_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
    return v;
}

struct State {
    float x[8];
    f32x2 xp[8];
    f32x16 acc[4];
    f32x4 acc4[8];
    f16x8 a, b;
    float c1, c2, fa, fb;
    f32x2 cp;
    DEV void init() {
        const float s = (float)threadIdx.x * 1e-3f;
        for (int i = 0; i < 8; ++i) {
            x[i] = s + i;
            xp[i] = f32x2{s, s + 1.f};
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int i = 0; i < 8; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < 8; ++e) {
            a[e] = (_Float16)(s * 0.01f);
            b[e] = (_Float16)0.5f;
        }
        c1 = 0.999f;
        c2 = 1e-3f;
        fa = s * 0.01f;
        fb = 0.5f;
        cp = f32x2{0.999f, 0.998f};
    }
    DEV float fold() {
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15");
        float r = 0.f;
        for (int i = 0; i < 8; ++i) r += x[i] + xp[i].x + xp[i].y;
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][15];
        for (int i = 0; i < 8; ++i) r += acc4[i][0] + acc4[i][3];
        return r;
    }
};

// ---- A: same wave -------------------------------------------------------------------------------------------------
template <int MK, int VK, int NF>
__global__ __launch_bounds__(256) void same_wave(int n, uint64_t* cyc, float* out) {
    extern __shared__ float pad[];
    State s;
    s.init();
    const uint64_t t0 = memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            mfma<MK>(s.acc[m], s.a, s.b, s.fa, s.fb);
#pragma unroll
            for (int f = 0; f < NF; ++f) filler<VK>(s.x[(m * NF + f) & 7], s.xp[(m * NF + f) & 7], s.c1, s.c2, s.cp);
        }
    }
    const uint64_t t1 = memtime();
    const float r = s.fold();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (r == 123.456f) out[0] = r + pad[0];
}

template <int MK, int VK, int NF>
static void run_same(int n, uint64_t* cyc, float* out, double& cycles_per_group, float& ms_out) {
    auto k = same_wave<MK, VK, NF>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    uint64_t bestc = ~0ull;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 100 * 1024, 0, n, cyc, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        uint64_t c;
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        best = std::min(best, ms);
        bestc = std::min(bestc, c);
    }
    cycles_per_group = (double)bestc / (4.0 * n);
    ms_out = best;
}

template <int MK, int VK>
static void sweep_same(int n, uint64_t* cyc, float* out) {
    double c[8];
    float ms[8];
    run_same<MK, VK, 0>(n, cyc, out, c[0], ms[0]);
    run_same<MK, VK, 2>(n, cyc, out, c[1], ms[1]);
    run_same<MK, VK, 4>(n, cyc, out, c[2], ms[2]);
    run_same<MK, VK, 5>(n, cyc, out, c[3], ms[3]);
    run_same<MK, VK, 6>(n, cyc, out, c[4], ms[4]);
    run_same<MK, VK, 8>(n, cyc, out, c[5], ms[5]);
    run_same<MK, VK, 12>(n, cyc, out, c[6], ms[6]);
    run_same<MK, VK, 16>(n, cyc, out, c[7], ms[7]);
    printf("A same-wave  %-14s %-14s cycles per {MFMA + NF fillers}, NF = 0 2 4 5 6 8 12 16: ",
           MK == 0 ? "no MFMA" : (MK == 1 ? "f16 32x32x16" : "f32 32x32x2"), kind_name[VK]);
    for (int i = 0; i < 8; ++i) printf("%6.1f ", c[i]);
    printf("\n");
}

// ---- B: two waves per SIMD, different roles -------------------------------------------------------------------------
template <int MK, int VK>
__global__ __launch_bounds__(512) void two_waves(int mode, int prio_m, int prio_v, int n_m, int n_v, uint64_t* cyc, uint32_t* ids, float* out) {
    extern __shared__ float pad[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    State s;
    s.init();
    const bool is_m = wave < 4;
    const int prio = is_m ? prio_m : prio_v;
    if (prio == 1) __builtin_amdgcn_s_setprio(1);
    if (prio == 2) __builtin_amdgcn_s_setprio(2);
    if (prio == 3) __builtin_amdgcn_s_setprio(3);
    __syncthreads();
    const uint64_t t0 = memtime();
    if (is_m) {
        if (mode & 1)
            for (int i = 0; i < n_m; ++i) {
#pragma unroll
                for (int m = 0; m < 4; ++m) mfma<MK>(s.acc[m], s.a, s.b, s.fa, s.fb);
            }
    } else {
        if (mode & 2)
            for (int i = 0; i < n_v; ++i) {
#pragma unroll
                for (int f = 0; f < 32; ++f) filler<VK>(s.x[f & 7], s.xp[f & 7], s.c1, s.c2, s.cp);
            }
    }
    const uint64_t t1 = memtime();
    const float r = s.fold();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {
        cyc[wave] = t1 - t0;
        ids[wave] = hw_id();
    }
    if (r == 123.456f) out[0] = r + pad[0];
}

template <int MK, int VK>
static void run_two(int n_m, int n_v, int prio_m, int prio_v, uint64_t* cyc, uint32_t* ids, float* out) {
    auto k = two_waves<MK, VK>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[4] = {0, 0, 0, 0};
    uint64_t cm[4] = {0, 0, 0, 0}, cv[4] = {0, 0, 0, 0};
    uint32_t hid[8];
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 100 * 1024, 0, mode, prio_m, prio_v, n_m, n_v, cyc, ids, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            best = std::min(best, t);
        }
        ms[mode] = best;
        uint64_t c[8];
        hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        hipMemcpy(hid, ids, 32, hipMemcpyDeviceToHost);
        cm[mode] = c[0];
        cv[mode] = c[4];
    }
    printf("B two-waves  %-14s %-14s prio m/v %d/%d | ms: mfma %.3f valu %.3f both %.3f (sum %.3f max %.3f) | cycles per MFMA alone %.1f both %.1f | per filler alone %.2f both %.2f | simd of waves:",
           MK == 1 ? "f16 32x32x16" : "f32 32x32x2", kind_name[VK], prio_m, prio_v, ms[1], ms[2], ms[3], ms[1] + ms[2], std::max(ms[1], ms[2]),
           (double)cm[1] / (4.0 * n_m), (double)cm[3] / (4.0 * n_m), (double)cv[2] / (32.0 * n_v), (double)cv[3] / (32.0 * n_v));
    for (int w = 0; w < 8; ++w) printf(" %u", (hid[w] >> 4) & 3u);
    printf("\n");
}

// ---- C: phases, W waves per SIMD all running {NA fillers ; NB MFMAs [+ NI fillers each]} ---------------------------------
template <int VK, int NA, int NB, int NI, int WAVES /*per workgroup*/, bool SMALL = false>
__global__ __launch_bounds__(WAVES * 64) void phases(int mode, int n, uint64_t* cyc, float* out) {
    extern __shared__ float pad[];
    State s;
    s.init();
    __syncthreads();
    const uint64_t t0 = memtime();
    for (int i = 0; i < n; ++i) {
        if (mode & 1) {
#pragma unroll
            for (int f = 0; f < NA; ++f) filler<VK>(s.x[f & 7], s.xp[f & 7], s.c1, s.c2, s.cp);
        }
        if (mode & 2) {
#pragma unroll
            for (int m = 0; m < NB; ++m) {
                if (SMALL) mfma_small(s.acc4[m & 7], s.a, s.b);
                else mfma<1>(s.acc[m & 3], s.a, s.b, s.fa, s.fb);
#pragma unroll
                for (int f = 0; f < NI; ++f) filler<VK>(s.x[(m * NI + f) & 7], s.xp[(m * NI + f) & 7], s.c1, s.c2, s.cp);
            }
        }
    }
    const uint64_t t1 = memtime();
    const float r = s.fold();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (r == 123.456f) out[0] = r + pad[0];
}

template <int VK, int NA, int NB, int NI, int WAVES, bool SMALL = false>
static void run_phases(int n, uint64_t* cyc, float* out) {
    auto k = phases<VK, NA, NB, NI, WAVES, SMALL>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms[4];
    for (int mode = 1; mode <= 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(WAVES * 64), 100 * 1024, 0, mode, n, cyc, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            best = std::min(best, t);
        }
        ms[mode] = best;
    }
    printf("C phases     %-14s %d waves/SIMD, per iteration %d fillers then %d MFMA(%s) with %d fillers each | ms: A %.3f  B %.3f  both %.3f (sum %.3f max %.3f)\n",
           kind_name[VK], WAVES / 4, NA, NB, SMALL ? "16x16x32 f16" : "32x32x16 f16", NI, ms[1], ms[2], ms[3], ms[1] + ms[2], std::max(ms[1], ms[2]));
}

int main(int argc, char** argv) {
    uint64_t* cyc;
    uint32_t* ids;
    float* out;
    hipMalloc(&cyc, 64);
    hipMalloc(&ids, 32);
    hipMalloc(&out, 4);
    const int n = 20000;
    if (argc > 1 && argv[1][0] == 'p') {
        // r03: packed-fp32 THROUGHPUT with 3 waves per SIMD (r02 measured the packed ops with one wave per SIMD, where every kind issues
        // once per ~4.75 cycles, and beside MFMA waves).  K2 issues 6 MFMAs per ~280 VALU: is a packed blend worth it there?
        run_phases<F_FMA, 512, 1, 0, 12>(1000, cyc, out);
        run_phases<F_PKFMA, 512, 1, 0, 12>(1000, cyc, out);
        run_phases<F_PKADD, 512, 1, 0, 12>(1000, cyc, out);
        run_phases<F_FMA, 272, 6, 0, 12>(2000, cyc, out);      // K2-shaped step, plain: 272 VALU + 6 MFMA
        run_phases<F_PKFMA, 272, 6, 0, 12>(2000, cyc, out);    // ... if every VALU were packed (the exclusion with the matrix pipe at K2's MFMA share)
        run_phases<F_PKFMA, 232, 6, 0, 12>(2000, cyc, out);    // ... 40 instructions fewer (the blend of 5 levels packed: 80 -> 40)
        run_phases<F_FMA, 232, 6, 0, 12>(2000, cyc, out);
        run_phases<F_FMA, 512, 120, 0, 12>(1000, cyc, out);    // K1-shaped, for the same box
        run_phases<F_PKFMA, 512, 120, 0, 12>(1000, cyc, out);
        return 0;
    }
    if (argc > 1 && argv[1][0] == 'm') {
        // r04: K1's wave-step {~900 VALU of hash phase and compositing ; 120 MFMAs with the ~480 MLP VALU between them} with the MLPs' products as
        // 32x32x16 tiles (today) and as twice as many 16x16x32 tiles (the candidate of DESIGN.md 9 item 2a): the same MACs, half the accumulator traffic
        for (int rep = 0; rep < 2; ++rep) {
            run_phases<F_FMA, 900, 120, 4, 12>(1000, cyc, out);
            run_phases<F_FMA, 900, 240, 2, 12, true>(1000, cyc, out);
            run_phases<F_FMA, 964, 216, 2, 12, true>(1000, cyc, out);   // ... density layer 2 without its zero rows (-24), 32 more half-rate swaps as 64 fillers
            // the same 1380 VALU + 120 MFMAs with the hash-phase VALU spread BETWEEN the MFMAs (what a software-pipelined K1 would issue: sample i + 1's
            // hash arithmetic inside sample i's MLP), at 3 and at 2 waves per SIMD (the pipelined kernel needs ~80 more registers: 2 waves)
            run_phases<F_FMA, 60, 120, 11, 12>(1000, cyc, out);
            run_phases<F_FMA, 900, 120, 4, 8>(1000, cyc, out);
            run_phases<F_FMA, 60, 120, 11, 8>(1000, cyc, out);
            run_phases<F_FMA, 0, 120, 0, 12>(1000, cyc, out);          // the MFMAs alone
            run_phases<F_FMA, 0, 240, 0, 12, true>(1000, cyc, out);
        }
        return 0;
    }
    // A: fillers alone (issue cost of each kind), then beside the f16 MFMA and the f32 MFMA
    sweep_same<0, F_FMA>(n, cyc, out);
    sweep_same<1, F_FMA>(n, cyc, out);
    sweep_same<2, F_FMA>(n, cyc, out);
    sweep_same<0, F_XAD>(n, cyc, out);
    sweep_same<1, F_XAD>(n, cyc, out);
    sweep_same<0, F_PKFMA>(n, cyc, out);
    sweep_same<1, F_PKFMA>(n, cyc, out);
    sweep_same<1, F_PKADD>(n, cyc, out);
    sweep_same<0, F_CVTPK>(n, cyc, out);
    sweep_same<1, F_CVTPK>(n, cyc, out);
    sweep_same<1, F_MUL24>(n, cyc, out);
    sweep_same<0, F_EXP>(n, cyc, out);
    sweep_same<1, F_EXP>(n, cyc, out);
    sweep_same<1, F_XOR>(n, cyc, out);
    sweep_same<1, F_FMAMIX>(n, cyc, out);
    sweep_same<1, F_MAXI>(n, cyc, out);
    sweep_same<1, F_FRACT>(n, cyc, out);
    sweep_same<1, F_MOV>(n, cyc, out);
    sweep_same<0, F_MIXLO>(n, cyc, out);
    sweep_same<1, F_MIXLO>(n, cyc, out);
    sweep_same<0, F_MIXHI>(n, cyc, out);
    sweep_same<1, F_MIXHI>(n, cyc, out);
    sweep_same<0, F_PKMAXH>(n, cyc, out);
    sweep_same<1, F_PKMAXH>(n, cyc, out);
    sweep_same<1, F_MIXCLAMP>(n, cyc, out);
    sweep_same<0, F_CVTI>(n, cyc, out);
    sweep_same<0, F_BITOP3>(n, cyc, out);
    sweep_same<0, F_FMAC>(n, cyc, out);
    sweep_same<0, F_SUB>(n, cyc, out);
    sweep_same<0, F_ADD3>(n, cyc, out);
    sweep_same<0, F_MAD24>(n, cyc, out);
    sweep_same<0, F_LSHL>(n, cyc, out);
    sweep_same<0, F_CNDMASK>(n, cyc, out);
    sweep_same<0, F_ADDF64>(n, cyc, out);
    sweep_same<0, F_CVTF64>(n, cyc, out);
    sweep_same<0, F_SWAP>(n, cyc, out);
    sweep_same<0, F_FLOOR>(n, cyc, out);
    sweep_same<0, F_RCP>(n, cyc, out);
    sweep_same<0, F_CVTU>(n, cyc, out);
    sweep_same<0, F_MOV>(n, cyc, out);
    sweep_same<0, F_FRACT>(n, cyc, out);
    sweep_same<0, F_MUL24>(n, cyc, out);
    sweep_same<0, F_MAXI>(n, cyc, out);
    sweep_same<1, F_CVTI>(n, cyc, out);
    sweep_same<1, F_BITOP3>(n, cyc, out);
    sweep_same<1, F_SWAP>(n, cyc, out);
    // B: different waves of one SIMD
    run_two<1, F_FMA>(40000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_FMA>(40000, 10000, 1, 0, cyc, ids, out);
    run_two<1, F_FMA>(40000, 10000, 0, 1, cyc, ids, out);
    run_two<1, F_FMA>(40000, 10000, 3, 0, cyc, ids, out);
    run_two<2, F_FMA>(20000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_XAD>(40000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_PKFMA>(40000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_CVTPK>(40000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_MIXLO>(40000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_PKMAXH>(40000, 10000, 0, 0, cyc, ids, out);
    run_two<1, F_FMAMIX>(40000, 10000, 0, 0, cyc, ids, out);
    // C: K1-shaped phases: ~510 VALU then 120 MFMA with 7 VALU each
    run_phases<F_FMA, 512, 120, 0, 12>(1000, cyc, out);
    run_phases<F_FMA, 512, 120, 7, 12>(1000, cyc, out);
    run_phases<F_FMA, 512, 120, 4, 12>(1000, cyc, out);
    run_phases<F_FMA, 1360, 120, 0, 12>(1000, cyc, out);
    run_phases<F_FMA, 512, 120, 7, 8>(1000, cyc, out);
    run_phases<F_FMA, 512, 120, 7, 4>(1000, cyc, out);
    run_phases<F_XAD, 512, 120, 7, 12>(1000, cyc, out);
    // throughput of single opcodes with 3 waves per SIMD (column A: 512 fillers per iteration): which opcodes are slower than full rate
    // when other waves could fill their issue gaps?
    run_phases<F_FMA, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CNDMASK, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_SWAP, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_MIXLO, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_EXP, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_MOV, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CVTF64, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_FMAMIX, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CVTPK, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_PKMAXH, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_FMAC, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_BITOP3, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CNDMASK64, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_BFI, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CMP, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CNDMASKC, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_MAXI, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_CVTI, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_ADDF64, 512, 1, 0, 12>(1000, cyc, out);
    run_phases<F_RCP, 512, 1, 0, 12>(1000, cyc, out);
    return 0;
}
