// sn_main.h -- the fused main-field render kernel (rows a13-a17 of SURVEY.md §8(a)).
//
// Mapping (DESIGN.md "Kernel K1"):
//   lane  = one ray; wave = an 8x8 pixel tile (64 rays); workgroup = 2x2 tiles (256 threads).
//   Every wave marches its 64 rays front to back, ONE sample per lane per step.  At each step
//     1. each lane hash-encodes its own sample (16 levels x 8 gathers of 8 B; neighbouring
//        pixels at equal depth share voxels, so the wave's 64 addresses per gather collapse to a
//        few cache lines at fine levels and to 1-8 at coarse levels),
//     2. the wave evaluates the density MLP (32->64->16) and the colour MLP (16 SH + 15 geo
//        [+ folded appearance bias] ->64->64->3) on the matrix cores.  Weights are the MFMA A
//        operand (read from an LDS image), activations the B operand (column = sample = lane&31),
//        so the C/D layout of one layer IS the B layout of the next: layers chain in registers
//        with no LDS round trip and no cross-lane traffic except one permlane32_swap per input
//        register of the first layer,
//     3. each lane composites its own ray (transmittance, rgb, accumulation, median / expected
//        depth) -- no scan is needed because depth order is the loop order.
//
// MFMA layout facts used (cdna_hip_programming.md §3):
//   v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]
//   A "k-slot" (t, h) of a layer = k-step t, lane half h.  rho(r) = (r&3)+8*(r>>2).
#pragma once
#include "sn_device.h"

// LDS weight image of the main field, float offsets.  Built on the host by sn_api.hip
// (build_main_image) -- keep the two in sync.
struct SnMainImg {
    static constexpr int W1 = 0;        // [rt=2][t4=4][lane=64][4]   32 -> 64
    static constexpr int W2 = 2048;     // [rt=1][t4=8][64][4]        64 -> 32 rows (16 real + dup)
    static constexpr int WC1 = 4096;    // [rt=2][t4=4][64][4]        (16 L2-rows + 16 SH) -> 64
    static constexpr int WC2 = 6144;    // [rt=2][t4=8][64][4]        64 -> 64
    static constexpr int B1 = 10240;    // [rt=2][h=2][16]
    static constexpr int B2 = 10304;    // [1][2][16]
    static constexpr int BC1 = 10336;   // [2][2][16]
    static constexpr int BC2 = 10400;   // [2][2][16]
    static constexpr int W3 = 10464;    // [n][h=2][32], n = 3 channels
    static constexpr int W3_ROWS = 3;
    static constexpr int B3 = W3 + W3_ROWS * 64;  // [4]: the 3 biases; [3] = 1 / (output scale of layer 2) of the split-precision image (h0 = row 0 * that)
    static constexpr int TOTAL = B3 + 4;  // 10 660 floats = 42 640 bytes; a multiple of 4
};

// acc[rt] (tile 0) / acc[rt] (tile 1) <- bias + W . op     (exact fp32 MFMA)
template <int RT, int KS>
SN_DEV void sn_mlp_layer_f32(const float* __restrict__ wimg, const float* __restrict__ bimg, const float* op0,
                             const float* op1, f32x16* acc0, f32x16* acc1, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4* b = (const f32x4*)(bimg + (rt * 2 + h) * 16);
        f32x4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        f32x16 v = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        acc0[rt] = v;
        acc1[rt] = v;
    }
#pragma unroll
    for (int t4 = 0; t4 < KS / 4; ++t4) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            f32x4 a = *(const f32x4*)(wimg + ((rt * (KS / 4) + t4) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], op0[4 * t4 + e], acc0[rt], 0, 0, 0);
                acc1[rt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], op1[4 * t4 + e], acc1[rt], 0, 0, 0);
            }
        }
    }
}

// Per-ray direction operands: 8 k-steps x 2 tiles.
struct SnShOps {
    float t0[8], t1[8];
    SN_DEV void build(const float d[3], int remap) {
        float c[16];
        sn_direction_encoding(d, remap, c);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float a = c[2 * u], b = c[2 * u + 1];
            sn_swap_halves(a, b);
            t0[u] = a;
            t1[u] = b;
        }
    }
};

// Full main-field evaluation of the wave's 64 samples.  feat[32]: this lane's own hash features.
// Returns this lane's own pre-activation density h0 and post-sigmoid rgb.
// GEO (stage kernel only): geo16[k] = layer-2 output k of the lane's own sample (k = 0: h0, k = 1..15: the geometry features nerfstudio
// calls `base_mlp_out` / `density_embedding`), gathered from the accumulator layout with 8 half-wave swaps.
template <bool GEO = false>
SN_DEV void sn_main_field_f32(const float* __restrict__ lds, float* feat, const SnShOps& sh, int lane, float& h0, float rgb[3],
                              float* geo16 = nullptr) {
    const bool upper = lane >= 32;
    // ---- layer 1: 32 -> 64, ReLU ---------------------------------------------------------
    float op0[32], op1[32];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float a = feat[2 * t], b = feat[2 * t + 1];
        sn_swap_halves(a, b);
        op0[t] = a;
        op1[t] = b;
    }
    f32x16 a0[2], a1[2];
    sn_mlp_layer_f32<2, 16>(lds + SnMainImg::W1, lds + SnMainImg::B1, op0, op1, a0, a1, lane);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            op0[rt * 16 + r] = sn_relu(a0[rt][r]);
            op1[rt * 16 + r] = sn_relu(a1[rt][r]);
        }
    // (scheduler fences between layers: otherwise every LDS weight read of the whole MLP is hoisted to the top and spills)
    __builtin_amdgcn_sched_barrier(0);
    // ---- layer 2: 64 -> 16 (rows 0..15 real; row 20 duplicates row 0 for the upper half) ----
    f32x16 g0[1], g1[1];
    sn_mlp_layer_f32<1, 32>(lds + SnMainImg::W2, lds + SnMainImg::B2, op0, op1, g0, g1, lane);
    h0 = upper ? g1[0][8] : g0[0][0];
    if (GEO) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float a = g0[0][r], b = g1[0][r];
            sn_swap_halves(a, b);  // every lane: a = row rho(r), b = row rho(r) + 4 of its OWN sample
            geo16[(r & 3) + 8 * (r >> 2)] = a;
            geo16[(r & 3) + 8 * (r >> 2) + 4] = b;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 1: (L2 rows 0..15 | SH16) -> 64, ReLU --------------------------------
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        op0[t] = g0[0][t];
        op1[t] = g1[0][t];
        op0[8 + t] = sh.t0[t];
        op1[8 + t] = sh.t1[t];
    }
    sn_mlp_layer_f32<2, 16>(lds + SnMainImg::WC1, lds + SnMainImg::BC1, op0, op1, a0, a1, lane);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            op0[rt * 16 + r] = sn_relu(a0[rt][r]);
            op1[rt * 16 + r] = sn_relu(a1[rt][r]);
        }
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 2: 64 -> 64, ReLU ----------------------------------------------------
    sn_mlp_layer_f32<2, 32>(lds + SnMainImg::WC2, lds + SnMainImg::BC2, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 3: 64 -> 3 on the VALU (a 32-row MFMA tile would be 90 % padding) -----
    const int h = lane >> 5;
    float p0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        const f32x4* w = (const f32x4*)(lds + SnMainImg::W3 + (n * 2 + h) * 32);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                f32x4 wv = w[rt * 4 + r4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x0 = sn_relu(a0[rt][r4 * 4 + e]);
                    float x1 = sn_relu(a1[rt][r4 * 4 + e]);
                    p0[n] = fmaf(wv[e], x0, p0[n]);
                    p1[n] = fmaf(wv[e], x1, p1[n]);
                }
            }
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        float a = p0[n], b = p1[n];
        sn_swap_halves(a, b);  // lower: own tile-0 partial + upper's tile-0 partial; upper: tile 1
        float x = a + b + lds[SnMainImg::B3 + n];
        rgb[n] = __builtin_amdgcn_rcpf(1.0f + sn_exp<true>(-x));  // v_exp_f32 / v_rcp_f32 (~1 ulp each) instead of libm exp and an IEEE divide
    }
}

// ==========================================================================================
// Split-precision variant ("fp16x2"): every fp32 operand x is split into fp16 hi + lo (x = hi + lo to ~2^-22), and
// each product group is three v_mfma_f32_32x32x16_f16 (hi.hi + hi.lo + lo.hi; the lo.lo term, ~2^-22 relative, is
// dropped) with fp32 accumulation.  3 x 1/16-cost MFMAs replace one fp32 MFMA: 120 MFMAs x 32 cycles per wave-step
// instead of 320 x 64.  Layout facts: A[i=l&31][k=8(l>>5)+e], B[k=8(l>>5)+e][j=l&31], e = 0..7 (one f16x8 = 4 VGPRs);
// C/D as for fp32.  A k-slot is (s = 16-wide k-step, h = lane half, e).
// Range: |x| beyond 65504 saturates (cvt_pkrtz) -- trained nerfacto activations are O(1..100).
// ==========================================================================================
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// byte offsets of the fp16x2 LDS image; weights [rt][s][hi|lo][lane][8 halves]
struct SnMainImgH {
    static constexpr int W1 = 0;          // 2 rt x 2 s x 2 x 1 KiB
    static constexpr int W2 = 8192;       // 1 x 4 x 2 KiB
    static constexpr int WC1 = 16384;     // 2 x 2 x 2 KiB
    static constexpr int WC2 = 24576;     // 2 x 4 x 2 KiB
    static constexpr int FP32 = 40960;    // then the fp32 tail, same sub-layout as SnMainImg from B1 on
    static constexpr int TAIL_FLOATS = SnMainImg::TOTAL - SnMainImg::B1;
    static constexpr int TOTAL_BYTES = FP32 + TAIL_FLOATS * 4;  // 42640
    static constexpr int B1 = 0, B2 = SnMainImg::B2 - SnMainImg::B1, BC1 = SnMainImg::BC1 - SnMainImg::B1,
                         BC2 = SnMainImg::BC2 - SnMainImg::B1, W3 = SnMainImg::W3 - SnMainImg::B1, B3 = SnMainImg::B3 - SnMainImg::B1;
};

// two fp32 -> packed fp16 hi pair and lo pair: hi = RTZ(a), lo = RTZ(a - hi).  The difference is formed by v_fma_mix_f32, which
// reads the fp16 half straight out of the packed register (fma(hi16, -1.0, a) in fp32, exact) -- 4 instructions per pair
// instead of mask, mask, convert, packed subtract, convert.  hipcc does not select fma_mix for this pattern, hence the asm
// (a plain VALU instruction: no hazard class of its own; checked on hardware by tools/probes/mix_probe).
// Tried r02: forming the low pair with v_fma_mixlo_f16 / v_fma_mixhi_f16 (fp16 results written straight into the halves of the packed
// register, 3 instructions per pair, 1444 -> 1356 VALU per wave-step) is SLOWER, 2.85 -> 2.98 ms: those two opcodes issue at half rate on
// gfx950 (8.5 cycles per wave64 against 4.7 for v_fma_mix_f32 / v_cvt_pkrtz / v_pk_max_f16; tools/probes/overlap2_probe.hip).
SN_DEV void sn_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
#ifdef SN_PROBE_NOSPLIT  // tools/probes/mlp_probe.hip: the MFMA part of the MLP alone (no instruction, dependencies kept)
    asm volatile("" : "=v"(hi), "=v"(lo) : "v"(a), "v"(b));
    return;
#endif
    hi = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(la, lb));
}

// ReLU folded into the split (K1 only: needs the range conditioning of sn_finalize_weights, which keeps every pre-activation below 2^10):
//   hi = max(RTZ16(a), 0) as ONE packed-f16 max for the pair;   lo = RTZ16(clamp(a - RTZ16(a), 0, 1)).
// RTZ rounds toward zero, so a - RTZ16(a) has the sign of a: for a < 0 both parts become 0, for a >= 0 the residual is in [0, ulp16(a)) and
// stays below 1 because |a| < 2048 -- the clamp of v_fma_mix_f32 is then exactly max(., 0).  5 instructions per pair instead of 6
// (two v_max_i32 + the plain split): -128 VALU per wave-step.  (r01 had tried the same fold without the range guarantee and dropped it:
// activations >= 2048 lose their low part.)
#ifndef SN_RELU_FOLD
#define SN_RELU_FOLD 1
#endif
SN_DEV void sn_split2_relu(float a, float b, uint32_t& hi, uint32_t& lo) {
    const uint32_t hr = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(a, b));
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0] clamp" : "=v"(la) : "v"(hr), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(lb) : "v"(hr), "v"(b));
    lo = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(la, lb));
    typedef _Float16 sn_h2 __attribute__((ext_vector_type(2)));
    const sn_h2 zero = {(_Float16)0.0f, (_Float16)0.0f};
    hi = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(sn_h2, hr), zero));  // v_pk_max_f16
}

struct SnOpH {  // one B operand (8 k-slots of one 32-sample tile), hi and lo parts
    u32x4 hi, lo;
    SN_DEV void set_relu(const float v[8]) {  // operands = relu(v), see sn_split2_relu
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t h, l;
            sn_split2_relu(v[2 * e], v[2 * e + 1], h, l);
            hi[e] = h;
            lo[e] = l;
        }
    }
    SN_DEV void set(const float v[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            uint32_t h, l;
            sn_split2(v[2 * e], v[2 * e + 1], h, l);
            hi[e] = h;
            lo[e] = l;
        }
    }
};

// Wave priority around the MFMA cluster of every k-step (cdna_hip_programming.md T5): with three waves per SIMD in different phases the
// arbiter otherwise serves the oldest wave's VALU first and the matrix pipe idles between a wave's clusters.  Measured r02, interleaved
// A/B on one box (tools/ab_lib.sh): 3.084 -> 2.980 ms (-3.4 %), the same for priority 1, 2 or 3; raising the priority of the whole MLP
// phase or of the hash phase instead: no effect; 2 waves/SIMD: -2.7 % alone, worse together with the priority.  0 disables.
#ifndef SN_MFMA_PRIO
#define SN_MFMA_PRIO 1
#endif
#ifndef SN_MFMA_H  // (tools/probes/mlp_probe.hip overrides it to time the VALU part of the MLP alone)
#define SN_MFMA_H(ACC, A, B) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, ACC, 0, 0, 0)
#endif

template <int RT, int KS>
SN_DEV void sn_mlp_layer_h(const char* __restrict__ wimg, const float* __restrict__ bimg, const SnOpH* op0, const SnOpH* op1,
                           f32x16* acc0, f32x16* acc1, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const f32x4* b = (const f32x4*)(bimg + (rt * 2 + h) * 16);
        f32x4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
        f32x16 v = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        acc0[rt] = v;
        acc1[rt] = v;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 ah[RT], al[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const char* base = wimg + (((rt * KS + s) * 2) * 64 + lane) * 16;
            ah[rt] = __builtin_bit_cast(f16x8, *(const u32x4*)base);
            al[rt] = __builtin_bit_cast(f16x8, *(const u32x4*)(base + 1024));
        }
        const f16x8 bh0 = __builtin_bit_cast(f16x8, op0[s].hi), bl0 = __builtin_bit_cast(f16x8, op0[s].lo);
        const f16x8 bh1 = __builtin_bit_cast(f16x8, op1[s].hi), bl1 = __builtin_bit_cast(f16x8, op1[s].lo);
        // small terms first, then hi.hi; independent accumulators interleaved
#if SN_MFMA_PRIO
        __builtin_amdgcn_s_setprio(SN_MFMA_PRIO);
#endif
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            SN_MFMA_H(acc0[rt], al[rt], bh0);
            SN_MFMA_H(acc1[rt], al[rt], bh1);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            SN_MFMA_H(acc0[rt], ah[rt], bl0);
            SN_MFMA_H(acc1[rt], ah[rt], bl1);
        }
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            SN_MFMA_H(acc0[rt], ah[rt], bh0);
            SN_MFMA_H(acc1[rt], ah[rt], bh1);
        }
#if SN_MFMA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
}

// per-ray direction operands, fp16x2 form: slot (h, e) <-> SH component 8h + e
struct SnShOpsH {
    SnOpH t0, t1;
    SN_DEV void build(const float d[3], int remap) {
        float c[16];
        sn_direction_encoding(d, remap, c);
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = c[e], b = c[8 + e];
            sn_swap_halves(a, b);
            v0[e] = a;
            v1[e] = b;
        }
        t0.set(v0);
        t1.set(v1);
    }
};

// ReLU + split of one 32-row accumulator tile into its two B operands (k-steps 2rt, 2rt+1).
// FOLD: the conditioned kernels fold the ReLU into the split (sn_split2_relu); the normals kernel, which splits unconditioned operands, does not.
template <bool FOLD = false>
SN_DEV void sn_acc_to_ops(const f32x16& acc, bool relu, SnOpH& s0, SnOpH& s1) {
    float v[16];
    if (FOLD && relu) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = acc[r];
        s0.set_relu(v);
        s1.set_relu(v + 8);
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = relu ? sn_relu(acc[r]) : acc[r];
    s0.set(v);
    s1.set(v + 8);
}

template <bool GEO = false>
SN_DEV void sn_main_field_h(const char* __restrict__ ldsb, float* feat, const SnShOpsH& sh, int lane, float& h0, float rgb[3],
                            float* geo16 = nullptr) {
    const bool upper = lane >= 32;
    const float* tail = (const float*)(ldsb + SnMainImgH::FP32);
    SnOpH op0[4], op1[4];
    // ---- layer 1: slot (s, h, e) <-> feature 16 s + 8 h + e ----
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = feat[16 * s + e], b = feat[16 * s + 8 + e];
            sn_swap_halves(a, b);
            v0[e] = a;
            v1[e] = b;
        }
        op0[s].set(v0);
        op1[s].set(v1);
    }
    f32x16 a0[2], a1[2];
    sn_mlp_layer_h<2, 2>(ldsb + SnMainImgH::W1, tail + SnMainImgH::B1, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a0[rt], true, op0[2 * rt], op0[2 * rt + 1]);
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a1[rt], true, op1[2 * rt], op1[2 * rt + 1]);
    }
    // ---- layer 2 ----
    f32x16 g0[1], g1[1];
    sn_mlp_layer_h<1, 4>(ldsb + SnMainImgH::W2, tail + SnMainImgH::B2, op0, op1, g0, g1, lane);
    // the layers run on power-of-two scaled activations (sn_api.hip plan_split_scales); the density row is scaled back here, the geo
    // rows carry their scale into colour layer 1, whose weights hold the inverse
    h0 = (upper ? g1[0][8] : g0[0][0]) * tail[SnMainImgH::B3 + 3];
    if (GEO) {  // (as in sn_main_field_f32; the rows carry the layer's power-of-two output scale)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float a = g0[0][r], b = g1[0][r];
            sn_swap_halves(a, b);
            geo16[(r & 3) + 8 * (r >> 2)] = a * tail[SnMainImgH::B3 + 3];
            geo16[(r & 3) + 8 * (r >> 2) + 4] = b * tail[SnMainImgH::B3 + 3];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- colour layer 1: k-step 0 <- layer-2 regs 0..7 (rows rho(e)+4h), k-step 1 <- SH ----
    {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v0[e] = g0[0][e];
            v1[e] = g1[0][e];
        }
        op0[0].set(v0);
        op1[0].set(v1);
        op0[1] = sh.t0;
        op1[1] = sh.t1;
    }
    sn_mlp_layer_h<2, 2>(ldsb + SnMainImgH::WC1, tail + SnMainImgH::BC1, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a0[rt], true, op0[2 * rt], op0[2 * rt + 1]);
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a1[rt], true, op1[2 * rt], op1[2 * rt + 1]);
    }
    // ---- colour layer 2 (two 32-row passes: halves the live accumulators, same MFMAs) + colour layer 3 ----
    const int h = lane >> 5;
    // plain fp32 FMAs (NOT v_pk_fma_f32: packed fp32 ops are mutually exclusive with the matrix pipe on gfx950 and would stall
    // behind the other waves' MFMAs -- tools/probes/overlap2_probe.hip); two accumulators per channel and tile for issue distance
    float p0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16 c0[1], c1[1];
        sn_mlp_layer_h<1, 4>(ldsb + SnMainImgH::WC2 + rt * 8192, tail + SnMainImgH::BC2 + rt * 32, op0, op1, c0, c1, lane);
        __builtin_amdgcn_sched_barrier(0);
        float r0[16], r1[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            r0[j] = sn_relu(c0[0][j]);
            r1[j] = sn_relu(c1[0][j]);
        }
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const f32x4* w = (const f32x4*)(tail + SnMainImgH::W3 + (n * 2 + h) * 32);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 wv = w[rt * 4 + r4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0[n] = fmaf(wv[e], r0[4 * r4 + e], p0[n]);
                    p1[n] = fmaf(wv[e], r1[4 * r4 + e], p1[n]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        float a = p0[n], b = p1[n];
        sn_swap_halves(a, b);
        float x = a + b + tail[SnMainImgH::B3 + n];
        rgb[n] = __builtin_amdgcn_rcpf(1.0f + sn_exp<true>(-x));  // v_exp_f32 / v_rcp_f32 (~1 ulp each) instead of libm exp and an IEEE divide
    }
}

// ==========================================================================================
// Single-fp16 form (PREC 2, r04; opt-in `precision="fp16"` for tiny-cuda-nn checkpoints ONLY -- never the default, never the headline):
// the arithmetic the library those checkpoints were trained with runs in (FullyFusedMLP: fp16 weights, fp16 activations between the
// layers), as far as this hardware's MFMA allows -- fp16 operands, ONE v_mfma_f32_32x32x16_f16 per product group (no hi / lo split),
// fp32 accumulation inside a layer, every layer's output rounded to fp16 (round-to-nearest-even, v_cvt_pk_f16_f32).  Weights are the hi
// planes of the split-precision image (= RNE16 of the range-conditioned weights; a power-of-two scale commutes with the rounding).
// Colour layer 3 (64 -> 3) runs on the matrix cores too (the library pads its output layer the same way): its fp16 A operand sits
// behind the image (SnMainImgF16).  48 MFMAs and ~220 conversion VALU per wave-step instead of 120 and ~670.
// oracle/tcnn_layout.py emulates exactly these roundings; what is NOT reproduced is the library's fp16 ACCUMULATION inside a layer.
// ==========================================================================================
struct SnMainImgF16 {
    static constexpr int W3H = SnMainImg::TOTAL * 4;   // byte offset: A operand of colour layer 3, [s = 4][lane = 64][8 halves], rows 0..2 real
    static constexpr int TAILF = W3H + 4096;           // float[4]: [0] = 1 / s5 (the power-of-two scale of that operand)
    static constexpr int TOTAL_BYTES = TAILF + 16;
    static constexpr int TOTAL_FLOATS = TOTAL_BYTES / 4;
};

// RNE (v_cvt_pkrtz_f16_f32 truncates): hipcc selects v_cvt_pk_f16_f32 for the vector conversion.  NOT inline asm: its operands are MFMA
// results, and the compiler inserts the wait states an MFMA write -> VALU read needs only in front of instructions it knows (found on
// hardware, r04: with an asm conversion the geometry rows came back as garbage that changed with unrelated code further down).
SN_DEV uint32_t sn_pk_f16(float a, float b) {
    typedef float sn_f2 __attribute__((ext_vector_type(2)));
    typedef _Float16 sn_h2v __attribute__((ext_vector_type(2)));
    const sn_f2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sn_h2v));
}
SN_DEV uint32_t sn_pk_f16_relu(float a, float b) {
    typedef _Float16 sn_h2 __attribute__((ext_vector_type(2)));
    const sn_h2 zero = {(_Float16)0.0f, (_Float16)0.0f};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(sn_h2, sn_pk_f16(a, b)), zero));  // v_pk_max_f16
}
SN_DEV float sn_round_f16(float x) { return (float)(_Float16)x; }

struct SnOpF {  // one B operand: 8 k-slots of one 32-sample tile, fp16
    u32x4 v;
    template <bool RELU>
    SN_DEV void set(const float x[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = RELU ? sn_pk_f16_relu(x[2 * e], x[2 * e + 1]) : sn_pk_f16(x[2 * e], x[2 * e + 1]);
    }
};

template <int RT, int KS>
SN_DEV void sn_mlp_layer_f16(const char* __restrict__ wimg, const float* __restrict__ bimg, const SnOpF* op0, const SnOpF* op1, f32x16* acc0,
                             f32x16* acc1, int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        f32x16 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bimg) {
            const f32x4* b = (const f32x4*)(bimg + (rt * 2 + h) * 16);
            f32x4 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3];
            v = f32x16{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w, b2.x, b2.y, b2.z, b2.w, b3.x, b3.y, b3.z, b3.w};
        }
        acc0[rt] = v;
        acc1[rt] = v;
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        f16x8 a[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a[rt] = __builtin_bit_cast(f16x8, *(const u32x4*)(wimg + (((rt * KS + s) * 2) * 64 + lane) * 16));  // the hi plane
        const f16x8 b0 = __builtin_bit_cast(f16x8, op0[s].v), b1 = __builtin_bit_cast(f16x8, op1[s].v);
#if SN_MFMA_PRIO
        __builtin_amdgcn_s_setprio(SN_MFMA_PRIO);
#endif
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            SN_MFMA_H(acc0[rt], a[rt], b0);
            SN_MFMA_H(acc1[rt], a[rt], b1);
        }
#if SN_MFMA_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    }
}

struct SnShOpsF {
    SnOpF t0, t1;
    SN_DEV void build(const float d[3], int remap) {
        float c[16];
        sn_direction_encoding(d, remap, c);
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = c[e], b = c[8 + e];
            sn_swap_halves(a, b);
            v0[e] = a;
            v1[e] = b;
        }
        t0.set<false>(v0);
        t1.set<false>(v1);
    }
};

// one 32-row accumulator tile -> the two fp16 operands of the next layer (k-steps 2 rt, 2 rt + 1), ReLU applied
SN_DEV void sn_acc_to_ops_f16(const f32x16& acc, SnOpF& s0, SnOpF& s1) {
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[r];
    s0.set<true>(v);
    s1.set<true>(v + 8);
}

template <bool GEO = false>
SN_DEV void sn_main_field_f16(const char* __restrict__ ldsb, float* feat, const SnShOpsF& sh, int lane, float& h0, float rgb[3], float* geo16 = nullptr) {
    const bool upper = lane >= 32;
    const float* tail = (const float*)(ldsb + SnMainImgH::FP32);
    SnOpF op0[4], op1[4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = feat[16 * s + e], b = feat[16 * s + 8 + e];
            sn_swap_halves(a, b);
            v0[e] = a;
            v1[e] = b;
        }
        op0[s].set<false>(v0);   // the grid's features enter the network as fp16
        op1[s].set<false>(v1);
    }
    f32x16 a0[2], a1[2];
    sn_mlp_layer_f16<2, 2>(ldsb + SnMainImgH::W1, tail + SnMainImgH::B1, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sn_acc_to_ops_f16(a0[rt], op0[2 * rt], op0[2 * rt + 1]);
        sn_acc_to_ops_f16(a1[rt], op1[2 * rt], op1[2 * rt + 1]);
    }
    f32x16 g0[1], g1[1];
    sn_mlp_layer_f16<1, 4>(ldsb + SnMainImgH::W2, tail + SnMainImgH::B2, op0, op1, g0, g1, lane);
    h0 = sn_round_f16(upper ? g1[0][8] : g0[0][0]) * tail[SnMainImgH::B3 + 3];
    if (GEO) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float a = g0[0][r], b = g1[0][r];
            sn_swap_halves(a, b);
            geo16[(r & 3) + 8 * (r >> 2)] = sn_round_f16(a) * tail[SnMainImgH::B3 + 3];
            geo16[(r & 3) + 8 * (r >> 2) + 4] = sn_round_f16(b) * tail[SnMainImgH::B3 + 3];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        float v0[8], v1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v0[e] = g0[0][e];
            v1[e] = g1[0][e];
        }
        op0[0].set<false>(v0);
        op1[0].set<false>(v1);
        op0[1] = sh.t0;
        op1[1] = sh.t1;
    }
    sn_mlp_layer_f16<2, 2>(ldsb + SnMainImgH::WC1, tail + SnMainImgH::BC1, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sn_acc_to_ops_f16(a0[rt], op0[2 * rt], op0[2 * rt + 1]);
        sn_acc_to_ops_f16(a1[rt], op1[2 * rt], op1[2 * rt + 1]);
    }
    SnOpF q0[4], q1[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16 c0[1], c1[1];
        sn_mlp_layer_f16<1, 4>(ldsb + SnMainImgH::WC2 + rt * 8192, tail + SnMainImgH::BC2 + rt * 32, op0, op1, c0, c1, lane);
        __builtin_amdgcn_sched_barrier(0);
        sn_acc_to_ops_f16(c0[0], q0[2 * rt], q0[2 * rt + 1]);
        sn_acc_to_ops_f16(c1[0], q1[2 * rt], q1[2 * rt + 1]);
    }
    // colour layer 3 on the matrix cores: rows 0..2 of a 32-row tile; lane j < 32 holds them in registers 0..2 (tile 0 = its own
    // sample, tile 1 = the sample of lane j + 32, which one swap per channel hands over)
    f32x16 r0, r1;
    {
        const char* w3 = ldsb + SnMainImgF16::W3H;
        f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        r0 = z;
        r1 = z;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 a = __builtin_bit_cast(f16x8, *(const u32x4*)(w3 + (s * 64 + lane) * 16));
            SN_MFMA_H(r0, a, __builtin_bit_cast(f16x8, q0[s].v));
            SN_MFMA_H(r1, a, __builtin_bit_cast(f16x8, q1[s].v));
        }
    }
    const float inv_s5 = *(const float*)(ldsb + SnMainImgF16::TAILF);
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        float a = r0[n], b = r1[n];
        sn_swap_halves(a, b);  // lower lanes keep tile 0's row n of their own sample; upper lanes receive tile 1's from their partner
        const float x = sn_round_f16(fmaf(a, inv_s5, tail[SnMainImgH::B3 + n]));
        rgb[n] = __builtin_amdgcn_rcpf(1.0f + sn_exp<true>(-x));
    }
}

// ------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------
// levels whose gathers are in flight together (64 VGPRs of loads at 4) and the occupancy the kernels are compiled for (168 VGPRs)
#ifndef SN_FAST_HASH
#define SN_FAST_HASH true  // fused kernels use the reduced-instruction hash arithmetic (sn_hash_corners_fast)
#endif
#ifndef SN_HASH_GROUP
#define SN_HASH_GROUP 4
#endif
#ifndef SN_MAIN_WAVES_PER_SIMD
#define SN_MAIN_WAVES_PER_SIMD 3
#endif
// how many of the leading de-hashed levels are kept in bilinear-coefficient form (sn_device.h SnDenseCopy::n_bc; 32 bytes per grid
// point: levels 0-8 of the main grid are 0.51 GB, level 9 alone would add 0.82 GB) -- host and kernels read the same constants
#ifndef SN_BC_MAIN
#define SN_BC_MAIN 9
#endif
#ifndef SN_BC_PROP
#define SN_BC_PROP 5
#endif
// torch grid: the hashed levels of the main field (those without a de-hashed copy) from x-paired tables -- one 16-byte gather per
// corner pair, 4 per level instead of 8 (sn_device.h "x-paired hash tables"); 0.47 GB for nerfacto's levels 11-15.  Re-measured r02 with the
// gather path as a co-bottleneck (same-box A/B): uniform sampler (the 800x800x64 bench) 2.88 vs 2.87-2.90 ms -- the 20 gathers saved are
// paid back by a lower sustained clock (1.69 vs 1.75 GHz: wider fetches, more fabric traffic), as r01 had found; behind the proposal
// sampler (samples concentrated near surfaces) the 1080p frame drops 15.82 -> 15.39 ms.  So: bins mode only, built only for models with
// proposal nets.
#ifndef SN_MAIN_PAIRS
#define SN_MAIN_PAIRS 1
#endif
// single-fp16 mode: the hashed levels of the grid's fp16 storage as x-pairs (four 8-byte gathers per level) instead of 4-byte rows (eight)
#ifndef SN_H16_GROUP
#define SN_H16_GROUP SN_HASH_GROUP   // levels per scheduler fence in the single-fp16 mode; measured r04 (tools/ab_libs.sh, tcnn fp16): 2 / 4 / 8 / 16 = 1.99 / 1.85 / 1.95 / 2.03 ms
#endif
#ifndef SN_H16_PAIRS
#define SN_H16_PAIRS 1
#endif
// r06: position arithmetic of the production instantiations.  1 = sn_sample_q_exact (the oracle's q bit for bit), 0 = sn_sample_q_fast
// (FMA positions + v_rcp_f32 contraction, <= 2 ulp).  UNIFORM: behind the initial sampler alone (MODE 0: a position depends on no computed
// weight, so exact positions make every voxel and blend offset the oracle's); BINS: behind the proposal sampler (MODE 1: the bins
// themselves already differ from the oracle's in the last bits).  Same-box A/B: profiles/r06_exact_positions_ab.txt.
#ifndef SN_EXACT_POS_UNIFORM
#define SN_EXACT_POS_UNIFORM 1
#endif
#ifndef SN_EXACT_POS_BINS
#define SN_EXACT_POS_BINS 0
#endif
#ifndef SN_STRIP_W
#define SN_STRIP_W 8  // r02 same-box A/B over widths 0 / 4 / 8 / 12 / 16: 2.826 / 2.803 / 2.805 / 2.817 / 2.823 ms (camera 0)
#endif

// Launch shape of a K1 instantiation: 4-wave workgroups of 2x2 tiles, SN_MAIN_WAVES_PER_SIMD of them per CU.
template <int PREC>
struct SnK1Shape {
    static constexpr int WG_WAVES = 4;
    static constexpr int TX = 2, TY = 2;             // tiles per workgroup
    static constexpr int THREADS = WG_WAVES * 64;
    static constexpr int WAVES_PER_SIMD = SN_MAIN_WAVES_PER_SIMD;
    static constexpr int HASH_GROUP = PREC == 2 ? SN_H16_GROUP : SN_HASH_GROUP;
};

struct SnMainParams {
    const float* origins;     // [H*W,3]
    const float* directions;  // [H*W,3]
    const float* nears;       // [H*W] or null
    const float* fars;        // [H*W] or null
    const float* sbins;       // [S+1] spacing bins (uniform mode), device
    const float* ebins;       // bins mode: euclidean bins, [tile][S+1][64]
    const float* table;       // main hash table [16 << log2_t, 2]
    const float* wimg;        // SnMainImg (global copy)
    float* rgb;
    float* depth;
    float* acc;
    float* exp_raw;           // un-clipped expected depth (clipped by sn_clip_expected_kernel) or null
    uint32_t* chunk_minmax;   // [2][n_chunks] ordered-uint mins then maxs of sample mid-points, or null
    int n_chunks;
    float scal[16];
    int height, width, n_samples;
    int tile_w_log2, tile_h_log2;  // tile = 2^a x 2^b pixels, a + b = 6
    int tiles_x, tiles_y;          // tiles per row / column
    int log2_t;
    float near_plane, far_plane, avg_density;
    int sh_remap;
    int chunk_rays;
    float feat_scale;   // torch grid: power-of-two scale of the hash features (carried by the de-hashed copies; applied here to the other levels)
    SnDenseCopy hquads;  // single-fp16 mode (PREC 2, tiny-cuda-nn grid, ND > 0): the fp16 storage of the grid (sn_device.h "fp16 STORAGE") ...
    const float* hrows;  // ... its hashed levels [ND, 16): fp16 x-pairs (SN_H16_PAIRS, entry numbers in hpinfo) or 4-byte rows
    uint32_t hrows_bytes;
    SnPairInfo hpinfo;
    const float* pairs;  // SN_MAIN_PAIRS: x-paired tables of the levels >= ND (pre-scaled by feat_scale), or null
    uint32_t pairs_bytes;
    SnPairInfo pinfo;
    SnGridLevels grid;  // ND > 0: R of the de-hashed coarse copies; GRID 1, ND = -1: dense-level resolutions of the tiny-cuda-nn grid
    SnDenseCopy dense;  // ND > 0
    // Split-depth tail (r03): workgroups [0, seg_first_block) march whole rays; every later block is a SEGMENT job -- workgroup
    // seg_first_block + j / n_seg of the image, samples [k seg_len, (k + 1) seg_len) with k = j % n_seg -- that stores (density, r, g, b)
    // per sample to seg_scratch [tail workgroup][wave][sample][lane] instead of compositing; sn_main_combine_kernel composites them in
    // sample order with the same arithmetic (bit-identical outputs).  n_seg <= 1: no segment jobs.
    int seg_first_block, n_seg, seg_len;
    f32x4* seg_scratch;
    int early_term;   // exact early termination of saturated waves (below); 0 = off (SN_EARLY_TERM=0: the A/B and bit-identity switch)
    unsigned long long* march_stats;  // SnRenderOpts.march_stats ([0]: wave-steps this kernel's early termination skipped) or null; STATS instantiations only
    int bg_mode;      // RGBRenderer background: 0 = the ray's last sample, 1 = the constant colour bg
    float bg[3];
    int spacing_uniform;  // SnRenderOpts.spacing_mode: the initial sampler's s(x) is the identity (sn_spacing)
    SnPosMap pm;          // SnFieldDesc.disable_scene_contraction (sn_sample_q_fast)
    // test instrumentation (DUMP = 1 instantiations only; sn_render_rays_debug)
    uint32_t* dump_fetch;  // [H*W][S][16][8] fetch records (sn_hash_encode) or null
    float* dump_q;         // [H*W][S][3] normalised positions that were hashed, or null
    int32_t* dump_median;  // [H*W] median index, or null
};

// XCD-aware, bijective block remap: the dispatcher places block b on XCD b % 8 (observed); give each
// XCD a contiguous run of workgroups so neighbouring image tiles share an L2.  Speed only.
SN_DEV int sn_xcd_remap(int b, int n) {
    const int nx = 8;
    int xcd = b % nx, k = b / nx;
    int qn = n / nx, rn = n % nx;
    int base = xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn;
    return base + k;
}

// logical workgroup index (dispatch order) -> workgroup coordinates in units of 2x2 tiles.  SN_STRIP_W > 0: column strips SN_STRIP_W
// workgroups wide, each walked row by row, so that the ~96 workgroups an XCD runs at a time cover a squarer patch of the image (more
// voxels shared in its L2) than two full-width rows
SN_DEV void sn_main_wg_coords(const SnMainParams& p, int logical_block, int& bx, int& by) {
    const int gbx = (p.tiles_x + 1) / 2, gby = (p.tiles_y + 1) / 2;
    const int blk = sn_xcd_remap(logical_block, gbx * gby);
    if (SN_STRIP_W > 0) {
        const int full = (gbx / SN_STRIP_W) * SN_STRIP_W * gby;  // workgroups inside complete strips
        if (blk < full) {
            const int strip = blk / (SN_STRIP_W * gby), r = blk % (SN_STRIP_W * gby);
            bx = strip * SN_STRIP_W + r % SN_STRIP_W;
            by = r / SN_STRIP_W;
        } else {
            const int w = gbx - (gbx / SN_STRIP_W) * SN_STRIP_W, r = blk - full;  // the narrower last strip
            bx = (gbx / SN_STRIP_W) * SN_STRIP_W + r % w;
            by = r / w;
        }
    } else {
        bx = blk % gbx;
        by = blk / gbx;
    }
}

// What a ray's compositing state becomes: outputs + the chunk-global [min, max] of the sample mid-points.  Shared by the main kernel
// (whole-ray workgroups) and sn_main_combine_kernel (the tail's segment jobs).  bin(k) = euclidean bin k of this lane's ray.
template <bool DUMP, typename BIN>
SN_DEV void sn_main_epilogue(const SnMainParams& p, SnComposite& comp, float r, float g, float b, BIN bin, int S, bool valid, int px, int py, int lane) {
    float out_rgb[3], depth, acc, exp_raw;
    if (p.bg_mode) {  // a constant background instead of the last sample's colour (wave-uniform)
        r = p.bg[0];
        g = p.bg[1];
        b = p.bg[2];
    }
    // the mid-points the outputs need -- first, last, median sample -- from the same bins with the loop's own arithmetic, once per ray
    const int mi = comp.median_index(S);
    const float first_mid = sn_mid(bin(0), bin(1));
    const float last_mid = sn_mid(bin(S - 1), bin(S));
    comp.finish_fused(S, sn_mid(bin(mi), bin(mi + 1)), r, g, b, out_rgb, depth, acc, exp_raw);
    if (valid) {
        const int64_t pix = (int64_t)py * p.width + px;
        if (p.rgb) {
            p.rgb[pix * 3 + 0] = out_rgb[0];
            p.rgb[pix * 3 + 1] = out_rgb[1];
            p.rgb[pix * 3 + 2] = out_rgb[2];
        }
        if (p.depth) p.depth[pix] = depth;
        if (DUMP && p.dump_median) p.dump_median[pix] = comp.median_idx;
        if (p.acc) p.acc[pix] = acc;
        if (p.exp_raw) p.exp_raw[pix] = exp_raw;
    }
    // chunk-global [min, max] of the sample mid-points (A17 quirk): bins are monotone along a ray, so a
    // ray contributes its first and last mid-point.  A tile straddles at most a few chunks.
    if (p.chunk_minmax) {
        const int my_chunk = valid ? (int)(((int64_t)py * p.width + px) / p.chunk_rays) : -1;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int c = __shfl(my_chunk, leader);
            const bool mine = valid && my_chunk == c;
            uint32_t lo = mine ? sn_float_ordered(first_mid) : 0xffffffffu;
            uint32_t hi = mine ? sn_float_ordered(last_mid) : 0u;
#pragma unroll
            for (int s = 32; s >= 1; s >>= 1) {
                lo = min(lo, (uint32_t)__shfl_xor((int)lo, s));
                hi = max(hi, (uint32_t)__shfl_xor((int)hi, s));
            }
            if (lane == leader) {
                atomicMin(&p.chunk_minmax[c], lo);
                atomicMax(&p.chunk_minmax[p.n_chunks + c], hi);
            }
            todo &= ~__ballot(mine);
        }
    }
}

template <int MODE /*0 uniform-in-s bins, 1 explicit bins*/, int PREC /*0 fp32 MFMA, 1 fp16 hi+lo split MFMA, 2 single fp16 (opt-in)*/,
          int GRID = 0 /*0 nerfstudio torch-path hash grid, 1 tiny-cuda-nn grid semantics*/,
          int ND = -1 /*GRID 1: number of leading dense levels, fixed at compile time (-1: run-time decision per level)*/,
          bool DUMP = false /*test instrumentation: record what every sample fetches (SnMainParams::dump_*)*/,
          bool ALT = false /*the non-default sampler / position map: SnMainParams::spacing_uniform and pm are honoured*/,
          bool STATS = false /*diagnostics: count the wave-steps the early termination skips (SnMainParams::march_stats).  Its own instantiation: an
                               atomic in the exit branch of the PRODUCTION kernel changed hipcc's schedule of the whole hash phase -- 4 instead of 16
                               gathers in flight, +10 % per launch (r05, found by a same-box A/B against the r04 library) */>
__global__ __launch_bounds__((SnK1Shape<PREC>::THREADS), (SnK1Shape<PREC>::WAVES_PER_SIMD))
void sn_render_main_kernel(SnMainParams p) {
    using SHAPE = SnK1Shape<PREC>;
    constexpr int NT = SHAPE::THREADS;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int su = ALT ? p.spacing_uniform : 0;          // (constants in the production instantiations: their code is what it was
    const SnPosMap* pm = ALT ? &p.pm : nullptr;          //  before the two options existed)
    // both weight images are SnMainImg::TOTAL floats (42 640 B; the single-fp16 mode appends colour layer 3's fp16 operand)
    constexpr int IMG_FLOATS = PREC == 2 ? SnMainImgF16::TOTAL_FLOATS : SnMainImg::TOTAL;   // (PREC 2: + colour layer 3's fp16 operand)
    for (int i = tid * 4; i < IMG_FLOATS; i += NT * 4) *(f32x4*)(lds + i) = *(const f32x4*)(p.wimg + i);
    // Uniform sampler without per-ray nears / fars (the collider's constants): the S + 1 euclidean bins are the same for every ray
    // of the frame.  They are computed once per workgroup -- with the same strict arithmetic, so bit-identical -- and read back as
    // LDS broadcasts, instead of ~20 VALU instructions (an IEEE division among them) per lane and step.
    float* etab = lds + IMG_FLOATS;
    const bool shared_bins = MODE == 0 && p.nears == nullptr;
    if (shared_bins) {
        const float sn = sn_spacing(p.near_plane, su), sf = sn_spacing(p.far_plane, su);
        for (int i = tid; i <= p.n_samples; i += NT) etab[i] = sn_euclid(p.sbins ? p.sbins[i] : (float)i / (float)p.n_samples, sn, sf, su);
    }
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroup = 2x2 tiles.  Blocks past seg_first_block are segment jobs of the launch's last, partly filled round of workgroups: the
    // same tiles, a slice of the samples each, so that the round fills the chip (SnMainParams "Split-depth tail")
    int logical_block = blockIdx.x, i_lo = 0, i_hi = p.n_samples;
    f32x4* seg_out = nullptr;  // wave-uniform: non-null <=> this workgroup stores samples instead of compositing them
    if (p.n_seg > 1 && (int)blockIdx.x >= p.seg_first_block) {
        // The dispatcher places block b on XCD b % 8 and sn_xcd_remap gives logical block L the tiles of XCD L % 8's band: the segments of a
        // tail workgroup must therefore stay on ITS XCD (seg_first_block is a multiple of 8), one after the other, so that they share
        // its L2 like a whole-ray workgroup does (spread over the XCDs, every segment re-fetched the tile's voxels into another L2: +8 %
        // instead of -5 % at 512x512).  The grid pads the tail to a multiple of 8 workgroups; the padding returns here.
        const int j = (int)blockIdx.x - p.seg_first_block, xcd = j & 7, m = j >> 3;
        const int tail_wg = (m / p.n_seg) * 8 + xcd, k = m % p.n_seg;
        logical_block = p.seg_first_block + tail_wg;
        if (logical_block >= ((p.tiles_x + SHAPE::TX - 1) / SHAPE::TX) * ((p.tiles_y + SHAPE::TY - 1) / SHAPE::TY)) return;
        i_lo = k * p.seg_len;
        i_hi = min(p.n_samples, i_lo + p.seg_len);
        seg_out = p.seg_scratch + ((int64_t)(tail_wg * SHAPE::WG_WAVES + wave) * p.n_samples) * 64 + lane;
        if (i_lo >= i_hi) return;  // (n_seg * seg_len may exceed S by a segment)
    }
    int bx, by;
    sn_main_wg_coords(p, logical_block, bx, by);
    const int tx = bx * SHAPE::TX + (wave % SHAPE::TX);
    const int ty = by * SHAPE::TY + (wave / SHAPE::TX);
    if (tx >= p.tiles_x || ty >= p.tiles_y) return;  // wave-uniform
    const int tw = 1 << p.tile_w_log2;
    const int px = (tx << p.tile_w_log2) + (lane & (tw - 1));
    const int py = (ty << p.tile_h_log2) + (lane >> p.tile_w_log2);
    const bool valid = px < p.width && py < p.height;
    const int cx = min(px, p.width - 1), cy = min(py, p.height - 1);
    const int64_t ray = (int64_t)cy * p.width + cx;

    float o[3], d[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = p.origins[ray * 3 + c];
        d[c] = p.directions[ray * 3 + c];
    }
    const float near = p.nears ? p.nears[ray] : p.near_plane;
    const float far = p.fars ? p.fars[ray] : p.far_plane;
    const float s_near = sn_spacing(near, su), s_far = sn_spacing(far, su);
    constexpr bool EXACTQ = !ALT && (MODE == 0 ? SN_EXACT_POS_UNIFORM != 0 : SN_EXACT_POS_BINS != 0);
    const float dh[3] = {d[0] * 0.5f, d[1] * 0.5f, d[2] * 0.5f};  // sn_sample_q_exact: (d t) / 2 == (d / 2) t
    SnShOps sh;
    SnShOpsH shh;
    SnShOpsF shf;
    if (PREC == 2) shf.build(d, p.sh_remap);
    else if (PREC == 0) sh.build(d, p.sh_remap);
    else shh.build(d, p.sh_remap);

    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.table, (16u << p.log2_t) * 8u);
    const int S = p.n_samples;
    const float* eb = nullptr;
    if (MODE == 1) eb = p.ebins + ((int64_t)(ty * p.tiles_x + tx) * (S + 1)) * 64 + lane;

    SnComposite comp;
    comp.init();
    // euclidean bin k of this lane's ray, with the loop's own arithmetic
    auto bin = [&](int k) -> float {
        return MODE == 0 ? (shared_bins ? etab[k] : sn_euclid(p.sbins ? p.sbins[k] : (float)k / (float)S, s_near, s_far, su)) : eb[(int64_t)k * 64];
    };
    float t0 = bin(i_lo);
    float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll 1
    for (int i = i_lo; i < i_hi; ++i) {
        // The LDS weight reads are loop-invariant; LICM would hoist them all (368 VGPRs) and spill.  A compiler-only
        // memory clobber per iteration keeps them inside the loop.
        asm volatile("" ::: "memory");
        const float t1 = bin(i + 1);
        float q[3];
        // ALT: the literal strict form (it also serves the box map); EXACTQ: the same q from sn_sample_q_exact; else the fast form
        const bool sel = ALT ? sn_sample_q(o, d, t0, t1, q, pm) : (EXACTQ ? sn_sample_q_exact(o, dh, t0, t1, q) : sn_sample_q_fast(o, d, t0, t1, q));
        // (the exact form's longer dependent chain in one scheduling region with the first level group made hipcc serialise that group --
        // load, wait, blend, level by level; fenced off, the group schedules like the other three: 16 gathers in flight)
        if (EXACTQ) __builtin_amdgcn_sched_barrier(0);
        float feat[32];
        {
            // plain table: the x-paired layout (sn_device.h) measured no gain here (r01: 4.11 vs 4.18 ms) -- splitting a level
            // into per-t tables loses the x-locality of the plain layout (16 consecutive x share a 128-B line)
            uint32_t* rec = nullptr;
            if (DUMP && valid) {
                const size_t smp = (size_t)ray * (size_t)S + (size_t)i;
                if (p.dump_fetch) rec = p.dump_fetch + smp * 128;
                if (p.dump_q) {
                    p.dump_q[smp * 3 + 0] = q[0];
                    p.dump_q[smp * 3 + 1] = q[1];
                    p.dump_q[smp * 3 + 2] = q[2];
                }
            }
            // grid arithmetic: 1 = nerfstudio torch grid (fused form), 3 = tiny-cuda-nn grid with de-hashed copies, 2 = tiny-cuda-nn grid read
            // from the uploaded table with the dense / hashed decision per level at run time (ND = -1: shapes the copies do not cover)
            constexpr int AR = GRID ? (ND > 0 ? 3 : 2) : (SN_FAST_HASH ? 1 : 0);
            constexpr int NBC = ND > SN_BC_MAIN ? SN_BC_MAIN : (ND > 0 ? ND : 0);
            if (PREC == 2 && GRID == 1 && ND > 0) {
                // single-fp16 mode: the tiny-cuda-nn grid from its fp16 storage (quads of the de-hashed levels, 4-byte rows of the hashed ones)
                sn_hash_encode_h16<16, (ND > 0 ? ND : 1), SHAPE::HASH_GROUP, SN_H16_PAIRS != 0>(&p.hquads, sn_table_rsrc(p.hrows, p.hrows_bytes), p.hpinfo, p.scal, p.log2_t,
                                                                                              q, feat);
            } else if (PREC == 2) {
                sn_hash_encode<16, SHAPE::HASH_GROUP, AR, ND, DUMP, NBC, 0, true>(rsrc, p.scal, p.log2_t, q, feat, &p.grid, &p.dense, rec, p.feat_scale);
            } else if (SN_MAIN_PAIRS && MODE == 1 && ND > 0 && ND < 16) {
                // de-hashed levels [0, ND), then the hashed levels [ND, 16) from the x-paired tables
                sn_hash_encode<(ND > 0 ? ND : 1), SHAPE::HASH_GROUP, AR, ND, DUMP, NBC>(rsrc, p.scal, p.log2_t, q, feat, &p.grid, &p.dense, rec, p.feat_scale);
                __builtin_amdgcn_sched_barrier(0);
                sn_hash_encode_pairs<16, SHAPE::HASH_GROUP, true, (ND > 0 && ND < 16 ? ND : 0), GRID == 1, DUMP>(sn_table_rsrc(p.pairs, p.pairs_bytes), p.pinfo,
                                                                                                           p.scal, p.log2_t, q, feat, rec);
            } else {
                sn_hash_encode<16, SHAPE::HASH_GROUP, AR, ND, DUMP, NBC>(rsrc, p.scal, p.log2_t, q, feat, &p.grid, &p.dense, rec, p.feat_scale);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        float h0, rgb[3];
        if (PREC == 0) {
            sn_main_field_f32(lds, feat, sh, lane, h0, rgb);
        } else if (PREC == 2) {
            sn_main_field_f16((const char*)lds, feat, shf, lane, h0, rgb);
        } else {
            sn_main_field_h((const char*)lds, feat, shh, lane, h0, rgb);
        }
        __builtin_amdgcn_sched_barrier(0);
        float density = p.avg_density * sn_exp<true>(h0) * (sel ? 1.0f : 0.0f);
        // A NaN position (e.g. the 1e10 sentinel of a ray that misses render_aabb overflows to inf/inf) is NaN all the
        // way through the reference's field; v_max-based ReLU would launder it, so restore it here -- by arithmetic, not selects
        // (sn_sample_q_fast): q * 0 is +-0 for a finite q and NaN for a NaN one, and x + +-0 == x.
        const float nan_term = fmaf(q[2], 0.0f, fmaf(q[1], 0.0f, q[0] * 0.0f));
        density += nan_term;
        r = rgb[0] + nan_term;
        g = rgb[1] + nan_term;
        b = rgb[2] + nan_term;
        if (seg_out) seg_out[(int64_t)i * 64] = f32x4{density, r, g, b};  // (wave-uniform branch) composited later, in sample order
        else comp.step_fused(t0, t1, density, r, g, b);
        t0 = t1;
        // EXACT early termination (r04).  Once the transmittance in front of a sample is exactly 0 in fp32 -- exp(-cumsum(tau)) underflows
        // beyond tau ~ 88: a few samples behind any trained surface -- every later weight of that ray is exactly +0 whatever the field
        // returns (alpha * 0, NaN -> 0 by nan_to_num; cumsum(tau) only grows, and a NaN sum gives a NaN transmittance whose weight
        // nan_to_num zeroes as well), so the sums, the median count and the expected depth cannot change any more.  When that holds for
        // ALL 64 rays of the wave the march jumps to the last sample, whose colour is the 'last_sample' background (rgb += c_last (1 -
        // sum w)).  Outputs are bit-identical (tests/test_gpu_early_term.py); the synthetic benchmark scene never saturates (its
        // densities are O(1): max cumsum(tau) < 88), so the check costs it one v_cmp and one branch per step.  Segment jobs store every
        // sample and the DUMP instantiations record every fetch: not for them.
        if (!DUMP && p.early_term && !seg_out && i < i_hi - 2 && __all(comp.last_trans == 0.0f)) {
            if (STATS && p.march_stats && lane == 0) atomicAdd(&p.march_stats[0], (unsigned long long)(i_hi - 2 - i));
            i = i_hi - 2;
            t0 = bin(i_hi - 1);
        }
    }
    if (seg_out) return;
    sn_main_epilogue<DUMP>(p, comp, r, g, b, bin, S, valid, px, py, lane);
}

// Composites the samples the segment jobs stored (SnMainParams "Split-depth tail"): one wave per tile of the tail workgroups, the
// samples in order through SnComposite::step_fused -- the arithmetic, and therefore every output bit, of a whole-ray workgroup.
template <int MODE>
__global__ __launch_bounds__(256) void sn_main_combine_kernel(SnMainParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    float* etab = lds;
    const bool shared_bins = MODE == 0 && p.nears == nullptr;
    if (shared_bins) {
        const float sn = sn_spacing(p.near_plane, p.spacing_uniform), sf = sn_spacing(p.far_plane, p.spacing_uniform);
        for (int i = tid; i <= p.n_samples; i += 256) etab[i] = sn_euclid(p.sbins ? p.sbins[i] : (float)i / (float)p.n_samples, sn, sf, p.spacing_uniform);
    }
    __syncthreads();
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    sn_main_wg_coords(p, p.seg_first_block + (int)blockIdx.x, bx, by);
    const int tx = bx * 2 + (wave & 1);
    const int ty = by * 2 + (wave >> 1);
    if (tx >= p.tiles_x || ty >= p.tiles_y) return;  // wave-uniform
    const int tw = 1 << p.tile_w_log2;
    const int px = (tx << p.tile_w_log2) + (lane & (tw - 1));
    const int py = (ty << p.tile_h_log2) + (lane >> p.tile_w_log2);
    const bool valid = px < p.width && py < p.height;
    const int64_t ray = (int64_t)min(py, p.height - 1) * p.width + min(px, p.width - 1);
    const float s_near = sn_spacing(p.nears ? p.nears[ray] : p.near_plane, p.spacing_uniform), s_far = sn_spacing(p.fars ? p.fars[ray] : p.far_plane, p.spacing_uniform);
    const int S = p.n_samples;
    const float* eb = nullptr;
    if (MODE == 1) eb = p.ebins + ((int64_t)(ty * p.tiles_x + tx) * (S + 1)) * 64 + lane;
    auto bin = [&](int k) -> float {
        return MODE == 0 ? (shared_bins ? etab[k] : sn_euclid(p.sbins ? p.sbins[k] : (float)k / (float)S, s_near, s_far, p.spacing_uniform)) : eb[(int64_t)k * 64];
    };
    const f32x4* in = p.seg_scratch + ((int64_t)((int)blockIdx.x * 4 + wave) * S) * 64 + lane;
    SnComposite comp;
    comp.init();
    float t0 = bin(0), r = 0.f, g = 0.f, b = 0.f;
    // the compositing chain is sequential, its inputs are not: 8 samples (and their bins) are fetched at a time, so that the wave waits
    // for memory S / 8 times instead of S times
    constexpr int B = 8;
    for (int i0 = 0; i0 < S; i0 += B) {
        f32x4 v[B];
        float t[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int i = min(i0 + k, S - 1);
            v[k] = in[(int64_t)i * 64];
            t[k] = bin(i + 1);
        }
#pragma unroll
        for (int k = 0; k < B; ++k)
            if (i0 + k < S) {  // wave-uniform
                r = v[k].y;
                g = v[k].z;
                b = v[k].w;
                comp.step_fused(t0, t[k], v[k].x, r, g, b);
                t0 = t[k];
            }
    }
    sn_main_epilogue<false>(p, comp, r, g, b, bin, S, valid, px, py, lane);
}

// expected_depth = clip(raw, chunk min, chunk max)
__global__ void sn_clip_expected_kernel(const float* raw, const uint32_t* chunk_minmax, int64_t n, int chunk_rays, int n_chunks,
                                        float* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = (int)(i / chunk_rays);
    float lo = sn_ordered_float(chunk_minmax[c]), hi = sn_ordered_float(chunk_minmax[n_chunks + c]);
    const float x = raw[i];
    // torch.clip keeps a NaN (a ray with no weight at all: 0 / 1e-10 x an infinite mid-point); fminf / fmaxf would return the bound
    out[i] = x != x ? x : fminf(fmaxf(x, lo), hi);
}
