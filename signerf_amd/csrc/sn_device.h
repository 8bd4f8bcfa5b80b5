// sn_device.h -- device-side building blocks shared by the fused render kernels and the
// stage-level kernels (gfx950 / CDNA4 only).
//
// Numerics contract (DESIGN.md "Numerics"):
//   * everything that decides an INTEGER (hash-grid corner coordinates, table rows, searchsorted
//     and median indices) is computed with un-fused IEEE fp32 ops in the operand order of the
//     reference's torch-CPU path, so the integers are bit-identical given bit-identical inputs;
//   * wherever torch-CPU uses cumsum (which accumulates fp32 data in fp64 and rounds each prefix
//     to fp32) we accumulate in fp64 as well;
//   * the remaining fp32 work (trilinear blend, MLPs, sums) is free to use FMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SN_DEV __device__ __forceinline__

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define SN_MAX_LEVELS_DEV 16

// ------------------------------------------------------------------------------------------
// strict (no-FMA) helpers
// ------------------------------------------------------------------------------------------

// s(x) of the initial sampler: UniformLinDispPiecewiseSampler, or -- `uniform`, SnRenderOpts.spacing_mode = 1 --
// UniformSampler's identity.  `uniform` is the same for every lane of a launch (a scalar branch).
SN_DEV float sn_spacing(float x, int uniform = 0) {
#pragma clang fp contract(off)
    if (uniform) return x;
    return x < 1.0f ? x / 2.0f : 1.0f - 1.0f / (2.0f * x);
}

// s^-1(y)
SN_DEV float sn_spacing_inv(float x, int uniform = 0) {
#pragma clang fp contract(off)
    if (uniform) return x;
    return x < 0.5f ? 2.0f * x : 1.0f / (2.0f - 2.0f * x);
}

// mid-point of a bin, un-fused (the reference's (start + end) / 2)
SN_DEV float sn_mid(float a, float b) {
#pragma clang fp contract(off)
    return (a + b) / 2.0f;
}

// spacing bin -> euclidean distance along the ray: s^-1(b * s_far + (1 - b) * s_near)
SN_DEV float sn_euclid(float b, float s_near, float s_far, int uniform = 0) {
#pragma clang fp contract(off)
    float x = b * s_far + (1.0f - b) * s_near;
    return sn_spacing_inv(x, uniform);
}

// The fields' selector ((q > 0) & (q < 1)).all(-1) as ONE v_min3 / v_max3 pair and two compares instead of six compares and five
// s_and (r03: 800x800x64 frame -1.9 %, 1080p nerfacto frame -3.9 %, same box, interleaved -- the VCC-form compares and their scalar
// chain cost more than their count says).  min / max skip a NaN operand, so a position with a NaN coordinate may pass where the six
// compares fail -- it is NaN through the field either way (q * m and density * m stay NaN for m = 0 or 1): identical outputs.
SN_DEV bool sn_in_unit_cube(const float q[3]) {
    return __builtin_fminf(__builtin_fminf(q[0], q[1]), q[2]) > 0.0f && __builtin_fmaxf(__builtin_fmaxf(q[0], q[1]), q[2]) < 1.0f;
}

// Position -> grid coordinate of the fields (SnFieldDesc.disable_scene_contraction): box = 0, SceneContraction(inf) then (p + 2) / 4
// (nerfacto's default); box = 1, SceneBox.get_normalized_positions: (p - lo) / len.  The same for every lane of a launch.
struct SnPosMap {
    int box;
    float lo[3], len[3];  // every kernel divides by len, as SceneBox does
};

// Frustums.get_positions + SceneContraction(inf) + (p+2)/4 + selector (A5, A6).
// Returns q (already multiplied by the selector) and the selector.
SN_DEV bool sn_sample_q(const float o[3], const float d[3], float start, float end, float q[3], const SnPosMap* pm = nullptr) {
#pragma clang fp contract(off)
    float t = start + end;
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = o[c] + (d[c] * t) / 2.0f;
    const bool box = pm && pm->box;
    if (!box) {
        float mag = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2]));
        if (!(mag < 1.0f)) {
            float k = 2.0f - (1.0f / mag);
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = k * (p[c] / mag);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = box ? (p[c] - pm->lo[c]) / pm->len[c] : (p[c] + 2.0f) / 4.0f;
    const bool sel = sn_in_unit_cube(q);
    float m = sel ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = q[c] * m;
    return sel;
}

// Fused-kernel form: the same map with v_rcp_f32 (1 ulp) instead of four IEEE divisions in the contraction.  Positions only
// feed the fields (float work); the bins themselves (sn_euclid) stay strict.  Saves ~36 VALU instructions per sample.
// NANFREE (the proposal kernel, whose one-tile MLP must not see a NaN operand -- sn_prop_h0): q comes back finite, a NaN position as
// q = 0, and *nanq carries it: +-0 for a finite position, NaN otherwise.  The NaN is dropped by the `clamp` modifier of the selector
// multiply itself (q m is in [0, 1); a compute kernel runs with DX10_CLAMP set: NaN clamps to 0) -- no instruction added, no select.
template <bool NANFREE = false>
// Beyond |p| ~ 1.7e7 (1 / |p| below half an ulp of 2) the exact contraction rounds onto the face q = 1, which the selector drops; the
// reciprocal form may land one ulp inside.  The default sampler never gets there (far_plane 1000; rays that miss a render box are NaN);
// the ALT instantiations (uniform sampler: such rays sit at a finite 1e10) use the strict sn_sample_q instead.
SN_DEV bool sn_sample_q_fast(const float o[3], const float d[3], float start, float end, float q[3], float* nanq = nullptr) {
    const float t = (start + end) * 0.5f;
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = fmaf(d[c], t, o[c]);
    const float mag = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2]));
    // No selects: r = min(1 / mag, 1) makes k = (2 - r) r exactly 1 inside the unit box, where the contraction is the identity (NaN
    // positions stay NaN: min returns 1, NaN * 1).  v_cndmask_b32 in its VCC form issues ~5x slower than other VALU instructions on gfx950
    // (tools/probes/overlap2_probe.hip, r02), so the fused kernels avoid per-step selects.
    {
        const float r = fminf(__builtin_amdgcn_rcpf(mag), 1.0f);
        const float k = (2.0f - r) * r;
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = fmaf(k * p[c], 0.25f, 0.5f);
    }
    const bool sel = sn_in_unit_cube(q);
    const float m = sel ? 1.0f : 0.0f;
    if (NANFREE) {
        *nanq = fmaf(q[2], 0.0f, fmaf(q[1], 0.0f, q[0] * 0.0f));
#pragma unroll
        for (int c = 0; c < 3; ++c) asm("v_mul_f32_e64 %0, %1, %2 clamp" : "=v"(q[c]) : "v"(q[c]), "v"(m));
        return sel;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = q[c] * m;
    return sel;
}

// r06 -- the STRICT map at (nearly) the fast form's price: q is sn_sample_q's, bit for bit, without its four IEEE divisions.
// The main kernel behind the UNIFORM sampler uses it: there a sample's position depends on no computed weight, so with strict position
// arithmetic every voxel and every blend offset of the fused kernel is the oracle's (tests/test_gpu_fused_indices.py: 0 flips).
//   * (d t) / 2 == (d / 2) t: a power-of-two scale commutes with the rounding of the product (dh = d * 0.5, formed once per ray);
//   * m = max(mag, 1): for mag < 1 the reference skips the contraction -- with m = 1 every quotient p / 1 and k = 2 - 1 / 1 = 1 is exact,
//     so the branch-free form returns p itself (NaN positions: min / max skip them, k p stays NaN in that coordinate; sn_main.h restores
//     the all-NaN sample from it as before);
//   * y = RN(1 / m) from v_rcp_f32 (1 ulp) + TWO Newton steps: equal to the IEEE quotient for every fp32 m whose significand is not all
//     ones, whichever neighbour of it the instruction returns (one step is not enough when v_rcp_f32 errs upwards: 32 of 2^23
//     significands; tools/recip_exhaustive.py emulates the fmas exactly over all 2^23 significands x {RN - 1 ulp, RN, RN + 1 ulp});
//     for an all-ones significand m = 2^e (2 - 2^-23) the quotient is 2^-(e+1) (1 + 2^-23) = bits 0x7F000000 - bits(m), substituted by
//     a select;
//   * p / m = RN(q0 + (p - q0 m) y), q0 = RN(p y): the correctly rounded quotient from the correctly rounded reciprocal + one exact
//     residual (Markstein; sn_proposal.h uses the same identity, tests/test_recip_division.py holds it against 480 M emulated pairs);
//   * (k p' + 2) / 4 == fma(k p', 0.25, 0.5): again a power-of-two scale.
// 32 VALU instead of the fast form's 16 and the literal form's ~52.  Checked on the hardware itself against sn_sample_q
// (sn_debug_sample_positions: random, near-halfway and all-ones cases; tests/test_gpu_stages.py).
SN_DEV bool sn_sample_q_exact(const float o[3], const float dh[3], float start, float end, float q[3]) {
#pragma clang fp contract(off)
    const float t = start + end;
    float p[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) p[c] = o[c] + dh[c] * t;
    const float mag = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2]));
    const float m = fmaxf(mag, 1.0f);
    float y = __builtin_amdgcn_rcpf(m);
    y = __builtin_fmaf(__builtin_fmaf(-m, y, 1.0f), y, y);
    y = __builtin_fmaf(__builtin_fmaf(-m, y, 1.0f), y, y);
    const uint32_t mb = __float_as_uint(m);
    y = (mb & 0x7fffffu) == 0x7fffffu ? __uint_as_float(0x7F000000u - mb) : y;  // (a select, not a branch: a branch here splits the hash phase's scheduling region)
    const float k = 2.0f - y;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float q0 = p[c] * y;
        const float pc = __builtin_fmaf(__builtin_fmaf(-q0, m, p[c]), y, q0);
        q[c] = __builtin_fmaf(k * pc, 0.25f, 0.5f);
    }
    const bool sel = sn_in_unit_cube(q);
    const float msel = sel ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = q[c] * msel;
    return sel;
}

// Same, from a world position (stage-level field_forward).
SN_DEV bool sn_position_q(const float pin[3], float q[3], const SnPosMap* pm = nullptr) {
#pragma clang fp contract(off)
    float p[3] = {pin[0], pin[1], pin[2]};
    const bool box = pm && pm->box;
    if (!box) {
        float mag = fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fabsf(p[2]));
        if (!(mag < 1.0f)) {
            float k = 2.0f - (1.0f / mag);
#pragma unroll
            for (int c = 0; c < 3; ++c) p[c] = k * (p[c] / mag);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = box ? (p[c] - pm->lo[c]) / pm->len[c] : (p[c] + 2.0f) / 4.0f;
    const bool sel = sn_in_unit_cube(q);
    float m = sel ? 1.0f : 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) q[c] = q[c] * m;
    return sel;
}

// ------------------------------------------------------------------------------------------
// hash grid (A7, torch path): every level hashed, corners ceil/floor, primes (1, 2654435761, 805459861)
// ------------------------------------------------------------------------------------------

struct SnHashLevel {
    uint32_t boff[8];  // BYTE offset (row * 8) within the level, nerfstudio corner order
    float off[3];      // scaled - floor(scaled)
};

// Integer part.  uint32 wrap-around arithmetic equals the reference's int64 products followed by
// `% 2**k` because only the low k <= 32 bits survive and coordinates are non-negative.
SN_DEV void sn_hash_corners(const float q[3], float scale, uint32_t mask, SnHashLevel& hl) {
#pragma clang fp contract(off)
    uint32_t f[3], c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float x = q[a] * scale;
        float fl = floorf(x);
        hl.off[a] = x - fl;
        f[a] = (uint32_t)(int)fl;
        c[a] = (uint32_t)(int)ceilf(x);
    }
    // Only the low log2(T) bits of the products survive `& mask`, so the primes can be reduced mod T first (T <= 2^24 is
    // enforced by sn_create); coordinates are < 2^24, and v_mul_u32_u24 returns the exact low 32 bits of a 24x24-bit product,
    // so the FULL-RATE 24-bit multiply gives the same masked bits as the reference's int64 product (v_mul_lo_u32 is quarter
    // rate).  Bit-exact: tests/test_gpu_stages.py compares every table row.
    // The rows are wanted as byte offsets (x 8): shifting the three xor components instead of the eight results folds the
    // shift of y and z into the (reduced) primes -- (P & mask) << 3 still fits 24 bits for T <= 2^21 -- and leaves two shifts.
    const uint32_t P1 = (2654435761u & mask) << 3, P2 = (805459861u & mask) << 3, m8 = mask << 3;
    const uint32_t yf = __umul24(f[1], P1), yc = __umul24(c[1], P1), zf = __umul24(f[2], P2), zc = __umul24(c[2], P2);
    const uint32_t xf = f[0] << 3, xc = c[0] << 3;
    // order: 0 ccc, 1 cfc, 2 ffc, 3 fcc, 4 ccf, 5 cff, 6 fff, 7 fcf
    hl.boff[0] = (xc ^ yc ^ zc) & m8;
    hl.boff[1] = (xc ^ yf ^ zc) & m8;
    hl.boff[2] = (xf ^ yf ^ zc) & m8;
    hl.boff[3] = (xf ^ yc ^ zc) & m8;
    hl.boff[4] = (xc ^ yc ^ zf) & m8;
    hl.boff[5] = (xc ^ yf ^ zf) & m8;
    hl.boff[6] = (xf ^ yf ^ zf) & m8;
    hl.boff[7] = (xf ^ yc ^ zf) & m8;
}

// Trilinear blend in the reference's association (x, then y, then z); FMA allowed.
SN_DEV f32x2 sn_hash_blend(const f32x2 v[8], const float off[3]) {
    float ox = off[0], oy = off[1], oz = off[2];
    float nx = 1.0f - ox, ny = 1.0f - oy, nz = 1.0f - oz;
    f32x2 f03 = v[0] * ox + v[3] * nx;
    f32x2 f12 = v[1] * ox + v[2] * nx;
    f32x2 f56 = v[5] * ox + v[6] * nx;
    f32x2 f47 = v[4] * ox + v[7] * nx;
    f32x2 f0312 = f03 * oy + f12 * ny;
    f32x2 f4756 = f47 * oy + f56 * ny;
    return f0312 * oz + f4756 * nz;
}

// Fused-kernel variants (same values, fewer instructions; the staged entry points keep the literal forms above so that
// tests/test_gpu_stages.py can compare table rows and features bit for bit):
//  * positions are normalised into [0, 1] before they get here, so floor == truncation (v_cvt_i32_f32) and x - floor(x) ==
//    v_fract_f32(x), both exact;
//  * the "ceil" corner is taken as floor + 1.  It differs from ceil(x) only when x is an integer, where its blend weight
//    x - floor(x) is exactly 0, so the blended value is unchanged (table entries are finite) -- and (c + 1) * P == c * P + P
//    turns the second multiply of each axis into an add;
//  * the blend is a + w * (b - a) (one packed add + one packed FMA per lerp, no 1 - w): 1-ulp differences from the
//    reference's a * w + b * (1 - w) association, far inside the 1e-3 budget of the rendered outputs.
SN_DEV void sn_hash_corners_fast(const float q[3], float scale, uint32_t mask, SnHashLevel& hl) {
    uint32_t f[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = q[a] * scale;
        hl.off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    const uint32_t P1 = (2654435761u & mask) << 3, P2 = (805459861u & mask) << 3, m8 = mask << 3;
    const uint32_t yf = __umul24(f[1], P1), zf = __umul24(f[2], P2), xf = f[0] << 3;
    const uint32_t yc = yf + P1, zc = zf + P2, xc = xf + 8u;
    hl.boff[0] = (xc ^ yc ^ zc) & m8;
    hl.boff[1] = (xc ^ yf ^ zc) & m8;
    hl.boff[2] = (xf ^ yf ^ zc) & m8;
    hl.boff[3] = (xf ^ yc ^ zc) & m8;
    hl.boff[4] = (xc ^ yc ^ zf) & m8;
    hl.boff[5] = (xc ^ yf ^ zf) & m8;
    hl.boff[6] = (xf ^ yf ^ zf) & m8;
    hl.boff[7] = (xf ^ yc ^ zf) & m8;
}

SN_DEV f32x2 sn_hash_blend_fast(const f32x2 v[8], const float off[3]) {
    // Plain (un-packed) fp32 instructions ON PURPOSE.  Measured r02 (tools/probes/overlap2_probe.hip, profiles/r02_overlap2_probe.txt):
    // on gfx950 v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 are mutually exclusive with the matrix pipe of their SIMD -- a packed op
    // waits for the MFMA in flight (of ANY wave of the SIMD) and the next MFMA waits for it -- while plain VALU ops (v_fma_f32,
    // v_sub_f32, integer, conversions) issue beside a running v_mfma_f32_32x32x16_f16 for free.  In the fused kernels the blend of one
    // wave runs beside the MLP of the others, so 2 plain ops beat 1 packed op.  (The library is built with -fno-slp-vectorize so that
    // hipcc does not re-pack them.)
    const float ox = off[0], oy = off[1], oz = off[2];
    f32x2 out;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float f03 = fmaf(v[0][c] - v[3][c], ox, v[3][c]);
        const float f12 = fmaf(v[1][c] - v[2][c], ox, v[2][c]);
        const float f56 = fmaf(v[5][c] - v[6][c], ox, v[6][c]);
        const float f47 = fmaf(v[4][c] - v[7][c], ox, v[7][c]);
        const float f0312 = fmaf(f03 - f12, oy, f12);
        const float f4756 = fmaf(f47 - f56, oy, f56);
        out[c] = fmaf(f0312 - f4756, oz, f4756);
    }
    return out;
}

// tiny-cuda-nn grid semantics (`implementation="tcnn"` checkpoints; UNPINNED -- oracle/tcnn_layout.py):
// x = fmaf(scale, q, 0.5) with the library's real-valued per-level scale, corners floor(x) and floor(x) + 1, and a level whose
// whole grid fits its table is indexed densely (x + y*res + z*res^2, wrapping modulo the level size as the library does),
// otherwise with the same xor hash.  The tables keep this library's uniform [L][T] layout; the importer puts a dense level's
// rows at the start of its slot.  The level table is wave-uniform (kernel arguments, 4 SGPRs).
struct SnGridLevels {
    uint32_t packed[SN_MAX_LEVELS_DEV / 4];  // 8 bits per level: its resolution if it is indexed densely, else 0 (dense => res <= 128)
};
SN_DEV uint32_t sn_grid_dense_res(const SnGridLevels& g, int l) {
    uint32_t w = g.packed[l >> 2];
    // opaque to the optimiser: otherwise the per-level resolution, its square and the level size of all 16 levels are hoisted
    // out of the sample loop, 48 live SGPRs that spill into VGPR lanes (measured: 190 spilled VGPRs); 5 SALU ops per level instead
    asm volatile("" : "+s"(w));
    return (w >> ((l & 3) * 8)) & 0xffu;
}

// DENSE: 1 / 0 = the level is dense / hashed (known at compile time), -1 = decided at run time from dense_res (wave-uniform
// branch; in the fused kernels that form spills ~190 VGPRs, so they are instantiated per number of leading dense levels).
template <int DENSE>
SN_DEV void sn_hash_corners_tcnn(const float q[3], float scale, uint32_t mask, uint32_t dense_res, SnHashLevel& hl) {
    uint32_t f[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = fmaf(scale, q[a], 0.5f);
        hl.off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    if (DENSE < 0 ? dense_res != 0u : DENSE == 1) {
        const uint32_t r = dense_res, r2 = r * r;
        const uint32_t dense_size = (r2 * r + 7u) & ~7u;  // rows of the level in the library's layout = the wrap modulus
        const uint32_t i000 = f[0] + __umul24(f[1], r) + __umul24(f[2], r2);
        // nerfstudio corner order 0 ccc, 1 cfc, 2 ffc, 3 fcc, 4 ccf, 5 cff, 6 fff, 7 fcf with c = +1, f = +0 (x y z)
        const uint32_t dx[8] = {1, 1, 0, 0, 1, 1, 0, 0}, dy[8] = {1, 0, 0, 1, 1, 0, 0, 1}, dz[8] = {1, 1, 1, 1, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t idx = i000 + dx[k] + dy[k] * r + dz[k] * r2;
            idx = idx >= dense_size ? idx - dense_size : idx;  // index % size: idx < 2 * size always
            hl.boff[k] = idx << 3;
        }
    } else {
        const uint32_t P1 = (2654435761u & mask) << 3, P2 = (805459861u & mask) << 3, m8 = mask << 3;
        const uint32_t yf = __umul24(f[1], P1), zf = __umul24(f[2], P2), xf = f[0] << 3;
        const uint32_t yc = yf + P1, zc = zf + P2, xc = xf + 8u;
        hl.boff[0] = (xc ^ yc ^ zc) & m8;
        hl.boff[1] = (xc ^ yf ^ zc) & m8;
        hl.boff[2] = (xf ^ yf ^ zc) & m8;
        hl.boff[3] = (xf ^ yc ^ zc) & m8;
        hl.boff[4] = (xc ^ yc ^ zf) & m8;
        hl.boff[5] = (xc ^ yf ^ zf) & m8;
        hl.boff[6] = (xf ^ yf ^ zf) & m8;
        hl.boff[7] = (xf ^ yc ^ zf) & m8;
    }
}

// a * b + c on the full-rate 24-bit multiplier, b wave-uniform (an SGPR: the one constant-bus operand a VOP3 instruction may carry).
// hipcc forms `v_mul_u32_u24; v_mul_u32_u24; v_lshlrev; v_add3` for x * 2^k + y * s1 + z * s2; two of these and the shift are 3.
SN_DEV uint32_t sn_mad24(uint32_t a, uint32_t b_uniform, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}

// Buffer resource over a hash table (base must be wave-uniform: a kernel argument).
SN_DEV __amdgpu_buffer_rsrc_t sn_table_rsrc(const float* table, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (int)bytes, 0x00020000);
}

// Cache policy of the three kinds of gather (the builtin's `aux` operand: 1 = sc0, 2 = nt, 16 = sc1; MI355X_MICROARCH.md: sc1 / nt loads are
// served by the L2 without allocating in the CU's L1).  Compile-time experiment knobs; 0 = default policy.
#ifndef SN_AUX_ROW
#define SN_AUX_ROW 0     // 8-byte rows of a hashed level
#endif
#ifndef SN_AUX_DENSE
#define SN_AUX_DENSE 0   // 16-byte fetches of the de-hashed copies
#endif
#ifndef SN_AUX_PAIR
#define SN_AUX_PAIR 0    // 16-byte fetches of the x-paired tables (K2)
#endif

SN_DEV f32x2 sn_table_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, uint32_t level_off_bytes) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)byte_off, (int)level_off_bytes, SN_AUX_ROW);
    f32x2 o;
    o.x = __uint_as_float(r.x);
    o.y = __uint_as_float(r.y);
    return o;
}

// Encode one point over L levels -> feat[2L], level-major.  `scal` must be wave-uniform.
// GROUP > 0 fences the instruction scheduler every GROUP levels: at most GROUP*8 gathers (GROUP*16 VGPRs) are in flight,
// which keeps the fused kernels inside their register budget (the scheduler otherwise hoists all L*8 loads).
// one 16-byte gather (two adjacent rows, or half a bilinear-coefficient entry)
SN_DEV f32x4 sn_table_load_pair(__amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off, uint32_t level_off_bytes) {
    typedef uint32_t u32x4_ __attribute__((ext_vector_type(4)));
    u32x4_ r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)level_off_bytes, SN_AUX_DENSE);
    return f32x4{__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w)};
}

// De-hashed copies of the COARSE levels of a grid (an MI355X data layout, not a change of arithmetic):
//     D_l[x + R y + R^2 z] = table[l][index_l(x, y, z)],   0 <= x, y, z < R
// index_l = the grid's own row function: nerfstudio's torch HashEncoding hashes every level (R = scale_l + 2); tiny-cuda-nn indexes a level
// whose whole grid fits the table densely, x + y res + z res^2 modulo the level size, and hashes the others (R = floor(scale_l + 0.5) + 2;
// its positions are scale q + 0.5, TCNN below).
// built once by sn_finalize_weights.  In D_l the x + 1 corner is the next row, so a level costs four 16-byte gathers instead
// of eight 8-byte ones and neighbouring voxels share cache lines.  Values are the table's own, so results are bit-identical
// to the hashed reads.  Which levels are copied is the host's choice (sn_api.hip, build_dense_copies).
struct SnDenseCopy {
    const float* base;     // all copied levels, back to back
    uint32_t bytes;
    uint32_t off[12];      // byte offset of level l's copy (l < number of copied levels <= 12)
    uint32_t res[12];      // R of level l
    // Levels [0, n_bc) are stored in BILINEAR-COEFFICIENT form, 32 bytes per grid point (x, y, z):
    //     A = v(x,y,z)   B = v(x+1,y,z) - A   C = v(x,y+1,z) - A   D = (v(x+1,y+1,z) - v(x+1,y,z)) - C          (2 features each)
    // so that the blend inside one z slice is A + ox B + oy (C + ox D): 3 FMAs per feature instead of 3 lerps of 2 instructions, and a
    // level costs 16 VALU of blend instead of 28 -- the fused kernels are VALU-issue bound (r02).  Same four 16-byte gathers per level
    // (two adjacent ones per slice).  The remaining copied levels keep plain rows (8 bytes, x + 1 = next row).  Orientation sets
    // (r01: y- / z-fast duplicates, +0.4 % for 3x the bytes, profiles/r02_dense_sweep.txt) are gone.
    uint32_t n_bc;
};

// rec (test instrumentation, sn_render_rays_debug): where non-null, receives the byte offsets of the four 16-byte fetches
// (order: y1 z1, y0 z1, y0 z0, y1 z0 -- each returns the x0 and x0 + 1 entries) within the buffer of copies.
template <bool TCNN = false>
SN_DEV f32x2 sn_hash_level_dense_copy(__amdgpu_buffer_rsrc_t rsrc, uint32_t level_off_bytes, const float q[3], float scale, uint32_t R,
                                      uint32_t* rec = nullptr) {
    uint32_t f[3];
    float off[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = TCNN ? fmaf(scale, q[a], 0.5f) : q[a] * scale;
        off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    // byte offsets directly: the x8 is folded into the (wave-uniform) strides; R <= 255, so 8 R^2 and every product fit 24 bits
    // (full-rate v_mul_u32_u24 / v_mad_u32_u24; v_mul_lo_u32 is quarter rate)
    const uint32_t R8 = R << 3, R28 = (R * R) << 3;
    const uint32_t b_ff = sn_mad24(f[2], R28, sn_mad24(f[1], R8, f[0] << 3));
    // the y + 1 / z + 1 strides are wave-uniform: they ride in the buffer instruction's SCALAR offset (3 s_add per level on the
    // scalar port) instead of three per-lane v_add
    const uint32_t o_cf = level_off_bytes + R8, o_fc = level_off_bytes + R28, o_cc = level_off_bytes + (R8 + R28);
    if (rec) {
        rec[0] = b_ff + o_cc;
        rec[1] = b_ff + o_fc;
        rec[2] = b_ff + level_off_bytes;
        rec[3] = b_ff + o_cf;
    }
    const f32x4 p_cc = sn_table_load_pair(rsrc, b_ff, o_cc);
    const f32x4 p_fc = sn_table_load_pair(rsrc, b_ff, o_fc);
    const f32x4 p_ff = sn_table_load_pair(rsrc, b_ff, level_off_bytes);
    const f32x4 p_cf = sn_table_load_pair(rsrc, b_ff, o_cf);
    f32x2 v[8];
    v[3] = f32x2{p_cc.x, p_cc.y};
    v[0] = f32x2{p_cc.z, p_cc.w};
    v[2] = f32x2{p_fc.x, p_fc.y};
    v[1] = f32x2{p_fc.z, p_fc.w};
    v[6] = f32x2{p_ff.x, p_ff.y};
    v[5] = f32x2{p_ff.z, p_ff.w};
    v[7] = f32x2{p_cf.x, p_cf.y};
    v[4] = f32x2{p_cf.z, p_cf.w};
    return sn_hash_blend_fast(v, off);
}

// One level in bilinear-coefficient form (SnDenseCopy::n_bc): per feature  f_z(ox, oy) = A + ox B + oy (C + ox D), then the z lerp.
// Algebraically the trilinear blend; rounding differs from the lerp form by a few ulp of the table magnitude (the blend was never
// bit-exact with the reference's association -- DESIGN.md "Numerics").
template <bool TCNN = false>
SN_DEV f32x2 sn_hash_level_dense_bc(__amdgpu_buffer_rsrc_t rsrc, uint32_t level_off_bytes, const float q[3], float scale, uint32_t R,
                                    uint32_t* rec = nullptr) {
    uint32_t f[3];
    float off[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = TCNN ? fmaf(scale, q[a], 0.5f) : q[a] * scale;
        off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    // 32-byte entries; R <= 255 keeps 32 R^2 and every product inside 24 bits
    const uint32_t R32 = R << 5, R2_32 = (R * R) << 5;
    const uint32_t b = sn_mad24(f[2], R2_32, sn_mad24(f[1], R32, f[0] << 5));  // shift + two v_mad_u32_u24
    const uint32_t o_z1 = level_off_bytes + R2_32;  // the z + 1 slice: a wave-uniform stride, carried by the scalar offset
    if (rec) {
        rec[0] = b + level_off_bytes;
        rec[1] = b + o_z1;
    }
#if defined(SN_EXP_ONE_LOAD)  // experiment (wrong images): one of the four fetches only -- how much of the kernel is the gather path?
    const f32x4 ab0 = sn_table_load_pair(rsrc, b, level_off_bytes), cd0 = ab0, ab1 = ab0, cd1 = ab0;
#elif defined(SN_EXP_TWO_LOAD)  // experiment (wrong images): two fetches and 8 extra VALU instructions -- the cost shape of a pair-cooperative fetch
    const f32x4 ab0 = sn_table_load_pair(rsrc, b, level_off_bytes), cd0 = sn_table_load_pair(rsrc, b + 16u, level_off_bytes);
    f32x4 ab1 = ab0, cd1 = cd0;
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                 : "+v"(ab1.x), "+v"(ab1.y), "+v"(ab1.z), "+v"(ab1.w) : "v"(off[0]), "v"(off[1]));
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                 : "+v"(cd1.x), "+v"(cd1.y), "+v"(cd1.z), "+v"(cd1.w) : "v"(off[0]), "v"(off[1]));
#else
    const f32x4 ab0 = sn_table_load_pair(rsrc, b, level_off_bytes), cd0 = sn_table_load_pair(rsrc, b + 16u, level_off_bytes);
    const f32x4 ab1 = sn_table_load_pair(rsrc, b, o_z1), cd1 = sn_table_load_pair(rsrc, b + 16u, o_z1);
#endif
    const float ox = off[0], oy = off[1], oz = off[2];
    f32x2 out;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float z0 = fmaf(ox, ab0[2 + c], fmaf(oy, fmaf(ox, cd0[2 + c], cd0[c]), ab0[c]));
        const float z1 = fmaf(ox, ab1[2 + c], fmaf(oy, fmaf(ox, cd1[2 + c], cd1[c]), ab1[c]));
        out[c] = fmaf(z1 - z0, oz, z0);
    }
    return out;
}

// The same level with its four fetches KEPT ACROSS STEPS of the marching loop (proposal kernel, r02): consecutive samples of a ray fall into
// the same voxel of a coarse level most of the time -- and so does the whole 8x8 tile (measured on the 1080p bench frame, 256 samples
// uniform in s: the wave stays inside its voxels on 92 / 86 / 76 / 63 / 44 % of the steps at resolutions 16 / 27 / 45 / 76 / 128) -- so
// the coefficients are held in registers and re-fetched only on steps where ANY lane changed its entry (wave-uniform branch; every
// lane then re-fetches its own).  Values and blend are those of sn_hash_level_dense_bc: bit-identical results, fewer gathers.
struct SnBcCache {
    uint32_t b;  // byte offset of the entry the held coefficients belong to (0xffffffff: nothing held)
    f32x4 ab0, cd0, ab1, cd1;
    // wave-uniform test switch (SN_PROP_CACHE_OFF=1): re-fetch on every step, i.e. the plain path.  Kept as a MASK or-ed into the held key
    // (all ones: the key never equals an entry offset) so that the per-step test stays one compare and one branch -- r03: compare ->
    // scalar chains in the marching loop cost more than their instruction count says (sn_in_unit_cube)
    uint32_t always_mask;
    SN_DEV void reset(bool always_refetch = false) {
        b = 0xffffffffu;
        always_mask = always_refetch ? 0xffffffffu : 0u;
    }
};
template <bool TCNN = false>
SN_DEV f32x2 sn_hash_level_dense_bc_cached(__amdgpu_buffer_rsrc_t rsrc, uint32_t level_off_bytes, const float q[3], float scale, uint32_t R,
                                           SnBcCache& c, uint32_t* rec = nullptr) {
    uint32_t f[3];
    float off[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = TCNN ? fmaf(scale, q[a], 0.5f) : q[a] * scale;
        off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    const uint32_t R32 = R << 5, R2_32 = (R * R) << 5;
    const uint32_t b = sn_mad24(f[2], R2_32, sn_mad24(f[1], R32, f[0] << 5));  // shift + two v_mad_u32_u24
    const uint32_t o_z1 = level_off_bytes + R2_32;
    if (rec) {
        rec[0] = b + level_off_bytes;
        rec[1] = b + o_z1;
    }
    if (__builtin_amdgcn_ballot_w64(b != c.b) != 0ull) {
        c.b = b | c.always_mask;
        c.ab0 = sn_table_load_pair(rsrc, b, level_off_bytes);
        c.cd0 = sn_table_load_pair(rsrc, b + 16u, level_off_bytes);
        c.ab1 = sn_table_load_pair(rsrc, b, o_z1);
        c.cd1 = sn_table_load_pair(rsrc, b + 16u, o_z1);
    }
    const float ox = off[0], oy = off[1], oz = off[2];
    f32x2 out;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float z0 = fmaf(ox, c.ab0[2 + k], fmaf(oy, fmaf(ox, c.cd0[2 + k], c.cd0[k]), c.ab0[k]));
        const float z1 = fmaf(ox, c.ab1[2 + k], fmaf(oy, fmaf(ox, c.cd1[2 + k], c.cd1[k]), c.ab1[k]));
        out[k] = fmaf(z1 - z0, oz, z0);
    }
    return out;
}

// row of grid point (x, y, z) inside level slot `lv`: dense_res = 0 -> the xor hash (torch grids: every level; tcnn: the hashed levels),
// else tiny-cuda-nn's dense index x + y res + z res^2 modulo the level size next_multiple(res^3, 8)
SN_DEV uint32_t sn_grid_row(uint32_t x, uint32_t y, uint32_t z, uint32_t mask, uint32_t dense_res) {
    if (dense_res) {
        const uint32_t r2 = dense_res * dense_res, size = (r2 * dense_res + 7u) & ~7u;
        return (x + y * dense_res + z * r2) % size;
    }
    return (x ^ (y * 2654435761u) ^ (z * 805459861u)) & mask;
}

// builds one level in bilinear-coefficient form: one thread per grid point, entry i <-> (x, y, z) = (i % R, i / R % R, i / R^2)
__global__ void sn_build_bc_copy_kernel(const float* __restrict__ table, float* __restrict__ dense, int level, int log2_t, uint32_t R, float scale,
                                        uint32_t dense_res) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * R * R) return;
    const uint32_t x = i % R, y = (i / R) % R, z = i / (R * R);
    const uint32_t mask = (1u << log2_t) - 1u;
    const f32x2* lv = (const f32x2*)table + ((uint64_t)level << log2_t);
    auto row = [&](uint32_t xx, uint32_t yy) {
        const f32x2 v = lv[sn_grid_row(xx, yy, z, mask, dense_res)];
        return f32x2{v.x * scale, v.y * scale};
    };
    const f32x2 v00 = row(x, y), v10 = row(x + 1u, y), v01 = row(x, y + 1u), v11 = row(x + 1u, y + 1u);
    f32x4* e = (f32x4*)dense + (uint64_t)i * 2u;
    {
#pragma clang fp contract(off)
        const f32x2 B = v10 - v00, Cc = v01 - v00, D = (v11 - v10) - Cc;  // plain IEEE subtractions (the tests rebuild them with torch)
        e[0] = f32x4{v00.x, v00.y, B.x, B.y};
        e[1] = f32x4{Cc.x, Cc.y, D.x, D.y};
    }
}

// builds D_l for one level in plain-row form: one thread per entry, entry i holds grid point (x, y, z) = (i % R, i / R % R, i / R^2)
// scale: an exact power of two applied to the copied values (the "feature scale" of the split-precision MLPs, sn_api.hip
// plan_split_scales: the consuming layer's weights carry its inverse)
__global__ void sn_build_dense_copy_kernel(const float* __restrict__ table, float* __restrict__ dense, int level, int log2_t, uint32_t R,
                                           float scale, uint32_t dense_res) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * R * R) return;
    const uint32_t x = i % R, y = (i / R) % R, z = i / (R * R);
    const uint32_t mask = (1u << log2_t) - 1u;
    const uint32_t row = sn_grid_row(x, y, z, mask, dense_res);
    const f32x2* lv = (const f32x2*)table + ((uint64_t)level << log2_t);
    const f32x2 v = lv[row];
    ((f32x2*)dense)[i] = f32x2{v.x * scale, v.y * scale};
}

// ------------------------------------------------------------------------------------------
// fp16 STORAGE of a tiny-cuda-nn grid (r04; the single-fp16 mode of the main field only, sn_main.h "single-fp16 mode")
//
// tiny-cuda-nn evaluates its grids on an fp16 copy of the parameters (`params.to(half)`); the single-fp16 mode does the same: every table
// value is rounded to fp16 ONCE (round to nearest even, subnormals kept -- torch's `.half()`), the blend itself stays fp32.  Stored as
// fp16 a row is 4 bytes, so
//   * a de-hashed level keeps QUADS: entry (x, y, z) = { v(x,y,z), v(x+1,y,z), v(x,y+1,z), v(x+1,y+1,z) }, 16 bytes -- the four corners of
//     a z slice in ONE gather, a level in TWO instead of four;
//   * a hashed level keeps its rows as they are, 4 bytes each (eight 4-byte gathers; same count as before, half the bytes).
// 62 gathers per sample instead of 84 -- the mode's MLP is cheap enough (48 MFMAs, no operand splits) that the gather path was what bound it.
// Values carry the feature scale (an exact power of two): stored = half(float(half(raw)) * scale), exact while in fp16's range (the host checks).
// ------------------------------------------------------------------------------------------
SN_DEV uint32_t sn_pack_h2_scaled(f32x2 v, float scale) {
    typedef _Float16 h2v __attribute__((ext_vector_type(2)));
    const _Float16 a = (_Float16)v.x, b = (_Float16)v.y;                       // the rounding of the mode (RNE, subnormals kept)
    const h2v r = {(_Float16)((float)a * scale), (_Float16)((float)b * scale)}; // exact: a power-of-two scale inside fp16's range
    return __builtin_bit_cast(uint32_t, r);
}

// one thread per quad entry i <-> grid point (x, y, z) = (i % R, i / R % R, i / R^2) of a copied level
__global__ void sn_build_quad_h16_kernel(const float* __restrict__ table, uint32_t* __restrict__ quads, int level, int log2_t, uint32_t R, float scale,
                                         uint32_t dense_res) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * R * R) return;
    const uint32_t x = i % R, y = (i / R) % R, z = i / (R * R);
    const uint32_t mask = (1u << log2_t) - 1u;
    const f32x2* lv = (const f32x2*)table + ((uint64_t)level << log2_t);
    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
    const u32x4q e = {sn_pack_h2_scaled(lv[sn_grid_row(x, y, z, mask, dense_res)], scale), sn_pack_h2_scaled(lv[sn_grid_row(x + 1u, y, z, mask, dense_res)], scale),
                      sn_pack_h2_scaled(lv[sn_grid_row(x, y + 1u, z, mask, dense_res)], scale),
                      sn_pack_h2_scaled(lv[sn_grid_row(x + 1u, y + 1u, z, mask, dense_res)], scale)};
    ((u32x4q*)quads)[i] = e;
}

// the rows of the levels [level0, level0 + n_levels) as they are, 4 bytes each
__global__ void sn_build_rows_h16_kernel(const float* __restrict__ table, uint32_t* __restrict__ rows, int level0, int n_levels, int log2_t, float scale) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ((uint64_t)n_levels << log2_t)) return;
    rows[i] = sn_pack_h2_scaled(((const f32x2*)table)[((uint64_t)level0 << log2_t) + i], scale);
}

// b + (a - b) w with a, b the fp16 halves C of two packed registers: two v_fma_mix_f32 (full rate; the fp16 operands are read in place,
// no conversions).  hipcc forms cvt, cvt, sub for the difference, hence the asm; its inputs come from buffer loads (s_waitcnt is the
// compiler's, which tracks asm operands -- the MFMA-result hazard of DESIGN.md "third hazard" does not apply to loads).
template <int C>
SN_DEV float sn_lerp_hh(uint32_t ra, uint32_t rb, float w) {
    float d, x;
    if (C == 0) {
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(ra), "v"(rb));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,0,1]" : "=v"(x) : "v"(d), "v"(w), "v"(rb));
    } else {
        asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(d) : "v"(ra), "v"(rb));
        asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(x) : "v"(d), "v"(w), "v"(rb));
    }
    return x;
}

// a de-hashed level from its quads: two 16-byte gathers (z, z + 1), the blend of sn_hash_blend_fast (x, then y, then z) on the fp16 corners
SN_DEV f32x2 sn_hash_level_quad_h16(__amdgpu_buffer_rsrc_t rsrc, uint32_t level_off_bytes, const float q[3], float scale, uint32_t R) {
    typedef uint32_t u32x4q __attribute__((ext_vector_type(4)));
    uint32_t f[3];
    float off[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = fmaf(scale, q[a], 0.5f);
        off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    const uint32_t R16 = R << 4, R2_16 = (R * R) << 4;   // R < 1024: 16 R^2 and both products stay inside 24 bits
    const uint32_t b = sn_mad24(f[2], R2_16, sn_mad24(f[1], R16, f[0] << 4));
    const uint32_t o_z1 = level_off_bytes + R2_16;       // the z + 1 slice: a wave-uniform stride in the scalar offset
    const u32x4q e0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)b, (int)level_off_bytes, SN_AUX_DENSE);
    const u32x4q e1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)b, (int)o_z1, SN_AUX_DENSE);
    const float ox = off[0], oy = off[1], oz = off[2];
    f32x2 out;
    {
        const float x0 = sn_lerp_hh<0>(e0.y, e0.x, ox), x1 = sn_lerp_hh<0>(e0.w, e0.z, ox), x2 = sn_lerp_hh<0>(e1.y, e1.x, ox), x3 = sn_lerp_hh<0>(e1.w, e1.z, ox);
        const float y0 = fmaf(x1 - x0, oy, x0), y1 = fmaf(x3 - x2, oy, x2);
        out.x = fmaf(y1 - y0, oz, y0);
    }
    {
        const float x0 = sn_lerp_hh<1>(e0.y, e0.x, ox), x1 = sn_lerp_hh<1>(e0.w, e0.z, ox), x2 = sn_lerp_hh<1>(e1.y, e1.x, ox), x3 = sn_lerp_hh<1>(e1.w, e1.z, ox);
        const float y0 = fmaf(x1 - x0, oy, x0), y1 = fmaf(x3 - x2, oy, x2);
        out.y = fmaf(y1 - y0, oz, y0);
    }
    return out;
}

// a hashed level from its 4-byte rows: tiny-cuda-nn's positions and xor hash (sn_hash_corners_tcnn<0>), byte offsets row * 4
SN_DEV f32x2 sn_hash_level_rows_h16(__amdgpu_buffer_rsrc_t rsrc, uint32_t level_off_bytes, const float q[3], float scale, uint32_t mask) {
    uint32_t f[3];
    float off[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = fmaf(scale, q[a], 0.5f);
        off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    const uint32_t P1 = (2654435761u & mask) << 2, P2 = (805459861u & mask) << 2, m4 = mask << 2;
    const uint32_t yf = __umul24(f[1], P1), zf = __umul24(f[2], P2), xf = f[0] << 2;
    const uint32_t yc = yf + P1, zc = zf + P2, xc = xf + 4u;
    auto row = [&](uint32_t boff) { return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)(boff & m4), (int)level_off_bytes, SN_AUX_ROW); };
    // (x, y, z + 1) ... in the pairing of the x lerps: [x+1 | x] at (y, z), (y+1, z), (y, z+1), (y+1, z+1)
    const uint32_t a_ff = row(xc ^ yf ^ zf), b_ff = row(xf ^ yf ^ zf), a_cf = row(xc ^ yc ^ zf), b_cf = row(xf ^ yc ^ zf);
    const uint32_t a_fc = row(xc ^ yf ^ zc), b_fc = row(xf ^ yf ^ zc), a_cc = row(xc ^ yc ^ zc), b_cc = row(xf ^ yc ^ zc);
    const float ox = off[0], oy = off[1], oz = off[2];
    f32x2 out;
    {
        const float x0 = sn_lerp_hh<0>(a_ff, b_ff, ox), x1 = sn_lerp_hh<0>(a_cf, b_cf, ox), x2 = sn_lerp_hh<0>(a_fc, b_fc, ox), x3 = sn_lerp_hh<0>(a_cc, b_cc, ox);
        const float y0 = fmaf(x1 - x0, oy, x0), y1 = fmaf(x3 - x2, oy, x2);
        out.x = fmaf(y1 - y0, oz, y0);
    }
    {
        const float x0 = sn_lerp_hh<1>(a_ff, b_ff, ox), x1 = sn_lerp_hh<1>(a_cf, b_cf, ox), x2 = sn_lerp_hh<1>(a_fc, b_fc, ox), x3 = sn_lerp_hh<1>(a_cc, b_cc, ox);
        const float y0 = fmaf(x1 - x0, oy, x0), y1 = fmaf(x3 - x2, oy, x2);
        out.y = fmaf(y1 - y0, oz, y0);
    }
    return out;
}

// max |x| over a buffer (bit pattern of the non-negative float, which orders like the value; NaN sorts above inf and so shows)
__global__ void sn_absmax_kernel(const float* __restrict__ x, size_t n, uint32_t* out) {
    uint32_t m = 0u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        m = max(m, __float_as_uint(x[i]) & 0x7fffffffu);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, sft));
    if ((threadIdx.x & 63) == 0) atomicMax(out, m);
}

// ARITH: 0 = the literal torch-path arithmetic, 1 = its fused-kernel form (sn_hash_corners_fast), 2 = tiny-cuda-nn grid
// semantics read from the uploaded table (sn_hash_corners_tcnn, dense / hashed decided per level at run time; `grid` must point at the
// level table), 3 = tiny-cuda-nn semantics in the fused kernels: levels [0, ND) from their de-hashed copies, the rest hashed.
// ND (ARITH 1 / 3): number of leading levels read from de-hashed copies (`dense`).
// DUMP / rec (test instrumentation behind sn_render_rays_debug; include/signerf_hip.h "SnDebugDump"): where `rec` is non-null the
// lane records, per level, the 8 words that identify what it fetched: a hashed level -> the byte offsets of its 8 rows within the
// level, nerfstudio corner order; a de-hashed level in plain-row form -> words 0..3 = byte offsets of the four 16-byte fetches within
// the buffer of copies, word 4 = 0xD0000000; in bilinear-coefficient form -> words 0, 1 = byte offsets of the z and z + 1 entries
// (32 bytes each), word 4 = 0xB0000000.  Taken from the very registers that feed the loads.
// NBC (ARITH 1, ND > 0): levels [0, NBC) of the de-hashed copies are in bilinear-coefficient form (SnDenseCopy::n_bc).
// NCACHE: levels [0, NCACHE) (all of them in bilinear-coefficient form) keep their fetches across calls in cache[] (SnBcCache above).
// R16 (the single-fp16 mode on a path that reads the uploaded fp32 table): every fetched row is rounded through fp16 first -- the same values as
// the fp16 storage above (sn_pack_h2_scaled without the scale, which these paths apply to the blended feature).
template <int L, int GROUP = 0, int ARITH = 0, int ND = -1, bool DUMP = false, int NBC = 0, int NCACHE = 0, bool R16 = false>
SN_DEV void sn_hash_encode(__amdgpu_buffer_rsrc_t rsrc, const float* scal, int log2_t, const float q[3], float* feat,
                           const SnGridLevels* grid = nullptr, const SnDenseCopy* dense = nullptr, uint32_t* rec = nullptr,
                           float plain_scale = 1.0f, SnBcCache* cache = nullptr) {
    constexpr bool FAST = ARITH != 0;
    const uint32_t mask = (1u << log2_t) - 1u;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (GROUP > 0 && l > 0 && (l % GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
        if ((ARITH == 1 || ARITH == 3) && ND > 0 && l < ND && l < 12) {  // de-hashed copy of a coarse level
            uint32_t R = dense->res[l];
            asm volatile("" : "+s"(R));  // keep the per-level strides out of the loop-invariant set (SGPR pressure, see sn_grid_dense_res)
            const __amdgpu_buffer_rsrc_t drsrc = sn_table_rsrc(dense->base, dense->bytes);
            uint32_t* lrec = DUMP && rec ? rec + 8 * l : nullptr;
            const f32x2 e = l < NBC ? (l < NCACHE ? sn_hash_level_dense_bc_cached<ARITH == 3>(drsrc, dense->off[l], q, scal[l], R, cache[l < NCACHE ? l : 0], lrec)
                                                  : sn_hash_level_dense_bc<ARITH == 3>(drsrc, dense->off[l], q, scal[l], R, lrec))
                                    : sn_hash_level_dense_copy<ARITH == 3>(drsrc, dense->off[l], q, scal[l], R, lrec);
            if (DUMP && rec) {
                if (l < NBC) rec[8 * l + 2] = rec[8 * l + 3] = 0u;
                rec[8 * l + 4] = l < NBC ? 0xB0000000u : 0xD0000000u;
                rec[8 * l + 5] = rec[8 * l + 6] = rec[8 * l + 7] = 0u;
            }
            feat[2 * l] = e.x;
            feat[2 * l + 1] = e.y;
            continue;
        }
        SnHashLevel hl;
        if (ARITH == 3) sn_hash_corners_tcnn<0>(q, scal[l], mask, 0u, hl);  // beyond the copies every level is hashed (the host checks)
        else if (ARITH == 2) sn_hash_corners_tcnn<-1>(q, scal[l], mask, sn_grid_dense_res(*grid, l), hl);
        else if (ARITH == 1) sn_hash_corners_fast(q, scal[l], mask, hl);
        else sn_hash_corners(q, scal[l], mask, hl);
        const uint32_t lvl = ((uint32_t)l << log2_t) * 8u;
        f32x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sn_table_load(rsrc, hl.boff[k], lvl);
        if (R16) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = f32x2{(float)(_Float16)v[k].x, (float)(_Float16)v[k].y};
        }
        if (DUMP && rec) {
#pragma unroll
            for (int k = 0; k < 8; ++k) rec[8 * l + k] = hl.boff[k];
        }
        f32x2 e = FAST ? sn_hash_blend_fast(v, hl.off) : sn_hash_blend(v, hl.off);
        // fused kernels: the levels read from the uploaded table itself get the feature scale that the de-hashed copies already carry
        // (an exact power of two; the default 1.0 folds away everywhere else)
        feat[2 * l] = ARITH != 0 ? e.x * plain_scale : e.x;
        feat[2 * l + 1] = ARITH != 0 ? e.y * plain_scale : e.y;
    }
}

// ------------------------------------------------------------------------------------------
// x-paired hash tables (an MI355X data layout, not a change of arithmetic)
//
// The two x-neighbours of a voxel corner, (xf, y, z) and (xf+1, y, z), hash to rows r and r ^ m_t with m_t = 2^(t+1) - 1,
// t = number of trailing one bits of xf.  The gather path is bound by L1 tag lookups per INSTRUCTION (>= 16 per 64-lane
// gather whatever the coalescing; measured r01), so the library keeps, per level and per t, a paired copy
//     P[l][t][r] = { table[l][r], table[l][r ^ m_t] }          (16 bytes)
// and fetches both corners with ONE dwordx4 gather: 4 loads per level instead of 8, identical values.  Cost: (bitlen(scale_l)
// + 1) copies of each level.  Measured r01: +11 % on the proposal kernel (coarse levels, 84 MB per net) but NOTHING on the main
// field (1.2 GB; per-t tables lose the plain layout's x-locality, 16 consecutive x per 128-B line), so only K2 uses it.
// ------------------------------------------------------------------------------------------
struct SnPairInfo {
    uint32_t base[SN_MAX_LEVELS_DEV];  // first 16-byte entry of level l's t = 0 table
};

SN_DEV f32x4 sn_pair_load(__amdgpu_buffer_rsrc_t rsrc, uint32_t entry) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(entry * 16u), 0, SN_AUX_PAIR);
    f32x4 o;
    o.x = __uint_as_float(r.x);
    o.y = __uint_as_float(r.y);
    o.z = __uint_as_float(r.z);
    o.w = __uint_as_float(r.w);
    return o;
}

// Same result as sn_hash_encode (bit for bit), from the paired tables.  FAST: the fused-kernel arithmetic of
// sn_hash_corners_fast / sn_hash_blend_fast -- with "ceil = floor + 1" the pair's second half is always the wanted x + 1
// corner, so the per-corner selects of the literal form disappear as well.
// L0 / TCNN: levels [L0, L) only, with tiny-cuda-nn's position x = fmaf(scale, q, 0.5) -- the hashed levels of a tcnn grid
// (their x + 1 corner is row ^ m_t exactly as in the torch grid, so the same paired tables serve them).
// DUMP / rec: as in sn_hash_encode; a level read from the paired tables records words 0..3 = the 16-byte ENTRY numbers of its four
// fetches (order: y1 z1, y0 z1, y0 z0, y1 z0), word 4 = 0xA0000000 | t.
template <int L, int GROUP = 0, bool FAST = false, int L0 = 0, bool TCNN = false, bool DUMP = false>
SN_DEV void sn_hash_encode_pairs(__amdgpu_buffer_rsrc_t prsrc, const SnPairInfo& pi, const float* scal, int log2_t, const float q[3],
                                 float* feat, uint32_t* rec = nullptr) {
    const uint32_t mask = (1u << log2_t) - 1u;
#pragma unroll
    for (int l = L0; l < L; ++l) {
        if (GROUP > 0 && l > 0 && (l % GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
        uint32_t f[3], c[3];
        float off[3];
        if (FAST) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float x = TCNN ? fmaf(scal[l], q[a], 0.5f) : q[a] * scal[l];
                off[a] = __builtin_amdgcn_fractf(x);
                f[a] = (uint32_t)(int)x;
                c[a] = f[a] + 1u;
            }
        } else {
#pragma clang fp contract(off)
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float x = q[a] * scal[l];
                const float fl = floorf(x);
                off[a] = x - fl;
                f[a] = (uint32_t)(int)fl;
                c[a] = (uint32_t)(int)ceilf(x);
            }
        }
        const uint32_t P1 = 2654435761u & mask, P2 = 805459861u & mask;  // see sn_hash_corners
        const uint32_t yf = __umul24(f[1], P1), zf = __umul24(f[2], P2);
        const uint32_t yc = FAST ? yf + P1 : __umul24(c[1], P1), zc = FAST ? zf + P2 : __umul24(c[2], P2);
        const uint32_t t = (uint32_t)__builtin_ctz(~f[0]);  // trailing ones of xf (xf < 2^31, so ~xf != 0)
        const uint32_t base = pi.base[l] + (t << log2_t);
        // pair k: .xy = floor-x corner, .zw = ceil-x corner of (y?, z?)
        const uint32_t e_cc = base + ((f[0] ^ yc ^ zc) & mask), e_fc = base + ((f[0] ^ yf ^ zc) & mask);
        const uint32_t e_ff = base + ((f[0] ^ yf ^ zf) & mask), e_cf = base + ((f[0] ^ yc ^ zf) & mask);
        if (DUMP && rec) {
            rec[8 * l + 0] = e_cc;
            rec[8 * l + 1] = e_fc;
            rec[8 * l + 2] = e_ff;
            rec[8 * l + 3] = e_cf;
            rec[8 * l + 4] = 0xA0000000u | t;
            rec[8 * l + 5] = rec[8 * l + 6] = rec[8 * l + 7] = 0u;
        }
        const f32x4 p_cc = sn_pair_load(prsrc, e_cc);  // corners 3 (fcc), 0 (ccc)
        const f32x4 p_fc = sn_pair_load(prsrc, e_fc);  // corners 2 (ffc), 1 (cfc)
        const f32x4 p_ff = sn_pair_load(prsrc, e_ff);  // corners 6 (fff), 5 (cff)
        const f32x4 p_cf = sn_pair_load(prsrc, e_cf);  // corners 7 (fcf), 4 (ccf)
        const bool same = !FAST && c[0] == f[0];  // x on a grid plane: the ceil corner IS the floor corner
        f32x2 v[8];
        v[3] = f32x2{p_cc.x, p_cc.y};
        v[0] = same ? v[3] : f32x2{p_cc.z, p_cc.w};
        v[2] = f32x2{p_fc.x, p_fc.y};
        v[1] = same ? v[2] : f32x2{p_fc.z, p_fc.w};
        v[6] = f32x2{p_ff.x, p_ff.y};
        v[5] = same ? v[6] : f32x2{p_ff.z, p_ff.w};
        v[7] = f32x2{p_cf.x, p_cf.y};
        v[4] = same ? v[7] : f32x2{p_cf.z, p_cf.w};
        const f32x2 e = FAST ? sn_hash_blend_fast(v, off) : sn_hash_blend(v, off);
        feat[2 * l] = e.x;
        feat[2 * l + 1] = e.y;
    }
}

// builds P[l][t][r] for one table; grid over (entry r, slot = level-local table index), launched per level
__global__ void sn_build_pairs_kernel(const float* __restrict__ table, float* __restrict__ pairs, int level, int log2_t, uint32_t base,
                                      int n_t, float scale) {
    const uint32_t T = 1u << log2_t;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)T * n_t) return;
    const uint32_t t = (uint32_t)(i >> log2_t), r = (uint32_t)i & (T - 1u);
    const uint32_t m = ((2u << t) - 1u) & (T - 1u);
    const f32x2* lv = (const f32x2*)table + ((uint64_t)level << log2_t);
    const f32x2 a = lv[r], b = lv[r ^ m];
    ((f32x4*)pairs)[(uint64_t)base + i] = f32x4{a.x * scale, a.y * scale, b.x * scale, b.y * scale};
}

// fp16 storage, hashed levels as x-PAIRS (r04): P16[l][t][r] = { half2 table[l][r], half2 table[l][r ^ m_t] }, 8 bytes -- the x-paired
// tables above on the fp16 rows of sn_pack_h2_scaled: one 8-byte gather per corner pair, four per level instead of eight.
__global__ void sn_build_pairs_h16_kernel(const float* __restrict__ table, uint32_t* __restrict__ pairs, int level, int log2_t, uint32_t base,
                                          int n_t, float scale) {
    const uint32_t T = 1u << log2_t;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)T * n_t) return;
    const uint32_t t = (uint32_t)(i >> log2_t), r = (uint32_t)i & (T - 1u);
    const uint32_t m = ((2u << t) - 1u) & (T - 1u);
    const f32x2* lv = (const f32x2*)table + ((uint64_t)level << log2_t);
    typedef uint32_t u32x2p __attribute__((ext_vector_type(2)));
    ((u32x2p*)pairs)[(uint64_t)base + i] = u32x2p{sn_pack_h2_scaled(lv[r], scale), sn_pack_h2_scaled(lv[r ^ m], scale)};
}

// a hashed level of a tiny-cuda-nn grid from its fp16 x-pairs: entry numbers as in sn_hash_encode_pairs (8-byte entries), the blend of
// sn_hash_level_rows_h16
SN_DEV f32x2 sn_hash_level_pairs_h16(__amdgpu_buffer_rsrc_t prsrc, uint32_t base_entry, const float q[3], float scale, uint32_t mask, int log2_t) {
    typedef uint32_t u32x2p __attribute__((ext_vector_type(2)));
    uint32_t f[3];
    float off[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float x = fmaf(scale, q[a], 0.5f);
        off[a] = __builtin_amdgcn_fractf(x);
        f[a] = (uint32_t)(int)x;
    }
    const uint32_t P1 = 2654435761u & mask, P2 = 805459861u & mask;
    const uint32_t yf = __umul24(f[1], P1), zf = __umul24(f[2], P2), yc = yf + P1, zc = zf + P2;
    const uint32_t t = (uint32_t)__builtin_ctz(~f[0]);  // trailing ones of xf
    const uint32_t base = base_entry + (t << log2_t);
    auto pair = [&](uint32_t h) { return (u32x2p)__builtin_amdgcn_raw_buffer_load_b64(prsrc, (int)((base + ((f[0] ^ h) & mask)) << 3), 0, SN_AUX_PAIR); };
    const u32x2p p_ff = pair(yf ^ zf), p_cf = pair(yc ^ zf), p_fc = pair(yf ^ zc), p_cc = pair(yc ^ zc);   // .x = floor-x corner, .y = ceil-x corner
    const float ox = off[0], oy = off[1], oz = off[2];
    f32x2 out;
    {
        const float x0 = sn_lerp_hh<0>(p_ff.y, p_ff.x, ox), x1 = sn_lerp_hh<0>(p_cf.y, p_cf.x, ox), x2 = sn_lerp_hh<0>(p_fc.y, p_fc.x, ox), x3 = sn_lerp_hh<0>(p_cc.y, p_cc.x, ox);
        const float y0 = fmaf(x1 - x0, oy, x0), y1 = fmaf(x3 - x2, oy, x2);
        out.x = fmaf(y1 - y0, oz, y0);
    }
    {
        const float x0 = sn_lerp_hh<1>(p_ff.y, p_ff.x, ox), x1 = sn_lerp_hh<1>(p_cf.y, p_cf.x, ox), x2 = sn_lerp_hh<1>(p_fc.y, p_fc.x, ox), x3 = sn_lerp_hh<1>(p_cc.y, p_cc.x, ox);
        const float y0 = fmaf(x1 - x0, oy, x0), y1 = fmaf(x3 - x2, oy, x2);
        out.y = fmaf(y1 - y0, oz, y0);
    }
    return out;
}

// all L levels of a tiny-cuda-nn grid from its fp16 storage: levels [0, ND) from the quads (`quads`: SnDenseCopy with n_bc = 0 and 16-byte
// entries), the rest from their x-pairs (PAIRS: `rows_rsrc` = the paired tables, `pi.base[l]` = first 8-byte entry of level l) or from the
// 4-byte rows (`rows_rsrc` = the levels [ND, L) back to back)
template <int L, int ND, int GROUP, bool PAIRS>
SN_DEV void sn_hash_encode_h16(const SnDenseCopy* quads, __amdgpu_buffer_rsrc_t rows_rsrc, const SnPairInfo& pi, const float* scal, int log2_t,
                               const float q[3], float* feat) {
    const uint32_t mask = (1u << log2_t) - 1u;
    const __amdgpu_buffer_rsrc_t qrsrc = sn_table_rsrc(quads->base, quads->bytes);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (GROUP > 0 && l > 0 && (l % GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
        f32x2 e;
        if (l < ND) {
            uint32_t R = quads->res[l < 12 ? l : 0];
            asm volatile("" : "+s"(R));
            e = sn_hash_level_quad_h16(qrsrc, quads->off[l < 12 ? l : 0], q, scal[l], R);
        } else if (PAIRS) {
            e = sn_hash_level_pairs_h16(rows_rsrc, pi.base[l], q, scal[l], mask, log2_t);
        } else {
            e = sn_hash_level_rows_h16(rows_rsrc, ((uint32_t)(l - ND) << log2_t) * 4u, q, scal[l], mask);
        }
        feat[2 * l] = e.x;
        feat[2 * l + 1] = e.y;
    }
}

// ------------------------------------------------------------------------------------------
// SH basis, degree 4 (A13)
// ------------------------------------------------------------------------------------------
SN_DEV void sn_sh16(float x, float y, float z, float* c) {
    float xx = x * x, yy = y * y, zz = z * z;
    c[0] = 0.28209479177387814f;
    c[1] = 0.4886025119029199f * y;
    c[2] = 0.4886025119029199f * z;
    c[3] = 0.4886025119029199f * x;
    c[4] = 1.0925484305920792f * x * y;
    c[5] = 1.0925484305920792f * y * z;
    c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
    c[7] = 1.0925484305920792f * x * z;
    c[8] = 0.5462742152960396f * (xx - yy);
    c[9] = 0.5900435899266435f * y * (3.0f * xx - yy);
    c[10] = 2.890611442640554f * x * y * z;
    c[11] = 0.4570457994644658f * y * (5.0f * zz - 1.0f);
    c[12] = 0.3731763325901154f * z * (5.0f * zz - 3.0f);
    c[13] = 0.4570457994644658f * x * (5.0f * zz - 1.0f);
    c[14] = 1.445305721320277f * z * (xx - yy);
    c[15] = 0.5900435899266435f * x * (xx - 3.0f * yy);
}

// direction -> the 16 SH inputs the colour MLP sees.  remap 0: basis evaluated on (d+1)/2 (torch
// fallback); remap 1: on ((d+1)/2)*2-1 (tinycudann).
SN_DEV void sn_direction_encoding(const float d[3], int remap, float* c) {
    float e[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        e[a] = (d[a] + 1.0f) / 2.0f;
        if (remap) e[a] = e[a] * 2.0f - 1.0f;
    }
    sn_sh16(e[0], e[1], e[2], c);
}

// ------------------------------------------------------------------------------------------
// per-ray compositing state (A10 + A17), one ray per lane, samples visited front to back
// exp for the fused kernels: v_exp_f32(x * log2 e), ~2 ulp plus |x| * 2^-24 relative (the libm form costs ~11 instructions;
// a fused main-kernel sample evaluates 6 of them).  The staged entry points keep expf.
template <bool FAST>
SN_DEV float sn_exp(float x) {
    return FAST ? __builtin_amdgcn_exp2f(x * 1.44269504088896341f) : expf(x);
}

// ------------------------------------------------------------------------------------------
struct SnComposite {
    double cum_tau;   // cumsum(delta*density) -- torch-CPU cumsum accumulates fp32 data in fp64
    double cum_w;     // cumsum(weights) for the median search
    float sum_w;      // accumulation
    float sum_wd;     // sum(w * mid) for the expected depth
    float c[3];       // sum(w * rgb)
    float median;     // mid-point of the first sample with cum_w >= 0.5
    int median_idx;
    bool found;

    SN_DEV void init() {
        cum_tau = 0.0;
        cum_w = 0.0;
        sum_w = 0.f;
        sum_wd = 0.f;
        c[0] = c[1] = c[2] = 0.f;
        median = 0.f;
        median_idx = 0;
        found = false;
        below = 0.f;
        last_trans = 1.0f;
    }

    // One sample.  Returns the weight.
    template <bool FAST = false>
    SN_DEV float step(int i, float start, float end, float density, float r, float g, float b) {
        float w, mid;
        {
#pragma clang fp contract(off)
            float delta = end - start;
            float tau = delta * density;
            float alpha = 1.0f - sn_exp<FAST>(-tau);
            float trans = sn_exp<FAST>(-(float)cum_tau);
            w = alpha * trans;
            if (w != w) w = 0.0f;  // nan_to_num
            cum_tau += (double)tau;
            mid = (start + end) / 2.0f;
            cum_w += (double)w;
        }
        if (!found && (float)cum_w >= 0.5f) {
            found = true;
            median = mid;
            median_idx = i;
        }
        // RGBRenderer (eval): torch.nan_to_num on the per-sample colours
        if (r != r) r = 0.0f;
        if (g != g) g = 0.0f;
        if (b != b) b = 0.0f;
        sum_w += w;
        sum_wd += w * mid;
        c[0] += w * r;
        c[1] += w * g;
        c[2] += w * b;
        return w;
    }

    // The main kernel's form of step<true>: the same sums without per-step selects (see sn_sample_q_fast).  Requires what that kernel
    // guarantees: delta, density >= 0 or NaN, so w >= 0 or NaN and max(w, 0) IS nan_to_num(w); colours in [0, 1] or NaN, likewise.
    // The median is not tracked but COUNTED: cumsum(w) is non-decreasing, so the first index with cum_w >= 0.5 equals the number of
    // steps with cum_w < 0.5; `below` accumulates clamp(2^100 (0.5 - cum_w), 0, 1), which is exactly 1 or 0 (|0.5 - cum_w| is 0 or
    // >= 2^-25).  The caller turns the count into the index and re-reads that bin (median_index(), sn_main.h).
    float below;
    float last_trans;  // transmittance in front of the sample step_fused composited last (exact early termination, sn_main.h)
    SN_DEV void step_fused(float start, float end, float density, float r, float g, float b) {
        float w, mid;
        {
#pragma clang fp contract(off)
            const float delta = end - start;
            const float tau = delta * density;
            const float alpha = 1.0f - sn_exp<true>(-tau);
            const float trans = sn_exp<true>(-(float)cum_tau);
            last_trans = trans;
            w = fmaxf(alpha * trans, 0.0f);
            cum_tau += (double)tau;
            mid = (start + end) / 2.0f;
            cum_w += (double)w;
        }
        below += __builtin_amdgcn_fmed3f(fmaf((float)cum_w, -0x1p100f, 0x1p99f), 0.0f, 1.0f);
        r = fmaxf(r, 0.0f);
        g = fmaxf(g, 0.0f);
        b = fmaxf(b, 0.0f);
        sum_w += w;
        sum_wd += w * mid;
        c[0] += w * r;
        c[1] += w * g;
        c[2] += w * b;
    }
    SN_DEV int median_index(int n) const { return min((int)below, n - 1); }
    // finish() for step_fused: the caller supplies the median sample's mid-point
    SN_DEV void finish_fused(int n, float median_mid, float r, float g, float b, float out_rgb[3], float& depth, float& acc, float& exp_raw) {
        found = true;
        median = median_mid;
        median_idx = median_index(n);
        finish(n, median_mid, r, g, b, out_rgb, depth, acc, exp_raw);
    }

    // After the last sample (index n-1, mid-point last_mid, colour r,g,b = 'last_sample' background).
    SN_DEV void finish(int n, float last_mid, float r, float g, float b, float out_rgb[3], float& depth, float& acc, float& exp_raw) {
        if (r != r) r = 0.0f;
        if (g != g) g = 0.0f;
        if (b != b) b = 0.0f;
        if (!found) {
            median = last_mid;
            median_idx = n - 1;
        }
        float bgw = 1.0f - sum_w;
        out_rgb[0] = fminf(fmaxf(c[0] + r * bgw, 0.0f), 1.0f);
        out_rgb[1] = fminf(fmaxf(c[1] + g * bgw, 0.0f), 1.0f);
        out_rgb[2] = fminf(fmaxf(c[2] + b * bgw, 0.0f), 1.0f);
        depth = median;
        acc = sum_w;
        exp_raw = sum_wd / (sum_w + 1e-10f);
    }
};

// order-preserving float <-> uint map for atomicMin/atomicMax on floats
SN_DEV uint32_t sn_float_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
SN_DEV float sn_ordered_float(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// ReLU as ONE integer max on the bit pattern (hipcc turns fmaxf(x, 0) on an MFMA result into TWO v_max_f32, a canonicalising
// one first, and folds v_med3(x, 0, inf) back into the same pair): negative floats are negative ints, -0.0 is INT_MIN, positive
// floats keep their bits.  NaN handling is restored separately where it matters (sn_main.h).
SN_DEV float sn_relu(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// Exchange halves: afterwards a = [lanes 0-31: own a | lanes 32-63: lower partner's b],
//                              b = [lanes 0-31: upper partner's a | lanes 32-63: own b].
// gfx950 hazard found on hardware (r01, tools/determinism_probe.py): v_permlane32_swap must not read a VGPR in the wait states
// right after a VALU instruction wrote it.  hipcc (ROCm 7.2) places the swap directly behind its producer; when the two issue
// back to back the swap sees the register's PREVIOUS contents in lanes 48-63 (the last 16-lane pass of the write) -- observed
// as ~5 of 10 000 tiles per frame with colour off by ~1e-3, different tiles every run, only in the operand built for the
// upper half-wave.  One wait state removed every failure under an amplified test (idle issue slots forced with s_nop in the
// other phases: ~300 bad tiles per frame without it, 0 with it); two are used.  The nop is tied to both operands so it sits
// between their producers and the swap whatever the scheduler does.
SN_DEV void sn_swap_halves(float& a, float& b) {
    asm volatile("s_nop 1" : "+v"(a), "+v"(b));
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}
