// sn_proposal.h -- the proposal-sampler kernel (rows a8-a12 of SURVEY.md §8(a)).
//
// Mapping (DESIGN.md "Kernel K2"): the SAME mapping as the main kernel -- lane = ray, wave = 8x8 pixel tile, every lane
// marching its own ray front to back.
//   * r01 measured the first design (one ray per wave, lanes spread ALONG the ray, wave scans + LDS binary search) to be
//     L1-tag-lookup bound: 19 cache lines per gather instruction (one per voxel the ray crosses), 24.96 G TCP accesses =
//     the kernel's cycle count.  With lanes ACROSS an 8x8 tile at equal depth a gather touches 1-2 lines at proposal
//     resolutions (<= 256^3).
//   * marching order = prefix order, so transmittance, CDF and median are running sums (fp64, mirroring torch-CPU cumsum):
//     no scans, no shuffles;
//   * per-ray weights are staged through an L2/HBM scratch in [sample][lane] order (coalesced 256-B rows; 2 KB per ray and
//     level, read back once) -- a 64 KB-per-wave LDS copy would cap occupancy at 2 waves per CU;
//   * inverse-CDF resampling is a per-lane merge of the sorted u grid with the running CDF (searchsorted(right) == "all u in
//     [cdf_i, cdf_{i+1})");
//   * the tiny density MLP (10->16->1, 352 FLOP) runs on the matrix cores in split precision (sn_prop_mlp_mfma).
// Waves are persistent over tiles (grid = what the chip holds) so the scratch is per wave, not per tile.  The final
// sample bins are written in [tile][bin][lane=ray] order, exactly the order in which sn_render_main_kernel<1> reads them.
#pragma once
#include "../../include/signerf_hip.h"
#include "sn_device.h"
#include "sn_main.h"  // fp16 hi+lo split helpers (sn_split2, f16x8)

#define SN_PROP_MAX_SAMPLES 256
#define SN_PROP_WAVES 4
#ifndef SN_PROP_WG_PER_CU
#define SN_PROP_WG_PER_CU 3  // = waves per SIMD the kernel is compiled for (<= 168 VGPRs)
#endif
// leading levels of a proposal net whose bilinear coefficients are kept in registers across the steps of the marching loop
// (sn_device.h SnBcCache; 16 VGPRs per level).  Same-box A/B on the 1080p nerfacto frame (r02, tools/ab_lib.sh, 4 rounds of 40 frames;
// K2 is ~57 % of the frame): no cache at 5 waves/SIMD (the previous configuration) 15.13 ms, no cache at 3 waves 15.38; 2 levels at
// 4 waves 14.75; 3 / 4 / 5 levels at 3 waves 14.53 / 14.52 / 14.54 (5 levels spill).
#ifndef SN_PROP_CACHE
#define SN_PROP_CACHE 4
#endif

// proposal-net MLP pack (floats): W0 [k=10][n=16] (k-major), b0 [16], W1 [16], b1
#define SN_PROP_W0 0
#define SN_PROP_B0 160
#define SN_PROP_W1 176
#define SN_PROP_B1 192
// ... followed by the matrix-core form of the same weights (SN_PROP_MFMA): two A operands of v_mfma_f32_32x32x16_f16 as fp16 hi / lo
// parts, [lane][8 halves] each (sn_prop_mlp_mfma below: rows 0..15 serve the rays of lanes 0..31 through k = 0..7, rows 16..31 the rays
// of lanes 32..63 through k = 8..15), and the layer-2 weights in accumulator order, [h][r] = W1[(r & 3) + 8 (r >> 2) + 4 h], r = 0..7
#define SN_PROP_MA1_HI 196
#define SN_PROP_MA1_LO 452
#define SN_PROP_MA2_HI 708
#define SN_PROP_MA2_LO 964
#define SN_PROP_MW1 1220
// ... and the linear half of layer 2 (sn_prop_mlp_mfma): [k < 10] = sum_r W1[r] W0[r][k] / 2 (per unit of the SCALED features), [10] =
// sum_r W1[r] b0[r] / 2 + b1
#define SN_PROP_LIN 1236
#define SN_PROP_PACK_FLOATS 1252

struct SnScal5 {
    float v[5];
};

SN_DEV void sn_swap_halves_u(uint32_t& a, uint32_t& b) {
    asm volatile("s_nop 1" : "+v"(a), "+v"(b));
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

// The 10 -> 16 -> 1 MLP of the wave's 64 samples on the matrix cores, split precision (sn_main.h): features as fp16 hi + lo, products
// hi.hi + hi.lo + lo.hi with fp32 accumulation.
// The layer has 16 hidden units and a 32x32x16 tile has 32 rows, so ONE tile serves all 64 rays with every lane feeding its OWN
// features -- no cross-lane traffic in front of the MFMAs (r02; r01 built two 32-ray tiles with 8 v_permlane32_swap, a half-rate
// instruction that also needs wait states in front, sn_swap_halves):
//     B[k = 8 h + e][column j] = the e-th operand of the ray in lane j + 32 h        (h = lane >> 5: just what the lane holds)
//     A[row i][k]              = W[i][k] for i < 16, k < 8;   W[i - 16][k - 8] for i >= 16, k >= 8;   0 elsewhere
//     => D[i][j] = hidden unit i of ray j (i < 16), hidden unit i - 16 of ray j + 32 (i >= 16).
// Two such k-steps: operands (feature 0..7), then (feature 8, feature 9, 1.0 for the bias, 0 ...).  6 MFMAs as before, one
// accumulator tile instead of two.  Layer 2: a lane holds 8 hidden units of ray j (accumulator registers 0..7) and 8 of ray j + 32
// (registers 8..15); one swap of the two partial sums completes both rays.
SN_DEV float sn_prop_mlp_mfma(const float* __restrict__ w, const float* feat, int lane) {
    u32x4 xh, xl, yh = {0u, 0x00003c00u /* (1.0h, 0) -> the bias slot e = 2 */, 0u, 0u}, yl = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        uint32_t h, l;
        sn_split2(feat[2 * m], feat[2 * m + 1], h, l);
        xh[m] = h;
        xl[m] = l;
    }
    {
        uint32_t h, l;
        sn_split2(feat[8], feat[9], h, l);
        yh[0] = h;
        yl[0] = l;
    }
    const f16x8 a1h = __builtin_bit_cast(f16x8, *(const u32x4*)(w + SN_PROP_MA1_HI + lane * 4));
    const f16x8 a1l = __builtin_bit_cast(f16x8, *(const u32x4*)(w + SN_PROP_MA1_LO + lane * 4));
    const f16x8 a2h = __builtin_bit_cast(f16x8, *(const u32x4*)(w + SN_PROP_MA2_HI + lane * 4));
    const f16x8 a2l = __builtin_bit_cast(f16x8, *(const u32x4*)(w + SN_PROP_MA2_LO + lane * 4));
    const f16x8 b1h = __builtin_bit_cast(f16x8, xh), b1l = __builtin_bit_cast(f16x8, xl);
    const f16x8 b2h = __builtin_bit_cast(f16x8, yh), b2l = __builtin_bit_cast(f16x8, yl);
    f32x16 c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // small terms first
    SN_MFMA_H(c, a1l, b1h);
    SN_MFMA_H(c, a2l, b2h);
    SN_MFMA_H(c, a1h, b1l);
    SN_MFMA_H(c, a2h, b2l);
    SN_MFMA_H(c, a1h, b1h);
    SN_MFMA_H(c, a2h, b2h);
    // layer 2: accumulator register r holds hidden unit (r & 3) + 8 ((r & 7) >> 2) + 4 h of ray j (r < 8) / ray j + 32 (r >= 8).
    // w relu(x) = (w x + w |x|) / 2: the |x| half is ONE fma per unit (the absolute value is a source modifier) instead of a max and an
    // fma; the linear half, sum_r w_r x_r, is a linear function of the ray's 10 features and is evaluated as such by the ray's own lane
    // (10 fma, coefficients folded on the host): 26 VALU instead of 32.
    const f32x4* w1 = (const f32x4*)(w + SN_PROP_MW1 + (lane >> 5) * 8);
    const f32x4 wa = w1[0], wb = w1[1];
    float p0 = 0.0f, p1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const float wv = r < 4 ? wa[r] : wb[r - 4];
        p0 = fmaf(wv, __builtin_fabsf(c[r]), p0);
        p1 = fmaf(wv, __builtin_fabsf(c[8 + r]), p1);
    }
    const f32x4* lw = (const f32x4*)(w + SN_PROP_LIN);
    const f32x4 l0 = lw[0], l1 = lw[1], l2 = lw[2];
    float lin = l2[2];
#pragma unroll
    for (int k = 0; k < 10; ++k) lin = fmaf(k < 4 ? l0[k] : (k < 8 ? l1[k - 4] : l2[k - 8]), feat[k], lin);
    sn_swap_halves(p0, p1);  // lower lane: its own half of ray j + the upper lane's; upper lane: both halves of ray j + 32
    return (p0 + p1) + lin;
}

// pre-activation density of one proposal net at normalised position q.
// The MLP (10 -> 16 -> 1) runs on the matrix cores in split precision (sn_prop_mlp_mfma above).  History (r01; the VALU form -- weights
// broadcast from LDS, two hidden units per v_pk_fma_f32 -- left the product in r05, tools/patches/): an exact-fp32 version (v_mfma_f32_32x32x2_f32,
// 11 MFMAs x 64 cycles) was 9 % SLOWER than the VALU form -- the f32-input MFMA runs at the vector rate and does not overlap with VALU
// work at all (re-measured r02, tools/probes/overlap2_probe.hip), so its cycles simply replace VALU cycles; the fp16 hi+lo form needs
// 6 MFMAs x 32 cycles for the whole layer (K = 16 in one k-step), which plain VALU instructions do overlap with, and measured 4 %
// faster on the 1080p nerfacto frame (17.5 -> 16.8 ms, same box).
// GRID = 1: tiny-cuda-nn grid semantics (positions scale q + 0.5; `grid` = the level table).  ND > 0 (both grids): levels [0, ND) from
// their de-hashed copies (`dense`), the rest from the x-paired tables `prsrc`; ND = -1: torch grid -> everything from the paired
// tables, tcnn grid -> everything from the uploaded table `plain` with the dense / hashed decision per level at run time (its values
// then take the feature scale here: `plain_scale`).
template <int GRID = 0, int ND = -1, bool DUMP = false, int NCACHE = 0>
SN_DEV float sn_prop_h0(__amdgpu_buffer_rsrc_t prsrc, const SnPairInfo& pi, const SnScal5& scal, int log2_t, const float* __restrict__ w,
                        const float qin[3], const SnGridLevels* grid = nullptr, __amdgpu_buffer_rsrc_t plain = __amdgpu_buffer_rsrc_t(),
                        const SnDenseCopy* dense = nullptr, uint32_t* rec = nullptr, float plain_scale = 1.0f, SnBcCache* cache = nullptr,
                        const float* nanq_in = nullptr) {
    // A NaN position (a ray that misses render_aabb / the viewer's crop box carries the 1e10 sentinel: its samples overflow to inf * 0)
    // is NaN through the reference's whole field.  HERE it must not reach the matrix cores: sn_prop_mlp_mfma serves the rays of lanes j
    // and j + 32 with ONE tile whose other half is multiplied by zero weights, and 0 * NaN = NaN would hand the partner lane's NaN to a
    // healthy ray (r03, found by tests/test_gpu_random_parity.py: hit rays next to missing ones lost all their proposal weights and were
    // resampled uniformly).  So the features of a NaN lane are computed at q = 0 (finite) and its NaN is restored on the result --
    // q * 0 is +-0 for a finite position and NaN for a NaN one.
    // The marching loop hands over a NaN-free q and the NaN separately (sn_sample_q_fast<true>: no instruction spent); the stage-level
    // entry point passes the raw position and pays three v_max here (max(NaN, 0) = 0; q >= 0 otherwise).
    float q[3] = {qin[0], qin[1], qin[2]};
    float nanq;
    if (nanq_in) {
        nanq = *nanq_in;
    } else {
        nanq = fmaf(qin[2], 0.0f, fmaf(qin[1], 0.0f, qin[0] * 0.0f));
#pragma unroll
        for (int c = 0; c < 3; ++c) q[c] = __builtin_fmaxf(qin[c], 0.0f);
    }
    float feat[10];
    if (ND > 0) {
        constexpr int NBCP = ND > SN_BC_PROP ? SN_BC_PROP : ND;
        sn_hash_encode<(ND > 0 ? ND : 1), 0, (GRID ? 3 : 1), ND, DUMP, NBCP, (NCACHE < NBCP ? NCACHE : NBCP)>(plain, scal.v, log2_t, q, feat, grid, dense, rec,
                                                                                                            1.0f, cache);
        if (ND < 5) sn_hash_encode_pairs<5, 0, true, (ND > 0 && ND < 5 ? ND : 0), GRID == 1, DUMP>(prsrc, pi, scal.v, log2_t, q, feat, rec);
    } else if (GRID == 1) {
        sn_hash_encode<5, 0, 2, -1>(plain, scal.v, log2_t, q, feat, grid, nullptr, nullptr, plain_scale);
    } else {
        sn_hash_encode_pairs<5, 0, true, 0, false, DUMP>(prsrc, pi, scal.v, log2_t, q, feat, rec);
    }
    float out = sn_prop_mlp_mfma(w, feat, (int)(threadIdx.x & 63));
    // the reference's field is NaN all the way for a NaN position (and a v_max-based ReLU would launder it anyway): restored by arithmetic
    // (+-0 or NaN), not by a select -- sn_sample_q_fast explains why
    return out + nanq;
}

// ---- PDFSampler, eval mode (A11), one ray per lane --------------------------------------------------------------------
// Streams the N weights of this lane's ray from `w` (stride 64 floats: [sample][lane]) and emits the M+1 new spacing bins
// through `emit(j, bin, idx)`.  sb(i) returns the existing spacing bin i (0..N).  u: [M+1] grid (LDS or global).
// searchsorted(cdf, u, side="right") = number of knots <= u, so u_j belongs to interval i iff cdf_i <= u_j < cdf_{i+1}.
struct SnPdfNorm {
    float padding, denom;
    // weights_sum = sum(w + pad); padding = relu(1e-5 - sum); denom = sum + padding; per-sample padding / N
    SN_DEV void set(double sum_wp, int N) {
#pragma clang fp contract(off)
        const float wsum = (float)sum_wp;
        const float pd = fmaxf(1e-5f - wsum, 0.0f);
        denom = wsum + pd;
        padding = pd / (float)N;
    }
};

// RECIP (the fused kernel, r02): the per-weight division num / denom has a loop-invariant denominator, so the IEEE quotient is formed
// from its correctly rounded reciprocal y = RN(1 / denom) (one IEEE division per ray) as  q0 = RN(num y);  e = num - q0 denom (exact,
// one fma);  q = RN(q0 + e y).  Markstein's theorem gives q == RN(num / denom) when q0 is a FAITHFUL rounding of the quotient and the
// significand of denom is not all ones; q0 = RN(num RN(1 / denom)) can be 1.5 ulp off when the quotient sits just below a power of two
// (ADVICE r02), so the identity is held as MEASURED, not proven: tests/test_recip_division.py emulates the three instructions exactly
// and finds 0 mismatches over random operands of the resampler's range and over quotients within 64 ulps below a power of two (r03:
// 80 M + 400 M pairs), and tests/test_gpu_render.py::test_resampler_reciprocal_division_is_bit_identical compares whole renders with
// SN_PDF_IEEE=1 (the plain divisions).  A denominator with an all-ones significand (1 in 8 M), or one outside [2^-60, 2^60] (residual
// underflow; never the case: 1e-5 <= denom <= ~260), sends the whole wave down the plain IEEE path.  3 instead of ~10 VALU per weight.
template <bool RECIP = false, typename SB, typename EMIT>
SN_DEV void sn_pdf_lane(const float* __restrict__ w, int wstride, int N, int M, const float* u, float pad, const SnPdfNorm& nm, SB sb,
                        EMIT emit, bool force_ieee = false) {
    int j = 0;
    double cum = 0.0;
    float c_prev = 0.0f, b_prev = sb(0);
    float uj = u[0];
    float inv_denom = 0.0f;
    bool recip = false;
    {
#pragma clang fp contract(off)
        if (RECIP) inv_denom = 1.0f / nm.denom;
    }
    if (RECIP) {
        const uint32_t bits = __float_as_uint(nm.denom);
        const bool unsafe = (bits & 0x7fffffu) == 0x7fffffu || !(nm.denom >= 0x1p-60f && nm.denom <= 0x1p60f);
        recip = !force_ieee && !__any(unsafe);  // wave-uniform
    }
    for (int i = 0; i < N; ++i) {
        float c_next, b_next = sb(i + 1);
        {
#pragma clang fp contract(off)
            const float num = (w[(int64_t)i * wstride] + pad) + nm.padding;
            float pdf;
            if (RECIP && recip) {
                const float q0 = num * inv_denom;
                pdf = __builtin_fmaf(__builtin_fmaf(-q0, nm.denom, num), inv_denom, q0);
            } else {
                pdf = num / nm.denom;
            }
            cum += (double)pdf;
            c_next = fminf(1.0f, (float)cum);
        }
        while (__any(j <= M && uj < c_next)) {
            if (j <= M && uj < c_next) {
                float t, v;
                {
#pragma clang fp contract(off)
                    t = (uj - c_prev) / (c_next - c_prev);
                    if (t != t) t = 0.0f;
                    t = fminf(fmaxf(t, 0.0f), 1.0f);
                    v = b_prev + t * (b_next - b_prev);
                }
                emit(j, v, i + 1);
                ++j;
                uj = u[min(j, M)];
            }
        }
        c_prev = c_next;
        b_prev = b_next;
    }
    // u_j >= cdf_N: idx = N + 1, below = above = N -> bins_g0 + t * 0 with t clipped to [0, 1]
    while (__any(j <= M)) {
        if (j <= M) {
            emit(j, b_prev, N + 1);
            ++j;
        }
    }
}

// ---- proposal kernel -----------------------------------------------------------------------------
struct SnPropParams {
    const float* origins;
    const float* directions;
    const float* nears;
    const float* fars;
    const float* sbins0;                  // [n_samples[0]+1] initial spacing bins, or null (i / n0)
    const float* pdf_u[SN_MAX_PROPOSALS]; // u grid of resampling step k, or null ((j + 0.5) / (m + 1))
    float* ebins_out;                     // [tile][n_final+1][64]
    float* prop_depth[SN_MAX_PROPOSALS];  // [H*W] or null
    // test instrumentation (DUMP = 1 instantiations only; sn_render_rays_debug)
    float feat_scale[SN_MAX_PROPOSALS];      // power-of-two feature scale of net k (carried by its copies / paired tables; applied to plain reads)
    uint32_t* dump_fetch[SN_MAX_PROPOSALS];  // [H*W][n_samples[k]][5][8] fetch records of net k, or null
    float* dump_q[SN_MAX_PROPOSALS];         // [H*W][n_samples[k]][3] hashed positions of net k, or null
    int32_t* dump_pdf[SN_MAX_PROPOSALS];     // [H*W][m_k + 1] searchsorted index of every u of resampling step k, or null
    float* scratch;                       // [n_waves][SN_PROP_SCRATCH_FLOATS]
    unsigned int* tile_counter;           // the tile queue: next tile to hand out (set to the number of waves before the launch), or null: static stride
    const float* tables[SN_MAX_PROPOSALS];     // plain tables (tiny-cuda-nn grid mode)
    uint32_t table_bytes[SN_MAX_PROPOSALS];
    SnGridLevels grid[SN_MAX_PROPOSALS];       // tcnn: dense-level resolutions; torch with de-hashed copies: their R
    SnDenseCopy dense[SN_MAX_PROPOSALS];       // torch grid, ND > 0: de-hashed copies of the leading levels
    const float* pairs[SN_MAX_PROPOSALS];  // x-paired tables (sn_device.h)
    SnPairInfo pinfo[SN_MAX_PROPOSALS];
    uint32_t pairs_bytes[SN_MAX_PROPOSALS];
    const float* wpack[SN_MAX_PROPOSALS];
    float scal[SN_MAX_PROPOSALS][5];
    int log2_t[SN_MAX_PROPOSALS];
    int n_samples[SN_MAX_PROPOSALS];
    int n_levels;  // 1 or 2
    int n_final;
    int height, width, tile_w_log2, tile_h_log2, tiles_x, tiles_y;
    float near_plane, far_plane, avg_density, hist_pad;
    int pdf_ieee;   // test switch (SN_PDF_IEEE=1): the resampler divides with the plain IEEE sequence instead of sn_pdf_lane's RECIP form
    int early_term; // exact early termination of saturated waves (sn_prop_level); 0 = off (SN_EARLY_TERM=0)
    unsigned long long* march_stats;  // SnRenderOpts.march_stats ([1 + LV]: wave-steps the early termination of level LV skipped) or null; STATS instantiation only
    int cache_off;  // test switch: the coefficient cache re-fetches on every step (tests/test_gpu_render.py compares the two bit for bit)
    int spacing_uniform;  // SnRenderOpts.spacing_mode (sn_spacing)
    SnPosMap pm;          // SnFieldDesc.disable_scene_contraction (sn_sample_q_fast)
};

// per-wave scratch: weights [256][64] + two spacing-bin arrays [257][64]
#define SN_PROP_SCRATCH_W 0
#define SN_PROP_SCRATCH_B0 (SN_PROP_MAX_SAMPLES * 64)
#define SN_PROP_SCRATCH_B1 (SN_PROP_SCRATCH_B0 + (SN_PROP_MAX_SAMPLES + 1) * 64)
#define SN_PROP_SCRATCH_FLOATS (SN_PROP_SCRATCH_B1 + (SN_PROP_MAX_SAMPLES + 1) * 64)

// shared per-workgroup LDS: the sampler grids
struct SnPropLds {
    float sb0[SN_PROP_MAX_SAMPLES + 4];
    float eb0[SN_PROP_MAX_SAMPLES + 4];  // euclidean bins of the initial sampler when the frame has no per-ray nears / fars (same for every ray)
    float u[SN_MAX_PROPOSALS][SN_PROP_MAX_SAMPLES + 4];
    // MLP weight packs.  They are wave-uniform, but the per-iteration compiler memory clobber (needed against LICM) makes
    // hipcc fetch global weights with VECTOR loads -- 49 extra TA instructions per sample, more than the 40 hash gathers
    // (measured r01: SQ_INSTS_SMEM ~ 0, VMEM reads 2.2x the expected count).  From LDS they are broadcast ds_reads.
    float wpack[SN_MAX_PROPOSALS][SN_PROP_PACK_FLOATS];
};

// One proposal level for this lane's ray: density net LV at the N samples whose spacing bins are sb(0..N); writes the
// weights to w[i * 64] and returns sum(w + pad) (fp64) and the level's median depth.
// eb_shared: LV 0 only -- the level's euclidean bins from LDS (frames without per-ray nears / fars), else null
template <int LV, int GRID, int ND, bool DUMP, bool ALT, bool STATS, typename SB>
SN_DEV void sn_prop_level(const SnPropParams& p, const float* wp, SB sb, float* __restrict__ w, int N, const float o[3], const float d[3], float s_near,
                          float s_far, double& sum_wp, float& median_out, int64_t dump_ray = -1, const float* eb_shared = nullptr) {
    const int su = ALT ? p.spacing_uniform : 0;
    const SnPosMap* pm = ALT ? &p.pm : nullptr;
    SnScal5 scal;
#pragma unroll
    for (int l = 0; l < 5; ++l) scal.v[l] = p.scal[LV][l];
    const int log2_t = p.log2_t[LV];
    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.pairs[LV], p.pairs_bytes[LV]);
    const __amdgpu_buffer_rsrc_t plain = sn_table_rsrc(p.tables[LV], p.table_bytes[LV]);
    SnPairInfo pi;
#pragma unroll
    for (int l = 0; l < 5; ++l) pi.base[l] = p.pinfo[LV].base[l];
    double cum_tau = 0.0, cum_w = 0.0, swp = 0.0;
    float below = 0.0f;  // number of steps with cumsum(w) < 0.5 = index of the median sample (SnComposite::step_fused)
    constexpr int NCACHE = ND > 0 ? (SN_PROP_CACHE < ND ? SN_PROP_CACHE : ND) : 0;
    SnBcCache cache[NCACHE > 0 ? NCACHE : 1];
#pragma unroll
    for (int l = 0; l < (NCACHE > 0 ? NCACHE : 1); ++l) cache[l].reset(p.cache_off != 0);
    float e0 = eb_shared ? eb_shared[0] : sn_euclid(sb(0), s_near, s_far, su);
#pragma unroll 1
    for (int i = 0; i < N; ++i) {
        // keep the (loop-invariant) MLP weight loads inside the loop: hoisted, they cost ~200 registers (see sn_main.h)
        asm volatile("" ::: "memory");
        const float e1 = eb_shared ? eb_shared[i + 1] : sn_euclid(sb(i + 1), s_near, s_far, su);
        float q[3], nanq;
        bool sel;
        if (ALT) {  // the strict position arithmetic (see sn_sample_q_fast), then the same NaN-free hand-over
            sel = sn_sample_q(o, d, e0, e1, q, pm);
            nanq = fmaf(q[2], 0.0f, fmaf(q[1], 0.0f, q[0] * 0.0f));
#pragma unroll
            for (int c = 0; c < 3; ++c) q[c] = __builtin_fmaxf(q[c], 0.0f);
        } else {
            sel = sn_sample_q_fast<true>(o, d, e0, e1, q, &nanq);
        }
        uint32_t* rec = nullptr;
        if (DUMP && dump_ray >= 0) {
            const size_t smp = (size_t)dump_ray * (size_t)N + (size_t)i;
            if (p.dump_fetch[LV]) rec = p.dump_fetch[LV] + smp * 40;
            if (p.dump_q[LV]) {  // the position as the reference holds it (NaN for a missing ray)
                p.dump_q[LV][smp * 3 + 0] = q[0] + nanq;
                p.dump_q[LV][smp * 3 + 1] = q[1] + nanq;
                p.dump_q[LV][smp * 3 + 2] = q[2] + nanq;
            }
        }
        const float h0 = sn_prop_h0<GRID, ND, DUMP, NCACHE>(rsrc, pi, scal, log2_t, wp, q, &p.grid[LV], plain, &p.dense[LV], rec, p.feat_scale[LV], cache,
                                                            &nanq);
        const float density = p.avg_density * sn_exp<true>(h0) * (sel ? 1.0f : 0.0f);
        float wt, trans;
        {
#pragma clang fp contract(off)
            const float tau = (e1 - e0) * density;
            trans = sn_exp<true>(-(float)cum_tau);
            wt = fmaxf((1.0f - sn_exp<true>(-tau)) * trans, 0.0f);  // nan_to_num: tau >= 0 or NaN, so wt >= 0 or NaN
            cum_tau += (double)tau;
            cum_w += (double)wt;
            swp += (double)(wt + p.hist_pad);
        }
        below += __builtin_amdgcn_fmed3f(fmaf((float)cum_w, -0x1p100f, 0x1p99f), 0.0f, 1.0f);
        w[(int64_t)i * 64] = wt;
        e0 = e1;
        // EXACT early termination (r04; sn_main.h explains why it is exact): the transmittance in front of this sample was exactly 0 for
        // every ray of the wave, so every later weight is exactly +0 -- written as such, and added to the padded sum one by one as the
        // full march does (fp64 additions of the same constant, the same roundings); cumsum(w) and the median count stay what they are.
        if (!DUMP && p.early_term && __all(trans == 0.0f)) {
            if (STATS && p.march_stats && (threadIdx.x & 63) == 0) atomicAdd(&p.march_stats[1 + LV], (unsigned long long)(N - 1 - i));
            for (int k = i + 1; k < N; ++k) {
#pragma clang fp contract(off)
                swp += (double)(0.0f + p.hist_pad);
                w[(int64_t)k * 64] = 0.0f;
            }
            break;
        }
    }
    sum_wp = swp;
    {
        const int mi = min((int)below, N - 1);
        const float ea = eb_shared ? eb_shared[mi] : sn_euclid(sb(mi), s_near, s_far, su);
        const float ec = eb_shared ? eb_shared[mi + 1] : sn_euclid(sb(mi + 1), s_near, s_far, su);
        median_out = sn_mid(ea, ec);
    }
}

// GRID 1: ND0 / ND1 = leading dense levels of the two nets (-1: run-time decision per level)
template <int GRID, int ND0 = -1, int ND1 = -1, bool DUMP = false, bool ALT = false, bool STATS = false /*diagnostics instantiation, see sn_main.h*/>
__global__ __launch_bounds__(64 * SN_PROP_WAVES, SN_PROP_WG_PER_CU) void sn_proposal_kernel(SnPropParams p) {
    __shared__ __attribute__((aligned(16))) SnPropLds L;
    const int su = ALT ? p.spacing_uniform : 0;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // sampler grids -> LDS (per-lane indexed later)
    const int n0 = p.n_samples[0];
    const bool shared_bins = p.nears == nullptr;
    for (int i = tid; i <= n0; i += 64 * SN_PROP_WAVES) {
        const float sb = p.sbins0 ? p.sbins0[i] : (float)i / (float)n0;
        L.sb0[i] = sb;
        if (shared_bins) L.eb0[i] = sn_euclid(sb, sn_spacing(p.near_plane, su), sn_spacing(p.far_plane, su), su);  // the same strict arithmetic as per lane
    }
    for (int k = 0; k < p.n_levels; ++k) {
        const int m = k + 1 < p.n_levels ? p.n_samples[k + 1] : p.n_final;
        for (int j = tid; j <= m; j += 64 * SN_PROP_WAVES) {
#pragma clang fp contract(off)
            L.u[k][j] = p.pdf_u[k] ? p.pdf_u[k][j] : ((float)j + 0.5f) / (float)(m + 1);
        }
        for (int j = tid; j < SN_PROP_PACK_FLOATS; j += 64 * SN_PROP_WAVES) L.wpack[k][j] = p.wpack[k][j];
    }
    __syncthreads();

    const int n_tiles = p.tiles_x * p.tiles_y;
    const int wave_global = blockIdx.x * SN_PROP_WAVES + wave;
    float* __restrict__ sc = p.scratch + (int64_t)wave_global * SN_PROP_SCRATCH_FLOATS + lane;
    float* __restrict__ W = sc + SN_PROP_SCRATCH_W;
    float* __restrict__ B0 = sc + SN_PROP_SCRATCH_B0;
    float* __restrict__ B1 = sc + SN_PROP_SCRATCH_B1;
    const int tw = 1 << p.tile_w_log2;

    // A wave's FIRST tile is its own index; further tiles come from a QUEUE (one atomic per wave and tile; the counter starts at the number of
    // waves), not from a fixed stride (r05).  On a trained scene the exact early termination makes a tile cost anything between 30 % and 100 %
    // of a full march (sky vs a surface in front of the camera), and with the static assignment -- wave w walks tiles w, w + n_waves, ... -- the
    // launch lasted as long as its unluckiest wave: simulated on the trained scene's analytic depth map, makespan 1.13 x the mean load at
    // 1920x1080 (10.5 tiles per wave), 1.02 x with the queue.  Measured, same box, trained scene, one stream (profiles/r05_k2_tile_queue_ab.txt):
    // 1920x1080 12.0-12.5 -> 11.7-12.0 ms (-2.5 ... -4 %), 800x800 -0.6 ... -3.5 %; random-weight scenes (every tile costs the same): unchanged.
    // Every output belongs to a tile, so the mapping changes no bit.  tile_counter == null (frames with no more tiles than waves: the host
    // does not even zero a counter): the static stride.
    const int n_waves = gridDim.x * SN_PROP_WAVES;
    int tile = wave_global;
#pragma unroll 1
    for (; tile < n_tiles;) {
        const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
        const int px = (tx << p.tile_w_log2) + (lane & (tw - 1));
        const int py = (ty << p.tile_h_log2) + (lane >> p.tile_w_log2);
        const bool valid = px < p.width && py < p.height;
        const int64_t ray = (int64_t)min(py, p.height - 1) * p.width + min(px, p.width - 1);
        float o[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = p.origins[ray * 3 + c];
            d[c] = p.directions[ray * 3 + c];
        }
        const float s_near = sn_spacing(p.nears ? p.nears[ray] : p.near_plane, su);
        const float s_far = sn_spacing(p.fars ? p.fars[ray] : p.far_plane, su);
        float* eb_tile = p.ebins_out + (int64_t)tile * (p.n_final + 1) * 64 + lane;
        const int64_t pix = (int64_t)py * p.width + px;

        // level 0: the initial (uniform in s) sampler
        double sum_wp;
        float med;
        const int64_t dump_ray = DUMP && valid ? pix : -1;
        sn_prop_level<0, GRID, ND0, DUMP, ALT, STATS>(p, L.wpack[0], [&](int i) { return L.sb0[i]; }, W, n0, o, d, s_near, s_far, sum_wp, med, dump_ray,
                                          shared_bins ? L.eb0 : nullptr);
        if (valid && p.prop_depth[0]) p.prop_depth[0][pix] = med;
        SnPdfNorm nm;
        nm.set(sum_wp, n0);
        // DUMP: the searchsorted index of every u (PDFSampler's `inds`), as the merge found it
        auto dump_idx = [&](int k, int m, int j, int idx) {
            if (DUMP && dump_ray >= 0 && p.dump_pdf[k]) p.dump_pdf[k][(size_t)dump_ray * (size_t)(m + 1) + (size_t)j] = idx;
        };
        if (p.n_levels == 1) {
            sn_pdf_lane<true>(W, 64, n0, p.n_final, L.u[0], p.hist_pad, nm, [&](int i) { return L.sb0[i]; }, [&](int j, float v, int idx) {
                eb_tile[(int64_t)j * 64] = sn_euclid(v, s_near, s_far, su);
                dump_idx(0, p.n_final, j, idx);
            }, p.pdf_ieee != 0);
        } else {
            const int n1 = p.n_samples[1];
            sn_pdf_lane<true>(W, 64, n0, n1, L.u[0], p.hist_pad, nm, [&](int i) { return L.sb0[i]; }, [&](int j, float v, int idx) {
                B0[(int64_t)j * 64] = v;
                dump_idx(0, n1, j, idx);
            }, p.pdf_ieee != 0);
            sn_prop_level<1, GRID, ND1, DUMP, ALT, STATS>(p, L.wpack[1], [&](int i) { return B0[(int64_t)i * 64]; }, W, n1, o, d, s_near, s_far, sum_wp, med, dump_ray);
            if (valid && p.prop_depth[1]) p.prop_depth[1][pix] = med;
            nm.set(sum_wp, n1);
            sn_pdf_lane<true>(W, 64, n1, p.n_final, L.u[1], p.hist_pad, nm, [&](int i) { return B0[(int64_t)i * 64]; }, [&](int j, float v, int idx) {
                eb_tile[(int64_t)j * 64] = sn_euclid(v, s_near, s_far, su);
                dump_idx(1, p.n_final, j, idx);
            }, p.pdf_ieee != 0);
        }
        (void)B1;
        if (p.tile_counter) {
            int next = 0;
            if (lane == 0) next = (int)atomicAdd(p.tile_counter, 1u);
            tile = __builtin_amdgcn_readfirstlane(next);
        } else {
            tile += n_waves;
        }
    }
}

// ---- stage kernels ---------------------------------------------------------------------------------
struct SnPropStageParams {
    const float* positions;
    int64_t n;
    const float* pairs;
    SnPairInfo pinfo;
    uint32_t pairs_bytes;
    const float* wpack;
    float scal[5];
    int log2_t;
    float avg_density;
    float* density;
    const float* table;  // plain table + level table: tiny-cuda-nn grid mode (grid_mode != 0)
    uint32_t table_bytes;
    int grid_mode;
    SnGridLevels grid;
    float feat_scale;  // tcnn grid mode: the pack's first layer carries 1 / this; the plain table's rows are multiplied by it here
    SnPosMap pm;
};

__global__ void sn_prop_field_stage_kernel(SnPropStageParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = i < p.n ? i : p.n - 1;
    const float pos[3] = {p.positions[j * 3], p.positions[j * 3 + 1], p.positions[j * 3 + 2]};
    float q[3];
    const bool sel = sn_position_q(pos, q, &p.pm);
    SnScal5 scal;
#pragma unroll
    for (int l = 0; l < 5; ++l) scal.v[l] = p.scal[l];
    float h0;
    if (p.grid_mode) h0 = sn_prop_h0<1, -1>(sn_table_rsrc(p.pairs, p.pairs_bytes), p.pinfo, scal, p.log2_t, p.wpack, q, &p.grid, sn_table_rsrc(p.table, p.table_bytes),
                                            nullptr, nullptr, p.feat_scale);
    else h0 = sn_prop_h0<0, -1>(sn_table_rsrc(p.pairs, p.pairs_bytes), p.pinfo, scal, p.log2_t, p.wpack, q);
    if (i < p.n) p.density[i] = p.avg_density * expf(h0) * (sel ? 1.0f : 0.0f);
}

struct SnPdfStageParams {
    const float* sbins;    // [R,N+1]
    const float* weights;  // [R,N]
    int64_t n_rays;
    int n_in, n_out;
    const float* u;
    float hist_pad;
    float* new_bins;  // [R,M+1]
    int32_t* inds;    // [R,M+1] or null
};

// one ray per lane, row-major inputs (test sizes only; the fused kernel streams [sample][lane] rows)
__global__ __launch_bounds__(64) void sn_pdf_stage_kernel(SnPdfStageParams p) {
    const int64_t r0 = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t r = r0 < p.n_rays ? r0 : p.n_rays - 1;
    const bool live = r0 < p.n_rays;
    const int N = p.n_in, M = p.n_out;
    const float* w = p.weights + r * N;
    const float* sb = p.sbins + r * (N + 1);
    double swp = 0.0;
    for (int i = 0; i < N; ++i) {
#pragma clang fp contract(off)
        swp += (double)(w[i] + p.hist_pad);
    }
    SnPdfNorm nm;
    nm.set(swp, N);
    sn_pdf_lane(w, 1, N, M, p.u, p.hist_pad, nm, [&](int i) { return sb[i]; }, [&](int j, float v, int idx) {
        if (live) {
            p.new_bins[r * (M + 1) + j] = v;
            if (p.inds) p.inds[r * (M + 1) + j] = idx;
        }
    });
}
