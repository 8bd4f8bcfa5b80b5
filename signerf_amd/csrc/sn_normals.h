// sn_normals.h -- analytic and predicted normals of the main field.
//
// What the reference computes when `predict_normals=True` (/root/reference/signerf/signerf_config.py:33) [NS-RECALL]:
//   normals      = -normalize(d h0 / d q)       h0 = PRE-activation density, q = the field's normalised sample location
//   pred_normals = normalize(tanh(W_h . MLP_{27->64->64->64}([posenc(p_world) (12) | geo (15)]) + b_h))
//   both rendered as  n = sum_i w_i n_i ;  n / (|n| + 1e-10) ;  (n + 1) / 2
// `DatasetGenerator.render_camera` never reads them (datasetgenerator.py:700-701), so they are a SEPARATE, lazily launched
// kernel: K1's register budget stays untouched and the generator does not pay for them.
//
// Same mapping as K1 (lane = ray, wave = 8x8 tile, one sample per lane per step), exact-fp32 MFMA MLPs in K1's register-chained
// layout (sn_main.h).  Per wave-step:
//   1. hash-encode (as K1), density MLP forward; the ReLU mask of layer 1 is kept;
//   2. reverse mode for d h0 / d feat: h0 = W2[0,:] . relu(z1) + b, so  g_feat = W1^T (W2[0,:] * [z1 > 0]).  The host folds
//      W2[0,:] into the transposed layer (image block WB), the B operand is the 0/1 mask in the layer-1 output layout, and one
//      32-row MFMA pass gives g_feat; 16 permlane32 swaps bring the 32 values of a ray into its own lane;
//   3. a second pass over the 16 levels re-gathers the corners (L1/L2 hits) and contracts the gradient of the trilinear blend
//      with g_feat:  g_q = sum_l scale_l * (g_feat[2l] * d f0 / d off + g_feat[2l+1] * d f1 / d off);
//   4. the pred-normal MLP has the shape of the colour MLP ([16 layer-2 rows | 16 per-lane inputs] -> 64 -> 64 -> 3): it runs
//      through the SAME image slots and code path as the colour MLP, with the position encoding in the SH slots and its last
//      two linear layers (64 -> 64, no activation, then the 64 -> 3 head) multiplied together on the host.
#pragma once
#include "sn_main.h"

struct SnNormImg {  // float offsets.  [0, SnMainImg::TOTAL) has SnMainImg's layout, the pred-normal MLP in the colour slots
    static constexpr int WB = SnMainImg::TOTAL;  // [rt=1][t4=8][64][4]: mask (64, layer-1 output order) -> d h0 / d feat (32 rows)
    static constexpr int ZB = WB + 2048;         // its bias image: 32 zeros
    static constexpr int TOTAL = ZB + 32;        // 12 740 floats = 50 960 B
};

// fp16x2 form: SnMainImgH (pred-normal MLP in the colour slots) followed by the reverse-pass layer in the same operand order
struct SnNormImgH {
    static constexpr int WB = SnMainImgH::TOTAL_BYTES;  // [s=4][hi|lo][lane][8 halves] = 8 KiB
    static constexpr int TOTAL_BYTES = WB + 8192;       // 50 832
};

struct SnNormalsParams {
    const float* origins;
    const float* directions;
    const float* nears;
    const float* fars;
    const float* sbins;  // uniform mode
    const float* ebins;  // bins mode: [tile][S+1][64]
    const float* table;
    const float* wimg;   // SnNormImg
    float* normals;      // [H*W,3] or null
    float* pred_normals; // [H*W,3] or null
    float scal[16];
    int height, width, n_samples;
    int tile_w_log2, tile_h_log2, tiles_x, tiles_y;
    int log2_t;
    float near_plane, far_plane, avg_density;
    SnGridLevels grid;
    SnDenseCopy dense;     // ND > 0: the main grid's de-hashed copies (r02: the normals kernel was gather-issue bound with 256 hashed gathers per step)
    float inv_feat_scale;  // ND > 0: the copies carry the power-of-two feature scale t0 of the split-precision render; the exact-fp32 image does not
    // Range conditioning of the split-precision form (PREC = 1; r03, as K1's -- sn_api.hip plan_split_scales): the fp16 hi+lo image holds
    // W1 s1 / t0, W2 s2 / s1, ... so the kernel feeds it features TIMES t0 (`feat_scale`: the copies' own values; the hashed levels are
    // multiplied), reads h0 through the image's 1 / s2 slot, and gets the reverse pass's d h0 / d feat times `grad_scale` (a power of two
    // that lifts the transposed layer W1^T diag(W2[0,:]) into fp16's normal range; the analytic normal is a direction, so the factor
    // only moves the 1e-12 clamp of F.normalize with it).  Both are 1 for the exact-fp32 form.
    float feat_scale;
    float grad_scale;
    float pe_rev_scale;  // position encoding: 1 = nerfstudio's torch NeRFEncoding, sin(2 pi x 2^k); 0.5 = tiny-cuda-nn's Frequency, sin(pi x 2^k)
    int spacing_uniform;  // SnRenderOpts.spacing_mode (sn_spacing)
    SnPosMap pm;          // SnFieldDesc.disable_scene_contraction (sn_sample_q)
};

// NeRFEncoding(in_dim 3, 2 frequencies 2^0, 2^1): [sin(2 pi x_a 2^k)] for (a, k) a-major, then the same with a pi/2 phase.
// v_sin_f32 takes its argument in revolutions, so sin(2 pi x 2^k) = v_sin(fract(x 2^k)): the power-of-two scaling and the
// range reduction are exact (libm sinf's reduction path costs ~200 spilled registers here).
// rev_scale = 0.5 gives tiny-cuda-nn's Frequency encoding, sin(pi x 2^k) and cos = the same a quarter revolution on (the scaling
// stays a power of two, hence exact); the slot ORDER is this one in both cases -- tcnn_import permutes the first layer's columns.
SN_DEV void sn_position_encoding(const float p[3], float pe[12], float rev_scale) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float rev = p[a] * ((float)(1 << k) * rev_scale);
            pe[a * 2 + k] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev));
            pe[6 + a * 2 + k] = __builtin_amdgcn_sinf(__builtin_amdgcn_fractf(rev + 0.25f));
        }
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) with v_exp_f32 / v_rcp_f32 (exp overflow -> 1, underflow -> -1)
SN_DEV float sn_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(sn_exp<true>(2.0f * x) + 1.0f); }

// Forward of the density MLP + reverse-mode d h0 / d feat + pred-normal MLP for the wave's 64 samples.
// feat[32]: this lane's hash features; pe[12]: its position encoding.  Out: this lane's h0, g_feat[32], pre-tanh pred normal x[3].
SN_DEV void sn_normals_field(const float* __restrict__ lds, const float* feat, const float* pe, int lane, float& h0, float* gfeat,
                             float x[3]) {
    const bool upper = lane >= 32;
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
    __builtin_amdgcn_sched_barrier(0);
    f32x16 g0[1], g1[1];
    sn_mlp_layer_f32<1, 32>(lds + SnMainImg::W2, lds + SnMainImg::B2, op0, op1, g0, g1, lane);
    h0 = upper ? g1[0][8] : g0[0][0];
    __builtin_amdgcn_sched_barrier(0);
    // ---- d h0 / d feat: rows rho(r) + 4h of tile 0 / tile 1, then into the owning lane ----------------------------------
    {
        // torch's ReLU backward is grad * (z > 0), and relu(z) > 0 <=> z > 0: the mask replaces the activations in place
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            op0[k] = op0[k] > 0.0f ? 1.0f : 0.0f;
            op1[k] = op1[k] > 0.0f ? 1.0f : 0.0f;
        }
        f32x16 b0[1], b1[1];
        sn_mlp_layer_f32<1, 32>(lds + SnNormImg::WB, lds + SnNormImg::ZB, op0, op1, b0, b1, lane);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = b0[0][r], b = b1[0][r];
            sn_swap_halves(a, b);  // lower lane j: own tile-0 row rho(r), lane j+32's tile-0 row rho(r)+4; upper: tile 1
            const int row = (r & 3) + 8 * (r >> 2);
            gfeat[row] = a;
            gfeat[row + 4] = b;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- pred-normal layer 1: (layer-2 rows 0..15 | position encoding in the SH slots) -> 64, ReLU --------------------------
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        op0[t] = g0[0][t];
        op1[t] = g1[0][t];
        float a = t < 6 ? pe[2 * t] : 0.0f, b = t < 6 ? pe[2 * t + 1] : 0.0f;
        if (t < 6) sn_swap_halves(a, b);
        op0[8 + t] = a;
        op1[8 + t] = b;
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
    sn_mlp_layer_f32<2, 32>(lds + SnMainImg::WC2, lds + SnMainImg::BC2, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (layer 3 . head): 64 -> 3 on the VALU, as colour layer 3 ---------------------------------------------------------------
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
                    p0[n] = fmaf(wv[e], sn_relu(a0[rt][r4 * 4 + e]), p0[n]);
                    p1[n] = fmaf(wv[e], sn_relu(a1[rt][r4 * 4 + e]), p1[n]);
                }
            }
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        float a = p0[n], b = p1[n];
        sn_swap_halves(a, b);
        x[n] = a + b + lds[SnMainImg::B3 + n];
    }
}

// Split-precision (fp16 hi + lo, sn_main.h) form of sn_normals_field: same layers through sn_mlp_layer_h.  The reverse pass reads
// a 0 / 1 mask -- exact in fp16 with a zero lo part -- so it needs only two of the three product terms.
SN_DEV void sn_normals_field_h(const char* __restrict__ ldsb, const float* feat, const float* pe, int lane, float& h0, float* gfeat,
                               float x[3]) {
    const bool upper = lane >= 32;
    const float* tail = (const float*)(ldsb + SnMainImgH::FP32);
    SnOpH op0[4], op1[4];
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
    // ReLU mask of layer 1 as fp16 pairs (1.0h = 0x3c00), in the operand order of the next layer.
    // r04: by integer arithmetic, not selects -- a > 0 <=> the float's bit pattern is a positive int32, so clamp(bits, 0, 1) (v_med3_i32)
    // is the 0 / 1 mask, and (m1 << 16 | m0) * 0x3c00 the fp16 pair: 4 instructions per pair instead of 2 compares + 2 VCC-form v_cndmask
    // + 1 v_or (back-to-back VCC selects issue ~5x slower than plain VALU on gfx950, tools/probes/overlap2_probe.hip).  Compiler builtins
    // only: these are the first readers of the layer's MFMA results, and hipcc inserts the MFMA-write -> VALU-read wait states in front
    // of instructions it knows, not in front of inline asm (sn_main.h sn_pk_f16 met exactly that).
    u32x4 m0[4], m1[4];
    {
        auto pair = [](float lo, float hi) {
            const int a = min(max(__float_as_int(lo), 0), 1), b = min(max(__float_as_int(hi), 0), 1);
            return (uint32_t)((b << 16) | a) * 0x3c00u;
        };
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                m0[2 * rt + j / 4][j % 4] = pair(a0[rt][2 * j], a0[rt][2 * j + 1]);
                m1[2 * rt + j / 4][j % 4] = pair(a1[rt][2 * j], a1[rt][2 * j + 1]);
            }
    }
    // (the normals images are conditioned to the same 2^10 pre-activation bound as K1's, sn_api.hip: the ReLU folds into the split)
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a0[rt], true, op0[2 * rt], op0[2 * rt + 1]);
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a1[rt], true, op1[2 * rt], op1[2 * rt + 1]);
    }
    f32x16 g0[1], g1[1];
    sn_mlp_layer_h<1, 4>(ldsb + SnMainImgH::W2, tail + SnMainImgH::B2, op0, op1, g0, g1, lane);
    h0 = (upper ? g1[0][8] : g0[0][0]) * tail[SnMainImgH::B3 + 3];  // the image's 1 / s2 (range conditioning)
    __builtin_amdgcn_sched_barrier(0);
    {
        f32x16 b0 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, b1 = b0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const char* base = ldsb + SnNormImgH::WB + ((s * 2) * 64 + lane) * 16;
            const f16x8 ah = __builtin_bit_cast(f16x8, *(const u32x4*)base);
            const f16x8 al = __builtin_bit_cast(f16x8, *(const u32x4*)(base + 1024));
            const f16x8 k0 = __builtin_bit_cast(f16x8, m0[s]), k1 = __builtin_bit_cast(f16x8, m1[s]);
            SN_MFMA_H(b0, al, k0);
            SN_MFMA_H(b1, al, k1);
            SN_MFMA_H(b0, ah, k0);
            SN_MFMA_H(b1, ah, k1);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float a = b0[r], b = b1[r];
            sn_swap_halves(a, b);
            const int row = (r & 3) + 8 * (r >> 2);
            gfeat[row] = a;
            gfeat[row + 4] = b;
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
        op0[0].set(v0);
        op1[0].set(v1);
        // slot (h, e) of k-step 1 <-> position-encoding component 8h + e (12 real, 4 zero)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = pe[e], b = e < 4 ? pe[8 + e] : 0.0f;
            sn_swap_halves(a, b);
            v0[e] = a;
            v1[e] = b;
        }
        op0[1].set(v0);
        op1[1].set(v1);
    }
    sn_mlp_layer_h<2, 2>(ldsb + SnMainImgH::WC1, tail + SnMainImgH::BC1, op0, op1, a0, a1, lane);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a0[rt], true, op0[2 * rt], op0[2 * rt + 1]);
        sn_acc_to_ops<SN_RELU_FOLD != 0>(a1[rt], true, op1[2 * rt], op1[2 * rt + 1]);
    }
    const int h = lane >> 5;
    float p0[3] = {0.f, 0.f, 0.f}, p1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f32x16 c0[1], c1[1];
        sn_mlp_layer_h<1, 4>(ldsb + SnMainImgH::WC2 + rt * 8192, tail + SnMainImgH::BC2 + rt * 32, op0, op1, c0, c1, lane);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const f32x4* w = (const f32x4*)(tail + SnMainImgH::W3 + (n * 2 + h) * 32);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const f32x4 wv = w[rt * 4 + r4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    p0[n] = fmaf(wv[e], sn_relu(c0[0][r4 * 4 + e]), p0[n]);
                    p1[n] = fmaf(wv[e], sn_relu(c1[0][r4 * 4 + e]), p1[n]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        float a = p0[n], b = p1[n];
        sn_swap_halves(a, b);
        x[n] = a + b + tail[SnMainImgH::B3 + n];
    }
}

// 1.0 for an in-voxel offset > 0, 0.0 for exactly 0: offsets are v_fract results in [0, 1), a positive one is >= 2^-24 x 2^-12, so
// clamp(off * 2^127) is exactly 0 or 1 -- one multiply with the clamp modifier instead of a compare and a select
SN_DEV float sn_nonzero01(float off) { return __builtin_amdgcn_fmed3f(off * 0x1p127f, 0.0f, 1.0f); }

// g_q += sum over levels of scale_l * (g_feat[2l], g_feat[2l+1]) . d(feature pair)/d(offset): the gradient of the trilinear
// blend (sn_hash_blend's association) w.r.t. the in-voxel offset; floor / ceil carry no gradient.
#ifndef SN_GRAD_GROUP
#define SN_GRAD_GROUP 4
#endif
// ND > 0 (torch grid): levels [0, ND) come from the de-hashed copies, [0, NBC) of them in bilinear-coefficient form, where the slopes
// are the coefficients themselves: per z slice d/d ox = B + oy D, d/d oy = C + ox D, and d/d oz = slice 1 - slice 0.  The copies hold
// feature_scale x value; `inv_scale` (its exact inverse) rides in the per-level factor.
// One level: the slopes d(feature pair)/d(in-voxel offset) -- and, from the SAME fetches, the feature pair itself (r04: K3 keeps the slopes of
// its finest levels from the forward pass instead of gathering those levels a second time).  val / dx / dy / dz carry whatever scale the
// fetched entries carry (the copies hold feature_scale x value); `sl` is the factor that turns a slope w.r.t. the offset into one w.r.t.
// the normalised position and divides that scale out again.
template <int GRID, int ND, int NBC>
SN_DEV void sn_hash_level_slopes(int l, __amdgpu_buffer_rsrc_t rsrc, const float* scal, int log2_t, const float q[3], const SnGridLevels* grid,
                                 const SnDenseCopy* dense, float inv_scale, float val[2], float dx[2], float dy[2], float dz[2], float off_out[3],
                                 float& sl_out) {
    const uint32_t mask = (1u << log2_t) - 1u;
    if (ND > 0 && l < NBC && l < 12) {
        uint32_t R = dense->res[l];
        asm volatile("" : "+s"(R));
        const __amdgpu_buffer_rsrc_t drsrc = sn_table_rsrc(dense->base, dense->bytes);
        uint32_t f[3];
        float off[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float x = q[a] * scal[l];
            off[a] = __builtin_amdgcn_fractf(x);
            f[a] = (uint32_t)(int)x;
        }
        const uint32_t R32 = R << 5, R2_32 = (R * R) << 5;
        const uint32_t b = sn_mad24(f[2], R2_32, sn_mad24(f[1], R32, f[0] << 5));
        const uint32_t o0 = dense->off[l], o1 = o0 + R2_32;
        const f32x4 ab0 = sn_table_load_pair(drsrc, b, o0), cd0 = sn_table_load_pair(drsrc, b + 16u, o0);
        const f32x4 ab1 = sn_table_load_pair(drsrc, b, o1), cd1 = sn_table_load_pair(drsrc, b + 16u, o1);
        const float ox = off[0], oy = off[1], oz = off[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float e0 = fmaf(ox, cd0[2 + c], cd0[c]), e1 = fmaf(ox, cd1[2 + c], cd1[c]);   // d slice / d oy
            const float z0 = fmaf(ox, ab0[2 + c], fmaf(oy, e0, ab0[c])), z1 = fmaf(ox, ab1[2 + c], fmaf(oy, e1, ab1[c]));
            const float h0 = fmaf(oy, cd0[2 + c], ab0[2 + c]), h1 = fmaf(oy, cd1[2 + c], ab1[2 + c]);  // d slice / d ox
            dx[c] = fmaf(h1 - h0, oz, h0);
            dy[c] = fmaf(e1 - e0, oz, e0);
            dz[c] = z1 - z0;
            val[c] = fmaf(dz[c], oz, z0);
        }
        off_out[0] = ox;
        off_out[1] = oy;
        off_out[2] = oz;
        sl_out = scal[l] * inv_scale;
        return;
    }
    f32x2 v[8];
    float ox, oy, oz, sl = scal[l];
    if (ND > 0 && l < ND && l < 12) {  // de-hashed copy in plain-row form: four 16-byte fetches (sn_hash_level_dense_copy)
        uint32_t R = dense->res[l];
        asm volatile("" : "+s"(R));
        const __amdgpu_buffer_rsrc_t drsrc = sn_table_rsrc(dense->base, dense->bytes);
        uint32_t f[3];
        float off[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float x = q[a] * scal[l];
            off[a] = __builtin_amdgcn_fractf(x);
            f[a] = (uint32_t)(int)x;
        }
        const uint32_t R8 = R << 3, R28 = (R * R) << 3;
        const uint32_t b_ff = sn_mad24(f[2], R28, sn_mad24(f[1], R8, f[0] << 3));
        const uint32_t o_ff = dense->off[l], o_cf = o_ff + R8, o_fc = o_ff + R28, o_cc = o_ff + (R8 + R28);
        const f32x4 p_cc = sn_table_load_pair(drsrc, b_ff, o_cc), p_fc = sn_table_load_pair(drsrc, b_ff, o_fc);
        const f32x4 p_ff = sn_table_load_pair(drsrc, b_ff, o_ff), p_cf = sn_table_load_pair(drsrc, b_ff, o_cf);
        v[3] = f32x2{p_cc.x, p_cc.y};
        v[0] = f32x2{p_cc.z, p_cc.w};
        v[2] = f32x2{p_fc.x, p_fc.y};
        v[1] = f32x2{p_fc.z, p_fc.w};
        v[6] = f32x2{p_ff.x, p_ff.y};
        v[5] = f32x2{p_ff.z, p_ff.w};
        v[7] = f32x2{p_cf.x, p_cf.y};
        v[4] = f32x2{p_cf.z, p_cf.w};
        ox = off[0];
        oy = off[1];
        oz = off[2];
        sl *= inv_scale;
    } else {
        SnHashLevel hl;
        if (GRID) sn_hash_corners_tcnn<-1>(q, scal[l], mask, sn_grid_dense_res(*grid, l), hl);
        else sn_hash_corners_fast(q, scal[l], mask, hl);
        const uint32_t lvl = ((uint32_t)l << log2_t) * 8u;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = sn_table_load(rsrc, hl.boff[k], lvl);
        ox = hl.off[0];
        oy = hl.off[1];
        oz = hl.off[2];
    }
    // corner order 0 ccc, 1 cfc, 2 ffc, 3 fcc, 4 ccf, 5 cff, 6 fff, 7 fcf; the "c" corner of an axis has weight off.
    // Plain fp32 instructions, lerps as a + w (b - a) (sn_hash_blend_fast explains both): the slopes are differences of lerps.
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const float x03 = v[0][c] - v[3][c], x12 = v[1][c] - v[2][c], x47 = v[4][c] - v[7][c], x56 = v[5][c] - v[6][c];
        const float xz1 = fmaf(x03 - x12, oy, x12), xz0 = fmaf(x47 - x56, oy, x56);   // d/d ox in the slices z + 1, z
        dx[c] = fmaf(xz1 - xz0, oz, xz0);
        const float f03 = fmaf(x03, ox, v[3][c]), f12 = fmaf(x12, ox, v[2][c]), f47 = fmaf(x47, ox, v[7][c]), f56 = fmaf(x56, ox, v[6][c]);
        const float y1 = f03 - f12, y0 = f47 - f56;                                    // d/d oy in the slices z + 1, z
        dy[c] = fmaf(y1 - y0, oz, y0);
        const float zc = fmaf(y1, oy, f12), zf = fmaf(y0, oy, f56);                    // the blend in the slices z + 1, z
        dz[c] = zc - zf;
        val[c] = fmaf(dz[c], oz, zf);
    }
    off_out[0] = ox;
    off_out[1] = oy;
    off_out[2] = oz;
    sl_out = sl;
}

// torch path: where scale * q is an integer in fp32 (about 6e-4 of the samples: ulp(x) / 1 at the fine levels), ceil(x) == floor(x), both
// corners of that axis are the SAME table row and autograd sees no slope along it; the kernels fetch floor + 1 (value-identical, weight 0),
// so the slope is dropped explicitly -- as a 0 / 1 factor, not a select.  tiny-cuda-nn grids have no such points.
template <int GRID, int ND = -1, int NBC = 0, int L0 = 0, int L1 = 16>
SN_DEV void sn_hash_encode_grad(__amdgpu_buffer_rsrc_t rsrc, const float* scal, int log2_t, const float q[3], const float* gfeat,
                                const SnGridLevels* grid, float g[3], const SnDenseCopy* dense = nullptr, float inv_scale = 1.0f) {
    if (L0 == 0) g[0] = g[1] = g[2] = 0.0f;
#pragma unroll
    for (int l = L0; l < L1; ++l) {
        if (l > L0 && (l % SN_GRAD_GROUP) == 0) __builtin_amdgcn_sched_barrier(0);
        float val[2], dx[2], dy[2], dz[2], off[3], sl;
        sn_hash_level_slopes<GRID, ND, NBC>(l, rsrc, scal, log2_t, q, grid, dense, inv_scale, val, dx, dy, dz, off, sl);
        const float ga = gfeat[2 * l] * sl, gb = gfeat[2 * l + 1] * sl;
        if (!GRID) {
            g[0] = fmaf(fmaf(ga, dx[0], gb * dx[1]), sn_nonzero01(off[0]), g[0]);
            g[1] = fmaf(fmaf(ga, dy[0], gb * dy[1]), sn_nonzero01(off[1]), g[1]);
            g[2] = fmaf(fmaf(ga, dz[0], gb * dz[1]), sn_nonzero01(off[2]), g[2]);
        } else {
            g[0] = fmaf(ga, dx[0], fmaf(gb, dx[1], g[0]));
            g[1] = fmaf(ga, dy[0], fmaf(gb, dy[1], g[1]));
            g[2] = fmaf(ga, dz[0], fmaf(gb, dz[1], g[2]));
        }
    }
}

// Forward pass of the levels [L0, 16) WITH their slopes (r04): the feature pairs go to feat[2l], feat[2l + 1] (times feat_mul: the hashed
// levels take the feature scale the copies already carry, or the copies lose it, as the precision's weight image expects) and the six
// position-slopes of every level -- already multiplied by the level's scale factor and the zero-offset 0 / 1 factors -- to keep[6 (l - L0) ..].
// The second pass then contracts them with the back-propagated feature gradient (sn_kept_slopes_contract) without touching the tables.
template <int GRID, int ND, int NBC, int L0>
SN_DEV void sn_hash_encode_keep(__amdgpu_buffer_rsrc_t rsrc, const float* scal, int log2_t, const float q[3], float* feat, float* keep,
                                const SnGridLevels* grid, const SnDenseCopy* dense, float inv_scale, float copy_mul, float plain_mul) {
#pragma unroll
    for (int l = L0; l < 16; ++l) {
        if (l > L0 && ((l - L0) % 2) == 0) __builtin_amdgcn_sched_barrier(0);
        float val[2], dx[2], dy[2], dz[2], off[3], sl;
        sn_hash_level_slopes<GRID, ND, NBC>(l, rsrc, scal, log2_t, q, grid, dense, inv_scale, val, dx, dy, dz, off, sl);
        const float fm = (ND > 0 && l < ND && l < 12) ? copy_mul : plain_mul;
        feat[2 * l] = val[0] * fm;
        feat[2 * l + 1] = val[1] * fm;
        const float kx = GRID ? sl : sl * sn_nonzero01(off[0]), ky = GRID ? sl : sl * sn_nonzero01(off[1]), kz = GRID ? sl : sl * sn_nonzero01(off[2]);
        float* k = keep + 6 * (l - L0);
        k[0] = dx[0] * kx;
        k[1] = dx[1] * kx;
        k[2] = dy[0] * ky;
        k[3] = dy[1] * ky;
        k[4] = dz[0] * kz;
        k[5] = dz[1] * kz;
    }
}

template <int L0>
SN_DEV void sn_kept_slopes_contract(const float* keep, const float* gfeat, float g[3]) {
#pragma unroll
    for (int l = L0; l < 16; ++l) {
        const float* k = keep + 6 * (l - L0);
        const float ga = gfeat[2 * l], gb = gfeat[2 * l + 1];
        g[0] = fmaf(ga, k[0], fmaf(gb, k[1], g[0]));
        g[1] = fmaf(ga, k[2], fmaf(gb, k[3], g[1]));
        g[2] = fmaf(ga, k[4], fmaf(gb, k[5], g[2]));
    }
}

// waves per SIMD the torch-grid kernels are compiled for (2: 197-210 VGPRs without spills; 3 caps them at 168)
#ifndef SN_NORMALS_WAVES
#define SN_NORMALS_WAVES 2
#endif
// first level whose slopes are kept from the forward pass (torch grid with de-hashed copies; 16 = none): 6 registers per kept level.
// Measured r04, same box, 800x800x64, tables x 1e-3 (profiles/r04_normals_keep_ab.txt): none 4.95 ms (168 gathers per wave-step, 200 VGPRs);
// levels 11-15: 4.37 (128 gathers, 228 VGPRs); 9-15: 4.30 (120, 240); 7-15: 4.12 (112, 252 -- the last level that fits two waves per SIMD
// without spills).  Three waves per SIMD (168 VGPRs, 31 spilled, nothing kept): 5.4 ms.
#ifndef SN_NORMALS_KEEP0
#define SN_NORMALS_KEEP0 7
#endif
template <int MODE /*0 uniform-in-s bins, 1 explicit bins*/, int GRID /*0 torch grid, 1 tiny-cuda-nn grid*/,
          int PREC /*0 exact fp32 MFMA, 1 fp16 hi+lo split MFMA*/, int ND = -1 /*torch grid: leading levels read from the de-hashed copies*/,
          bool ALT = false /*the non-default sampler / position map (SnNormalsParams::spacing_uniform, pm)*/>
// (the run-time dense / hashed branch of the tiny-cuda-nn grid needs more registers than 2 waves per SIMD leave: 1 wave there)
__global__ __launch_bounds__(256, GRID ? 1 : SN_NORMALS_WAVES) void sn_normals_kernel(SnNormalsParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int su = ALT ? p.spacing_uniform : 0;
    const SnPosMap* pm = ALT ? &p.pm : nullptr;
    for (int i = tid * 4; i < (PREC ? SnNormImgH::TOTAL_BYTES / 4 : SnNormImg::TOTAL); i += 256 * 4) *(f32x4*)(lds + i) = *(const f32x4*)(p.wimg + i);
    __syncthreads();

    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gbx = (p.tiles_x + 1) >> 1, gby = (p.tiles_y + 1) >> 1;
    const int blk = sn_xcd_remap(blockIdx.x, gbx * gby);
    const int tx = (blk % gbx) * 2 + (wave & 1);
    const int ty = (blk / gbx) * 2 + (wave >> 1);
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
    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.table, (16u << p.log2_t) * 8u);
    const int S = p.n_samples;
    const float* eb = nullptr;
    if (MODE == 1) eb = p.ebins + ((int64_t)(ty * p.tiles_x + tx) * (S + 1)) * 64 + lane;

    SnComposite comp;  // weights only: NormalsRenderer sums w * n WITHOUT the nan_to_num RGBRenderer applies to colours (a ray that misses
    comp.init();       // the render box has NaN normals and zero weights: its rendered normal is NaN in the reference, not 0.5)
    float an_acc[3] = {0.f, 0.f, 0.f};
    float pn[3] = {0.f, 0.f, 0.f};
    float t0 = MODE == 0 ? sn_euclid(p.sbins ? p.sbins[0] : 0.0f, s_near, s_far, su) : eb[0];
#pragma unroll 1
    for (int i = 0; i < S; ++i) {
        asm volatile("" ::: "memory");  // keeps the loop-invariant LDS weight reads inside the loop (sn_main.h)
        const float t1 = MODE == 0 ? sn_euclid(p.sbins ? p.sbins[i + 1] : (float)(i + 1) / (float)S, s_near, s_far, su)
                                   : eb[(int64_t)(i + 1) * 64];
        float q[3];
        // the STRICT position arithmetic (not K1's rcp form): the gradient is discontinuous across voxel faces, and a position
        // one ulp off lands in the neighbouring voxel of a fine level about once per thousand samples
        const bool sel = sn_sample_q(o, d, t0, t1, q, pm);
        float feat[32];
        constexpr int NBC = ND > SN_BC_MAIN ? SN_BC_MAIN : (ND > 0 ? ND : 0);
        // r04: the finest levels [KEEP0, 16) are fetched ONCE -- value and slopes from the same gathers (sn_hash_encode_keep), the slopes
        // (6 registers per level) waiting across the MLPs -- instead of twice; the kernel sat at 0.94 of the gather-issue roof with 168
        // gathers per wave-step (DESIGN K3) and has registers to spare at two waves per SIMD
        constexpr int KEEP0 = (ND > 0 && !GRID) ? SN_NORMALS_KEEP0 : 16;
        float keep[KEEP0 < 16 ? 6 * (16 - KEEP0) : 1];
        if (ND > 0) {
            // values of the copies are the table's own times the feature scale t0 (an exact power of two).  Split precision: the
            // conditioned image expects t0 x feature, so the copies' values go in as they are and the hashed levels are multiplied
            // (sn_hash_encode's plain_scale); exact fp32: divided out again here
            if (KEEP0 > 0)
                sn_hash_encode<(KEEP0 < 16 ? (KEEP0 > 0 ? KEEP0 : 1) : 16), 4, 1, ND, false, NBC>(rsrc, p.scal, p.log2_t, q, feat, &p.grid, &p.dense, nullptr,
                                                                                                  PREC ? p.feat_scale : 1.0f);
            if (!PREC) {
#pragma unroll
                for (int k = 0; k < 2 * ((ND > 0 ? ND : 0) < KEEP0 ? (ND > 0 ? ND : 0) : KEEP0); ++k) feat[k] *= p.inv_feat_scale;
            }
            if (KEEP0 < 16) {
                __builtin_amdgcn_sched_barrier(0);
                sn_hash_encode_keep<GRID, ND, NBC, KEEP0>(rsrc, p.scal, p.log2_t, q, feat, keep, &p.grid, &p.dense, p.inv_feat_scale,
                                                          PREC ? 1.0f : p.inv_feat_scale, PREC ? p.feat_scale : 1.0f);
            }
        } else {
            sn_hash_encode<16, 4, (GRID ? 2 : 1), -1>(rsrc, p.scal, p.log2_t, q, feat, &p.grid, nullptr, nullptr, PREC ? p.feat_scale : 1.0f);
        }
        float pe[12];
        {
            const float tm = (t0 + t1) * 0.5f;
            const float pw[3] = {fmaf(d[0], tm, o[0]), fmaf(d[1], tm, o[1]), fmaf(d[2], tm, o[2])};
            sn_position_encoding(pw, pe, p.pe_rev_scale);
        }
        __builtin_amdgcn_sched_barrier(0);
        float h0, gfeat[32], x[3];
        if (PREC) sn_normals_field_h((const char*)lds, feat, pe, lane, h0, gfeat, x);
        else sn_normals_field(lds, feat, pe, lane, h0, gfeat, x);
        __builtin_amdgcn_sched_barrier(0);
        float g[3];
        // opaque copies of q: otherwise the compiler keeps the first pass's 128 corner offsets alive across the MLPs to reuse
        // them here (~120 spilled registers) instead of recomputing them
        asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]));
        sn_hash_encode_grad<GRID, ND, NBC, 0, KEEP0>(rsrc, p.scal, p.log2_t, q, gfeat, &p.grid, g, &p.dense, p.inv_feat_scale);
        if (KEEP0 < 16) sn_kept_slopes_contract<KEEP0>(keep, gfeat, g);
        __builtin_amdgcn_sched_barrier(0);
        // Field.get_normals: -F.normalize(grad) = -grad / max(|grad|, 1e-12)
        const float gl = fmaxf(sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), PREC ? 1e-12f * p.grad_scale : 1e-12f);  // (g carries grad_scale)
        const float igl = -__builtin_amdgcn_rcpf(gl);  // (1-ulp reciprocals: the per-sample normals are weighted, summed and renormalised)
        const float an[3] = {g[0] * igl, g[1] * igl, g[2] * igl};
        // PredNormalsFieldHead: tanh, then F.normalize
        // (a NaN position -- a ray that misses the render box -- is NaN through the reference's MLPs; the v_max-based ReLUs here launder it, so
        // it is restored on the result: q * 0 is +-0 for a finite position and NaN for a NaN one)
        const float nan_term = fmaf(q[2], 0.0f, fmaf(q[1], 0.0f, q[0] * 0.0f));
        const float tx3[3] = {sn_tanh(x[0]) + nan_term, sn_tanh(x[1]) + nan_term, sn_tanh(x[2]) + nan_term};
        const float itl = __builtin_amdgcn_rcpf(fmaxf(sqrtf(tx3[0] * tx3[0] + tx3[1] * tx3[1] + tx3[2] * tx3[2]), 1e-12f));
        const float density = p.avg_density * sn_exp<true>(h0) * (sel ? 1.0f : 0.0f);
        const float w = comp.step<true>(i, t0, t1, density, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            an_acc[c] = fmaf(w, an[c], an_acc[c]);
            pn[c] = fmaf(w * itl, tx3[c], pn[c]);
        }
        t0 = t1;
    }
    if (valid) {
        const int64_t pix = (int64_t)py * p.width + px;
        // NormalsRenderer (normalize=True): n / (|n| + 1e-10); NormalsShader: (n + 1) / 2
        if (p.normals) {
            const float l = sqrtf(an_acc[0] * an_acc[0] + an_acc[1] * an_acc[1] + an_acc[2] * an_acc[2]) + 1e-10f;
#pragma unroll
            for (int c = 0; c < 3; ++c) p.normals[pix * 3 + c] = (an_acc[c] / l + 1.0f) / 2.0f;
        }
        if (p.pred_normals) {
            const float l = sqrtf(pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2]) + 1e-10f;
#pragma unroll
            for (int c = 0; c < 3; ++c) p.pred_normals[pix * 3 + c] = (pn[c] / l + 1.0f) / 2.0f;
        }
    }
}
