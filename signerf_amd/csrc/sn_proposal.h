// sn_proposal.h -- the proposal-sampler kernel (rows a8-a12 of SURVEY.md §8(a)).
//
// Mapping (DESIGN.md "Kernel K2"): one wave owns one 8x8 pixel tile and walks its 64 rays ONE RAY AT A
// TIME with the 64 lanes spread over that ray's samples (256 / 96 samples = 4 / 1.5 lane-rounds):
//   * consecutive samples of a ray fall into the same or neighbouring coarse voxels (5 levels,
//     res 16..256), so a wave's gather addresses coalesce;
//   * transmittance and CDF are prefix sums ALONG the ray = wave scans (DPP/permute shuffles, fp64 to
//     mirror torch-CPU's cumsum), the inverse-CDF lookup is a per-lane binary search in a 1 KB
//     per-wave LDS array ("LDS staging of per-ray samples");
//   * the tiny density MLP (10->16->1, 352 FLOP) runs on the VALU with wave-uniform weights in SGPRs.
// The final sample bins are written in [tile][bin][lane=ray] order, which is exactly the coalesced
// order in which sn_render_main_kernel<1> (lane = ray) consumes them.
#pragma once
#include "../../include/signerf_hip.h"
#include "sn_device.h"

#define SN_PROP_MAX_SAMPLES 256
#define SN_PROP_WAVES 4

// proposal-net MLP pack (floats): W0 [16][10], b0 [16], W1 [16], b1
#define SN_PROP_W0 0
#define SN_PROP_B0 160
#define SN_PROP_W1 176
#define SN_PROP_B1 192
#define SN_PROP_PACK_FLOATS 196

struct SnScal5 {
    float v[5];
};

// pre-activation density of one proposal net at normalised position q
SN_DEV float sn_prop_h0(__amdgpu_buffer_rsrc_t rsrc, const SnScal5& scal, int log2_t, const float* __restrict__ w, const float q[3]) {
    float feat[10];
    sn_hash_encode<5>(rsrc, scal.v, log2_t, q, feat);
    float out = w[SN_PROP_B1];
#pragma unroll
    for (int n = 0; n < 16; ++n) {
        float a = w[SN_PROP_B0 + n];
#pragma unroll
        for (int k = 0; k < 10; ++k) a = fmaf(w[SN_PROP_W0 + n * 10 + k], feat[k], a);
        out = fmaf(w[SN_PROP_W1 + n], fmaxf(a, 0.0f), out);
    }
    // v_max-based ReLU launders NaN; the reference's field is NaN all the way for a NaN position
    if ((q[0] != q[0]) | (q[1] != q[1]) | (q[2] != q[2])) out = __builtin_nanf("");
    return out;
}

// ---- wave-level primitives ---------------------------------------------------------------------
SN_DEV double sn_shfl_up_f64(double v, int delta) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_up(lo, delta);
    hi = __shfl_up(hi, delta);
    return __hiloint2double(hi, lo);
}
SN_DEV double sn_shfl_f64(double v, int src) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl(lo, src);
    hi = __shfl(hi, src);
    return __hiloint2double(hi, lo);
}
// inclusive prefix sum over the 64 lanes
SN_DEV double sn_wave_scan_f64(double v, int lane) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        double n = sn_shfl_up_f64(v, s);
        if (lane >= s) v += n;
    }
    return v;
}
SN_DEV double sn_wave_sum_f64(double v) {
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        int lo = __double2loint(v), hi = __double2hiint(v);
        lo = __shfl_xor(lo, s);
        hi = __shfl_xor(hi, s);
        v += __hiloint2double(hi, lo);
    }
    return v;
}
// Orders this wave's LDS traffic (a wave's DS ops complete in issue order; only the compiler must be fenced).
SN_DEV void sn_wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- PDFSampler, eval mode (A11), for ONE ray held in this wave's LDS arrays ---------------------
// sb [N+1] spacing bins, wl [N] weights -> nb [M+1] new spacing bins; cdf [N+1] scratch.
// u: global [M+1] or null (fallback (k+0.5)/(M+1)).  inds: optional global [M+1] int32.
SN_DEV void sn_pdf_wave(const float* sb, const float* wl, float* cdf, float* nb, int N, int M, const float* __restrict__ u, float pad,
                        int lane, int32_t* inds) {
    // weights + padding, their sum
    double part = 0.0;
    for (int i = lane; i < N; i += 64) {
#pragma clang fp contract(off)
        part += (double)(wl[i] + pad);
    }
    float wsum = (float)sn_wave_sum_f64(part);
    float padding, denom;
    {
#pragma clang fp contract(off)
        padding = fmaxf(1e-5f - wsum, 0.0f);
        denom = wsum + padding;
        padding = padding / (float)N;
    }
    // cdf = min(1, cumsum(pdf)), prepend 0
    double carry = 0.0;
    for (int c = 0; c < N; c += 64) {
        const int i = c + lane;
        float pdf = 0.0f;
        if (i < N) {
#pragma clang fp contract(off)
            pdf = ((wl[i] + pad) + padding) / denom;
        }
        double incl = sn_wave_scan_f64((double)pdf, lane) + carry;
        if (i < N) cdf[i + 1] = fminf(1.0f, (float)incl);
        carry = sn_shfl_f64(incl, 63);
    }
    if (lane == 0) cdf[0] = 0.0f;
    sn_wave_lds_fence();
    // inverse CDF
    for (int j = lane; j <= M; j += 64) {
        float uj;
        if (u) uj = u[j];
        else {
#pragma clang fp contract(off)
            uj = ((float)j + 0.5f) / (float)(M + 1);
        }
        // searchsorted(cdf, u, side="right") = number of entries <= u
        int lo = 0, hi = N + 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (cdf[mid] <= uj) lo = mid + 1;
            else hi = mid;
        }
        const int idx = lo;
        const int below = min(max(idx - 1, 0), N), above = min(max(idx, 0), N);
        float t, v;
        {
#pragma clang fp contract(off)
            const float c0 = cdf[below], c1 = cdf[above], b0 = sb[below], b1 = sb[above];
            t = (uj - c0) / (c1 - c0);
            if (t != t) t = 0.0f;
            t = fminf(fmaxf(t, 0.0f), 1.0f);
            v = b0 + t * (b1 - b0);
        }
        nb[j] = v;
        if (inds) inds[j] = idx;
    }
    sn_wave_lds_fence();
}

// ---- proposal kernel -----------------------------------------------------------------------------
struct SnPropParams {
    const float* origins;
    const float* directions;
    const float* nears;
    const float* fars;
    const float* sbins0;                  // [n_samples[0]+1] initial spacing bins
    const float* pdf_u[SN_MAX_PROPOSALS]; // u grid of resampling step k, or null
    float* ebins_out;                     // [tile][n_final+1][64]
    float* prop_depth[SN_MAX_PROPOSALS];  // [H*W] or null
    const float* table[SN_MAX_PROPOSALS];
    const float* wpack[SN_MAX_PROPOSALS];
    float scal[SN_MAX_PROPOSALS][5];
    int log2_t[SN_MAX_PROPOSALS];
    int n_samples[SN_MAX_PROPOSALS];
    int n_levels;  // 1 or 2
    int n_final;
    int height, width, tile_w_log2, tile_h_log2, tiles_x, tiles_y;
    float near_plane, far_plane, avg_density, hist_pad;
};

struct SnPropLds {
    float a[SN_PROP_MAX_SAMPLES + 4];  // spacing bins (ping)
    float b[SN_PROP_MAX_SAMPLES + 4];  // spacing bins (pong)
    float w[SN_PROP_MAX_SAMPLES + 4];
    float cdf[SN_PROP_MAX_SAMPLES + 4];
};

// One proposal level for one ray: evaluate density net LV at the N samples given by spacing bins sb,
// write weights to wl, return the median depth of this level (prop_depth_LV).
template <int LV>
SN_DEV float sn_prop_level(const SnPropParams& p, const float* sb, float* wl, int N, const float o[3], const float d[3], float s_near,
                           float s_far, int lane) {
    SnScal5 scal;
#pragma unroll
    for (int l = 0; l < 5; ++l) scal.v[l] = p.scal[LV][l];
    const int log2_t = p.log2_t[LV];
    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.table[LV], (5u << log2_t) * 8u);
    const float* __restrict__ wp = p.wpack[LV];
    double carry_tau = 0.0, carry_w = 0.0;
    bool found = false;
    float median = 0.0f, last_mid = 0.0f;
    for (int c = 0; c < N; c += 64) {
        const int i = min(c + lane, N - 1);
        const bool live = c + lane < N;
        const float e0 = sn_euclid(sb[i], s_near, s_far), e1 = sn_euclid(sb[i + 1], s_near, s_far);
        float q[3];
        const bool sel = sn_sample_q(o, d, e0, e1, q);
        const float h0 = sn_prop_h0(rsrc, scal, log2_t, wp, q);
        const float density = p.avg_density * expf(h0) * (sel ? 1.0f : 0.0f);
        float tau, mid;
        {
#pragma clang fp contract(off)
            tau = live ? (e1 - e0) * density : 0.0f;
            mid = (e0 + e1) / 2.0f;
        }
        const double incl = sn_wave_scan_f64((double)tau, lane) + carry_tau;
        float w;
        {
#pragma clang fp contract(off)
            const float excl = (float)(incl - (double)tau);
            w = (1.0f - expf(-tau)) * expf(-excl);
            if (w != w) w = 0.0f;
            if (!live) w = 0.0f;
        }
        // NOTE: (incl - tau) re-derives the exclusive prefix; fp64 sums of fp32 data are exact for any
        // realistic dynamic range, so this equals torch's sequential cumsum of the preceding terms.
        if (live) wl[i] = w;
        carry_tau = sn_shfl_f64(incl, 63);
        const double cw = sn_wave_scan_f64((double)w, lane) + carry_w;
        carry_w = sn_shfl_f64(cw, 63);
        const unsigned long long hit = __ballot(live && (float)cw >= 0.5f);
        if (!found && hit) {
            found = true;
            median = __shfl(mid, __ffsll((long long)hit) - 1);
        }
        const int last_lane = min(N - 1 - c, 63);
        last_mid = __shfl(mid, last_lane);
    }
    sn_wave_lds_fence();
    return found ? median : last_mid;
}

__global__ __launch_bounds__(64 * SN_PROP_WAVES) void sn_proposal_kernel(SnPropParams p) {
    __shared__ SnPropLds lds_all[SN_PROP_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SnPropLds& L = lds_all[wave];
    const int tile = blockIdx.x * SN_PROP_WAVES + wave;
    if (tile >= p.tiles_x * p.tiles_y) return;
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int tw = 1 << p.tile_w_log2;
    // lane-resident copy of this tile's 64 rays
    const int px = (tx << p.tile_w_log2) + (lane & (tw - 1));
    const int py = (ty << p.tile_h_log2) + (lane >> p.tile_w_log2);
    const bool valid = px < p.width && py < p.height;
    const int64_t ray = (int64_t)min(py, p.height - 1) * p.width + min(px, p.width - 1);
    float ro[3], rd[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ro[c] = p.origins[ray * 3 + c];
        rd[c] = p.directions[ray * 3 + c];
    }
    const float rnear = p.nears ? p.nears[ray] : p.near_plane;
    const float rfar = p.fars ? p.fars[ray] : p.far_plane;
    float my_depth0 = 0.0f, my_depth1 = 0.0f;
    float* eb_tile = p.ebins_out + (int64_t)tile * (p.n_final + 1) * 64;

    for (int r = 0; r < 64; ++r) {
        float o[3], d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            o[c] = __shfl(ro[c], r);
            d[c] = __shfl(rd[c], r);
        }
        const float s_near = sn_spacing(__shfl(rnear, r)), s_far = sn_spacing(__shfl(rfar, r));
        // level 0: the initial (uniform in s) sampler
        const int n0 = p.n_samples[0];
        for (int i = lane; i <= n0; i += 64) L.a[i] = p.sbins0 ? p.sbins0[i] : (float)i / (float)n0;
        sn_wave_lds_fence();
        const float dep0 = sn_prop_level<0>(p, L.a, L.w, n0, o, d, s_near, s_far, lane);
        if (lane == r) my_depth0 = dep0;
        const float* cur = L.a;
        int ncur = n0;
        if (p.n_levels > 1) {
            const int n1 = p.n_samples[1];
            sn_pdf_wave(L.a, L.w, L.cdf, L.b, n0, n1, p.pdf_u[0], p.hist_pad, lane, nullptr);
            const float dep1 = sn_prop_level<1>(p, L.b, L.w, n1, o, d, s_near, s_far, lane);
            if (lane == r) my_depth1 = dep1;
            cur = L.b;
            ncur = n1;
        }
        // final resampling -> the main field's bins
        float* nxt = cur == L.a ? L.b : L.a;
        sn_pdf_wave(cur, L.w, L.cdf, nxt, ncur, p.n_final, p.pdf_u[p.n_levels - 1], p.hist_pad, lane, nullptr);
        for (int j = lane; j <= p.n_final; j += 64) eb_tile[(int64_t)j * 64 + r] = sn_euclid(nxt[j], s_near, s_far);
        sn_wave_lds_fence();
    }
    if (valid) {
        const int64_t pix = (int64_t)py * p.width + px;
        if (p.prop_depth[0]) p.prop_depth[0][pix] = my_depth0;
        if (p.n_levels > 1 && p.prop_depth[1]) p.prop_depth[1][pix] = my_depth1;
    }
}

// ---- stage kernels ---------------------------------------------------------------------------------
struct SnPropStageParams {
    const float* positions;
    int64_t n;
    const float* table;
    const float* wpack;
    float scal[5];
    int log2_t;
    float avg_density;
    float* density;
};

__global__ void sn_prop_field_stage_kernel(SnPropStageParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t j = i < p.n ? i : p.n - 1;
    const float pos[3] = {p.positions[j * 3], p.positions[j * 3 + 1], p.positions[j * 3 + 2]};
    float q[3];
    const bool sel = sn_position_q(pos, q);
    SnScal5 scal;
#pragma unroll
    for (int l = 0; l < 5; ++l) scal.v[l] = p.scal[l];
    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.table, (5u << p.log2_t) * 8u);
    const float h0 = sn_prop_h0(rsrc, scal, p.log2_t, p.wpack, q);
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

__global__ __launch_bounds__(64 * SN_PROP_WAVES) void sn_pdf_stage_kernel(SnPdfStageParams p) {
    __shared__ SnPropLds lds_all[SN_PROP_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SnPropLds& L = lds_all[wave];
    const int64_t r = (int64_t)blockIdx.x * SN_PROP_WAVES + wave;
    if (r >= p.n_rays) return;
    const int N = p.n_in, M = p.n_out;
    for (int i = lane; i <= N; i += 64) L.a[i] = p.sbins[r * (N + 1) + i];
    for (int i = lane; i < N; i += 64) L.w[i] = p.weights[r * N + i];
    sn_wave_lds_fence();
    sn_pdf_wave(L.a, L.w, L.cdf, L.b, N, M, p.u, p.hist_pad, lane, p.inds ? p.inds + r * (M + 1) : nullptr);
    for (int j = lane; j <= M; j += 64) p.new_bins[r * (M + 1) + j] = L.b[j];
}
