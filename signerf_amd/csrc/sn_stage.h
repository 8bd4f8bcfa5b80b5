// sn_stage.h -- ray generation, slab tests and the stage-level kernels behind the parity entry
// points of include/signerf_hip.h.  They call the SAME device functions as the fused kernels.
#pragma once
#include "sn_device.h"
#include "sn_main.h"

// ------------------------------------------------------------------------------------------
// row a5: Cameras.generate_rays: pin-hole and fisheye cameras, optional OPENCV radial-tangential
// un-distortion (the cameras of the original dataset, datasetgenerator.py:274-275), optional explicit image coordinates
// ------------------------------------------------------------------------------------------
struct SnRayGenParams {
    float c2w[12];
    float fx, fy, cx, cy;
    int height, width;
    int camera_type;      // 1 perspective, 2 fisheye (nerfstudio CameraType values)
    int has_distortion;
    float dist[6];        // k1 k2 k3 k4 p1 p2
    const float* coords;  // optional [n,2] (y, x); nullptr = pixel centres of the full image
    int64_t n;
    float* origins;
    float* directions;
    float* pixel_area;
    float* directions_norm;
    int has_aabb;
    float aabb[6];
    float* nears;
    float* fars;
};

// nerfstudio's radial_and_tangential_undistort [NS]: 10 Newton steps from the distorted point, step = 0 where |det J| <= 1e-3.
// Un-fused IEEE fp32 in the operand order of the torch expressions (oracle/nerfacto.py::radial_and_tangential_undistort).
SN_DEV void sn_undistort(const float* kk, float xd, float yd, float& xo, float& yo) {
#pragma clang fp contract(off)
    const float k1 = kk[0], k2 = kk[1], k3 = kk[2], k4 = kk[3], p1 = kk[4], p2 = kk[5];
    float x = xd, y = yd;
    for (int it = 0; it < 10; ++it) {
        const float r = x * x + y * y;
        const float d = 1.0f + r * (k1 + r * (k2 + r * (k3 + r * k4)));
        const float fx = ((d * x + ((2.0f * p1) * x) * y) + p2 * (r + (2.0f * x) * x)) - xd;
        const float fy = ((d * y + ((2.0f * p2) * x) * y) + p1 * (r + (2.0f * y) * y)) - yd;
        const float d_r = k1 + r * (2.0f * k2 + r * (3.0f * k3 + (r * 4.0f) * k4));
        const float d_x = (2.0f * x) * d_r;
        const float d_y = (2.0f * y) * d_r;
        const float fx_x = ((d + d_x * x) + (2.0f * p1) * y) + (6.0f * p2) * x;
        const float fx_y = (d_y * x + (2.0f * p1) * x) + (2.0f * p2) * y;
        const float fy_x = (d_x * y + (2.0f * p2) * y) + (2.0f * p1) * x;
        const float fy_y = ((d + d_y * y) + (2.0f * p2) * x) + (6.0f * p1) * y;
        const float den = fy_x * fx_y - fx_x * fy_y;
        const float xn = fx * fy_y - fy * fx_y;
        const float yn = fy * fx_x - fx * fy_x;
        const bool ok = fabsf(den) > 1e-3f;
        x = x + (ok ? xn / den : 0.0f);
        y = y + (ok ? yn / den : 0.0f);
    }
    xo = x;
    yo = y;
}

// image-plane point -> camera-frame direction -> world direction (d_world = R . d_cam), normalised.
// TYPE: nerfstudio CameraType value -- 1 PERSPECTIVE (u, v, -1); 2 FISHEYE (equidistant: the image-plane radius is the angle from the axis);
// 3 EQUIRECTANGULAR (the viewer's preview, signerf/interface/viewer.py:307-319; [NS-RECALL] M: theta = -pi u, phi = pi (1/2 - v),
// d = (-sin theta sin phi, cos phi, -cos theta sin phi))
template <int TYPE>
SN_DEV void sn_cam_dir(const float* c2w, float u, float v, float out[3], float& norm) {
#pragma clang fp contract(off)
    float a = u, b = v, c = -1.0f;
    if (TYPE == 2) {
        float th = sqrtf(u * u + v * v);
        th = fminf(fmaxf(th, 0.0f), 3.14159265358979323846f);
        const float st = sinf(th);
        a = (u * st) / th;
        b = (v * st) / th;
        c = -cosf(th);
    } else if (TYPE == 3) {
        const float theta = -3.14159265358979323846f * u, phi = 3.14159265358979323846f * (0.5f - v);
        const float sp = sinf(phi);
        a = -sinf(theta) * sp;
        b = cosf(phi);
        c = -cosf(theta) * sp;
    }
    float w[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) w[i] = (a * c2w[i * 4 + 0] + b * c2w[i * 4 + 1]) + c * c2w[i * 4 + 2];
    float n = sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
    // nerfstudio's camera_utils.normalize_with_norm floors the norm at its module constant _EPS = np.finfo(float).eps * 4 (8.88e-16 [NS-RECALL, M-H];
    // r01-r04 used 1e-20): it only matters for a direction shorter than that -- a degenerate camera matrix -- and the fixture decides it by data
    n = fmaxf(n, 8.8817841970012523e-16f);
#pragma unroll
    for (int i = 0; i < 3; ++i) out[i] = w[i] / n;
    norm = n;
}

template <int TYPE, bool DISTORT>
__global__ void sn_generate_rays_kernel(SnRayGenParams p) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    float x, y;
    if (p.coords) {
        y = p.coords[i * 2 + 0];
        x = p.coords[i * 2 + 1];
    } else {
        const int iy = (int)(i / p.width), ix = (int)(i % p.width);
        x = (float)ix + 0.5f;
        y = (float)iy + 0.5f;
    }
    float u = (x - p.cx) / p.fx, v = -((y - p.cy) / p.fy);
    float ux = (x - p.cx + 1.0f) / p.fx, vx = v;
    float uy = u, vy = -((y - p.cy + 1.0f) / p.fy);
    if (DISTORT) {
        const float u0 = u, v0 = v;
        sn_undistort(p.dist, u0, v0, u, v);
        sn_undistort(p.dist, ux, v0, ux, vx);
        sn_undistort(p.dist, u0, vy, uy, vy);
    }
    float d[3], dx[3], dy[3], nrm, n1, n2;
    sn_cam_dir<TYPE>(p.c2w, u, v, d, nrm);
    sn_cam_dir<TYPE>(p.c2w, ux, vx, dx, n1);
    sn_cam_dir<TYPE>(p.c2w, uy, vy, dy, n2);
    float o[3] = {p.c2w[3], p.c2w[7], p.c2w[11]};
    if (p.origins) {
        p.origins[i * 3 + 0] = o[0];
        p.origins[i * 3 + 1] = o[1];
        p.origins[i * 3 + 2] = o[2];
    }
    if (p.directions) {
        p.directions[i * 3 + 0] = d[0];
        p.directions[i * 3 + 1] = d[1];
        p.directions[i * 3 + 2] = d[2];
    }
    if (p.pixel_area) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float e = d[c] - dx[c], f = d[c] - dy[c];
            a = c == 0 ? e * e : a + e * e;
            b = c == 0 ? f * f : b + f * f;
        }
        p.pixel_area[i] = sqrtf(a) * sqrtf(b);
    }
    if (p.directions_norm) p.directions_norm[i] = nrm;
    if (p.has_aabb && p.nears && p.fars) {
        // nerfstudio intersect_aabb: clamped slab test, invalid -> 1e10 (A1)
        float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float a = (p.aabb[c] - o[c]) / d[c];
            float b = (p.aabb[3 + c] - o[c]) / d[c];
            tmin = fmaxf(tmin, fminf(a, b));
            tmax = fminf(tmax, fmaxf(a, b));
        }
        tmin = fminf(fmaxf(tmin, 0.0f), 1e10f);
        tmax = fminf(fmaxf(tmax, 0.0f), 1e10f);
        if (tmax <= tmin) {
            tmin = 1e10f;
            tmax = 1e10f;
        }
        p.nears[i] = tmin;
        p.fars[i] = tmax;
    }
}

// ------------------------------------------------------------------------------------------
// row a4: intersect_with_aabb (signerf/utils/intersection.py:5-56): 1/(d + 1e-6), no clamping
// ------------------------------------------------------------------------------------------
struct SnAabb {
    float v[6];
};

__global__ void sn_intersect_with_aabb_kernel(const float* origins, const float* directions, int64_t n, SnAabb box,
                                              float* nears, float* fars) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float nr = -INFINITY, fr = INFINITY;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float o = origins[i * 3 + c];
        float inv = 1.0f / (directions[i * 3 + c] + 1e-6f);
        float a = (box.v[c] - o) * inv;
        float b = (box.v[3 + c] - o) * inv;
        nr = fmaxf(nr, fminf(a, b));
        fr = fminf(fr, fmaxf(a, b));
    }
    nears[i] = nr;
    fars[i] = fr;
}

// ------------------------------------------------------------------------------------------
// viewer crop: nerfstudio's intersect_obb [NS] -- rays into the box frame (world2box = inverse of [R | T]),
// then the clamped slab test of intersect_aabb against [-S/2, S/2]; invalid -> 1e10 for both
// ------------------------------------------------------------------------------------------
struct SnObb {
    float w2b[12];  // 3x4 row-major
    float half[3];
};

__global__ void sn_intersect_obb_kernel(const float* origins, const float* directions, int64_t n, SnObb box, float* nears, float* fars) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float o[3] = {origins[i * 3], origins[i * 3 + 1], origins[i * 3 + 2]};
    const float d[3] = {directions[i * 3], directions[i * 3 + 1], directions[i * 3 + 2]};
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* m = box.w2b + 4 * c;
        const float ob = ((m[0] * o[0] + m[1] * o[1]) + m[2] * o[2]) + m[3];
        const float db = (m[0] * d[0] + m[1] * d[1]) + m[2] * d[2];
        const float a = (-box.half[c] - ob) / db, b = (box.half[c] - ob) / db;
        tmin = fmaxf(tmin, fminf(a, b));
        tmax = fminf(tmax, fmaxf(a, b));
    }
    tmin = fminf(fmaxf(tmin, 0.0f), 1e10f);
    tmax = fminf(fmaxf(tmax, 0.0f), 1e10f);
    if (tmax <= tmin) tmin = tmax = 1e10f;
    nears[i] = tmin;
    fars[i] = tmax;
}

// ------------------------------------------------------------------------------------------
// row a13: hash encoding of explicit normalised positions
// ------------------------------------------------------------------------------------------
struct SnHashStageParams {
    const float* q;
    int64_t n;
    const float* table;
    float scal[16];
    int num_levels, log2_t;
    float* features;   // [n, 2L]
    int32_t* indices;  // [n, L, 8] or null
    const float* pairs;  // x-paired tables: when non-null the features come from them (must equal the plain path bit for bit)
    SnPairInfo pinfo;
    uint32_t pairs_bytes;
    int grid_mode;       // 1: tiny-cuda-nn grid semantics (level table in `grid`)
    SnGridLevels grid;
    float inv_pair_scale;  // the paired tables hold the rows times a power of two (feature scale); this undoes it exactly
};

__global__ void sn_hash_encode_kernel(SnHashStageParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const float q[3] = {p.q[i * 3], p.q[i * 3 + 1], p.q[i * 3 + 2]};
    const uint32_t mask = (1u << p.log2_t) - 1u;
    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.table, ((uint32_t)p.num_levels << p.log2_t) * 8u);
    for (int l = 0; l < p.num_levels; ++l) {
        SnHashLevel hl;
        if (p.grid_mode) sn_hash_corners_tcnn<-1>(q, p.scal[l], mask, sn_grid_dense_res(p.grid, l), hl);
        else sn_hash_corners(q, p.scal[l], mask, hl);
        const uint32_t lvl = ((uint32_t)l << p.log2_t) * 8u;
        f32x2 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            v[k] = sn_table_load(rsrc, hl.boff[k], lvl);
            if (p.indices) p.indices[(i * p.num_levels + l) * 8 + k] = (int32_t)((hl.boff[k] >> 3) + ((uint32_t)l << p.log2_t));
        }
        f32x2 e = p.grid_mode ? sn_hash_blend_fast(v, hl.off) : sn_hash_blend(v, hl.off);
        if (!p.pairs) {
            p.features[i * 2 * p.num_levels + 2 * l] = e.x;
            p.features[i * 2 * p.num_levels + 2 * l + 1] = e.y;
        }
    }
    if (p.pairs) {
        const __amdgpu_buffer_rsrc_t prsrc = sn_table_rsrc(p.pairs, p.pairs_bytes);
        float feat[32];
        if (p.num_levels == 16) sn_hash_encode_pairs<16>(prsrc, p.pinfo, p.scal, p.log2_t, q, feat);
        else sn_hash_encode_pairs<5>(prsrc, p.pinfo, p.scal, p.log2_t, q, feat);
        for (int k = 0; k < 2 * p.num_levels; ++k) p.features[i * 2 * p.num_levels + k] = feat[k] * p.inv_pair_scale;
    }
}

// ------------------------------------------------------------------------------------------
// rows a14/a15: main field on explicit world positions (wave-level MFMA path, one point per lane)
// ------------------------------------------------------------------------------------------
struct SnFieldStageParams {
    const float* positions;   // [n,3] world
    const float* directions;  // [n,3] or null
    int64_t n;
    const float* table;
    const float* wimg;
    float scal[16];
    int log2_t;
    float avg_density;
    int sh_remap;
    float* density;  // [n]
    float* rgb;      // [n,3] or null
    float* geo;      // [n,15] or null: layer-2 outputs 1..15 (nerfstudio's `base_mlp_out`)
    int grid_mode;   // 1: tiny-cuda-nn grid semantics
    SnGridLevels grid;
    float feat_scale;  // power-of-two feature scale whose inverse the first layer's weights carry (both images)
    SnPosMap pm;
};

template <int PREC>
__global__ __launch_bounds__(256, 2) void sn_main_field_stage_kernel(SnFieldStageParams p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    for (int i = tid * 4; i < (PREC == 2 ? SnMainImgF16::TOTAL_FLOATS : SnMainImg::TOTAL); i += 256 * 4) *(f32x4*)(lds + i) = *(const f32x4*)(p.wimg + i);
    __syncthreads();
    const int lane = tid & 63;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + tid;
    const int64_t j = i < p.n ? i : p.n - 1;
    const float pos[3] = {p.positions[j * 3], p.positions[j * 3 + 1], p.positions[j * 3 + 2]};
    float d[3] = {0.f, 0.f, 1.f};
    if (p.directions) {
        d[0] = p.directions[j * 3];
        d[1] = p.directions[j * 3 + 1];
        d[2] = p.directions[j * 3 + 2];
    }
    SnShOps sh;
    SnShOpsH shh;
    SnShOpsF shf;
    if (PREC == 0) sh.build(d, p.sh_remap);
    else if (PREC == 2) shf.build(d, p.sh_remap);
    else shh.build(d, p.sh_remap);
    float q[3];
    const bool sel = sn_position_q(pos, q, &p.pm);
    const __amdgpu_buffer_rsrc_t rsrc = sn_table_rsrc(p.table, (16u << p.log2_t) * 8u);
    float feat[32];
    if (p.grid_mode && PREC == 2) sn_hash_encode<16, 0, 2, -1, false, 0, 0, true>(rsrc, p.scal, p.log2_t, q, feat, &p.grid);  // fp16 grid values (sn_device.h "fp16 STORAGE")
    else if (p.grid_mode) sn_hash_encode<16, 0, 2>(rsrc, p.scal, p.log2_t, q, feat, &p.grid);
    else sn_hash_encode<16>(rsrc, p.scal, p.log2_t, q, feat);
#pragma unroll
    for (int k = 0; k < 32; ++k) feat[k] *= p.feat_scale;
    float h0, rgb[3], geo16[16];
    if (PREC == 0) sn_main_field_f32<true>(lds, feat, sh, lane, h0, rgb, geo16);
    else if (PREC == 2) sn_main_field_f16<true>((const char*)lds, feat, shf, lane, h0, rgb, geo16);
    else sn_main_field_h<true>((const char*)lds, feat, shh, lane, h0, rgb, geo16);
    const bool qnan = (q[0] != q[0]) | (q[1] != q[1]) | (q[2] != q[2]);
    if (qnan) h0 = rgb[0] = rgb[1] = rgb[2] = __builtin_nanf("");
    if (i < p.n && p.geo) {
#pragma unroll
        for (int k = 0; k < 15; ++k) p.geo[i * 15 + k] = qnan ? __builtin_nanf("") : geo16[1 + k];
    }
    if (i < p.n) {
        p.density[i] = p.avg_density * expf(h0) * (sel ? 1.0f : 0.0f);
        if (p.rgb) {
            p.rgb[i * 3] = rgb[0];
            p.rgb[i * 3 + 1] = rgb[1];
            p.rgb[i * 3 + 2] = rgb[2];
        }
    }
}

// ------------------------------------------------------------------------------------------
// rows a10 + a17 on explicit per-sample inputs (one ray per thread)
// ------------------------------------------------------------------------------------------
struct SnCompositeStageParams {
    const float* bins;     // [R,S+1]
    const float* density;  // [R,S]
    const float* rgb_s;    // [R,S,3]
    int64_t n_rays;
    int n_samples;
    float* weights;
    float* rgb;
    float* depth;
    int32_t* median_index;
    float* acc;
    float* exp_raw;
    uint32_t* minmax;  // [2]
};

__global__ void sn_composite_kernel(SnCompositeStageParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_rays) return;
    const int S = p.n_samples;
    const float* b = p.bins + i * (S + 1);
    SnComposite comp;
    comp.init();
    float r = 0.f, g = 0.f, bl = 0.f, first_mid = 0.f, last_mid = 0.f;
    for (int s = 0; s < S; ++s) {
        const float t0 = b[s], t1 = b[s + 1];
        r = p.rgb_s[(i * S + s) * 3];
        g = p.rgb_s[(i * S + s) * 3 + 1];
        bl = p.rgb_s[(i * S + s) * 3 + 2];
        const float w = comp.step(s, t0, t1, p.density[i * S + s], r, g, bl);
        if (p.weights) p.weights[i * S + s] = w;
        {
#pragma clang fp contract(off)
            last_mid = (t0 + t1) / 2.0f;
        }
        if (s == 0) first_mid = last_mid;
    }
    float out[3], depth, acc, er;
    comp.finish(S, last_mid, r, g, bl, out, depth, acc, er);
    if (p.rgb) {
        p.rgb[i * 3] = out[0];
        p.rgb[i * 3 + 1] = out[1];
        p.rgb[i * 3 + 2] = out[2];
    }
    if (p.depth) p.depth[i] = depth;
    if (p.median_index) p.median_index[i] = comp.median_idx;
    if (p.acc) p.acc[i] = acc;
    if (p.exp_raw) {
        p.exp_raw[i] = er;
        atomicMin(&p.minmax[0], sn_float_ordered(first_mid));
        atomicMax(&p.minmax[1], sn_float_ordered(last_mid));
    }
}

// ------------------------------------------------------------------------------------------
// test instrumentation (r06): the three position maps of sn_device.h on explicit samples, one sample per thread.  The fused kernel
// behind the uniform sampler uses sn_sample_q_exact, whose q must BE sn_sample_q's (the literal IEEE form): compared bit for bit on the
// hardware by tests/test_gpu_stages.py (random, near-halfway and all-ones-significand cases).  The wave-uniform branch of the exact form
// needs whole waves: the grid is padded and inactive lanes compute on a dummy sample.
// ------------------------------------------------------------------------------------------
__global__ void sn_debug_sample_positions_kernel(const float* origins, const float* directions, const float* starts, const float* ends, int64_t n,
                                                 float* q_strict, float* q_exact, float* q_fast) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const int64_t j = live ? i : 0;
    float o[3], d[3], dh[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = origins[j * 3 + c];
        d[c] = directions[j * 3 + c];
        dh[c] = d[c] * 0.5f;
    }
    const float t0 = starts[j], t1 = ends[j];
    float qs[3], qe[3], qf[3];
    sn_sample_q(o, d, t0, t1, qs);
    sn_sample_q_exact(o, dh, t0, t1, qe);
    sn_sample_q_fast(o, d, t0, t1, qf);
    if (!live) return;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (q_strict) q_strict[i * 3 + c] = qs[c];
        if (q_exact) q_exact[i * 3 + c] = qe[c];
        if (q_fast) q_fast[i * 3 + c] = qf[c];
    }
}
