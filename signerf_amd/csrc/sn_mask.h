// sn_mask.h -- SURVEY.md §8(f) row 1: the step right after the render on every camera, "aabb" masking mode
// (/root/reference/signerf/datasetgenerator/datasetgenerator.py:758-818).  The reference round-trips the mask through the
// CPU for cv2.dilate (:776-778) and syncs on `torch.sum(visible_mask) > 1e-6` (:770); here everything stays on the device:
//   K-a  slab test (intersection.py:5-56) + visibility mask + count / masked-depth min,max (atomics)
//   K-b  per-row prefix counts of the mask
//   K-c  elliptical dilation as "any set pixel in a per-row run" (2 prefix lookups per structuring-element row) fused with
//        the condition image 1 - clamp((depth - dmin) / (dmax - dmin)), and the "nothing visible -> zeros" branch.
#pragma once
#include "sn_device.h"

#define SN_MASK_MAX_K 256  // largest structuring element (the reference uses (50, 50), (20, 20) before; 2 x 256 shorts of kernel arguments)

struct SnEllipse {
    int kw, kh, ax, ay;       // size and anchor (cv2 default anchor = ksize / 2)
    short j1[SN_MASK_MAX_K];  // per row: columns [j1, j2) are set
    short j2[SN_MASK_MAX_K];
};

struct SnMaskParams {
    const float* origins;
    const float* directions;
    const float* depth;
    int height, width;
    float aabb[6];
    int inverse_mask;
    int dilate;  // 0: mask = visible mask
    SnEllipse el;
    int has_manual_depth;
    float manual_min, manual_range, depth_radius;  // manual_range = (float)((double)max - (double)min), formed on the host
    uint8_t* vis;      // [H*W] scratch
    int32_t* prefix;   // [H][W+1] scratch
    uint32_t* stats;   // [0] count, [1] ordered min depth, [2] ordered max depth
    uint8_t* mask;     // [H*W] out
    float* condition;  // [H*W] out
};

// Grid-stride over the pixels, ONE set of atomics per workgroup: the three statistics are single words that every contribution has to
// reach, and r02's profile of the 58-view loop showed the earlier one-set-per-wave form (10 000 waves x 3 same-address atomics at
// 800x800) serialised in L2 for 339 us per view -- 7 % of a view, 20x the dilation kernel.  Integer count and ordered-uint min / max:
// order-independent, so the results are unchanged.
#define SN_MASK_VIS_BLOCKS 512
__global__ __launch_bounds__(256) void sn_mask_visible_kernel(SnMaskParams p) {
    __shared__ uint32_t red[3][4];
    const int64_t n = (int64_t)p.height * p.width;
    uint32_t cnt = 0u, lo = 0xffffffffu, hi = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
        float nr = -INFINITY, fr = INFINITY;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float o = p.origins[i * 3 + c];
            const float inv = 1.0f / (p.directions[i * 3 + c] + 1e-6f);
            const float a = (p.aabb[c] - o) * inv, b = (p.aabb[3 + c] - o) * inv;
            nr = fmaxf(nr, fminf(a, b));
            fr = fminf(fr, fmaxf(a, b));
        }
        const float dep = p.depth[i];
        const bool non_empty = (nr < fr) && (nr > 0.0f);  // FIXME in the reference: cameras inside the box are ignored
        bool vis = (nr < dep) && (dep < fr) && non_empty;
        if (p.inverse_mask) vis = !vis;
        p.vis[i] = vis ? 1 : 0;
        cnt += vis ? 1u : 0u;
        if (vis && (dep * 1.0f > 0.0f)) {  // depth[(depth * visible_mask) > 0]
            const uint32_t od = sn_float_ordered(dep);
            lo = min(lo, od);
            hi = max(hi, od);
        }
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
        cnt += (uint32_t)__shfl_xor((int)cnt, s);
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, s));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, s));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wave] = cnt;
        red[1][wave] = lo;
        red[2][wave] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        cnt = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        lo = min(min(red[1][0], red[1][1]), min(red[1][2], red[1][3]));
        hi = max(max(red[2][0], red[2][1]), max(red[2][2], red[2][3]));
        if (cnt) atomicAdd(&p.stats[0], cnt);
        if (lo != 0xffffffffu) {
            atomicMin(&p.stats[1], lo);
            atomicMax(&p.stats[2], hi);
        }
    }
}

// one wave per row: prefix[y][x] = number of set pixels in row y at columns < x
__global__ __launch_bounds__(64) void sn_mask_prefix_kernel(SnMaskParams p) {
    const int y = blockIdx.x, lane = threadIdx.x;
    const uint8_t* row = p.vis + (int64_t)y * p.width;
    int32_t* out = p.prefix + (int64_t)y * (p.width + 1);
    int carry = 0;
    if (lane == 0) out[0] = 0;
    for (int c = 0; c < p.width; c += 64) {
        const int x = c + lane;
        int v = x < p.width ? row[x] : 0;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            const int t = __shfl_up(v, s);
            if (lane >= s) v += t;
        }
        if (x < p.width) out[x + 1] = carry + v;
        carry += __shfl(v, 63);
    }
}

__global__ void sn_mask_condition_kernel(SnMaskParams p) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)p.height * p.width;
    if (i >= n) return;
    const uint32_t count = p.stats[0];
    if (count == 0) {  // is_visible == False: zero mask, zero condition (datasetgenerator.py:812-818)
        p.mask[i] = 0;
        if (p.condition) p.condition[i] = 0.0f;
        return;
    }
    const int y = (int)(i / p.width), x = (int)(i % p.width);
    bool m = p.vis[i] != 0;
    if (p.dilate) {
        // cv2.dilate: dst(x,y) = max over set (i,j) of src(x + j - ax, y + i - ay); pixels outside the image do not contribute
        m = false;
        for (int r = 0; r < p.el.kh && !m; ++r) {
            const int yy = y + r - p.el.ay;
            if (yy < 0 || yy >= p.height) continue;
            const int a = max(x + p.el.j1[r] - p.el.ax, 0), b = min(x + p.el.j2[r] - p.el.ax, p.width);
            if (a >= b) continue;
            const int32_t* pr = p.prefix + (int64_t)yy * (p.width + 1);
            m = pr[b] - pr[a] > 0;
        }
    }
    p.mask[i] = m ? 1 : 0;
    if (p.condition) {
#pragma clang fp contract(off)
        float dmin, range;
        if (p.has_manual_depth) {
            dmin = p.manual_min;
            range = p.manual_range;
        } else {
            // torch.min / torch.max of an empty selection raise in the reference; here an empty selection (all visible depths
            // <= 0) leaves the sentinels, which map to -inf / +inf -> condition NaN-free but meaningless.  Not reachable with
            // positive depths.
            dmin = sn_ordered_float(p.stats[1]) - p.depth_radius;
            const float dmax = sn_ordered_float(p.stats[2]) + p.depth_radius;
            range = dmax - dmin;
        }
        const float dn = (p.depth[i] - dmin) / range;
        // torch.clamp keeps a NaN (the depth of a ray that missed render_aabb, inf / inf of an unbounded selection); fminf / fmaxf would drop it
        p.condition[i] = 1.0f - (dn != dn ? dn : fminf(fmaxf(dn, 0.0f), 1.0f));
    }
}

// tensor_to_image (image_tensor_converter.py:21-23,28-30): x * 255 then numpy's astype(uint8) -- truncation toward zero,
// no clamp; out-of-range values wrap like the x86 cvttss2si + byte-truncate sequence numpy compiles to.
__global__ void sn_tensor_to_uint8_kernel(const float* __restrict__ in, int64_t n, uint8_t* __restrict__ out) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = in[i] * 255.0f;
    int iv;
    if (v != v || v >= 2147483648.0f || v < -2147483648.0f) iv = (int)0x80000000;  // cvttss2si "integer indefinite"
    else iv = (int)v;
    out[i] = (uint8_t)(iv & 0xff);
}

// ------------------------------------------------------------------------------------------
// F.interpolate(mode="bilinear", align_corners=False) between [h, w, C] windows of channel-last images (SURVEY §8(f) row 1:
// the 1/2 downscale + paste into the reference sheet, datasetgenerator.py:526-539, and the upscale of the edited cell, :586).
// PyTorch's arithmetic (ATen UpSample.h, area_pixel_compute_source_index + guard_index_and_lambda), fp32:
//   scale = in / out;  src = max(fma(scale, dst + 0.5, -0.5), 0);  i0 = min(int(src), in - 1);  i1 = min(i0 + 1, in - 1);
//   l1 = clamp(src - i0, 0, 1);  l0 = 1 - l1;   out = h0 * (w0 * p00 + w1 * p01) + h1 * (w0 * p10 + w1 * p11)
// ------------------------------------------------------------------------------------------
struct SnResizeParams {
    const void* src;
    float* dst;
    int src_u8;  // source elements are uint8 (the 0/1 mask) instead of fp32
    int src_h, src_w, dst_h, dst_w, channels;
    int64_t src_row_stride, dst_row_stride;  // in elements
    int threshold;                           // write (value > 0.5) as 1.0 / 0.0 (mask_scaled, :527)
};

SN_DEV void sn_resize_axis(int dst_index, int in_size, int out_size, int& i0, int& i1, float& l0, float& l1) {
#pragma clang fp contract(off)
    const float scale = (float)in_size / (float)out_size;
    float src = fmaf(scale, (float)dst_index + 0.5f, -0.5f);  // one rounding, as the compiled ATen kernels (x86 FMA / nvcc fmad) do
    if (src < 0.0f) src = 0.0f;
    i0 = min((int)src, in_size - 1);
    i1 = min(i0 + 1, in_size - 1);
    l1 = fminf(fmaxf(src - (float)i0, 0.0f), 1.0f);
    l0 = 1.0f - l1;
}

__global__ void sn_resize_bilinear_kernel(SnResizeParams p) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)p.dst_h * p.dst_w * p.channels;
    if (i >= n) return;
    const int c = (int)(i % p.channels);
    const int x = (int)((i / p.channels) % p.dst_w);
    const int y = (int)(i / ((int64_t)p.channels * p.dst_w));
    int y0, y1, x0, x1;
    float h0, h1, w0, w1;
    sn_resize_axis(y, p.src_h, p.dst_h, y0, y1, h0, h1);
    sn_resize_axis(x, p.src_w, p.dst_w, x0, x1, w0, w1);
    auto at = [&](int yy, int xx) -> float {
        const int64_t o = (int64_t)yy * p.src_row_stride + (int64_t)xx * p.channels + c;
        return p.src_u8 ? (float)((const uint8_t*)p.src)[o] : ((const float*)p.src)[o];
    };
    const float top = w0 * at(y0, x0) + w1 * at(y0, x1);
    const float bot = w0 * at(y1, x0) + w1 * at(y1, x1);
    float v = h0 * top + h1 * bot;
    if (p.threshold) v = v > 0.5f ? 1.0f : 0.0f;
    p.dst[(int64_t)y * p.dst_row_stride + (int64_t)x * p.channels + c] = v;
}
