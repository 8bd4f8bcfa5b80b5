// sn_api.hip -- C ABI of libsignerf_hip.so (see include/signerf_hip.h for the contract and the
// reference interfaces each entry point stands in for).  gfx950 only.
#include "../../include/signerf_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "sn_device.h"
#include "sn_main.h"
#include "sn_mask.h"
#include "sn_normals.h"
#include "sn_proposal.h"
#include "sn_stage.h"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void* ptr = nullptr;
    size_t bytes = 0;
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
};

}  // namespace

struct SnContext {
    SnFieldDesc desc;
    int device = 0;
    int n_cus = 256;  // compute units of the device (the split-depth tail plan of the main kernel sizes its last round from it)
    std::mutex mu;  // guards weights + error text only; render calls do not take it
    std::string error;
    std::map<std::string, std::vector<float>> host;  // small tensors (MLP layers, appearance mean)
    DevBuf table_main;
    DevBuf table_prop[SN_MAX_PROPOSALS];
    DevBuf pairs_prop[SN_MAX_PROPOSALS];  // x-paired copies of the proposal tables (sn_device.h), rebuilt by sn_finalize_weights
    SnPairInfo pinfo_prop[SN_MAX_PROPOSALS];
    DevBuf wimg_main;                   // SnMainImg (fp32 MFMA operands)
    DevBuf wimg_main_h;                 // SnMainImgH (fp16 hi+lo MFMA operands)
    DevBuf wpack_prop[SN_MAX_PROPOSALS]; // SnPropPack
    DevBuf wimg_normals;                // SnNormImg (sn_normals.h); built when the weights are finalized
    DevBuf wimg_normals_h;              // SnNormImgH: its fp16 hi+lo form
    bool has_pred_normals = false;      // field.mlp_pred_normals.* / field.field_head_pred_normals.* were uploaded
    DevBuf pairs_main;            // x-paired copies of the main grid's levels that have no de-hashed copy (SN_MAIN_PAIRS, sn_main.h)
    SnPairInfo pinfo_main{};
    DevBuf dense_main;            // de-hashed copies of the coarse levels of a torch-path main grid (sn_device.h SnDenseCopy)
    SnDenseCopy dense_info{};
    SnGridLevels dense_res{};     // their resolutions R = scale + 2, packed like a tcnn level table
    int nd_torch = 0;             // number of copied levels
    DevBuf hquads_main, hrows_main;   // SnFieldDesc.half_grid: fp16 storage of a tiny-cuda-nn main grid (sn_device.h "fp16 STORAGE")
    SnDenseCopy hquads_info{};        // (quads of the levels [0, nd_torch); hrows_main: x-pairs / rows of the levels [nd_torch, L))
    SnPairInfo hpinfo_main{};
    float table_absmax_main = 0.0f;   // max |value| of the uploaded main table (sn_finalize_weights)
    DevBuf dense_prop[SN_MAX_PROPOSALS];
    SnDenseCopy dense_info_prop[SN_MAX_PROPOSALS]{};
    SnGridLevels dense_res_prop[SN_MAX_PROPOSALS]{};
    int nd_prop[SN_MAX_PROPOSALS] = {0, 0};
    // Range conditioning of the split-precision (fp16 hi + lo) MLPs, decided by sn_finalize_weights (plan_split_scales):
    float feat_scale_main = 1.0f;                      // power of two carried by the main grid's de-hashed copies (1 for tcnn grids)
    float feat_scale_prop[SN_MAX_PROPOSALS] = {1.0f, 1.0f};  // ... by a proposal net's de-hashed copies and paired tables
    bool split_ok = true;        // false: some scaled weight leaves the fp16 range -> precision 1 requests render with exact fp32 MFMA
    std::string split_why;
    bool normals_split_ok = true;  // the normals kernel's own conditioned operands fit fp16 (sn_finalize_weights)
    float grad_scale_normals = 1.0f;  // power of two carried by the split-precision reverse-pass layer of the normals kernel
    bool finalized = false;
    std::atomic<uint64_t> weights_epoch{0};  // advanced by every sn_upload_weights / sn_finalize_weights (the workspace stamps carry it)
    uint64_t id = 0;                         // process-unique handle number (a stamp must not match a NEW handle at a recycled address)
    SnPosMap pos_map{};  // SnFieldDesc.disable_scene_contraction + aabb, as the kernels take it (sn_create)
    // ordering of weight uploads against renders in flight (RenderGuard below): the completion event of the LAST render of every
    // stream that rendered with this handle (a stream is in-order, so its last render covers its earlier ones)
    static constexpr size_t kMaxRenderStreams = 64;
    std::mutex ev_mu;
    std::map<hipStream_t, hipEvent_t> render_ev;
    std::vector<hipEvent_t> spare_ev;
    hipEvent_t weights_ev = nullptr;
    bool weights_ev_recorded = false;
    // diagnostic / test switches of the environment, read when the handle is created, when its weights are finalized and by
    // sn_debug_reload_env -- not by every render call (getenv on the render path of several threads)
    struct Switches {
        std::atomic<int> render_chain{0}, prop_cache_off{0}, pdf_ieee{0},  tail_split_off{0}, early_term{1};
    } sw;
};

#ifndef SN_DENSE_LEVELS_DEFAULT
#define SN_DENSE_LEVELS_DEFAULT 11
#endif

namespace {

// Renders are NOT ordered against each other: a handle is re-entrant (no state of a render lives in it, scratch is the caller's),
// so renders issued from several host threads / HIP streams overlap on the GPU.  History (r01, tools/concurrency_probe.py): two
// renders in flight on two hardware queues corrupted lanes 48-63 of a few tiles per frame; the cause was an instruction hazard in
// the proposal MLP that only the mixed-kernel issue pattern exposes (sn_proposal.h, sn_prop_h0), fixed there; r02 removed the packed-
// fp32 instructions of that hazard family from the fused kernels altogether.  The device-side render chain that r01 kept "as a second
// line of defence" is now an OPT-IN diagnostic: SN_RENDER_CHAIN=1 makes every render of the process wait for the previous one's
// completion event on its device (the tests run both ways).
struct RenderChain {
    std::mutex mu;
    hipEvent_t ev[16] = {};
    bool recorded[16] = {};
};
RenderChain g_chain;

// What IS ordered, per handle: weight uploads against the renders that read the buffers they overwrite.
//   * every render makes its stream wait for the handle's last upload / finalize (weights_ev) and, when its kernels are enqueued,
//     records its STREAM's render event (SnContext::render_ev: one event per stream, re-recorded by every render of that stream);
//   * sn_upload_weights / sn_finalize_weights make their stream wait for all of those before touching a buffer.
int env_int(const char* name) {
    const char* e = getenv(name);
    return e ? atoi(e) : 0;
}
void load_switches(SnHandle h) {
    h->sw.render_chain = env_int("SN_RENDER_CHAIN") != 0;
    h->sw.prop_cache_off = env_int("SN_PROP_CACHE_OFF") != 0;  // test switch, see SnPropParams::cache_off
    h->sw.pdf_ieee = env_int("SN_PDF_IEEE") != 0;              // test switch, see SnPropParams::pdf_ieee
    {
        const char* e = getenv("SN_TAIL_SPLIT");                  // test / A-B switch: SN_TAIL_SPLIT=0 renders the last round of workgroups whole
        h->sw.tail_split_off = e && atoi(e) == 0;
    }
    {
        const char* e = getenv("SN_EARLY_TERM");   // 0: every sample of every ray is evaluated (bit-identical outputs; A/B and test switch)
        h->sw.early_term = e ? (atoi(e) != 0) : 1;
    }
}

struct RenderGuard {
    SnHandle h;
    hipStream_t st;
    int dev;
    bool chain;
    RenderGuard(SnHandle h_, hipStream_t s) : h(h_), st(s), dev(h_->device & 15) {
        chain = h->sw.render_chain.load(std::memory_order_relaxed) != 0;
        {
            std::lock_guard<std::mutex> g(h->ev_mu);
            if (h->weights_ev_recorded) (void)hipStreamWaitEvent(st, h->weights_ev, 0);
        }
        if (chain) {
            std::lock_guard<std::mutex> g(g_chain.mu);
            if (g_chain.recorded[dev]) (void)hipStreamWaitEvent(st, g_chain.ev[dev], 0);
        }
    }
    ~RenderGuard() {
        {
            std::lock_guard<std::mutex> g(h->ev_mu);
            auto it = h->render_ev.find(st);
            if (it == h->render_ev.end()) {
                hipEvent_t ev = nullptr;
                if (h->render_ev.size() >= SnContext::kMaxRenderStreams) {
                    // too many distinct streams (streams created and dropped per call): retire another stream's entry WITHOUT losing
                    // its render -- this stream first waits for it, so the event recorded below completes after both (ADVICE r02:
                    // an evicted event must not be dropped silently)
                    auto victim = h->render_ev.begin();
                    (void)hipStreamWaitEvent(st, victim->second, 0);
                    ev = victim->second;
                    h->render_ev.erase(victim);
                } else if (!h->spare_ev.empty()) {
                    ev = h->spare_ev.back();
                    h->spare_ev.pop_back();
                } else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
                    ev = nullptr;
                }
                if (ev) it = h->render_ev.emplace(st, ev).first;
            }
            if (it != h->render_ev.end() && hipEventRecord(it->second, st) != hipSuccess) {
                // cannot track this render: fall back to the only safe thing
                (void)hipStreamSynchronize(st);
            } else if (it == h->render_ev.end()) {
                (void)hipStreamSynchronize(st);
            }
        }
        if (chain) {
            std::lock_guard<std::mutex> g(g_chain.mu);
            if (!g_chain.ev[dev] && hipEventCreateWithFlags(&g_chain.ev[dev], hipEventDisableTiming) != hipSuccess) return;
            if (hipEventRecord(g_chain.ev[dev], st) == hipSuccess) g_chain.recorded[dev] = true;
        }
    }
};

// upload / finalize side of the per-handle ordering: wait (stream-side) for the last render of EVERY stream
void wait_for_renders(SnHandle h, hipStream_t st) {
    std::lock_guard<std::mutex> g(h->ev_mu);
    for (auto& kv : h->render_ev) (void)hipStreamWaitEvent(st, kv.second, 0);
}
void mark_weights_written(SnHandle h, hipStream_t st) {
    std::lock_guard<std::mutex> g(h->ev_mu);
    if (!h->weights_ev && hipEventCreateWithFlags(&h->weights_ev, hipEventDisableTiming) != hipSuccess) return;
    if (hipEventRecord(h->weights_ev, st) == hipSuccess) h->weights_ev_recorded = true;
}

thread_local std::string g_error_copy;  // sn_last_error hands out a per-thread copy: another thread may rewrite the handle's text

int fail(SnHandle h, int code, const std::string& msg);

// include/signerf_hip.h "ABI evolution": the caller's struct begins with the sizeof ITS header gives it.  Exactly that many bytes are
// copied into the library's own (zeroed) struct: fields the caller does not know keep their zero defaults, nothing behind the caller's
// struct is read.  min_size = the struct's size in the first versioned header (r06).
template <typename T>
int adopt_struct(SnHandle h, const T* in, size_t min_size, T& out, const char* what) {
    memset((void*)&out, 0, sizeof(T));
    uint32_t sz = 0;
    memcpy(&sz, (const void*)in, sizeof(sz));
    if (sz < min_size || sz > sizeof(T))
        return fail(h, SN_ERR_INVALID, std::string(what) + ".struct_size = " + std::to_string(sz) + ": this library (SN_ABI_VERSION " + std::to_string(SN_ABI_VERSION) +
                                           ") knows sizes " + std::to_string(min_size) + " .. " + std::to_string(sizeof(T)) +
                                           (sz == 0 ? " -- struct_size was not set (set it to sizeof of the struct in your header)"
                                                    : sz > sizeof(T) ? " -- the caller was built against a newer header than the library" : ""));
    memcpy((void*)&out, (const void*)in, sz);
    out.struct_size = (uint32_t)sizeof(T);
    return SN_OK;
}
// Smallest struct_size accepted.  r06's header is the first versioned one, so no binding with a shorter struct exists yet; the lower bounds
// are nevertheless set where the r04 / r05 appendices begin, so that the short-struct path -- the one a future appendix will rely on -- is a
// path the tests execute today (tests/test_cabi.py, tests/test_gpu_abi.py: a caller that declares the shorter size and has garbage, a wild
// pointer included, in the memory behind it).
constexpr size_t kFieldDescMin = offsetof(SnFieldDesc, dense_levels);      // ... up to the r03 fields (disable_scene_contraction, aabb)
constexpr size_t kRenderOptsMin = offsetof(SnRenderOpts, march_stats);     // ... up to the r03 fields (background, spacing_mode)
constexpr size_t kMaskOptsMin = offsetof(SnMaskOpts, additional_depth_radius) + sizeof(float);
constexpr size_t kDebugLayoutMin = offsetof(SnDebugLayout, table_bytes);   // ... up to feature_scale

// SnRenderOpts.reuse_final_bins: what the final bins in a workspace belong to, kept on the HOST per workspace address (the library
// cannot read a workspace back without a device sync).  sn_render_rays stamps the workspace it wrote bins into; every other entry point
// that is handed a workspace clears its stamp; sn_render_normals with reuse_final_bins compares.  Process-wide, so that a render of
// ANOTHER handle into the same memory invalidates the stamp as well.
struct WorkspaceStamp {
    uint64_t handle_id, weights_epoch, serial;
    int32_t height, width, nprop, n_prop_samples[SN_MAX_PROPOSALS], n_nerf, spacing_mode;
    float near_plane, far_plane;
    const void *origins, *directions, *nears, *fars, *sbins, *pdf_u[SN_MAX_PROPOSALS];
    int device;
};
struct StampTable {
    std::mutex mu;
    std::unordered_map<const void*, WorkspaceStamp> m;
    uint64_t serial = 0;
    static constexpr size_t kMax = 1024;
};
StampTable g_stamps;
std::atomic<uint64_t> g_next_handle_id{1};

WorkspaceStamp make_stamp(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars, int32_t height,
                          int32_t width, const SnRenderOpts& o) {
    WorkspaceStamp s;
    memset(&s, 0, sizeof(s));
    s.handle_id = h->id;
    s.weights_epoch = h->weights_epoch.load(std::memory_order_relaxed);
    s.height = height;
    s.width = width;
    s.nprop = o.num_proposal_iterations;
    for (int i = 0; i < SN_MAX_PROPOSALS; ++i) {
        s.n_prop_samples[i] = i < o.num_proposal_iterations ? o.num_proposal_samples[i] : 0;
        s.pdf_u[i] = i < o.num_proposal_iterations ? o.pdf_u[i] : nullptr;
    }
    s.n_nerf = o.num_nerf_samples;
    s.spacing_mode = o.spacing_mode;
    s.near_plane = o.near_plane;
    s.far_plane = o.far_plane;
    s.origins = origins;
    s.directions = directions;
    s.nears = nears;
    s.fars = fars;
    s.sbins = o.initial_spacing_bins;
    s.device = h->device;
    return s;
}
void stamp_workspace(const void* ws, WorkspaceStamp s) {
    std::lock_guard<std::mutex> g(g_stamps.mu);
    if (g_stamps.m.size() >= StampTable::kMax && !g_stamps.m.count(ws)) {  // bounded: drop the oldest entry
        auto oldest = g_stamps.m.begin();
        for (auto it = g_stamps.m.begin(); it != g_stamps.m.end(); ++it)
            if (it->second.serial < oldest->second.serial) oldest = it;
        g_stamps.m.erase(oldest);
    }
    s.serial = ++g_stamps.serial;
    g_stamps.m[ws] = s;
}
void clear_stamp(const void* ws) {
    if (!ws) return;
    std::lock_guard<std::mutex> g(g_stamps.mu);
    g_stamps.m.erase(ws);
}
// empty string = the workspace holds the bins this call would compute; otherwise what differs
std::string stamp_mismatch(const void* ws, const WorkspaceStamp& want) {
    std::lock_guard<std::mutex> g(g_stamps.mu);
    auto it = g_stamps.m.find(ws);
    if (it == g_stamps.m.end()) return "no sn_render_rays call of this process left final bins in this workspace (or another call has used it since)";
    const WorkspaceStamp& s = it->second;
    if (s.handle_id != want.handle_id || s.device != want.device) return "the bins in this workspace were written by another handle";
    if (s.weights_epoch != want.weights_epoch) return "the handle's weights changed after the render that wrote these bins";
    if (s.height != want.height || s.width != want.width) return "the bins belong to a frame of another size";
    if (s.nprop != want.nprop || s.n_nerf != want.n_nerf || memcmp(s.n_prop_samples, want.n_prop_samples, sizeof(s.n_prop_samples)) != 0)
        return "the bins were sampled with other sample counts";
    if (s.spacing_mode != want.spacing_mode || memcmp(&s.near_plane, &want.near_plane, 4) != 0 || memcmp(&s.far_plane, &want.far_plane, 4) != 0)
        return "the bins were sampled with another initial sampler / collider planes";
    if (s.origins != want.origins || s.directions != want.directions || s.nears != want.nears || s.fars != want.fars)
        return "the bins belong to another ray bundle (origins / directions / nears / fars pointers differ)";
    if (s.sbins != want.sbins || memcmp(s.pdf_u, want.pdf_u, sizeof(s.pdf_u)) != 0) return "the bins were sampled on other sampler grids";
    return "";
}

int fail(SnHandle h, int code, const std::string& msg) {
    if (h) {
        std::lock_guard<std::mutex> g(h->mu);
        h->error = msg;
    } else {
        g_create_error = msg;
    }
    return code;
}

#define SN_HIP(h, expr)                                                                              \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return fail(h, SN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

bool check_hashmlp(const SnHashMlpDesc& d, int levels, int hidden, int out, std::string& why) {
    if (d.num_levels != levels) why = "num_levels must be " + std::to_string(levels);
    else if (d.features_per_level != 2) why = "features_per_level must be 2";
    else if (d.log2_hashmap_size < 4 || d.log2_hashmap_size > 21) why = "log2_hashmap_size out of range [4,21]";
    else if (d.hidden_dim != hidden) why = "hidden_dim must be " + std::to_string(hidden);
    else if (d.num_layers != 2) why = "num_layers must be 2";
    else if (d.out_dim != out) why = "out_dim must be " + std::to_string(out);
    else if (d.grid_mode != 0 && d.grid_mode != 1) why = "grid_mode must be 0 (torch HashEncoding) or 1 (tiny-cuda-nn)";
    else return true;
    return false;
}

// tiny-cuda-nn level table (grid_mode 1): resolution = ceil(scale) + 1; a level is dense when its whole grid fits the table
SnGridLevels grid_levels(const SnHashMlpDesc& d) {
    SnGridLevels g;
    memset(&g, 0, sizeof(g));
    if (d.grid_mode != 1) return g;
    const uint64_t T = 1ull << d.log2_hashmap_size;
    for (int l = 0; l < d.num_levels; ++l) {
        const uint64_t res = (uint64_t)ceilf(d.scalings[l]) + 1;
        const uint64_t n = res * res * res;
        if (n <= T && res <= 255) g.packed[l >> 2] |= (uint32_t)res << ((l & 3) * 8);
    }
    return g;
}

// De-hashed copies of the leading levels of a torch-path grid (sn_device.h, SnDenseCopy).  A level is copied while its copy
// stays under `cap_mb` (a fine level's copy is R^3 / T times larger than its hashed slot) and 8 R^2 fits 24 bits.  Measured r01:
// the main grid (T = 2^19, one pass of 48-64 samples per ray) still gains from level 10 (R = 408, 543 MB; same-box 3.85 / 3.55 /
// 3.33 / 3.30 / 3.26 ms for 0 / 8 / 9 / 10 / 11 copied levels), the proposal nets (352 samples per ray over the coarse levels)
// lose with the 137 MB copy of the second net's finest level (frame 18.8 vs 17.5 ms) -- hence the two caps.
// how many of the `want` leading levels fit the per-level cap (the first level above it ends the run; build_dense_copies applies the same rule)
int dense_levels_under_cap(const SnHashMlpDesc& d, int want, uint64_t cap_mb) {
    int nd = 0;
    for (int l = 0; l < want && l < d.num_levels && l < 12; ++l) {
        const uint64_t r = d.grid_mode == 0 ? (uint64_t)d.scalings[l] + 2 : (uint64_t)floorf(d.scalings[l] + 0.5f) + 2;
        if (r > 700 || r * r * r * 8 > cap_mb * 1000 * 1000) break;
        ++nd;
    }
    return nd;
}

int build_dense_copies(SnHandle h, const SnHashMlpDesc& d, const DevBuf& table, int want, uint64_t cap_mb, DevBuf& buf, SnDenseCopy& info,
                       SnGridLevels& res, int& nd_out, hipStream_t st, int bc_levels, float scale = 1.0f) {
    nd_out = 0;
    memset(&info, 0, sizeof(info));
    memset(&res, 0, sizeof(res));
    if (want <= 0 || !table.ptr) {
        buf.release();
        return SN_OK;
    }
    const SnGridLevels tcnn_dense = grid_levels(d);  // tiny-cuda-nn: resolution of the levels it indexes densely (all 0 for torch grids)
    uint64_t bytes = 0;
    uint32_t R[12];
    int nd = 0;
    for (int l = 0; l < want && l < d.num_levels && l < 12; ++l) {
        // grid points per axis a sample can touch: torch x = q scale in [0, scale) -> floor + 1 <= scale + 1; tcnn x = scale q + 0.5
        const uint64_t r = d.grid_mode == 0 ? (uint64_t)d.scalings[l] + 2 : (uint64_t)floorf(d.scalings[l] + 0.5f) + 2;
        if (r > 700 || r * r * r * 8 > cap_mb * 1000 * 1000) break;  // 32 R^2 (bilinear-coefficient entries) must fit 24 bits
        R[l] = (uint32_t)r;
        info.off[l] = (uint32_t)bytes;
        // bilinear-coefficient levels: 32 bytes per grid point; plain-row levels: 8, plus one spare row (the last entry's 16-byte read)
        bytes += l < bc_levels ? r * r * r * 32 : (r * r * r + 1) * 8;
        bytes = (bytes + 255) & ~255ull;
        ++nd;
    }
    if (nd == 0) return SN_OK;
    if (bytes >= 0xf0000000ull) return fail(h, SN_ERR_INVALID, "de-hashed copies exceed the 32-bit buffer-offset range");
    if (buf.bytes != bytes) {
        buf.release();
        SN_HIP(h, hipMalloc(&buf.ptr, bytes));
        buf.bytes = bytes;
    }
    SN_HIP(h, hipMemsetAsync(buf.ptr, 0, bytes, st));
    for (int l = 0; l < nd; ++l) {
        const uint32_t n = R[l] * R[l] * R[l];
        float* dst = (float*)((char*)buf.ptr + info.off[l]);
        const uint32_t dres = (tcnn_dense.packed[l >> 2] >> ((l & 3) * 8)) & 0xffu;
        if (l < bc_levels)
            hipLaunchKernelGGL(sn_build_bc_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)table.ptr, dst, l, d.log2_hashmap_size, R[l], scale, dres);
        else
            hipLaunchKernelGGL(sn_build_dense_copy_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)table.ptr, dst, l, d.log2_hashmap_size, R[l], scale, dres);
        info.res[l] = R[l];
    }
    SN_HIP(h, hipGetLastError());
    info.base = (const float*)buf.ptr;
    info.bytes = (uint32_t)bytes;
    info.n_bc = (uint32_t)std::min(nd, std::max(bc_levels, 0));
    nd_out = nd;
    return SN_OK;
}

// fp16 storage of a tiny-cuda-nn grid for the single-fp16 mode (SnFieldDesc.half_grid; sn_device.h "fp16 STORAGE"): quads of the `nd` levels
// that have a de-hashed copy (same R), 4-byte rows of the others.  `absmax` = max |table value| (finite, checked by the caller).
int build_half_grid(SnHandle h, const SnHashMlpDesc& d, const DevBuf& table, const SnDenseCopy& copies, int nd, DevBuf& quads, SnDenseCopy& qinfo,
                    DevBuf& rows, SnPairInfo& pinfo, hipStream_t st, float scale, float absmax) {
    memset(&qinfo, 0, sizeof(qinfo));
    memset(&pinfo, 0, sizeof(pinfo));
    if (nd <= 0 || nd > 12 || !table.ptr) {
        quads.release();
        rows.release();
        return SN_OK;
    }
    if (!((double)absmax * scale <= 65504.0))
        return fail(h, SN_ERR_INVALID, "half_grid: the scaled table values leave fp16's range (max |value| " + std::to_string(absmax) + " x feature scale " +
                                           std::to_string(scale) + ")");
    const SnGridLevels tcnn_dense = grid_levels(d);
    uint64_t bytes = 0;
    for (int l = 0; l < nd; ++l) {
        const uint64_t r = copies.res[l];
        if (r >= 1024) return fail(h, SN_ERR_INVALID, "half_grid: a copied level is wider than 1023 grid points");
        qinfo.off[l] = (uint32_t)bytes;
        qinfo.res[l] = (uint32_t)r;
        bytes += r * r * r * 16;
        bytes = (bytes + 255) & ~255ull;
    }
    if (bytes >= 0xf0000000ull) return fail(h, SN_ERR_INVALID, "half_grid: the quads exceed the 32-bit buffer-offset range");
    if (quads.bytes != bytes) {
        quads.release();
        SN_HIP(h, hipMalloc(&quads.ptr, bytes));
        quads.bytes = bytes;
    }
    for (int l = 0; l < nd; ++l) {
        const uint32_t n = qinfo.res[l] * qinfo.res[l] * qinfo.res[l];
        const uint32_t dres = (tcnn_dense.packed[l >> 2] >> ((l & 3) * 8)) & 0xffu;
        hipLaunchKernelGGL(sn_build_quad_h16_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (const float*)table.ptr,
                           (uint32_t*)((char*)quads.ptr + qinfo.off[l]), l, d.log2_hashmap_size, qinfo.res[l], scale, dres);
    }
    qinfo.base = (const float*)quads.ptr;
    qinfo.bytes = (uint32_t)bytes;
    qinfo.n_bc = 0;
    const int rest = d.num_levels - nd;
#if SN_H16_PAIRS
    // the hashed levels as x-pairs: per level one table per count t of trailing one bits of the floor-x coordinate (build_pairs), 8-byte entries
    const uint32_t T = 1u << d.log2_hashmap_size;
    uint64_t entries = 0;
    int n_t[SN_MAX_LEVELS] = {};
    for (int l = nd; l < d.num_levels; ++l) {
        int bits = 0;
        for (uint32_t s = (uint32_t)ceilf(d.scalings[l]) + 1u; s; s >>= 1) ++bits;
        n_t[l] = bits + 1;
        pinfo.base[l] = (uint32_t)entries;
        entries += (uint64_t)n_t[l] * T;
    }
    const uint64_t rbytes = std::max<uint64_t>(entries * 8, 256);
    if (rbytes >= (1ull << 32)) return fail(h, SN_ERR_INVALID, "half_grid: the paired fp16 tables exceed the 4 GiB buffer-descriptor range");
#else
    const uint64_t rbytes = std::max<uint64_t>(((uint64_t)std::max(rest, 0) << d.log2_hashmap_size) * 4, 256);
#endif
    if (rows.bytes != rbytes) {
        rows.release();
        SN_HIP(h, hipMalloc(&rows.ptr, rbytes));
        rows.bytes = rbytes;
    }
#if SN_H16_PAIRS
    for (int l = nd; l < d.num_levels; ++l) {
        const uint64_t n = (uint64_t)n_t[l] * T;
        hipLaunchKernelGGL(sn_build_pairs_h16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)table.ptr, (uint32_t*)rows.ptr, l,
                           d.log2_hashmap_size, pinfo.base[l], n_t[l], scale);
    }
#else
    if (rest > 0) {
        const uint64_t n = (uint64_t)rest << d.log2_hashmap_size;
        hipLaunchKernelGGL(sn_build_rows_h16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)table.ptr, (uint32_t*)rows.ptr, nd, rest,
                           d.log2_hashmap_size, scale);
    }
#endif
    SN_HIP(h, hipGetLastError());
    return SN_OK;
}

// number of leading dense levels if the dense levels form a prefix of the level list, else -1
int leading_dense(const SnHashMlpDesc& d) {
    const SnGridLevels g = grid_levels(d);
    int nd = 0;
    bool ended = false;
    for (int l = 0; l < d.num_levels; ++l) {
        const bool dense = ((g.packed[l >> 2] >> ((l & 3) * 8)) & 0xffu) != 0u;
        if (dense && ended) return -1;
        if (dense) ++nd;
        else ended = true;
    }
    return nd;
}

inline int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// Host-side construction of the LDS weight image consumed by sn_main_field_f32 (sn_main.h).
// W1 [64,32], b1 [64]; W2 [16,64], b2 [16]; Wc1 [64,63], bc1 [64]; Wc2 [64,64], bc2; Wc3 [3,64], bc3 [3];
// app [A] = mean appearance embedding (folded into the colour-layer-1 bias; A14).
std::vector<float> build_main_image(const SnFieldDesc& d, const float* W1, const float* b1, const float* W2, const float* b2,
                                    const float* Wc1, const float* bc1, const float* Wc2, const float* bc2, const float* Wc3,
                                    const float* bc3, const float* app) {
    std::vector<float> img(SnMainImg::TOTAL, 0.0f);
    const int geo = d.geo_feat_dim;           // 15
    const int sh = d.sh_levels * d.sh_levels;  // 16
    const int cin = sh + geo + d.appearance_embed_dim;
    auto put = [&](int base, int KS, int rt, int t, int lane, float v) {
        img[base + ((rt * (KS / 4) + t / 4) * 64 + lane) * 4 + (t % 4)] = v;
    };
    // layer 1: slot (t,h) <-> feature 2t+h
    for (int rt = 0; rt < 2; ++rt)
        for (int t = 0; t < 16; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                int row = rt * 32 + (lane & 31), h = lane >> 5;
                put(SnMainImg::W1, 16, rt, t, lane, W1[row * 32 + 2 * t + h]);
            }
    // layer 2: 32 padded rows: 0 = h0, 1..15 = geo, 20 = h0 again (so that lanes 32-63 find their
    // sample's density in their own half), rest zero.  slot (t = rt'*16 + r, h) <-> hidden rt'*32 + rho(r) + 4h
    auto l2src = [&](int row) { return row < 16 ? row : (row == 20 ? 0 : -1); };
    for (int t = 0; t < 32; ++t)
        for (int lane = 0; lane < 64; ++lane) {
            int row = lane & 31, h = lane >> 5, src = l2src(row);
            int hid = (t / 16) * 32 + rho(t % 16) + 4 * h;
            put(SnMainImg::W2, 32, 0, t, lane, src >= 0 ? W2[src * 64 + hid] : 0.0f);
        }
    // colour layer 1: k-steps 0..7 <- layer-2 rows rho(t)+4h (row 0 = h0 is not an input; rows 1..15 = geo 0..14
    // = colour inputs sh+0 .. sh+14); k-steps 8..15 <- SH component 2(t-8)+h = colour input 2(t-8)+h
    for (int rt = 0; rt < 2; ++rt)
        for (int t = 0; t < 16; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                int row = rt * 32 + (lane & 31), h = lane >> 5;
                float v = 0.0f;
                if (t < 8) {
                    int l2row = rho(t) + 4 * h;
                    if (l2row >= 1 && l2row <= geo) v = Wc1[row * cin + sh + (l2row - 1)];
                } else {
                    int s = 2 * (t - 8) + h;
                    if (s < sh) v = Wc1[row * cin + s];
                }
                put(SnMainImg::WC1, 16, rt, t, lane, v);
            }
    // colour layer 2
    for (int rt = 0; rt < 2; ++rt)
        for (int t = 0; t < 32; ++t)
            for (int lane = 0; lane < 64; ++lane) {
                int row = rt * 32 + (lane & 31), h = lane >> 5;
                int hid = (t / 16) * 32 + rho(t % 16) + 4 * h;
                put(SnMainImg::WC2, 32, rt, t, lane, Wc2[row * 64 + hid]);
            }
    // bias images [rt][h][r] -> bias[rt*32 + rho(r) + 4h]
    auto bias_img = [&](int base, int RT, auto&& f) {
        for (int rt = 0; rt < RT; ++rt)
            for (int h = 0; h < 2; ++h)
                for (int r = 0; r < 16; ++r) img[base + (rt * 2 + h) * 16 + r] = f(rt * 32 + rho(r) + 4 * h);
    };
    bias_img(SnMainImg::B1, 2, [&](int n) { return b1[n]; });
    bias_img(SnMainImg::B2, 1, [&](int row) {
        int src = l2src(row);
        return src >= 0 ? b2[src] : 0.0f;
    });
    bias_img(SnMainImg::BC1, 2, [&](int n) {
        float acc = bc1[n];
        for (int a = 0; a < d.appearance_embed_dim; ++a) acc += Wc1[n * cin + sh + geo + a] * app[a];
        return acc;
    });
    bias_img(SnMainImg::BC2, 2, [&](int n) { return bc2[n]; });
    // colour layer 3 (VALU): [n][h][rt*16 + r] = Wc3[n][rt*32 + rho(r) + 4h]
    for (int n = 0; n < 3; ++n)
        for (int h = 0; h < 2; ++h)
            for (int rt = 0; rt < 2; ++rt)
                for (int r = 0; r < 16; ++r)
                    img[SnMainImg::W3 + (n * 2 + h) * 32 + rt * 16 + r] = Wc3[n * 64 + rt * 32 + rho(r) + 4 * h];
    for (int n = 0; n < 3; ++n) img[SnMainImg::B3 + n] = bc3[n];
    return img;
}


// ---- fp16 helpers (host) -------------------------------------------------------------------------------------
uint16_t f32_to_f16_rne(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7bffu);  // saturate instead of inf (matches cvt_pkrtz on the device side)
    if (x < 0x38800000u) {                                     // subnormal half (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;
        const int shift = 126 - (int)(x >> 23);                // 14..24
        uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        uint32_t half = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), mid = 1u << (shift - 1);
        if (rem > mid || (rem == mid && (half & 1u))) ++half;
        return (uint16_t)(sign | half);
    }
    uint32_t half = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (half & 1u))) ++half;
    return (uint16_t)(sign | half);
}

float f16_to_f32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, x;
    if (e == 0) {
        if (m == 0) x = sign;
        else {
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            x = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
    else x = sign | ((e + 112) << 23) | (m << 13);
    float f;
    memcpy(&f, &x, 4);
    return f;
}

// fp16x2 LDS image consumed by sn_main_field_h: same rows / folded bias as build_main_image, other k-slot order.
std::vector<float> build_main_image_h(const SnFieldDesc& d, const float* W1, const float* W2, const float* Wc1, const float* Wc2,
                                      const std::vector<float>& img32) {
    std::vector<float> out(SnMainImg::TOTAL, 0.0f);
    uint16_t* hw = (uint16_t*)out.data();
    const int geo = d.geo_feat_dim, sh = d.sh_levels * d.sh_levels, cin = sh + geo + d.appearance_embed_dim;
    auto put = [&](int base_bytes, int KS, int rt, int s, int lane, int e, float w) {
        const uint16_t hi = f32_to_f16_rne(w);
        const uint16_t lo = f32_to_f16_rne(w - f16_to_f32(hi));
        const size_t off = (size_t)base_bytes / 2 + ((size_t)((rt * KS + s) * 2) * 64 + lane) * 8 + e;
        hw[off] = hi;
        hw[off + 512] = lo;  // the lo plane follows 64 lanes x 8 halves later
    };
    auto l2src = [&](int row) { return row < 16 ? row : (row == 20 ? 0 : -1); };
    for (int lane = 0; lane < 64; ++lane) {
        const int h = lane >> 5, i = lane & 31;
        for (int e = 0; e < 8; ++e) {
            for (int rt = 0; rt < 2; ++rt)
                for (int s = 0; s < 2; ++s) put(SnMainImgH::W1, 2, rt, s, lane, e, W1[(rt * 32 + i) * 32 + 16 * s + 8 * h + e]);
            for (int s = 0; s < 4; ++s) {  // hidden unit of slot (s = 2 rt' + s', h, e): rt'*32 + rho(8 s' + e) + 4h
                const int hid = (s / 2) * 32 + rho(8 * (s % 2) + e) + 4 * h, src = l2src(i);
                put(SnMainImgH::W2, 4, 0, s, lane, e, src >= 0 ? W2[src * 64 + hid] : 0.0f);
                for (int rt = 0; rt < 2; ++rt) put(SnMainImgH::WC2, 4, rt, s, lane, e, Wc2[(rt * 32 + i) * 64 + hid]);
            }
            for (int rt = 0; rt < 2; ++rt) {
                const int row = rt * 32 + i;
                const int l2row = rho(e) + 4 * h;  // k-step 0: layer-2 rows
                put(SnMainImgH::WC1, 2, rt, 0, lane, e, (l2row >= 1 && l2row <= geo) ? Wc1[row * cin + sh + (l2row - 1)] : 0.0f);
                const int comp = 8 * h + e;        // k-step 1: SH components
                put(SnMainImgH::WC1, 2, rt, 1, lane, e, comp < sh ? Wc1[row * cin + comp] : 0.0f);
            }
        }
    }
    memcpy((char*)out.data() + SnMainImgH::FP32, img32.data() + SnMainImg::B1, (size_t)SnMainImgH::TAIL_FLOATS * 4);
    return out;
}

// ---- range conditioning of the split-precision MLPs ---------------------------------------------------------------------------
// Every fp32 operand of the "fp16x2" path is carried as fp16 hi + lo.  That is fp32-grade (2^-22 relative) only while the operand sits
// in [2^-3, 65504]: below, lo drops into fp16's subnormals (absolute resolution 2^-24, e.g. ~13 bits for a value of 5e-4 -- nerfstudio
// initialises its tables at 1e-3); above, cvt_pkrtz saturates.  Scaling by a power of two is exact, so sn_finalize_weights moves
// every layer into the upper part of the range once, on the host, at no run-time cost:
//   features   f' = t0 f        t0 = 2^floor(log2(2^10 / max|table|)); the de-hashed copies / paired tables store t0 * row (levels read
//                               from the uploaded table are multiplied in the kernel), the first layer's weights carry 1 / t0
//   layer l    z_l' = s_l z_l   s_l = 2^floor(log2(2^10 / B_l)), B_l = interval bound of |z_l| over all inputs with |f| <= max|table|;
//                               W_l' = W_l s_l / s_(l-1), b_l' = b_l s_l; ReLU commutes with s_l > 0; the last consumer divides it out
// B_l is a true bound, so no activation can saturate; the largest weights of a layer land in [2^-1, 2^4] by construction.  The only
// failure left is a scaled weight outside the fp16 range (a unit whose inputs are bounded ~0 next to ordinary ones): the handle then
// renders precision-1 requests with the exact fp32 MFMA path (sn_effective_precision reports it).
float pow2_floor(double x) {
    if (!(x > 0.0) || !std::isfinite(x)) return 1.0f;
    // exponent clamped to +-80: scales stay finite fp32 numbers with finite products and reciprocals whatever the parameters are
    // (a table whose largest entry is below 2^-70 is a zero field for every practical purpose)
    return (float)std::ldexp(1.0, std::max(-80, std::min(80, (int)std::floor(std::log2(x)))));
}

struct MainSplitPlan {
    float t0 = 1.0f, s1 = 1.0f, s2 = 1.0f, s3 = 1.0f, s4 = 1.0f;
    bool ok = true;
    std::string why;
    double max_bound = 0.0;  // largest interval bound of an (unscaled) activation
    double B2[16] = {};      // interval bounds of the (unscaled) layer-2 outputs: row 0 = h0, rows 1..15 = the geo features
};

MainSplitPlan plan_split_scales(const SnFieldDesc& d, float table_absmax, bool scale_features, const float* W1, const float* b1, const float* W2,
                                const float* b2, const float* Wc1, const float* bc1, const float* Wc2, const float* bc2, const float* app) {
    MainSplitPlan pl;
    const int geo = d.geo_feat_dim, sh = d.sh_levels * d.sh_levels, cin = sh + geo + d.appearance_embed_dim;
    if (!std::isfinite(table_absmax)) {
        pl.ok = false;
        pl.why = "the hash table holds non-finite values";
        return pl;
    }
    const double M0 = table_absmax;
    pl.t0 = scale_features && M0 > 0.0 ? pow2_floor(1024.0 / M0) : 1.0f;
    std::vector<double> B1(64), B2(16), Bc1(64), Bc2(64);
    double m1 = 0, m2 = 0, m3 = 0, m4 = 0;
    for (int n = 0; n < 64; ++n) {
        double a = std::fabs(b1[n]);
        for (int k = 0; k < 32; ++k) a += std::fabs(W1[n * 32 + k]) * M0;
        B1[n] = a;
        m1 = std::max(m1, a);
    }
    for (int r = 0; r < 16; ++r) {
        double a = std::fabs(b2[r]);
        for (int n = 0; n < 64; ++n) a += std::fabs(W2[r * 64 + n]) * B1[n];
        B2[r] = a;
        pl.B2[r] = a;
        m2 = std::max(m2, a);
    }
    for (int n = 0; n < 64; ++n) {
        double a = std::fabs(bc1[n]);
        for (int e = 0; e < d.appearance_embed_dim; ++e) a += std::fabs(Wc1[n * cin + sh + geo + e] * app[e]);
        for (int c = 0; c < sh; ++c) a += std::fabs(Wc1[n * cin + c]) * 3.0;  // |SH component| < 3 for degree 4 on [-1, 1]^3
        for (int j = 0; j < geo; ++j) a += std::fabs(Wc1[n * cin + sh + j]) * B2[1 + j];
        Bc1[n] = a;
        m3 = std::max(m3, a);
    }
    for (int n = 0; n < 64; ++n) {
        double a = std::fabs(bc2[n]);
        for (int k = 0; k < 64; ++k) a += std::fabs(Wc2[n * 64 + k]) * Bc1[k];
        Bc2[n] = a;
        m4 = std::max(m4, a);
    }
    if (!std::isfinite(m1) || !std::isfinite(m2) || !std::isfinite(m3) || !std::isfinite(m4)) {
        pl.ok = false;
        pl.why = "non-finite MLP parameters";
        return pl;
    }
    pl.max_bound = std::max(std::max(m1, m2), std::max(m3, m4));
    // target: every scaled pre-activation below 2^10 -- far inside fp16's range, and below 2048, which the ReLU folded into the operand
    // split needs (sn_main.h sn_split2_relu: the low part a - RTZ16(a) must stay below 1 for its clamp to be a plain max(., 0))
    const double target = SN_RELU_FOLD ? 1024.0 : 16384.0;
    pl.s1 = m1 > 0 ? pow2_floor(target / m1) : 1.0f;
    pl.s2 = m2 > 0 ? pow2_floor(target / m2) : 1.0f;
    pl.s3 = m3 > 0 ? pow2_floor(target / m3) : 1.0f;
    pl.s4 = m4 > 0 ? pow2_floor(target / m4) : 1.0f;
    return pl;
}

// true if every element of a scaled operand fits fp16 (|x| <= 65504) -- the split saturates beyond
bool fits_half(const std::vector<float>& v) {
    for (float x : v)
        if (!(std::fabs(x) <= 65504.0f)) return false;
    return true;
}

const std::vector<float>* find(SnHandle h, const std::string& name, size_t count) {
    auto it = h->host.find(name);
    if (it == h->host.end() || it->second.size() != count) return nullptr;
    return &it->second;
}

// Builds the x-paired copy of one hash table (sn_device.h): per level l, (bitlen(scale_l) + 1) tables of T 16-byte entries.
int build_pairs(SnHandle h, const SnHashMlpDesc& d, const DevBuf& table, DevBuf& pairs, SnPairInfo& info, hipStream_t st, float scale = 1.0f,
                int first_level = 0) {
    const uint32_t T = 1u << d.log2_hashmap_size;
    uint64_t entries = 0;
    int n_t[SN_MAX_LEVELS];
    const SnGridLevels gl = grid_levels(d);
    for (int l = 0; l < d.num_levels; ++l) {
        int bits = 0;
        // largest floor coordinate: floor(scale) in the torch grid, at most ceil(scale) with tiny-cuda-nn's +0.5
        for (uint32_t s = (uint32_t)ceilf(d.scalings[l]) + 1u; s; s >>= 1) ++bits;
        n_t[l] = bits + 1;
        if (((gl.packed[l >> 2] >> ((l & 3) * 8)) & 0xffu) != 0u) n_t[l] = 0;  // dense tcnn level: read from the plain table
        if (l < first_level) n_t[l] = 0;                                        // levels that are read from de-hashed copies
        info.base[l] = (uint32_t)entries;
        entries += (uint64_t)n_t[l] * T;
    }
    if (entries == 0) entries = 1;
    for (int l = d.num_levels; l < SN_MAX_LEVELS; ++l) info.base[l] = 0;
    const uint64_t bytes = entries * 16;
    if (bytes >= (1ull << 32)) return fail(h, SN_ERR_INVALID, "paired hash tables exceed the 4 GiB buffer-descriptor range");
    if (pairs.bytes != bytes) {
        pairs.release();
        SN_HIP(h, hipMalloc(&pairs.ptr, bytes));
        pairs.bytes = bytes;
    }
    for (int l = 0; l < d.num_levels; ++l) {
        const uint64_t n = (uint64_t)n_t[l] * T;
        if (n == 0) continue;
        hipLaunchKernelGGL(sn_build_pairs_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)table.ptr,
                           (float*)pairs.ptr, l, d.log2_hashmap_size, info.base[l], n_t[l], scale);
    }
    SN_HIP(h, hipGetLastError());
    return SN_OK;
}

struct TileGeom {
    int tw_log2, th_log2, tiles_x, tiles_y;
};

TileGeom tile_geometry(int height, int width) {
    TileGeom g;
    if (height >= 8) {
        g.tw_log2 = 3;
        g.th_log2 = 3;
    } else {
        g.tw_log2 = 6;
        g.th_log2 = 0;
    }
    g.tiles_x = (width + (1 << g.tw_log2) - 1) >> g.tw_log2;
    g.tiles_y = (height + (1 << g.th_log2) - 1) >> g.th_log2;
    return g;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct WorkspacePlan {
    size_t off_exp_raw, off_minmax, off_ebins, off_prop_scratch, off_prop_counter, off_seg, total;
    int prop_blocks;
    int n_chunks;
    int seg_first_block, n_seg, seg_len;  // split-depth tail of the main kernel (n_seg <= 1: none)
};

// Split-depth tail of the main kernel (sn_main.h SnMainParams).  A launch is whole workgroups (2x2 tiles, 4 waves) on a fixed number of
// slots (CUs x SN_MAIN_WAVES_PER_SIMD workgroups = 768): its last round may hold a handful of workgroups -- a 64x64 viewer frame is 16 of
// them, the tail of a 640x640 frame 64 -- each still marching all S samples with the chip nearly empty.  Such a tail's workgroups are cut
// into n segment jobs of ceil(S / n) samples, so that the round fills more of the chip and is ~1/n as long; a small kernel composites
// the stored samples in order (bit-identical).
// WHEN (measured r03, frames back to back on one stream, profiles/r03_tail_split.txt): it pays when the tail is SMALL -- at most 1/8 of
// the slots: 64x64 0.79 -> 0.25 ms, 128x128 0.69 -> 0.41, 640x640 2.06 -> 1.88 (-9 %).  A tail that already fills a quarter of the chip
// gains nothing or loses: its lone waves step ~2x faster than waves that share a SIMD AND run at the boost clock (the chip is power-
// limited when full), segment jobs pay the prologue n times -- 800x800 (196 of 768) -1 % per launch but +1 % with two frames in flight,
// 200x200 (169) +11 %, 512x512 (256) +8 %.  So: tails above slots / 8 stay whole.
// n minimises a cost model in units of sample steps: rounds of jobs x (samples per job + ~1.5 steps of prologue), a partly filled round
// priced at 0.25 + 0.75 x fill.
struct TailPlan {
    int first_block, n_seg, seg_len;
};
TailPlan plan_tail(int total_wgs, int n_cus, int S, bool enabled) {
    TailPlan t{total_wgs, 1, S};
    const int slots = n_cus * SN_MAIN_WAVES_PER_SIMD;
    if (!enabled || slots <= 0 || S < 8) return t;
    const int tail = total_wgs % slots;
    if (tail == 0 || tail > slots / 8) return t;
    auto cost = [&](int n) {
        const long jobs = (long)tail * n;
        const double len = (double)((S + n - 1) / n) + 1.5;
        const long full = jobs / slots, rest = jobs % slots;
        return (double)full * len + (rest ? len * (0.25 + 0.75 * (double)rest / slots) : 0.0) + 0.5 * (n - 1);  // (+ a little per extra segment: scratch, launch)
    };
    int best = 1;
    double best_cost = cost(1);
    for (int n = 2; n <= 8 && (S + n - 1) / n >= 4; ++n)
        if (cost(n) < best_cost - 1e-9) {
            best = n;
            best_cost = cost(n);
        }
    if (best == 1 || best_cost > 0.9 * cost(1)) return t;  // not worth a second kernel
    t.first_block = total_wgs - tail;
    if (t.first_block % 8 != 0) return TailPlan{total_wgs, 1, S};  // (the XCD-affine job order needs whole rows of 8 in front; true for 256 CUs)
    t.n_seg = best;
    t.seg_len = (S + best - 1) / best;
    return t;
}

WorkspacePlan plan_workspace(int height, int width, const SnRenderOpts& o, int n_cus, bool tail_split) {
    WorkspacePlan w;
    const size_t n = (size_t)height * width;
    const TileGeom g = tile_geometry(height, width);
    size_t off = 0;
    w.off_exp_raw = off;
    off += align256(n * 4);
    w.n_chunks = (int)((n + (size_t)o.chunk_rays - 1) / (size_t)o.chunk_rays);
    w.off_minmax = off;
    off += align256((size_t)w.n_chunks * 8);
    w.off_ebins = off;
    w.off_prop_scratch = off;
    w.off_prop_counter = off;
    w.prop_blocks = 0;
    if (o.num_proposal_iterations > 0) {
        off += align256((size_t)g.tiles_x * g.tiles_y * 64 * (o.num_nerf_samples + 1) * 4);
        // persistent proposal waves: what the chip holds (256 CUs x SN_PROP_WG_PER_CU workgroups of SN_PROP_WAVES waves), at most one per tile
        const int ntiles = g.tiles_x * g.tiles_y;
        w.prop_blocks = std::min((ntiles + SN_PROP_WAVES - 1) / SN_PROP_WAVES, 256 * SN_PROP_WG_PER_CU);
        w.off_prop_scratch = off;
        off += align256((size_t)w.prop_blocks * SN_PROP_WAVES * SN_PROP_SCRATCH_FLOATS * 4);
        w.off_prop_counter = off;   // the proposal kernel's tile queue (one uint32, zeroed per launch)
        off += 256;
    }
    {
        const int gbx = (g.tiles_x + 1) / 2, gby = (g.tiles_y + 1) / 2;
        const TailPlan t = plan_tail(gbx * gby, n_cus, o.num_nerf_samples, tail_split);
        w.seg_first_block = t.first_block;
        w.n_seg = t.n_seg;
        w.seg_len = t.seg_len;
        w.off_seg = off;
        if (t.n_seg > 1) off += align256((size_t)(gbx * gby - t.first_block) * 4 * (size_t)o.num_nerf_samples * 64 * 16);  // (density, r, g, b) per sample
    }
    w.total = off;
    return w;
}

// The non-default sampler / position map run in the ALT instantiations of K1 / K2 / K3 (run-time-generic: uploaded tables, no de-hashed
// copies, STRICT position arithmetic), so that the production kernels' code does not depend on them.  A far plane beyond 1e7 goes there
// too: out there the exact contraction rounds onto the face q = 1 (dropped by the selector) and the production kernels' reciprocal
// form may not (sn_sample_q_fast).
bool needs_generic_kernels(SnHandle h, const SnRenderOpts* opts) {
    return opts->spacing_mode != 0 || h->pos_map.box != 0 || !(opts->far_plane <= 1.0e7f);
}

bool valid_opts(const SnFieldDesc& d, const SnRenderOpts& o, std::string& why) {
    if (o.num_proposal_iterations < 0 || o.num_proposal_iterations > d.num_proposals) why = "num_proposal_iterations exceeds the proposal nets of this handle";
    else if (o.num_nerf_samples < 1 || o.num_nerf_samples > 1024) why = "num_nerf_samples out of range [1,1024]";
    else if (o.chunk_rays < 1) why = "chunk_rays must be positive";
    else if (o.precision < 0 || o.precision > 2) why = "precision must be 0 (fp32), 1 (split fp16) or 2 (single fp16, tiny-cuda-nn grids)";
    else if (o.precision == 2 && d.main_field.grid_mode != 1)
        why = "precision 2 (single fp16) is the arithmetic of tiny-cuda-nn checkpoints: it needs main_field.grid_mode = 1";
    else if (o.background_mode != 0 && o.background_mode != 1) why = "background_mode must be 0 (last sample) or 1 (constant colour)";
    else if (o.spacing_mode != 0 && o.spacing_mode != 1) why = "spacing_mode must be 0 (piecewise) or 1 (uniform)";
    else {
        for (int i = 0; i < o.num_proposal_iterations; ++i)
            if (o.num_proposal_samples[i] < 2 || o.num_proposal_samples[i] > SN_PROP_MAX_SAMPLES) {
                why = "num_proposal_samples out of range [2," + std::to_string(SN_PROP_MAX_SAMPLES) + "]";
                return false;
            }
        return true;
    }
    return false;
}

}  // namespace

extern "C" {

int sn_abi_version(void) { return SN_ABI_VERSION; }

int sn_create(const SnFieldDesc* desc_in, SnHandle* out) {
    if (!desc_in || !out) return fail(nullptr, SN_ERR_INVALID, "sn_create: null argument");
    SnFieldDesc desc_own;
    if (int rc = adopt_struct(nullptr, desc_in, kFieldDescMin, desc_own, "sn_create: SnFieldDesc")) return rc;
    const SnFieldDesc* desc = &desc_own;
    std::string why;
    if (!check_hashmlp(desc->main_field, 16, 64, 16, why)) return fail(nullptr, SN_ERR_INVALID, "main field: " + why);
    if (desc->geo_feat_dim != 15 || desc->hidden_dim_color != 64 || desc->sh_levels != 4)
        return fail(nullptr, SN_ERR_INVALID, "unsupported colour head: need geo_feat_dim 15, hidden_dim_color 64, sh_levels 4");
    if (desc->appearance_embed_dim < 0 || desc->appearance_embed_dim > 256)
        return fail(nullptr, SN_ERR_INVALID, "appearance_embed_dim out of range");
    if (desc->num_proposals < 0 || desc->num_proposals > SN_MAX_PROPOSALS)
        return fail(nullptr, SN_ERR_INVALID, "num_proposals out of range");
    for (int i = 0; i < desc->num_proposals; ++i)
        if (!check_hashmlp(desc->proposals[i], 5, 16, 1, why))
            return fail(nullptr, SN_ERR_INVALID, "proposal net " + std::to_string(i) + ": " + why);
    if (desc->disable_scene_contraction != 0 && desc->disable_scene_contraction != 1)
        return fail(nullptr, SN_ERR_INVALID, "disable_scene_contraction must be 0 or 1");
    if (desc->disable_scene_contraction)
        for (int k = 0; k < 3; ++k)
            if (!(desc->aabb[3 + k] - desc->aabb[k] > 0.0f) || !std::isfinite(desc->aabb[3 + k] - desc->aabb[k]))
                return fail(nullptr, SN_ERR_INVALID, "disable_scene_contraction needs a scene box with positive finite extents (SnFieldDesc.aabb)");
    SnContext* c = new SnContext();
    c->desc = *desc;
    c->pos_map.box = desc->disable_scene_contraction;
    for (int k = 0; k < 3; ++k) {
        const float len = desc->disable_scene_contraction ? desc->aabb[3 + k] - desc->aabb[k] : 1.0f;  // (aabb[1] - aabb[0] in fp32, as SceneBox does)
        c->pos_map.lo[k] = desc->disable_scene_contraction ? desc->aabb[k] : 0.0f;
        c->pos_map.len[k] = len;
    }
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        return fail(nullptr, SN_ERR_HIP, "hipGetDevice failed (no HIP device?)");
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && cus > 0) c->n_cus = cus;
    }
    load_switches(c);
    c->id = g_next_handle_id.fetch_add(1);
    *out = c;
    return SN_OK;
}

int sn_debug_reload_env(SnHandle h) {
    if (!h) return SN_ERR_INVALID;
    load_switches(h);
    return SN_OK;
}

int sn_destroy(SnHandle h) {
    if (!h) return SN_OK;
    for (auto& kv : h->render_ev) {
        (void)hipEventSynchronize(kv.second);  // a render may still be reading the buffers released below
        (void)hipEventDestroy(kv.second);
    }
    for (hipEvent_t ev : h->spare_ev) (void)hipEventDestroy(ev);
    if (h->weights_ev) (void)hipEventDestroy(h->weights_ev);
    h->table_main.release();
    h->dense_main.release();
    h->hquads_main.release();
    h->hrows_main.release();
    h->pairs_main.release();
    h->wimg_main.release();
    h->wimg_main_h.release();
    h->wimg_normals.release();
    h->wimg_normals_h.release();
    for (int i = 0; i < SN_MAX_PROPOSALS; ++i) {
        h->table_prop[i].release();
        h->pairs_prop[i].release();
        h->dense_prop[i].release();
        h->wpack_prop[i].release();
    }
    delete h;
    return SN_OK;
}

const char* sn_last_error(SnHandle h) {
    if (!h) return g_create_error.c_str();
    std::lock_guard<std::mutex> g(h->mu);
    g_error_copy = h->error;  // valid until this THREAD's next sn_last_error call, whatever other threads do to the handle
    return g_error_copy.c_str();
}

int sn_upload_weights(SnHandle h, const char* name, const void* data, size_t bytes, SnStream stream) {
    if (!h || !name || !data) return fail(h, SN_ERR_INVALID, "sn_upload_weights: null argument");
    hipStream_t st = (hipStream_t)stream;
    const std::string n(name);
    h->weights_epoch.fetch_add(1);  // (bins a workspace holds were sampled with the previous weights: SnRenderOpts.reuse_final_bins)
    wait_for_renders(h, st);  // renders in flight on other streams still read the buffers this call overwrites (or frees)
    auto upload_table = [&](DevBuf& buf, const SnHashMlpDesc& d) -> int {
        const size_t want = ((size_t)d.num_levels << d.log2_hashmap_size) * 2 * sizeof(float);
        if (bytes != want) return fail(h, SN_ERR_INVALID, n + ": expected " + std::to_string(want) + " bytes, got " + std::to_string(bytes));
        if (buf.bytes != want) {
            buf.release();
            SN_HIP(h, hipMalloc(&buf.ptr, want));
            buf.bytes = want;
        }
        SN_HIP(h, hipMemcpyAsync(buf.ptr, data, want, hipMemcpyDefault, st));
        SN_HIP(h, hipStreamSynchronize(st));
        mark_weights_written(h, st);
        {
            std::lock_guard<std::mutex> g(h->mu);
            h->finalized = false;  // the paired copies must be rebuilt
        }
        return SN_OK;
    };
    if (n == "field.mlp_base.encoder.hash_table") return upload_table(h->table_main, h->desc.main_field);
    for (int i = 0; i < h->desc.num_proposals; ++i)
        if (n == "proposal_networks." + std::to_string(i) + ".mlp_base.encoder.hash_table")
            return upload_table(h->table_prop[i], h->desc.proposals[i]);
    if (n.rfind("field.", 0) != 0 && n.rfind("proposal_networks.", 0) != 0)
        return fail(h, SN_ERR_INVALID, "unknown parameter name: " + n);
    if (bytes % sizeof(float) != 0 || bytes > (1u << 22)) return fail(h, SN_ERR_INVALID, n + ": bad size");
    std::vector<float> v(bytes / sizeof(float));
    SN_HIP(h, hipMemcpyAsync(v.data(), data, bytes, hipMemcpyDefault, st));
    SN_HIP(h, hipStreamSynchronize(st));
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->host[n] = std::move(v);
        h->finalized = false;
    }
    return SN_OK;
}

int sn_finalize_weights(SnHandle h, SnStream stream) {
    if (!h) return SN_ERR_INVALID;
    hipStream_t st = (hipStream_t)stream;
    h->weights_epoch.fetch_add(1);
    load_switches(h);
    wait_for_renders(h, st);
    const SnFieldDesc& d = h->desc;
    if (!h->table_main.ptr) return fail(h, SN_ERR_STATE, "missing field.mlp_base.encoder.hash_table");
    const int cin = d.sh_levels * d.sh_levels + d.geo_feat_dim + d.appearance_embed_dim;
    struct Need {
        const char* name;
        size_t count;
    };
    const Need needs[] = {
        {"field.mlp_base.mlp.layers.0.weight", 64 * 32}, {"field.mlp_base.mlp.layers.0.bias", 64},
        {"field.mlp_base.mlp.layers.1.weight", 16 * 64}, {"field.mlp_base.mlp.layers.1.bias", 16},
        {"field.mlp_head.layers.0.weight", (size_t)64 * cin}, {"field.mlp_head.layers.0.bias", 64},
        {"field.mlp_head.layers.1.weight", 64 * 64}, {"field.mlp_head.layers.1.bias", 64},
        {"field.mlp_head.layers.2.weight", 3 * 64}, {"field.mlp_head.layers.2.bias", 3},
    };
    const std::vector<float>* t[10];
    for (int i = 0; i < 10; ++i) {
        t[i] = find(h, needs[i].name, needs[i].count);
        if (!t[i]) return fail(h, SN_ERR_STATE, std::string("missing or mis-sized parameter ") + needs[i].name);
    }
    std::vector<float> zero_app((size_t)d.appearance_embed_dim, 0.0f);
    const std::vector<float>* app = &zero_app;
    if (d.appearance_embed_dim > 0) {
        app = find(h, "field.embedding_appearance.mean", (size_t)d.appearance_embed_dim);
        if (!app) return fail(h, SN_ERR_STATE, "missing field.embedding_appearance.mean");
    }
    // max |row| of every hash table (device reduction): the input bound of the range conditioning below
    float absmax_main = 0.0f, absmax_prop[SN_MAX_PROPOSALS] = {0.0f, 0.0f};
    {
        uint32_t* d_m = nullptr;
        SN_HIP(h, hipMalloc((void**)&d_m, 4 * (1 + SN_MAX_PROPOSALS)));
        SN_HIP(h, hipMemsetAsync(d_m, 0, 4 * (1 + SN_MAX_PROPOSALS), st));
        hipLaunchKernelGGL(sn_absmax_kernel, dim3(1024), dim3(256), 0, st, (const float*)h->table_main.ptr, h->table_main.bytes / 4, d_m);
        for (int i = 0; i < d.num_proposals; ++i)
            if (h->table_prop[i].ptr)
                hipLaunchKernelGGL(sn_absmax_kernel, dim3(256), dim3(256), 0, st, (const float*)h->table_prop[i].ptr, h->table_prop[i].bytes / 4, d_m + 1 + i);
        uint32_t bits[1 + SN_MAX_PROPOSALS];
        hipError_t e = hipMemcpyAsync(bits, d_m, sizeof(bits), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        (void)hipFree(d_m);
        if (e != hipSuccess) return fail(h, SN_ERR_HIP, std::string("sn_finalize_weights: table scan: ") + hipGetErrorString(e));
        memcpy(&absmax_main, &bits[0], 4);
        h->table_absmax_main = absmax_main;
        for (int i = 0; i < SN_MAX_PROPOSALS; ++i) memcpy(&absmax_prop[i], &bits[1 + i], 4);
    }
    const MainSplitPlan pl = plan_split_scales(d, absmax_main, true, t[0]->data(), t[1]->data(), t[2]->data(), t[3]->data(), t[4]->data(),
                                               t[5]->data(), t[6]->data(), t[7]->data(), app->data());
    h->feat_scale_main = pl.t0;
    h->split_ok = pl.ok;
    h->split_why = pl.why;
    h->normals_split_ok = pl.ok;  // refined below, once the normals kernel's own conditioned operands exist
    auto scaled = [](const std::vector<float>& v, double f) {
        std::vector<float> o(v.size());
        for (size_t i = 0; i < v.size(); ++i) o[i] = (float)((double)v[i] * f);  // f is a power of two: exact (barring under / overflow)
        return o;
    };
    // exact-fp32 image: only the feature scale (the de-hashed copies carry it), W1 / t0 -- bit-identical results
    const std::vector<float> W1f = scaled(*t[0], 1.0 / pl.t0);
    std::vector<float> img = build_main_image(d, W1f.data(), t[1]->data(), t[2]->data(), t[3]->data(), t[4]->data(), t[5]->data(),
                                              t[6]->data(), t[7]->data(), t[8]->data(), t[9]->data(), app->data());
    if (!h->wimg_main.ptr) {
        SN_HIP(h, hipMalloc(&h->wimg_main.ptr, img.size() * 4));
        h->wimg_main.bytes = img.size() * 4;
    }
    SN_HIP(h, hipMemcpyAsync(h->wimg_main.ptr, img.data(), img.size() * 4, hipMemcpyHostToDevice, st));
    // split-precision image: every layer in its conditioned range (plan_split_scales)
    const int sh_n = d.sh_levels * d.sh_levels;
    const std::vector<float> W1s = scaled(*t[0], (double)pl.s1 / pl.t0), b1s = scaled(*t[1], pl.s1);
    const std::vector<float> W2s = scaled(*t[2], (double)pl.s2 / pl.s1), b2s = scaled(*t[3], pl.s2);
    std::vector<float> Wc1s = scaled(*t[4], pl.s3);  // SH and appearance columns; the geo columns take s3 / s2
    for (int n = 0; n < 64; ++n)
        for (int j = 0; j < d.geo_feat_dim; ++j) Wc1s[(size_t)n * cin + sh_n + j] = (float)((double)(*t[4])[(size_t)n * cin + sh_n + j] * ((double)pl.s3 / pl.s2));
    const std::vector<float> bc1s = scaled(*t[5], pl.s3);
    const std::vector<float> Wc2s = scaled(*t[6], (double)pl.s4 / pl.s3), bc2s = scaled(*t[7], pl.s4);
    const std::vector<float> Wc3s = scaled(*t[8], 1.0 / pl.s4);
    if (h->split_ok && !(fits_half(W1s) && fits_half(W2s) && fits_half(Wc1s) && fits_half(Wc2s))) {
        h->split_ok = false;
        h->split_why = "a range-conditioned MLP weight leaves the fp16 range";
    }
    std::vector<float> img_s = build_main_image(d, W1s.data(), b1s.data(), W2s.data(), b2s.data(), Wc1s.data(), bc1s.data(), Wc2s.data(),
                                                bc2s.data(), Wc3s.data(), t[9]->data(), app->data());
    img_s[SnMainImg::B3 + 3] = 1.0f / pl.s2;
    std::vector<float> imgh = build_main_image_h(d, W1s.data(), W2s.data(), Wc1s.data(), Wc2s.data(), img_s);
    {
        // single-fp16 mode (sn_main.h SnMainImgF16): colour layer 3 as an fp16 A operand behind the image -- rows 0..2 = the three output
        // channels, k-slot (s, h, e) <-> hidden unit (s / 2) 32 + rho(8 (s % 2) + e) + 4 h (the operand order colour layer 2's output is
        // converted into), lifted by the power of two s5 so that its largest entry sits in [128, 256) (the weights already carry 1 / s4)
        imgh.resize(SnMainImgF16::TOTAL_FLOATS, 0.0f);
        double m = 0.0;
        for (float w : Wc3s) m = std::max(m, (double)std::fabs(w));
        const float s5 = m > 0 && std::isfinite(m) ? pow2_floor(256.0 / m) : 1.0f;
        uint16_t* hw = (uint16_t*)((char*)imgh.data() + SnMainImgF16::W3H);
        for (int sk = 0; sk < 4; ++sk)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = lane & 31, hh = lane >> 5;
                    const int hid = (sk / 2) * 32 + rho(8 * (sk % 2) + e) + 4 * hh;
                    hw[((size_t)sk * 64 + lane) * 8 + e] = row < 3 ? f32_to_f16_rne((float)((double)Wc3s[row * 64 + hid] * s5)) : (uint16_t)0;
                }
        *(float*)((char*)imgh.data() + SnMainImgF16::TAILF) = 1.0f / s5;
    }
    if (h->wimg_main_h.ptr && h->wimg_main_h.bytes != imgh.size() * 4) h->wimg_main_h.release();
    if (!h->wimg_main_h.ptr) {
        SN_HIP(h, hipMalloc(&h->wimg_main_h.ptr, imgh.size() * 4));
        h->wimg_main_h.bytes = imgh.size() * 4;
    }
    SN_HIP(h, hipMemcpyAsync(h->wimg_main_h.ptr, imgh.data(), imgh.size() * 4, hipMemcpyHostToDevice, st));
    {
        // Normals image (sn_normals.h): the density MLP of the main image, the pred-normal MLP (if uploaded) in the colour slots,
        // and the transposed layer of the reverse pass.  The pred-normal MLP's last linear layer (64 -> 64, no activation) and
        // PredNormalsFieldHead's Linear(64 -> 3) are multiplied together here.
        const std::string pn = "field.mlp_pred_normals.layers.", hd = "field.field_head_pred_normals.net.";
        const int pin = 12 + d.geo_feat_dim;
        const std::vector<float>*w0 = find(h, pn + "0.weight", (size_t)64 * pin), *c0 = find(h, pn + "0.bias", 64),
                                 *w1 = find(h, pn + "1.weight", 64 * 64), *c1 = find(h, pn + "1.bias", 64),
                                 *w2 = find(h, pn + "2.weight", 64 * 64), *c2 = find(h, pn + "2.bias", 64),
                                 *wh = find(h, hd + "weight", 3 * 64), *ch = find(h, hd + "bias", 3);
        h->has_pred_normals = w0 && c0 && w1 && c1 && w2 && c2 && wh && ch;
        SnFieldDesc dn = d;
        dn.appearance_embed_dim = 0;
        const int sh = d.sh_levels * d.sh_levels, cin_n = sh + d.geo_feat_dim;
        std::vector<float> P1((size_t)64 * cin_n, 0.0f), z64(64, 0.0f), z6464(64 * 64, 0.0f), Wf(3 * 64, 0.0f), bf(3, 0.0f);
        if (h->has_pred_normals) {
            for (int n = 0; n < 64; ++n) {
                for (int k = 0; k < 12; ++k) P1[(size_t)n * cin_n + k] = (*w0)[(size_t)n * pin + k];  // position encoding -> SH slots
                for (int k = 0; k < d.geo_feat_dim; ++k) P1[(size_t)n * cin_n + sh + k] = (*w0)[(size_t)n * pin + 12 + k];
            }
            for (int n = 0; n < 3; ++n) {
                double b = (*ch)[n];
                for (int j = 0; j < 64; ++j) b += (double)(*wh)[n * 64 + j] * (double)(*c2)[j];
                bf[n] = (float)b;
                for (int k = 0; k < 64; ++k) {
                    double a = 0.0;
                    for (int j = 0; j < 64; ++j) a += (double)(*wh)[n * 64 + j] * (double)(*w2)[j * 64 + k];
                    Wf[n * 64 + k] = (float)a;
                }
            }
        }
        std::vector<float> nimg = build_main_image(dn, t[0]->data(), t[1]->data(), t[2]->data(), t[3]->data(), P1.data(),
                                                   h->has_pred_normals ? c0->data() : z64.data(),
                                                   h->has_pred_normals ? w1->data() : z6464.data(),
                                                   h->has_pred_normals ? c1->data() : z64.data(), Wf.data(), bf.data(), nullptr);
        nimg.resize(SnNormImg::TOTAL, 0.0f);
        // reverse pass: row f <- sum over hidden j of W1[j][f] * W2[0][j] * mask_j; slot (t, h) <-> hidden (t/16)*32 + rho(t%16) + 4h
        for (int t32 = 0; t32 < 32; ++t32)
            for (int lane = 0; lane < 64; ++lane) {
                const int f = lane & 31, hh = lane >> 5;
                const int hid = (t32 / 16) * 32 + rho(t32 % 16) + 4 * hh;
                nimg[SnNormImg::WB + ((t32 / 4) * 64 + lane) * 4 + (t32 % 4)] = (*t[0])[hid * 32 + f] * (*t[2])[hid];
            }
        if (!h->wimg_normals.ptr) {
            SN_HIP(h, hipMalloc(&h->wimg_normals.ptr, nimg.size() * 4));
            h->wimg_normals.bytes = nimg.size() * 4;
        }
        SN_HIP(h, hipMemcpyAsync(h->wimg_normals.ptr, nimg.data(), nimg.size() * 4, hipMemcpyHostToDevice, st));
        // fp16 hi+lo form, RANGE-CONDITIONED like the main image (r03; r02 split the unconditioned matrices and fell back to exact fp32
        // as soon as max|table| < 1/8 -- i.e. for every real checkpoint: nerfstudio initialises its tables at 1e-3, tiny-cuda-nn at
        // 1e-4).  The density MLP's layers are the main image's (W1 s1 / t0, b1 s1, W2 s2 / s1, b2 s2: features come in times t0, h0
        // leaves through the 1 / s2 slot); the pred-normal MLP gets its own output scales from interval bounds over |pe| <= 1 and the
        // geo bounds B2: layer 1 -> s3n, layer 2 -> s4n, the fp32 (layer 3 . head) weights carry 1 / s4n; the reverse-pass layer
        // W1^T diag(W2[0,:]) is lifted by gsc so that its largest entry sits at ~2^10.
        const double tgt = SN_RELU_FOLD ? 1024.0 : 16384.0;
        double m3n = 0.0, m4n = 0.0, mwb = 0.0;
        std::vector<double> Bp1(64, 0.0);
        if (h->has_pred_normals) {
            for (int n = 0; n < 64; ++n) {
                double a = std::fabs((*c0)[n]);
                for (int k = 0; k < 12; ++k) a += std::fabs((*w0)[(size_t)n * pin + k]);
                for (int j = 0; j < d.geo_feat_dim; ++j) a += std::fabs((*w0)[(size_t)n * pin + 12 + j]) * pl.B2[1 + j];
                Bp1[n] = a;
                m3n = std::max(m3n, a);
            }
            for (int n = 0; n < 64; ++n) {
                double a = std::fabs((*c1)[n]);
                for (int k = 0; k < 64; ++k) a += std::fabs((*w1)[(size_t)n * 64 + k]) * Bp1[k];
                m4n = std::max(m4n, a);
            }
        }
        for (int hid = 0; hid < 64; ++hid)
            for (int f = 0; f < 32; ++f) mwb = std::max(mwb, (double)std::fabs((*t[0])[hid * 32 + f] * (*t[2])[hid]));
        const bool finite_n = std::isfinite(m3n) && std::isfinite(m4n) && std::isfinite(mwb);
        const float s3n = finite_n && m3n > 0 ? pow2_floor(tgt / m3n) : 1.0f, s4n = finite_n && m4n > 0 ? pow2_floor(tgt / m4n) : 1.0f;
        const float gsc = finite_n && mwb > 0 ? pow2_floor(1024.0 / mwb) : 1.0f;
        h->grad_scale_normals = gsc;
        std::vector<float> P1n(P1.size(), 0.0f);
        for (int n = 0; n < 64; ++n) {
            for (int k = 0; k < sh; ++k) P1n[(size_t)n * cin_n + k] = (float)((double)P1[(size_t)n * cin_n + k] * s3n);
            for (int k = 0; k < d.geo_feat_dim; ++k) P1n[(size_t)n * cin_n + sh + k] = (float)((double)P1[(size_t)n * cin_n + sh + k] * ((double)s3n / pl.s2));
        }
        const std::vector<float> c0n = scaled(h->has_pred_normals ? *c0 : z64, s3n);
        const std::vector<float> w1n = scaled(h->has_pred_normals ? *w1 : z6464, (double)s4n / s3n), c1n = scaled(h->has_pred_normals ? *c1 : z64, s4n);
        const std::vector<float> Wfn = scaled(Wf, 1.0 / s4n);
        std::vector<float> nimg_s = build_main_image(dn, W1s.data(), b1s.data(), W2s.data(), b2s.data(), P1n.data(), c0n.data(), w1n.data(), c1n.data(),
                                                     Wfn.data(), bf.data(), nullptr);
        nimg_s[SnMainImg::B3 + 3] = 1.0f / pl.s2;
        std::vector<float> wbs((size_t)64 * 32);
        for (int hid = 0; hid < 64; ++hid)
            for (int f = 0; f < 32; ++f) wbs[(size_t)hid * 32 + f] = (float)((double)((*t[0])[hid * 32 + f] * (*t[2])[hid]) * gsc);
        h->normals_split_ok = pl.ok && finite_n && fits_half(W1s) && fits_half(W2s) && fits_half(P1n) && fits_half(w1n) && fits_half(wbs);
        std::vector<float> nh = build_main_image_h(dn, W1s.data(), W2s.data(), P1n.data(), w1n.data(), nimg_s);
        nh.resize(SnNormImgH::TOTAL_BYTES / 4, 0.0f);
        {
            uint16_t* hw = (uint16_t*)nh.data();
            for (int s4 = 0; s4 < 4; ++s4)
                for (int lane = 0; lane < 64; ++lane)
                    for (int e = 0; e < 8; ++e) {
                        const int f = lane & 31, hh = lane >> 5;
                        const int hid = (s4 / 2) * 32 + rho(8 * (s4 % 2) + e) + 4 * hh;
                        const float w = wbs[(size_t)hid * 32 + f];
                        const uint16_t hi = f32_to_f16_rne(w), lo = f32_to_f16_rne(w - f16_to_f32(hi));
                        const size_t off = (size_t)SnNormImgH::WB / 2 + ((size_t)(s4 * 2) * 64 + lane) * 8 + e;
                        hw[off] = hi;
                        hw[off + 512] = lo;
                    }
        }
        if (!h->wimg_normals_h.ptr) {
            SN_HIP(h, hipMalloc(&h->wimg_normals_h.ptr, nh.size() * 4));
            h->wimg_normals_h.bytes = nh.size() * 4;
        }
        SN_HIP(h, hipMemcpyAsync(h->wimg_normals_h.ptr, nh.data(), nh.size() * 4, hipMemcpyHostToDevice, st));
    }
    for (int i = 0; i < d.num_proposals; ++i) {
        const std::string pre = "proposal_networks." + std::to_string(i) + ".mlp_base.";
        if (!h->table_prop[i].ptr) return fail(h, SN_ERR_STATE, "missing " + pre + "encoder.hash_table");
        const std::vector<float>* w0 = find(h, pre + "mlp.layers.0.weight", 16 * 10);
        const std::vector<float>* b0 = find(h, pre + "mlp.layers.0.bias", 16);
        const std::vector<float>* w1 = find(h, pre + "mlp.layers.1.weight", 16);
        const std::vector<float>* b1 = find(h, pre + "mlp.layers.1.bias", 1);
        if (!w0 || !b0 || !w1 || !b1) return fail(h, SN_ERR_STATE, "missing or mis-sized MLP parameter under " + pre);
        // range conditioning of the net's one matrix-core layer (see plan_split_scales): features carry t0p (stored in the net's
        // de-hashed copies and paired tables), the hidden layer s1p; both are divided out by the weights around them
        double t0p = 1.0, s1p = 1.0;
        if (std::isfinite(absmax_prop[i]) && absmax_prop[i] > 0.0f) {
            const double M = absmax_prop[i];
            t0p = pow2_floor(1024.0 / M);
            double m1 = 0.0;
            for (int n = 0; n < 16; ++n) {
                double a = std::fabs((*b0)[n]);
                for (int k = 0; k < 10; ++k) a += std::fabs((*w0)[n * 10 + k]) * M;
                m1 = std::max(m1, a);
            }
            if (std::isfinite(m1) && m1 > 0.0) s1p = pow2_floor(16384.0 / m1);
            bool fits = true;
            for (int n = 0; n < 16; ++n) {
                fits = fits && std::fabs((*b0)[n] * s1p) <= 65504.0;
                for (int k = 0; k < 10; ++k) fits = fits && std::fabs((*w0)[n * 10 + k] * s1p / t0p) <= 65504.0;
            }
            if (!fits) t0p = s1p = 1.0;  // leave this net unconditioned (it only places samples)
        }
        h->feat_scale_prop[i] = (float)t0p;
        std::vector<float> pack(SN_PROP_PACK_FLOATS, 0.0f);
        // W0 is stored k-major ([k][n]) so that two neighbouring hidden units share a register pair (v_pk_fma_f32)
        for (int n = 0; n < 16; ++n)
            for (int k = 0; k < 10; ++k) pack[SN_PROP_W0 + k * 16 + n] = (float)((*w0)[n * 10 + k] / t0p);
        memcpy(pack.data() + SN_PROP_B0, b0->data(), 16 * 4);
        memcpy(pack.data() + SN_PROP_W1, w1->data(), 16 * 4);
        pack[SN_PROP_B1] = (*b1)[0];
        // matrix-core form (sn_prop_mlp_mfma): two A operands [lane][e], A[row = lane & 31][k = 8 (lane >> 5) + e], fp16 hi / lo
        // (lo = RNE(x - hi)).  Rows 0..15 are the hidden units for the rays of lanes 0..31 (k = 0..7), rows 16..31 the same units for
        // the rays of lanes 32..63 (k = 8..15); the other half of every row is zero.  Operand 1: e <-> W0[unit][e]; operand 2:
        // e = 0, 1 <-> W0[unit][8], W0[unit][9], e = 2 <-> b0[unit].
        {
            uint16_t* a1hi = (uint16_t*)(pack.data() + SN_PROP_MA1_HI);
            uint16_t* a1lo = (uint16_t*)(pack.data() + SN_PROP_MA1_LO);
            uint16_t* a2hi = (uint16_t*)(pack.data() + SN_PROP_MA2_HI);
            uint16_t* a2lo = (uint16_t*)(pack.data() + SN_PROP_MA2_LO);
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 8; ++e) {
                    const int row = lane & 31, half = lane >> 5, unit = row & 15;
                    float x1 = 0.0f, x2 = 0.0f;
                    if ((row >> 4) == half) {
                        x1 = (float)((*w0)[unit * 10 + e] * s1p / t0p);
                        x2 = e < 2 ? (float)((*w0)[unit * 10 + 8 + e] * s1p / t0p) : (e == 2 ? (float)((*b0)[unit] * s1p) : 0.0f);
                    }
                    const uint16_t h1 = f32_to_f16_rne(x1), h2 = f32_to_f16_rne(x2);
                    a1hi[lane * 8 + e] = h1;
                    a1lo[lane * 8 + e] = f32_to_f16_rne(x1 - f16_to_f32(h1));
                    a2hi[lane * 8 + e] = h2;
                    a2lo[lane * 8 + e] = f32_to_f16_rne(x2 - f16_to_f32(h2));
                }
            // layer 2 as (w x + w |x|) / 2: MW1 holds w / 2 in accumulator order (per unit of the s1p-scaled accumulators), LIN the linear half
            for (int hh = 0; hh < 2; ++hh)
                for (int r = 0; r < 8; ++r) pack[SN_PROP_MW1 + hh * 8 + r] = (float)(0.5 * (*w1)[(r & 3) + 8 * (r >> 2) + 4 * hh] / s1p);
            double cl = 0.0;
            for (int n = 0; n < 16; ++n) cl += (double)(*w1)[n] * (double)(*b0)[n];
            for (int k = 0; k < 10; ++k) {
                double v = 0.0;
                for (int n = 0; n < 16; ++n) v += (double)(*w1)[n] * (double)(*w0)[n * 10 + k];
                pack[SN_PROP_LIN + k] = (float)(0.5 * v / t0p);
            }
            pack[SN_PROP_LIN + 10] = (float)(0.5 * cl + (double)(*b1)[0]);
        }
        if (!h->wpack_prop[i].ptr) {
            SN_HIP(h, hipMalloc(&h->wpack_prop[i].ptr, pack.size() * 4));
            h->wpack_prop[i].bytes = pack.size() * 4;
        }
        SN_HIP(h, hipMemcpyAsync(h->wpack_prop[i].ptr, pack.data(), pack.size() * 4, hipMemcpyHostToDevice, st));
    }
    // de-hashed copies of the coarse levels
    {
        // the budget comes from the descriptor (SnFieldDesc.dense_levels / dense_copy_cap_mb: 0 = default, -1 = no copies).  The main kernel
        // is instantiated for SN_DENSE_LEVELS_DEFAULT (11) and SN_BC_MAIN (9: the coefficient-form levels alone, 0.51 GB) copied levels:
        // what the budget and the per-level cap allow is rounded DOWN to one of those counts (or to none)
        const int asked = d.dense_levels == 0 ? SN_DENSE_LEVELS_DEFAULT : std::max(0, d.dense_levels);
        const int want = std::max(0, std::min(asked, 12));
        const uint64_t cap_main = d.dense_copy_cap_mb > 0 ? (uint64_t)d.dense_copy_cap_mb : 600;
        int want_main = dense_levels_under_cap(d.main_field, want, cap_main);
        want_main = want_main >= SN_DENSE_LEVELS_DEFAULT ? SN_DENSE_LEVELS_DEFAULT : (want_main >= SN_BC_MAIN ? SN_BC_MAIN : 0);
        if (int rc = build_dense_copies(h, d.main_field, h->table_main, want_main, cap_main, h->dense_main, h->dense_info, h->dense_res, h->nd_torch, st,
                                        SN_BC_MAIN, h->feat_scale_main))
            return rc;
        for (int i = 0; i < d.num_proposals; ++i)
            if (int rc = build_dense_copies(h, d.proposals[i], h->table_prop[i], want, 100, h->dense_prop[i], h->dense_info_prop[i],
                                            h->dense_res_prop[i], h->nd_prop[i], st, SN_BC_PROP, h->feat_scale_prop[i]))
                return rc;
        // single-fp16 mode: the main grid once more in fp16 storage (tiny-cuda-nn grids whose copies cover the densely indexed levels)
        const char* hg = getenv("SN_HALF_GRID");   // (diagnostics: 0 forces the fp32-table path of the mode)
        const int td = d.main_field.grid_mode == 1 ? leading_dense(d.main_field) : -1;
        if (d.half_grid == 1 && !(hg && atoi(hg) == 0) && d.main_field.grid_mode == 1 && h->split_ok && h->nd_torch > 0 && td >= 0 && td <= h->nd_torch) {
            if (int rc = build_half_grid(h, d.main_field, h->table_main, h->dense_info, h->nd_torch, h->hquads_main, h->hquads_info, h->hrows_main, h->hpinfo_main,
                                         st, h->feat_scale_main, h->table_absmax_main))
                return rc;
        } else {
            h->hquads_main.release();
            h->hrows_main.release();
            memset(&h->hquads_info, 0, sizeof(h->hquads_info));
        }
    }
#if SN_MAIN_PAIRS
    // main grid, torch semantics: the levels beyond the de-hashed ones from x-paired tables (4 gathers per level instead of 8)
    if (h->nd_torch > 0 && d.num_proposals > 0) {  // read by the bins-mode kernel only (sn_main.h)
        if (int rc = build_pairs(h, d.main_field, h->table_main, h->pairs_main, h->pinfo_main, st, h->feat_scale_main, h->nd_torch)) return rc;
    } else {
        h->pairs_main.release();
    }
#endif
    for (int i = 0; i < d.num_proposals; ++i)
        if (int rc = build_pairs(h, d.proposals[i], h->table_prop[i], h->pairs_prop[i], h->pinfo_prop[i], st, h->feat_scale_prop[i])) return rc;
    SN_HIP(h, hipGetLastError());
    SN_HIP(h, hipStreamSynchronize(st));
    mark_weights_written(h, st);
    {
        std::lock_guard<std::mutex> g(h->mu);
        h->finalized = true;
    }
    return SN_OK;
}

int sn_generate_rays_camera(const SnCameraDesc* cam, const float* coords, int64_t n_coords, float* origins, float* directions,
                            float* pixel_area, float* directions_norm, const float* aabb, float* nears, float* fars, SnStream stream) {
    if (!cam || cam->height <= 0 || cam->width <= 0 || (coords && n_coords < 0))
        return fail(nullptr, SN_ERR_INVALID, "sn_generate_rays_camera: bad argument");
    if (cam->camera_type != SN_CAMERA_PERSPECTIVE && cam->camera_type != SN_CAMERA_FISHEYE && cam->camera_type != SN_CAMERA_EQUIRECTANGULAR)
        return fail(nullptr, SN_ERR_INVALID, "sn_generate_rays_camera: camera_type " + std::to_string(cam->camera_type) +
                                                 " is not supported (1 = PERSPECTIVE, 2 = FISHEYE, 3 = EQUIRECTANGULAR)");
    SnRayGenParams p;
    memcpy(p.c2w, cam->c2w, sizeof(p.c2w));
    p.fx = cam->fx;
    p.fy = cam->fy;
    p.cx = cam->cx;
    p.cy = cam->cy;
    p.height = cam->height;
    p.width = cam->width;
    p.camera_type = cam->camera_type;
    p.has_distortion = cam->has_distortion != 0 && cam->camera_type != SN_CAMERA_EQUIRECTANGULAR;  // (nerfstudio never un-distorts equirectangular images)
    memcpy(p.dist, cam->distortion, sizeof(p.dist));
    p.coords = coords;
    p.n = coords ? n_coords : (int64_t)cam->height * cam->width;
    p.origins = origins;
    p.directions = directions;
    p.pixel_area = pixel_area;
    p.directions_norm = directions_norm;
    p.has_aabb = aabb != nullptr;
    if (aabb) memcpy(p.aabb, aabb, sizeof(p.aabb));
    else memset(p.aabb, 0, sizeof(p.aabb));
    p.nears = nears;
    p.fars = fars;
    if (p.n == 0) return SN_OK;
    const dim3 grid((unsigned)((p.n + 255) / 256)), block(256);
    const bool fish = cam->camera_type == SN_CAMERA_FISHEYE;
    if (cam->camera_type == SN_CAMERA_EQUIRECTANGULAR) hipLaunchKernelGGL((sn_generate_rays_kernel<3, false>), grid, block, 0, (hipStream_t)stream, p);
    else if (fish && p.has_distortion) hipLaunchKernelGGL((sn_generate_rays_kernel<2, true>), grid, block, 0, (hipStream_t)stream, p);
    else if (fish) hipLaunchKernelGGL((sn_generate_rays_kernel<2, false>), grid, block, 0, (hipStream_t)stream, p);
    else if (p.has_distortion) hipLaunchKernelGGL((sn_generate_rays_kernel<1, true>), grid, block, 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((sn_generate_rays_kernel<1, false>), grid, block, 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_generate_rays launch: ") + hipGetErrorString(e));
    return SN_OK;
}

int sn_generate_rays(const float* c2w, float fx, float fy, float cx, float cy, int32_t height, int32_t width, float* origins,
                     float* directions, float* pixel_area, float* directions_norm, const float* aabb, float* nears, float* fars,
                     SnStream stream) {
    if (!c2w || height <= 0 || width <= 0) return fail(nullptr, SN_ERR_INVALID, "sn_generate_rays: bad argument");
    SnCameraDesc cam;
    memset(&cam, 0, sizeof(cam));
    memcpy(cam.c2w, c2w, sizeof(cam.c2w));
    cam.fx = fx;
    cam.fy = fy;
    cam.cx = cx;
    cam.cy = cy;
    cam.height = height;
    cam.width = width;
    cam.camera_type = SN_CAMERA_PERSPECTIVE;
    return sn_generate_rays_camera(&cam, nullptr, 0, origins, directions, pixel_area, directions_norm, aabb, nears, fars, stream);
}

int sn_intersect_with_aabb(const float* origins, const float* directions, int64_t n_rays, const float* aabb, float* nears,
                           float* fars, SnStream stream) {
    if (!origins || !directions || !aabb || !nears || !fars || n_rays < 0)
        return fail(nullptr, SN_ERR_INVALID, "sn_intersect_with_aabb: bad argument");
    if (n_rays == 0) return SN_OK;
    SnAabb box;
    memcpy(box.v, aabb, sizeof(box.v));
    hipLaunchKernelGGL(sn_intersect_with_aabb_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       origins, directions, n_rays, box, nears, fars);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_intersect_with_aabb launch: ") + hipGetErrorString(e));
    return SN_OK;
}

int sn_intersect_obb(const float* origins, const float* directions, int64_t n_rays, const float* world2box, const float* size,
                     float* nears, float* fars, SnStream stream) {
    if (!origins || !directions || !world2box || !size || !nears || !fars || n_rays < 0)
        return fail(nullptr, SN_ERR_INVALID, "sn_intersect_obb: bad argument");
    if (n_rays == 0) return SN_OK;
    SnObb box;
    memcpy(box.w2b, world2box, sizeof(box.w2b));
    for (int c = 0; c < 3; ++c) box.half[c] = size[c] / 2.0f;
    hipLaunchKernelGGL(sn_intersect_obb_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, origins,
                       directions, n_rays, box, nears, fars);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_intersect_obb launch: ") + hipGetErrorString(e));
    return SN_OK;
}

int sn_debug_sample_positions(const float* origins, const float* directions, const float* starts, const float* ends, int64_t n,
                              float* q_strict, float* q_exact, float* q_fast, SnStream stream) {
    if (!origins || !directions || !starts || !ends || n < 0) return fail(nullptr, SN_ERR_INVALID, "sn_debug_sample_positions: bad argument");
    if (n == 0) return SN_OK;
    hipLaunchKernelGGL(sn_debug_sample_positions_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, origins,
                       directions, starts, ends, n, q_strict, q_exact, q_fast);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_debug_sample_positions launch: ") + hipGetErrorString(e));
    return SN_OK;
}

size_t sn_workspace_bytes(SnHandle h, int32_t height, int32_t width, const SnRenderOpts* opts_in) {
    if (!h || !opts_in || height <= 0 || width <= 0) return 0;
    SnRenderOpts own;
    if (adopt_struct(h, opts_in, kRenderOptsMin, own, "sn_workspace_bytes: SnRenderOpts")) return 0;  // (the text is in sn_last_error)
    const SnRenderOpts* opts = &own;
    if (opts->chunk_rays < 1) return 0;
    return plan_workspace(height, width, *opts, h->n_cus, true).total;  // (the size with the tail split on covers both settings of SN_TAIL_SPLIT)
}

// K2: proposal sampler (rows a8-a12) -> final euclidean bins [tile][S+1][64] in the workspace.  Shared by the colour render and
// the normals render.
static int launch_proposals(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars,
                            int32_t height, int32_t width, const SnRenderOpts* opts, const WorkspacePlan& wp, const TileGeom& g, char* ws,
                            float* prop_depth_0, float* prop_depth_1, hipStream_t st, float** ebins_out, const SnDebugDump* dump = nullptr) {
    const SnFieldDesc& d = h->desc;
    const int nprop = opts->num_proposal_iterations;
    const float* d_sbins = opts->initial_spacing_bins;
    float* d_ebins = (float*)(ws + wp.off_ebins);
    SnPropParams pp;
    memset(&pp, 0, sizeof(pp));
    pp.origins = origins;
    pp.directions = directions;
    pp.nears = nears;
    pp.fars = fars;
    pp.sbins0 = d_sbins;
    for (int i = 0; i < SN_MAX_PROPOSALS; ++i) pp.pdf_u[i] = opts->pdf_u[i];
    pp.ebins_out = d_ebins;
    pp.scratch = (float*)(ws + wp.off_prop_scratch);
    {
        // the tile queue of the persistent waves (sn_proposal.h): only when some wave gets more than one tile
        const int n_waves = wp.prop_blocks * SN_PROP_WAVES, n_tiles = g.tiles_x * g.tiles_y;
        pp.tile_counter = nullptr;
        if (n_tiles > n_waves) {
            pp.tile_counter = (unsigned int*)(ws + wp.off_prop_counter);
            SN_HIP(h, hipMemsetD32Async((hipDeviceptr_t)pp.tile_counter, n_waves, 1, st));
        }
    }
    pp.cache_off = h->sw.prop_cache_off.load(std::memory_order_relaxed);
    pp.early_term = h->sw.early_term.load(std::memory_order_relaxed);
    pp.march_stats = (unsigned long long*)opts->march_stats;
    pp.spacing_uniform = opts->spacing_mode;
    pp.pm = h->pos_map;
    pp.pdf_ieee = h->sw.pdf_ieee.load(std::memory_order_relaxed);
    pp.prop_depth[0] = prop_depth_0;
    pp.prop_depth[1] = prop_depth_1;
    pp.height = height;
    pp.width = width;
    pp.tile_w_log2 = g.tw_log2;
    pp.tile_h_log2 = g.th_log2;
    pp.tiles_x = g.tiles_x;
    pp.tiles_y = g.tiles_y;
    pp.n_levels = nprop;
    for (int i = 0; i < nprop; ++i) {
        pp.pairs[i] = (const float*)h->pairs_prop[i].ptr;
        pp.pinfo[i] = h->pinfo_prop[i];
        pp.pairs_bytes[i] = (uint32_t)h->pairs_prop[i].bytes;
        pp.wpack[i] = (const float*)h->wpack_prop[i].ptr;
        pp.tables[i] = (const float*)h->table_prop[i].ptr;
        pp.table_bytes[i] = (uint32_t)h->table_prop[i].bytes;
        pp.grid[i] = grid_levels(d.proposals[i]);
        pp.feat_scale[i] = h->feat_scale_prop[i];
        pp.dense[i] = h->dense_info_prop[i];
        pp.log2_t[i] = d.proposals[i].log2_hashmap_size;
        for (int l = 0; l < 5; ++l) pp.scal[i][l] = d.proposals[i].scalings[l];
        pp.n_samples[i] = opts->num_proposal_samples[i];
    }
    pp.n_final = opts->num_nerf_samples;
    pp.near_plane = opts->near_plane;
    pp.far_plane = opts->far_plane;
    pp.avg_density = d.average_init_density;
    pp.hist_pad = d.histogram_padding;
    const dim3 pgrid((unsigned)wp.prop_blocks), pblock(64 * SN_PROP_WAVES);
    // the non-default sampler / position map (SnRenderOpts.spacing_mode, SnFieldDesc.disable_scene_contraction) run in their own
    // instantiations -- the run-time-generic ones: uploaded tables, no de-hashed copies -- so the production kernels' code does not change
    const bool alt = needs_generic_kernels(h, opts);
    if (dump && alt) return fail(h, SN_ERR_INVALID, "sn_render_rays_debug: the dump exists for the default sampler and scene contraction only");
    if (pp.march_stats && (dump || alt || d.proposals[0].grid_mode != 0 || !(nprop == 2 && h->nd_prop[0] == 5 && h->nd_prop[1] == 4)))
        return fail(h, SN_ERR_INVALID, "SnRenderOpts.march_stats: the counting instantiation of the proposal kernel exists for the default variant only (torch grid, 2 nets, 5 + 4 de-hashed levels, default sampler)");
    if (alt) {
        if (d.proposals[0].grid_mode == 1) hipLaunchKernelGGL((sn_proposal_kernel<1, -1, -1, false, true>), pgrid, pblock, 0, st, pp);
        else hipLaunchKernelGGL((sn_proposal_kernel<0, -1, -1, false, true>), pgrid, pblock, 0, st, pp);
    } else
    if (dump) {
        // the instrumented instantiation exists for the production variant of nerfacto's proposal nets only
        if (!(d.proposals[0].grid_mode == 0 && nprop == 2 && h->nd_prop[0] == 5 && h->nd_prop[1] == 4))
            return fail(h, SN_ERR_INVALID, "sn_render_rays_debug: the proposal-kernel dump exists for the default variant only (torch grid, 2 nets, 5 + 4 de-hashed levels)");
        for (int i = 0; i < SN_MAX_PROPOSALS; ++i) {
            pp.dump_fetch[i] = dump->prop_fetch[i];
            pp.dump_pdf[i] = dump->pdf_index[i];
            pp.dump_q[i] = dump->prop_q[i];
        }
        for (int i = 0; i < nprop; ++i) pp.grid[i] = h->dense_res_prop[i];
        hipLaunchKernelGGL((sn_proposal_kernel<0, 5, 4, true>), pgrid, pblock, 0, st, pp);
    } else
    if (d.proposals[0].grid_mode == 1) {
        // nerfacto's proposal nets (max_res 128 / 256): 5 and 4 levels have de-hashed copies, which must cover every level tiny-cuda-nn
        // indexes densely (3 and 2 at T = 2^17); other shapes read the uploaded tables with the per-level run-time decision
        const int td0 = leading_dense(d.proposals[0]), td1 = nprop > 1 ? leading_dense(d.proposals[1]) : 0;
        if (nprop == 2 && h->nd_prop[0] == 5 && h->nd_prop[1] == 4 && td0 >= 0 && td0 <= 5 && td1 >= 0 && td1 <= 4) {
            for (int i = 0; i < nprop; ++i) pp.grid[i] = h->dense_res_prop[i];
            hipLaunchKernelGGL((sn_proposal_kernel<1, 5, 4>), pgrid, pblock, 0, st, pp);
        } else {
            hipLaunchKernelGGL((sn_proposal_kernel<1, -1, -1>), pgrid, pblock, 0, st, pp);
        }
    } else if (nprop == 2 && h->nd_prop[0] == 5 && h->nd_prop[1] == 4) {
        // nerfacto's nets (max_res 128 / 256): every level but the finest of the second net has a de-hashed copy
        for (int i = 0; i < nprop; ++i) pp.grid[i] = h->dense_res_prop[i];
        if (pp.march_stats) hipLaunchKernelGGL((sn_proposal_kernel<0, 5, 4, false, false, true>), pgrid, pblock, 0, st, pp);   // (the counting instantiation)
        else hipLaunchKernelGGL((sn_proposal_kernel<0, 5, 4>), pgrid, pblock, 0, st, pp);
    } else {
        hipLaunchKernelGGL((sn_proposal_kernel<0, -1, -1>), pgrid, pblock, 0, st, pp);
    }
    SN_HIP(h, hipGetLastError());
    *ebins_out = d_ebins;
    return SN_OK;
}

static int render_rays_impl(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars, int32_t height,
                            int32_t width, const SnRenderOpts* opts, float* rgb, float* depth, float* accumulation, float* expected_depth,
                            float* prop_depth_0, float* prop_depth_1, const SnDebugDump* dump, SnStream stream) {
    if (!h) return SN_ERR_INVALID;
    if (!origins || !directions || !opts || height <= 0 || width <= 0) return fail(h, SN_ERR_INVALID, "sn_render_rays: bad argument");
    SnRenderOpts opts_own;
    if (int rc = adopt_struct(h, opts, kRenderOptsMin, opts_own, "sn_render_rays: SnRenderOpts")) return rc;
    opts = &opts_own;
    if ((nears == nullptr) != (fars == nullptr)) return fail(h, SN_ERR_INVALID, "sn_render_rays: nears and fars must both be given or both be NULL");
    if (!h->finalized) return fail(h, SN_ERR_STATE, "sn_render_rays: weights not finalized");
    std::string why;
    if (!valid_opts(h->desc, *opts, why)) return fail(h, SN_ERR_INVALID, "sn_render_rays: " + why);
    WorkspacePlan wp = plan_workspace(height, width, *opts, h->n_cus, true);
    if (!opts->workspace || opts->workspace_bytes < wp.total)
        return fail(h, SN_ERR_WORKSPACE, "sn_render_rays: workspace too small, need " + std::to_string(wp.total) + " bytes");
    hipStream_t st = (hipStream_t)stream;
    RenderGuard guard(h, st);  // waits for the handle's last weight upload; records this render for later uploads (opt-in: process-wide chain)
    char* ws = (char*)opts->workspace;
    const SnFieldDesc& d = h->desc;
    const TileGeom g = tile_geometry(height, width);
    const int64_t n = (int64_t)height * width;
    const int nprop = opts->num_proposal_iterations;

    const float* d_sbins = opts->initial_spacing_bins;  // null => kernels fall back to i/N
    uint32_t* d_minmax = nullptr;
    float* d_exp_raw = nullptr;
    if (expected_depth) {
        d_minmax = (uint32_t*)(ws + wp.off_minmax);
        d_exp_raw = (float*)(ws + wp.off_exp_raw);
        SN_HIP(h, hipMemsetAsync(d_minmax, 0xff, (size_t)wp.n_chunks * 4, st));
        SN_HIP(h, hipMemsetAsync(d_minmax + wp.n_chunks, 0x00, (size_t)wp.n_chunks * 4, st));
    }

    // every refusal that depends on the options comes BEFORE the first launch (a call must not return SN_ERR_INVALID with the
    // proposal kernel already enqueued and its counters half updated)
    if (opts->march_stats) {
        const bool tcnn_ = d.main_field.grid_mode == 1, alt_ = needs_generic_kernels(h, opts);
        const bool split_ = opts->precision >= 1 && h->split_ok, half1_ = split_ && opts->precision == 2;
        if (half1_ || dump || alt_)
            return fail(h, SN_ERR_INVALID, "SnRenderOpts.march_stats: no counting instantiation for this variant (single fp16 / instrumented / generic sampler)");
        if (!(split_ && !tcnn_ && h->nd_torch == SN_DENSE_LEVELS_DEFAULT))
            return fail(h, SN_ERR_INVALID, "SnRenderOpts.march_stats: the counting instantiation of the main kernel exists for the default variant only (torch grid, 11 de-hashed levels, precision 1)");
    }
    clear_stamp(ws);  // whatever this workspace held is being overwritten
    float* d_ebins = nullptr;
    if (nprop > 0)
        if (int rc = launch_proposals(h, origins, directions, nears, fars, height, width, opts, wp, g, ws, prop_depth_0, prop_depth_1, st, &d_ebins, dump))
            return rc;

    SnMainParams p;
    memset(&p, 0, sizeof(p));
    p.origins = origins;
    p.directions = directions;
    p.nears = nears;
    p.fars = fars;
    p.sbins = d_sbins;
    p.ebins = d_ebins;
    p.spacing_uniform = opts->spacing_mode;
    p.pm = h->pos_map;
    p.table = (const float*)h->table_main.ptr;
    const bool split = opts->precision >= 1 && h->split_ok;  // (a handle whose weights cannot be range-conditioned renders in exact fp32)
    const bool half1 = split && opts->precision == 2;       // single fp16 (sn_main.h sn_main_field_f16): tiny-cuda-nn grids only (valid_opts)
    p.wimg = (const float*)(split ? h->wimg_main_h.ptr : h->wimg_main.ptr);
    p.feat_scale = h->feat_scale_main;
    p.hquads = h->hquads_info;
    p.hrows = (const float*)h->hrows_main.ptr;
    p.hrows_bytes = (uint32_t)h->hrows_main.bytes;
    p.hpinfo = h->hpinfo_main;
    p.pairs = (const float*)h->pairs_main.ptr;
    p.pairs_bytes = (uint32_t)h->pairs_main.bytes;
    p.pinfo = h->pinfo_main;
    p.rgb = rgb;
    p.depth = depth;
    p.acc = accumulation;
    p.exp_raw = d_exp_raw;
    p.chunk_minmax = d_minmax;
    p.n_chunks = wp.n_chunks;
    for (int l = 0; l < 16; ++l) p.scal[l] = d.main_field.scalings[l];
    p.height = height;
    p.width = width;
    p.n_samples = opts->num_nerf_samples;
    p.tile_w_log2 = g.tw_log2;
    p.tile_h_log2 = g.th_log2;
    p.tiles_x = g.tiles_x;
    p.tiles_y = g.tiles_y;
    p.log2_t = d.main_field.log2_hashmap_size;
    p.near_plane = opts->near_plane;
    p.far_plane = opts->far_plane;
    p.avg_density = d.average_init_density;
    p.sh_remap = d.sh_remap;
    p.chunk_rays = opts->chunk_rays;
    p.bg_mode = opts->background_mode;
    p.early_term = h->sw.early_term.load(std::memory_order_relaxed);
    p.march_stats = (unsigned long long*)opts->march_stats;
    for (int c = 0; c < 3; ++c) p.bg[c] = opts->background_rgb[c];
    const bool tcnn = d.main_field.grid_mode == 1;
    // de-hashed copies are used when they cover every level tiny-cuda-nn indexes densely (always true for torch grids and for nerfacto's
    // tcnn shapes); otherwise the run-time variant (ND = -1) reads the uploaded table
    const int td = tcnn ? leading_dense(d.main_field) : 0;
    const bool alt = needs_generic_kernels(h, opts);  // (launch_proposals: the generic instantiations)
    const bool use_copies = !alt && h->nd_torch > 0 && (!tcnn || (td >= 0 && td <= h->nd_torch));
    p.grid = grid_levels(d.main_field);
    if (use_copies) {
        p.grid = h->dense_res;
        p.dense = h->dense_info;
    }
    const int gbx = (g.tiles_x + 1) / 2, gby = (g.tiles_y + 1) / 2;
    // weight image + (uniform sampler) the frame's S + 1 euclidean bins
    const size_t etab_bytes = nprop == 0 ? ((size_t)opts->num_nerf_samples + 1 + 3) / 4 * 16 : 0;
    const size_t lds_bytes = (half1 ? (size_t)SnMainImgF16::TOTAL_BYTES : (size_t)SnMainImg::TOTAL * 4) + etab_bytes;
    // split-depth tail (plan_tail): the workgroups of the last, partly filled round become n_seg segment jobs each
    const bool tail_split = !dump && !half1 && !h->sw.tail_split_off.load(std::memory_order_relaxed) && wp.n_seg > 1;
    p.seg_first_block = gbx * gby;
    p.n_seg = 1;
    p.seg_len = opts->num_nerf_samples;
    int n_tail = 0;
    if (tail_split) {
        n_tail = gbx * gby - wp.seg_first_block;
        p.seg_first_block = wp.seg_first_block;
        p.n_seg = wp.n_seg;
        p.seg_len = wp.seg_len;
        p.seg_scratch = (f32x4*)(ws + wp.off_seg);
    }
    const dim3 grid((unsigned)(gbx * gby - n_tail + ((n_tail + 7) / 8 * 8) * p.n_seg)), block(256);  // (segment jobs: the tail padded to whole XCD rows)
#define SN_LAUNCH_MAIN(MODE, PREC, GRID, ND) \
    hipLaunchKernelGGL((sn_render_main_kernel<MODE, PREC, GRID, ND>), grid, block, lds_bytes, st, p)
    const int nd_launch = use_copies ? h->nd_torch : -1;
#define SN_LAUNCH_MAIN_ND(MODE, PREC, GRID)                          \
    switch (nd_launch) { /* the copy counts sn_finalize_weights can leave a handle with */ \
        case SN_BC_MAIN: SN_LAUNCH_MAIN(MODE, PREC, GRID, SN_BC_MAIN); break;       \
        case SN_DENSE_LEVELS_DEFAULT: SN_LAUNCH_MAIN(MODE, PREC, GRID, SN_DENSE_LEVELS_DEFAULT); break;     \
        default: SN_LAUNCH_MAIN(MODE, PREC, GRID, -1); break;     \
    }
#define SN_LAUNCH_MAIN_TORCH(MODE, PREC) SN_LAUNCH_MAIN_ND(MODE, PREC, 0)
#define SN_LAUNCH_MAIN_TCNN(MODE, PREC) SN_LAUNCH_MAIN_ND(MODE, PREC, 1)
    if (p.march_stats && (half1 || dump || alt))
        return fail(h, SN_ERR_INVALID, "SnRenderOpts.march_stats: no counting instantiation for this variant (single fp16 / instrumented / generic sampler)");
    if (half1) {
        // single-fp16 mode: the tiny-cuda-nn grid's kernels, with the de-hashed copies when the handle has the default 11 of them
        if (dump || alt) return fail(h, SN_ERR_INVALID, "precision 2 (single fp16) has no instrumented / generic-sampler instantiation");
        // ND = 11: the grid from its fp16 storage (SnFieldDesc.half_grid); otherwise the run-time variant on the uploaded fp32 table with every
        // row rounded through fp16 on the fly -- the same values
        const bool hgrid = nd_launch == 11 && h->hquads_main.ptr && h->hrows_main.ptr;
        if (!hgrid) p.grid = grid_levels(d.main_field);
        if (nprop > 0) {
            if (hgrid) SN_LAUNCH_MAIN(1, 2, 1, 11); else SN_LAUNCH_MAIN(1, 2, 1, -1);
        } else {
            if (hgrid) SN_LAUNCH_MAIN(0, 2, 1, 11); else SN_LAUNCH_MAIN(0, 2, 1, -1);
        }
    } else
    if (dump) {
        // instrumented instantiations: the production variant (torch grid, 11 de-hashed levels), both samplers, both precisions
        if (tcnn || h->nd_torch != 11 || alt)
            return fail(h, SN_ERR_INVALID, "sn_render_rays_debug: the main-kernel dump exists for the default variant only (torch grid, 11 de-hashed levels, default sampler and scene contraction)");
        p.dump_fetch = dump->main_fetch;
        p.dump_q = dump->main_q;
        p.dump_median = dump->median_index;
        if (nprop > 0) {
            if (split) hipLaunchKernelGGL((sn_render_main_kernel<1, 1, 0, 11, true>), grid, block, lds_bytes, st, p);
            else hipLaunchKernelGGL((sn_render_main_kernel<1, 0, 0, 11, true>), grid, block, lds_bytes, st, p);
        } else {
            if (split) hipLaunchKernelGGL((sn_render_main_kernel<0, 1, 0, 11, true>), grid, block, lds_bytes, st, p);
            else hipLaunchKernelGGL((sn_render_main_kernel<0, 0, 0, 11, true>), grid, block, lds_bytes, st, p);
        }
    } else if (alt) {
#define SN_LAUNCH_MAIN_ALT(MODE, PREC, GRID) hipLaunchKernelGGL((sn_render_main_kernel<MODE, PREC, GRID, -1, false, true>), grid, block, lds_bytes, st, p)
        if (nprop > 0) {
            if (split) { if (tcnn) SN_LAUNCH_MAIN_ALT(1, 1, 1); else SN_LAUNCH_MAIN_ALT(1, 1, 0); }
            else { if (tcnn) SN_LAUNCH_MAIN_ALT(1, 0, 1); else SN_LAUNCH_MAIN_ALT(1, 0, 0); }
        } else {
            if (split) { if (tcnn) SN_LAUNCH_MAIN_ALT(0, 1, 1); else SN_LAUNCH_MAIN_ALT(0, 1, 0); }
            else { if (tcnn) SN_LAUNCH_MAIN_ALT(0, 0, 1); else SN_LAUNCH_MAIN_ALT(0, 0, 0); }
        }
#undef SN_LAUNCH_MAIN_ALT
    } else
    if (p.march_stats) {
        // the counting instantiations (diagnostics): the production variant of nerfacto's torch grid, split precision, both samplers
        if (!(split && !tcnn && nd_launch == SN_DENSE_LEVELS_DEFAULT))
            return fail(h, SN_ERR_INVALID, "SnRenderOpts.march_stats: the counting instantiation of the main kernel exists for the default variant only (torch grid, 11 de-hashed levels, precision 1)");
        if (nprop > 0) hipLaunchKernelGGL((sn_render_main_kernel<1, 1, 0, SN_DENSE_LEVELS_DEFAULT, false, false, true>), grid, block, lds_bytes, st, p);
        else hipLaunchKernelGGL((sn_render_main_kernel<0, 1, 0, SN_DENSE_LEVELS_DEFAULT, false, false, true>), grid, block, lds_bytes, st, p);
    } else
    if (nprop > 0) {
        if (split) { if (tcnn) { SN_LAUNCH_MAIN_TCNN(1, 1) } else { SN_LAUNCH_MAIN_TORCH(1, 1) } }
        else { if (tcnn) { SN_LAUNCH_MAIN_TCNN(1, 0) } else { SN_LAUNCH_MAIN_TORCH(1, 0) } }
    } else {
        if (split) { if (tcnn) { SN_LAUNCH_MAIN_TCNN(0, 1) } else { SN_LAUNCH_MAIN_TORCH(0, 1) } }
        else { if (tcnn) { SN_LAUNCH_MAIN_TCNN(0, 0) } else { SN_LAUNCH_MAIN_TORCH(0, 0) } }
    }
#undef SN_LAUNCH_MAIN_TCNN
#undef SN_LAUNCH_MAIN_TORCH
#undef SN_LAUNCH_MAIN_ND
#undef SN_LAUNCH_MAIN
    SN_HIP(h, hipGetLastError());
    if (tail_split) {
        if (nprop > 0) hipLaunchKernelGGL((sn_main_combine_kernel<1>), dim3((unsigned)n_tail), block, 0, st, p);
        else hipLaunchKernelGGL((sn_main_combine_kernel<0>), dim3((unsigned)n_tail), block, etab_bytes, st, p);
        SN_HIP(h, hipGetLastError());
    }
    if (expected_depth) {
        hipLaunchKernelGGL(sn_clip_expected_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_exp_raw, d_minmax, n,
                           opts->chunk_rays, wp.n_chunks, expected_depth);
        SN_HIP(h, hipGetLastError());
    }
    if (nprop > 0) stamp_workspace(ws, make_stamp(h, origins, directions, nears, fars, height, width, *opts));  // SnRenderOpts.reuse_final_bins
    return SN_OK;
}

int sn_render_rays(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars, int32_t height,
                   int32_t width, const SnRenderOpts* opts, float* rgb, float* depth, float* accumulation, float* expected_depth,
                   float* prop_depth_0, float* prop_depth_1, SnStream stream) {
    return render_rays_impl(h, origins, directions, nears, fars, height, width, opts, rgb, depth, accumulation, expected_depth, prop_depth_0,
                            prop_depth_1, nullptr, stream);
}

int sn_render_rays_debug(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars, int32_t height,
                         int32_t width, const SnRenderOpts* opts, float* rgb, float* depth, float* accumulation, float* expected_depth,
                         float* prop_depth_0, float* prop_depth_1, const SnDebugDump* dump, SnStream stream) {
    if (!dump) return fail(h, SN_ERR_INVALID, "sn_render_rays_debug: null dump");
    return render_rays_impl(h, origins, directions, nears, fars, height, width, opts, rgb, depth, accumulation, expected_depth, prop_depth_0,
                            prop_depth_1, dump, stream);
}

int sn_effective_precision(SnHandle h, int32_t requested, int32_t kernel) {
    if (!h || requested < 0 || requested > 2 || (kernel != 0 && kernel != 1)) return -1;
    if (!h->finalized) return -1;
    if (requested == 0) return 0;
    if (requested == 2) {  // single fp16: the render / field kernels of a tiny-cuda-nn grid; the normals kernel keeps the split form
        if (h->desc.main_field.grid_mode != 1) return -1;
        if (kernel == 0) return h->split_ok ? 2 : 0;
    }
    return (kernel == 0 ? h->split_ok : h->normals_split_ok) ? 1 : 0;
}

int sn_debug_layout(SnHandle h, int32_t which, SnDebugLayout* out) {
    if (!h || !out) return fail(h, SN_ERR_INVALID, "sn_debug_layout: null argument");
    if (which < -1 || which >= h->desc.num_proposals) return fail(h, SN_ERR_INVALID, "sn_debug_layout: bad field selector");
    if (!h->finalized) return fail(h, SN_ERR_STATE, "sn_debug_layout: weights not finalized");
    // the caller's struct may be SHORTER than this library's (include/signerf_hip.h "ABI evolution"): fill in our own, copy out its size
    SnDebugLayout* const caller_out = out;
    SnDebugLayout own;
    if (int rc = adopt_struct(h, caller_out, kDebugLayoutMin, own, "sn_debug_layout: SnDebugLayout")) return rc;
    uint32_t caller_size = 0;
    memcpy(&caller_size, (const void*)caller_out, sizeof(caller_size));
    out = &own;
    memset(out, 0, sizeof(*out));
    out->struct_size = caller_size;
    const SnDenseCopy& dc = which < 0 ? h->dense_info : h->dense_info_prop[which];
    out->n_dense = which < 0 ? h->nd_torch : h->nd_prop[which];
    for (int l = 0; l < 12; ++l) {
        out->dense_res[l] = dc.res[l];
        out->dense_off[l] = dc.off[l];
    }
    out->n_bc = (int32_t)dc.n_bc;
    out->dense_bytes = out->n_dense > 0 ? (which < 0 ? h->dense_main.bytes : h->dense_prop[which].bytes) : 0;
    out->feature_scale = which < 0 ? h->feat_scale_main : h->feat_scale_prop[which];
    const SnPairInfo& pi = which < 0 ? h->pinfo_main : h->pinfo_prop[which];
    for (int l = 0; l < SN_MAX_LEVELS; ++l) out->pair_base[l] = pi.base[l];
    out->pair_bytes = which < 0 ? h->pairs_main.bytes : h->pairs_prop[which].bytes;  // (main field: only for models with proposal nets)
    out->table_bytes = which < 0 ? h->table_main.bytes : h->table_prop[which].bytes;
    uint64_t total = h->table_main.bytes + h->wimg_main.bytes + h->wimg_main_h.bytes + h->wimg_normals.bytes + h->wimg_normals_h.bytes +
                     h->pairs_main.bytes + h->dense_main.bytes + h->hquads_main.bytes + h->hrows_main.bytes;
    out->half_grid_bytes = which < 0 ? h->hquads_main.bytes + h->hrows_main.bytes : 0;
    for (int i = 0; i < SN_MAX_PROPOSALS; ++i)
        total += h->table_prop[i].bytes + h->pairs_prop[i].bytes + h->wpack_prop[i].bytes + h->dense_prop[i].bytes;
    out->handle_bytes = total;
    memcpy((void*)caller_out, (const void*)&own, caller_size);
    return SN_OK;
}

int sn_debug_read(SnHandle h, int32_t which, int32_t what, void* dst, size_t bytes, SnStream stream) {
    if (!h || !dst) return fail(h, SN_ERR_INVALID, "sn_debug_read: null argument");
    if (which < -1 || which >= h->desc.num_proposals) return fail(h, SN_ERR_INVALID, "sn_debug_read: bad field selector");
    if (!h->finalized) return fail(h, SN_ERR_STATE, "sn_debug_read: weights not finalized");
    const DevBuf* src = nullptr;
    if (what == 0) src = which < 0 ? &h->dense_main : &h->dense_prop[which];
    else if (what == 1) src = which < 0 ? &h->pairs_main : &h->pairs_prop[which];
    if (!src || !src->ptr) return fail(h, SN_ERR_INVALID, "sn_debug_read: no such buffer");
    if (bytes != src->bytes) return fail(h, SN_ERR_INVALID, "sn_debug_read: expected " + std::to_string(src->bytes) + " bytes");
    SN_HIP(h, hipMemcpyAsync(dst, src->ptr, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SN_OK;
}

// one wave per workgroup: shader-clock cycles (s_memtime) elapsed while the constant-rate wall clock (s_memrealtime) advances by `ticks`.
// EIGHT workgroups, one per XCD (the dispatcher places block b on XCD b % 8): the XCDs of one part do not run at one clock under load
// (measured r04, profiles/r04_power_data_activity.txt: 1.84 .. 1.99 GHz side by side, persistently), so a single wave reports whichever die it
// lands on; the sums over the eight give the mean.
__global__ void sn_clock_probe_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    while (wall_clock64() - r0 < ticks) __builtin_amdgcn_s_sleep(32);
    const unsigned long long c1 = __builtin_readcyclecounter();
    const unsigned long long r1 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicAdd(&out[0], c1 - c0);
        atomicAdd(&out[1], r1 - r0);
    }
}

int sn_clock_probe(uint64_t* out, double seconds, SnStream stream) {
    if (!out || !(seconds > 0.0) || seconds > 1.0) return fail(nullptr, SN_ERR_INVALID, "sn_clock_probe: bad argument (0 < seconds <= 1)");
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0)
        return fail(nullptr, SN_ERR_HIP, "sn_clock_probe: cannot read the wall-clock rate of the device");
    const unsigned long long rate = (unsigned long long)khz * 1000ull;
    const unsigned long long ticks = (unsigned long long)(seconds * (double)rate);
    hipError_t e = hipMemsetAsync(out, 0, 16, (hipStream_t)stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(sn_clock_probe_kernel, dim3(8), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out, ticks);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out + 2, &rate, 8, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_clock_probe: ") + hipGetErrorString(e));
    return SN_OK;
}

int sn_render_normals(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars, int32_t height,
                      int32_t width, const SnRenderOpts* opts, float* normals, float* pred_normals, SnStream stream) {
    if (!h) return SN_ERR_INVALID;
    if (!origins || !directions || !opts || height <= 0 || width <= 0) return fail(h, SN_ERR_INVALID, "sn_render_normals: bad argument");
    SnRenderOpts opts_own;
    if (int rc = adopt_struct(h, opts, kRenderOptsMin, opts_own, "sn_render_normals: SnRenderOpts")) return rc;
    opts = &opts_own;
    if ((nears == nullptr) != (fars == nullptr)) return fail(h, SN_ERR_INVALID, "sn_render_normals: nears and fars must both be given or both be NULL");
    if (!h->finalized) return fail(h, SN_ERR_STATE, "sn_render_normals: weights not finalized");
    if (pred_normals && !h->has_pred_normals)
        return fail(h, SN_ERR_STATE, "sn_render_normals: field.mlp_pred_normals.* / field.field_head_pred_normals.net.* not uploaded");
    std::string why;
    if (!valid_opts(h->desc, *opts, why)) return fail(h, SN_ERR_INVALID, "sn_render_normals: " + why);
    if (!normals && !pred_normals) return SN_OK;
    const WorkspacePlan wp = plan_workspace(height, width, *opts, h->n_cus, true);
    if (!opts->workspace || opts->workspace_bytes < wp.total)
        return fail(h, SN_ERR_WORKSPACE, "sn_render_normals: workspace too small, need " + std::to_string(wp.total) + " bytes");
    hipStream_t st = (hipStream_t)stream;
    RenderGuard guard(h, st);
    char* ws = (char*)opts->workspace;
    const SnFieldDesc& d = h->desc;
    const TileGeom g = tile_geometry(height, width);
    const int nprop = opts->num_proposal_iterations;
    float* d_ebins = nullptr;
    if (nprop > 0) {
        const WorkspaceStamp want = make_stamp(h, origins, directions, nears, fars, height, width, *opts);
        if (opts->reuse_final_bins) {
            // the bins of the preceding sn_render_rays on this workspace: the host-side stamp of that call must describe THIS call (r06;
            // what no stamp can see -- foreign writes to the memory, rays changed in place, device ordering -- stays the caller's word)
            const std::string why_not = stamp_mismatch(ws, want);
            if (!why_not.empty()) return fail(h, SN_ERR_STATE, "sn_render_normals: reuse_final_bins: " + why_not);
            d_ebins = (float*)(ws + wp.off_ebins);
        } else {
            clear_stamp(ws);
            if (int rc = launch_proposals(h, origins, directions, nears, fars, height, width, opts, wp, g, ws, nullptr, nullptr, st, &d_ebins)) return rc;
            stamp_workspace(ws, want);  // (the proposal kernel is deterministic: these ARE the bins a colour render of the same call would leave)
        }
    } else {
        clear_stamp(ws);
    }
    SnNormalsParams p;
    memset(&p, 0, sizeof(p));
    p.origins = origins;
    p.directions = directions;
    p.nears = nears;
    p.fars = fars;
    p.sbins = opts->initial_spacing_bins;
    p.ebins = d_ebins;
    p.spacing_uniform = opts->spacing_mode;
    p.pm = h->pos_map;
    p.table = (const float*)h->table_main.ptr;
    const bool split = opts->precision >= 1 && h->normals_split_ok;  // (precision 2: the normals kernel keeps the split form)
    p.wimg = (const float*)(split ? h->wimg_normals_h.ptr : h->wimg_normals.ptr);
    p.normals = normals;
    p.pred_normals = pred_normals;
    for (int l = 0; l < 16; ++l) p.scal[l] = d.main_field.scalings[l];
    p.height = height;
    p.width = width;
    p.n_samples = opts->num_nerf_samples;
    p.tile_w_log2 = g.tw_log2;
    p.tile_h_log2 = g.th_log2;
    p.tiles_x = g.tiles_x;
    p.tiles_y = g.tiles_y;
    p.log2_t = d.main_field.log2_hashmap_size;
    p.near_plane = opts->near_plane;
    p.far_plane = opts->far_plane;
    p.avg_density = d.average_init_density;
    p.grid = grid_levels(d.main_field);
    p.pe_rev_scale = d.main_field.grid_mode == 1 ? 0.5f : 1.0f;  // tiny-cuda-nn's Frequency encoding runs at pi 2^k
    const int gbx = (g.tiles_x + 1) / 2, gby = (g.tiles_y + 1) / 2;
    const size_t lds_bytes = split ? (size_t)SnNormImgH::TOTAL_BYTES : (size_t)SnNormImg::TOTAL * 4;
    const dim3 grid((unsigned)(gbx * gby)), block(256);
    const bool tcnn = d.main_field.grid_mode == 1;
#define SN_LAUNCH_NORMALS(MODE, GRID, ND)                                                                      \
    if (split) hipLaunchKernelGGL((sn_normals_kernel<MODE, GRID, 1, ND>), grid, block, lds_bytes, st, p);      \
    else hipLaunchKernelGGL((sn_normals_kernel<MODE, GRID, 0, ND>), grid, block, lds_bytes, st, p)
    // nerfacto's torch grid with its default 11 de-hashed levels reads them (168 gathers per step instead of 256); other shapes and the
    // tiny-cuda-nn grid read the uploaded table
    const bool alt = needs_generic_kernels(h, opts);
    const bool copies = !alt && !tcnn && h->nd_torch == 11 && h->dense_main.ptr;
    if (copies) {
        p.dense = h->dense_info;
        p.inv_feat_scale = 1.0f / h->feat_scale_main;
    }
    p.feat_scale = split ? h->feat_scale_main : 1.0f;
    p.grad_scale = split ? h->grad_scale_normals : 1.0f;
#define SN_LAUNCH_NORMALS_ALT(MODE, GRID)                                                                         \
    if (split) hipLaunchKernelGGL((sn_normals_kernel<MODE, GRID, 1, -1, true>), grid, block, lds_bytes, st, p);   \
    else hipLaunchKernelGGL((sn_normals_kernel<MODE, GRID, 0, -1, true>), grid, block, lds_bytes, st, p)
    if (alt) {
        if (nprop > 0) { if (tcnn) { SN_LAUNCH_NORMALS_ALT(1, 1); } else { SN_LAUNCH_NORMALS_ALT(1, 0); } }
        else { if (tcnn) { SN_LAUNCH_NORMALS_ALT(0, 1); } else { SN_LAUNCH_NORMALS_ALT(0, 0); } }
    } else
    if (nprop > 0) {
        if (tcnn) { SN_LAUNCH_NORMALS(1, 1, -1); } else if (copies) { SN_LAUNCH_NORMALS(1, 0, 11); } else { SN_LAUNCH_NORMALS(1, 0, -1); }
    } else {
        if (tcnn) { SN_LAUNCH_NORMALS(0, 1, -1); } else if (copies) { SN_LAUNCH_NORMALS(0, 0, 11); } else { SN_LAUNCH_NORMALS(0, 0, -1); }
    }
#undef SN_LAUNCH_NORMALS_ALT
#undef SN_LAUNCH_NORMALS
    SN_HIP(h, hipGetLastError());
    return SN_OK;
}

int sn_hash_encode(SnHandle h, int32_t which, const float* q, int64_t n, float* features, int32_t* indices, SnStream stream) {
    if (!h) return SN_ERR_INVALID;
    if (!q || !features || n < 0) return fail(h, SN_ERR_INVALID, "sn_hash_encode: bad argument");
    if (which < -1 || which >= h->desc.num_proposals) return fail(h, SN_ERR_INVALID, "sn_hash_encode: bad field selector");
    const SnHashMlpDesc& d = which < 0 ? h->desc.main_field : h->desc.proposals[which];
    const DevBuf& tb = which < 0 ? h->table_main : h->table_prop[which];
    if (!tb.ptr) return fail(h, SN_ERR_STATE, "sn_hash_encode: hash table not uploaded");
    if (n == 0) return SN_OK;
    SnHashStageParams p;
    p.q = q;
    p.n = n;
    p.table = (const float*)tb.ptr;
    for (int l = 0; l < 16; ++l) p.scal[l] = l < d.num_levels ? d.scalings[l] : 0.0f;
    p.num_levels = d.num_levels;
    p.log2_t = d.log2_hashmap_size;
    p.features = features;
    p.indices = indices;
    // Proposal nets: when the weights are finalized the features come from the x-paired tables (the production layout of K2)
    const bool use_pairs = which >= 0 && h->finalized && d.grid_mode == 0;
    p.grid_mode = d.grid_mode;
    p.grid = grid_levels(d);
    p.pairs = use_pairs ? (const float*)h->pairs_prop[which].ptr : nullptr;
    if (which >= 0) p.pinfo = h->pinfo_prop[which];
    else memset(&p.pinfo, 0, sizeof(p.pinfo));
    p.pairs_bytes = which >= 0 ? (uint32_t)h->pairs_prop[which].bytes : 0u;
    p.inv_pair_scale = which >= 0 ? 1.0f / h->feat_scale_prop[which] : 1.0f;
    hipLaunchKernelGGL(sn_hash_encode_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    SN_HIP(h, hipGetLastError());
    return SN_OK;
}

int sn_field_forward(SnHandle h, int32_t which, const float* positions, const float* directions, int64_t n, int32_t precision,
                     float* density, float* rgb, SnStream stream) {
    return sn_field_forward_geo(h, which, positions, directions, n, precision, density, rgb, nullptr, stream);
}

int sn_field_forward_geo(SnHandle h, int32_t which, const float* positions, const float* directions, int64_t n, int32_t precision,
                         float* density, float* rgb, float* geo, SnStream stream) {
    if (!h) return SN_ERR_INVALID;
    if (geo && which >= 0) return fail(h, SN_ERR_INVALID, "sn_field_forward_geo: only the main field (which = -1) has geometry features");
    if (!positions || !density || n < 0) return fail(h, SN_ERR_INVALID, "sn_field_forward: bad argument");
    if (which < -1 || which >= h->desc.num_proposals) return fail(h, SN_ERR_INVALID, "sn_field_forward: bad field selector");
    if (!h->finalized) return fail(h, SN_ERR_STATE, "sn_field_forward: weights not finalized");
    if (precision < 0 || precision > 2) return fail(h, SN_ERR_INVALID, "sn_field_forward: precision must be 0, 1 or 2");
    if (precision == 2 && h->desc.main_field.grid_mode != 1)
        return fail(h, SN_ERR_INVALID, "precision 2 (single fp16) is the arithmetic of tiny-cuda-nn checkpoints: grid_mode 1 only");
    if (n == 0) return SN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (which < 0) {
        SnFieldStageParams p;
        p.pm = h->pos_map;
        p.positions = positions;
        p.directions = directions;
        p.n = n;
        p.table = (const float*)h->table_main.ptr;
        if (precision >= 1 && !h->split_ok) precision = 0;
        p.wimg = (const float*)(precision == 0 ? h->wimg_main.ptr : h->wimg_main_h.ptr);
        p.feat_scale = h->feat_scale_main;
        for (int l = 0; l < 16; ++l) p.scal[l] = h->desc.main_field.scalings[l];
        p.log2_t = h->desc.main_field.log2_hashmap_size;
        p.avg_density = h->desc.average_init_density;
        p.sh_remap = h->desc.sh_remap;
        p.density = density;
        p.rgb = rgb;
        p.geo = geo;
        p.grid_mode = h->desc.main_field.grid_mode;
        p.grid = grid_levels(h->desc.main_field);
        if (precision == 0)
            hipLaunchKernelGGL(sn_main_field_stage_kernel<0>, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)SnMainImg::TOTAL * 4, st, p);
        else if (precision == 2)
            hipLaunchKernelGGL(sn_main_field_stage_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)SnMainImgF16::TOTAL_BYTES, st, p);
        else
            hipLaunchKernelGGL(sn_main_field_stage_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)SnMainImg::TOTAL * 4, st, p);
    } else {
        SnPropStageParams p;
        p.pm = h->pos_map;
        p.positions = positions;
        p.n = n;
        p.pairs = (const float*)h->pairs_prop[which].ptr;
        p.pinfo = h->pinfo_prop[which];
        p.pairs_bytes = (uint32_t)h->pairs_prop[which].bytes;
        p.wpack = (const float*)h->wpack_prop[which].ptr;
        for (int l = 0; l < 5; ++l) p.scal[l] = h->desc.proposals[which].scalings[l];
        p.log2_t = h->desc.proposals[which].log2_hashmap_size;
        p.avg_density = h->desc.average_init_density;
        p.density = density;
        p.table = (const float*)h->table_prop[which].ptr;
        p.table_bytes = (uint32_t)h->table_prop[which].bytes;
        p.grid_mode = h->desc.proposals[which].grid_mode;
        p.grid = grid_levels(h->desc.proposals[which]);
        p.feat_scale = h->feat_scale_prop[which];
        hipLaunchKernelGGL(sn_prop_field_stage_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    }
    SN_HIP(h, hipGetLastError());
    return SN_OK;
}

int sn_composite(const float* euclid_bins, const float* density, const float* rgb_samples, int64_t n_rays, int32_t n_samples,
                 float* weights, float* rgb, float* depth, int32_t* median_index, float* accumulation, float* expected_depth,
                 SnStream stream) {
    if (!euclid_bins || !density || !rgb_samples || n_rays < 0 || n_samples < 1)
        return fail(nullptr, SN_ERR_INVALID, "sn_composite: bad argument");
    if (n_rays == 0) return SN_OK;
    hipStream_t st = (hipStream_t)stream;
    SnCompositeStageParams p;
    p.bins = euclid_bins;
    p.density = density;
    p.rgb_s = rgb_samples;
    p.n_rays = n_rays;
    p.n_samples = n_samples;
    p.weights = weights;
    p.rgb = rgb;
    p.depth = depth;
    p.median_index = median_index;
    p.acc = accumulation;
    p.exp_raw = nullptr;
    p.minmax = nullptr;
    uint32_t* mm = nullptr;
    if (expected_depth) {
        // expected_depth doubles as the raw buffer; min/max live in a small stream-ordered allocation
        if (hipMallocAsync((void**)&mm, 8, st) != hipSuccess) return fail(nullptr, SN_ERR_HIP, "sn_composite: hipMallocAsync failed");
        if (hipMemsetAsync(mm, 0xff, 4, st) != hipSuccess || hipMemsetAsync(mm + 1, 0x00, 4, st) != hipSuccess)
            return fail(nullptr, SN_ERR_HIP, "sn_composite: memset failed");
        p.exp_raw = expected_depth;
        p.minmax = mm;
    }
    hipLaunchKernelGGL(sn_composite_kernel, dim3((unsigned)((n_rays + 127) / 128)), dim3(128), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && expected_depth) {
        hipLaunchKernelGGL(sn_clip_expected_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, st, expected_depth, mm, n_rays,
                           0x7fffffff, 1, expected_depth);
        e = hipGetLastError();
        (void)hipFreeAsync(mm, st);
    }
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_composite launch: ") + hipGetErrorString(e));
    return SN_OK;
}

int sn_pdf_sample(const float* spacing_bins, const float* weights, int64_t n_rays, int32_t n_in, int32_t n_out, const float* u,
                  float histogram_padding, float* new_bins, int32_t* inds, SnStream stream) {
    if (!spacing_bins || !weights || !u || !new_bins || n_rays < 0 || n_in < 1 || n_in > SN_PROP_MAX_SAMPLES || n_out < 1 ||
        n_out > SN_PROP_MAX_SAMPLES)
        return fail(nullptr, SN_ERR_INVALID, "sn_pdf_sample: bad argument");
    if (n_rays == 0) return SN_OK;
    SnPdfStageParams p;
    p.sbins = spacing_bins;
    p.weights = weights;
    p.n_rays = n_rays;
    p.n_in = n_in;
    p.n_out = n_out;
    p.u = u;
    p.hist_pad = histogram_padding;
    p.new_bins = new_bins;
    p.inds = inds;
    hipLaunchKernelGGL(sn_pdf_stage_kernel, dim3((unsigned)((n_rays + 63) / 64)), dim3(64), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_pdf_sample launch: ") + hipGetErrorString(e));
    return SN_OK;
}


size_t sn_mask_workspace_bytes(int32_t height, int32_t width) {
    if (height <= 0 || width <= 0) return 0;
    const size_t n = (size_t)height * width;
    return align256(n) + align256((size_t)height * (width + 1) * 4) + 256;
}

int sn_aabb_mask_condition(const float* origins, const float* directions, const float* depth, int32_t height, int32_t width,
                           const float* aabb, const SnMaskOpts* opts, uint8_t* mask, float* condition, void* workspace,
                           size_t workspace_bytes, SnStream stream) {
    if (!origins || !directions || !depth || !aabb || !opts || !mask || height <= 0 || width <= 0)
        return fail(nullptr, SN_ERR_INVALID, "sn_aabb_mask_condition: bad argument");
    SnMaskOpts opts_own;
    if (int rc = adopt_struct(nullptr, opts, kMaskOptsMin, opts_own, "sn_aabb_mask_condition: SnMaskOpts")) return rc;
    opts = &opts_own;
    if (opts->dilate_w < 0 || opts->dilate_h < 0 || opts->dilate_w > SN_MASK_MAX_K || opts->dilate_h > SN_MASK_MAX_K ||
        ((opts->dilate_w == 0) != (opts->dilate_h == 0)))
        return fail(nullptr, SN_ERR_INVALID, "sn_aabb_mask_condition: dilation size must be 0 or within [1," + std::to_string(SN_MASK_MAX_K) + "] in both dimensions");
    if (!workspace || workspace_bytes < sn_mask_workspace_bytes(height, width))
        return fail(nullptr, SN_ERR_WORKSPACE, "sn_aabb_mask_condition: workspace too small");
    clear_stamp(workspace);  // (a caller may hand the mask step the memory a render used: its bins are gone then)
    hipStream_t st = (hipStream_t)stream;
    const size_t n = (size_t)height * width;
    SnMaskParams p;
    memset(&p, 0, sizeof(p));
    p.origins = origins;
    p.directions = directions;
    p.depth = depth;
    p.height = height;
    p.width = width;
    memcpy(p.aabb, aabb, sizeof(p.aabb));
    p.inverse_mask = opts->inverse_mask;
    p.dilate = opts->dilate_w > 0;
    if (p.dilate) {
        // cv2.getStructuringElement(cv2.MORPH_ELLIPSE, (w, h)): row i covers [c - dx, c + dx + 1) with
        // dx = round(c * sqrt((r*r - dy*dy) / (r*r))), r = h/2, c = w/2, dy = i - r (rows with |dy| > r are empty)
        const int kw = opts->dilate_w, kh = opts->dilate_h;
        const int r = kh / 2, c = kw / 2;
        const double inv_r2 = r ? 1.0 / ((double)r * r) : 0.0;
        p.el.kw = kw;
        p.el.kh = kh;
        p.el.ax = kw / 2;
        p.el.ay = kh / 2;
        for (int i = 0; i < kh; ++i) {
            int j1 = 0, j2 = 0;
            if (kw == 1 && kh == 1) j2 = 1;  // a 1x1 element is a rectangle
            else {
                const int dy = i - r;
                if (std::abs(dy) <= r) {
                    const int dx = (int)std::nearbyint(c * std::sqrt((r * r - dy * dy) * inv_r2));
                    j1 = std::max(c - dx, 0);
                    j2 = std::min(c + dx + 1, kw);
                }
            }
            p.el.j1[i] = (short)j1;
            p.el.j2[i] = (short)j2;
        }
    }
    p.has_manual_depth = opts->has_manual_depth;
    p.manual_min = (float)opts->manual_min;
    p.manual_range = (float)(opts->manual_max - opts->manual_min);
    p.depth_radius = opts->additional_depth_radius;
    char* ws = (char*)workspace;
    p.vis = (uint8_t*)ws;
    p.prefix = (int32_t*)(ws + align256(n));
    p.stats = (uint32_t*)(ws + align256(n) + align256((size_t)height * (width + 1) * 4));
    p.mask = mask;
    p.condition = condition;
    hipError_t e = hipMemsetAsync(p.stats, 0, 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(p.stats + 1, 0xff, 4, st);
    if (e == hipSuccess) e = hipMemsetAsync(p.stats + 2, 0, 4, st);
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_aabb_mask_condition memset: ") + hipGetErrorString(e));
    hipLaunchKernelGGL(sn_mask_visible_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, SN_MASK_VIS_BLOCKS)), dim3(256), 0, st, p);
    if (p.dilate) hipLaunchKernelGGL(sn_mask_prefix_kernel, dim3((unsigned)height), dim3(64), 0, st, p);
    hipLaunchKernelGGL(sn_mask_condition_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_aabb_mask_condition launch: ") + hipGetErrorString(e));
    return SN_OK;
}


int sn_tensor_to_uint8(const float* in, int64_t n, uint8_t* out, SnStream stream) {
    if (!in || !out || n < 0) return fail(nullptr, SN_ERR_INVALID, "sn_tensor_to_uint8: bad argument");
    if (n == 0) return SN_OK;
    hipLaunchKernelGGL(sn_tensor_to_uint8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, n, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_tensor_to_uint8 launch: ") + hipGetErrorString(e));
    return SN_OK;
}

int sn_resize_bilinear(const void* src, int32_t src_u8, int32_t src_h, int32_t src_w, int64_t src_row_stride, int32_t channels, float* dst,
                       int32_t dst_h, int32_t dst_w, int64_t dst_row_stride, int32_t threshold, SnStream stream) {
    if (!src || !dst || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || channels < 1 ||
        src_row_stride < (int64_t)src_w * channels || dst_row_stride < (int64_t)dst_w * channels)
        return fail(nullptr, SN_ERR_INVALID, "sn_resize_bilinear: bad argument");
    SnResizeParams p;
    p.src = src;
    p.dst = dst;
    p.src_u8 = src_u8 != 0;
    p.src_h = src_h;
    p.src_w = src_w;
    p.dst_h = dst_h;
    p.dst_w = dst_w;
    p.channels = channels;
    p.src_row_stride = src_row_stride;
    p.dst_row_stride = dst_row_stride;
    p.threshold = threshold != 0;
    const int64_t n = (int64_t)dst_h * dst_w * channels;
    hipLaunchKernelGGL(sn_resize_bilinear_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(nullptr, SN_ERR_HIP, std::string("sn_resize_bilinear launch: ") + hipGetErrorString(e));
    return SN_OK;
}

}  // extern "C"
