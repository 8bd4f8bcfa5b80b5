/*
 * signerf_hip.h -- C ABI of libsignerf_hip.so, the MI355X (gfx950) implementation of the
 * SIGNeRF reference-sheet render path.
 *
 * The reference is pure Python: the path is reached through two nerfstudio calls in
 *   /root/reference/signerf/datasetgenerator/datasetgenerator.py:691   camera.generate_rays(camera_indices=0, aabb_box=graph.render_aabb)
 *   /root/reference/signerf/datasetgenerator/datasetgenerator.py:694   graph.get_outputs_for_camera_ray_bundle(camera_ray_bundle)
 * followed by the in-tree slab test
 *   /root/reference/signerf/datasetgenerator/datasetgenerator.py:763   intersect_with_aabb(rays_o, rays_d, self.aabb)
 * A maintainer binds this library with ctypes from the Model / Cameras plugin objects
 * (INTEGRATION.md shows the stub).  Every entry point below names the reference interface
 * it stands in for.
 *
 * Conventions
 *   - plain C: no C++ types, no exceptions cross the boundary, no torch types.
 *   - every function returns an int status: 0 = ok, non-zero = error; the text is available
 *     from sn_last_error() (mirrors the RuntimeError raised at datasetgenerator.py:697-698).
 *   - all tensor arguments are raw DEVICE pointers (fp32 unless stated) owned by the caller
 *     (PyTorch-ROCm); the library borrows them for the duration of the call.  Weights are copied
 *     once into library-owned device buffers by sn_upload_weights().
 *   - work is enqueued on the caller's HIP stream (`stream` = hipStream_t); no hidden sync.
 *   - a handle is re-entrant: render calls keep no state in the handle; scratch memory is
 *     caller-provided (SnRenderOpts.workspace), sized by sn_workspace_bytes().  Concurrent
 *     render calls on one handle (GUI thread + viewer thread, interface.py:83-116,
 *     viewer.py:334-336) run unordered and may overlap on the GPU.  Weight uploads ARE ordered
 *     against them on the device: sn_upload_weights / sn_finalize_weights wait (stream-side)
 *     for the handle's renders in flight, later renders wait for the upload.  The host-side
 *     sequence "upload ..., finalize" itself must not be interleaved with render CALLS of other
 *     threads (the Python shim holds a lock around it).
 *   - SN_RENDER_CHAIN=1 in the environment (diagnostics) additionally makes every render of the
 *     process wait for the previous one on its device.  Off by default.
 *   - sn_last_error returns a per-thread copy of the text: valid until the calling thread's next
 *     sn_last_error call.
 *   - SIX environment switches, all off / at their defaults in normal use; none of them changes what a render means -- each turns an
 *     optimisation off so that a test can show it bit-identical to its plain form (read at sn_create / sn_finalize_weights and by
 *     sn_debug_reload_env, never by a render call):
 *       SN_EARLY_TERM=0       every sample of every ray is evaluated (default: a wave whose 64 rays all have an exactly-zero
 *                             transmittance skips the samples that can no longer change any output; tests/test_gpu_early_term.py)
 *       SN_TAIL_SPLIT=0       the last, partly filled round of the main kernel's workgroups marches whole rays (default: cut into
 *                             segment jobs + an ordered combine; tests/test_gpu_render.py)
 *       SN_PROP_CACHE_OFF=1   the proposal kernel re-fetches its cached level coefficients on every step (tests/test_gpu_render.py)
 *       SN_PDF_IEEE=1         its resampler divides with plain IEEE divisions instead of the reciprocal + exact-residual form (same file)
 *       SN_HALF_GRID=0        SnFieldDesc.half_grid is ignored: a precision-2 render rounds the rows of the uploaded table on the fly
 *                             (read by sn_finalize_weights; tests/test_gpu_fp16_mode.py)
 *       SN_RENDER_CHAIN=1     (above) renders of the process serialised on the device
 *     The measured-and-rejected variants of earlier rounds (4 waves per SIMD, colour layer 3 on the matrix cores, phase ablations,
 *     reciprocal CDF divisions, XCD tile orders ...) live in tools/patches/, each with the profile that rejected it -- not in the library.
 *   - architecture limits (sn_create returns SN_ERR_INVALID otherwise -- the kernels are written
 *     for nerfacto's shapes): main field 16 levels x 2 features, hidden 64, out 16, 2 layers,
 *     log2_hashmap_size in [4, 21]; colour head 15 geo features + SH degree 4 (+ <= 256
 *     appearance dims) -> 64 -> 64 -> 3; proposal nets 5 levels x 2 features, hidden 16, out 1.
 */
#ifndef SIGNERF_HIP_H
#define SIGNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_MAX_LEVELS 16
#define SN_MAX_PROPOSALS 2

/* ---- ABI evolution (r06) -----------------------------------------------------------------------------------------------------
 * The option structs below have grown by appended fields in every round.  Two guards keep a binding that was compiled against one
 * header safe in front of a library built from another:
 *   - sn_abi_version() returns the SN_ABI_VERSION the LIBRARY was built with; a binding compares it with the macro of its own header
 *     before anything else (the Python shim does so at load time).  It changes whenever a signature changes or a struct changes other
 *     than by appending fields.
 *   - every struct a caller fills in and that may still grow -- SnFieldDesc, SnRenderOpts, SnMaskOpts, and SnDebugLayout, which the library
 *     fills in -- begins with `uint32_t struct_size` = sizeof(that struct) IN THE CALLER'S HEADER.  The library reads (writes, for
 *     SnDebugLayout) exactly that many bytes: fields the caller's header does not know take their zero defaults (every appended field is
 *     defined so that zero means "as before"), pointers among them stay NULL -- the library never reads or writes through memory
 *     behind the caller's struct.  A struct_size below the smallest layout the library accepts (0: never set) or above the library's own
 *     sizeof (a caller newer than the library) is refused with SN_ERR_INVALID and a text that names the accepted range.  (Accepted today:
 *     from the layout without the r04 / r05 appendices -- SnFieldDesc up to `aabb`, SnRenderOpts up to `spacing_mode`, SnDebugLayout up to
 *     `feature_scale` -- to the full r06 layout.)
 *   SnHashMlpDesc (embedded), SnCameraDesc and SnDebugDump are frozen at their r06 layout; changing them bumps SN_ABI_VERSION. */
#define SN_ABI_VERSION 6
int sn_abi_version(void);

#define SN_OK 0
#define SN_ERR_INVALID 1     /* bad argument / unsupported architecture */
#define SN_ERR_HIP 2         /* a HIP runtime call failed */
#define SN_ERR_STATE 3       /* weights missing / not finalized; reuse_final_bins on a workspace that does not hold this frame's bins */
#define SN_ERR_WORKSPACE 4   /* workspace too small */

typedef struct SnContext* SnHandle;
typedef void* SnStream; /* hipStream_t */

/* One hash-grid + MLP stack (nerfstudio MLPWithHashEncoding). */
typedef struct SnHashMlpDesc {
    int32_t num_levels;         /* L */
    int32_t features_per_level; /* F, must be 2 */
    int32_t log2_hashmap_size;  /* table rows per level = 1 << this */
    int32_t hidden_dim;         /* 64 for the main field, 16 for proposal nets */
    int32_t num_layers;         /* must be 2 (one hidden layer) */
    int32_t out_dim;            /* 16 (1 + geo_feat_dim) main, 1 proposal */
    float scalings[SN_MAX_LEVELS]; /* grid_mode 0: floor(base_res * growth**l), computed by the host exactly as HashEncoding does;
                                    * grid_mode 1: exp2f(l * log2f(growth)) * base_res - 1 (fp32), tiny-cuda-nn's grid_scale */
    int32_t grid_mode;          /* 0 = nerfstudio's torch HashEncoding (every level hashed, ceil/floor corners);
                                 * 1 = tiny-cuda-nn HashGrid semantics for `implementation="tcnn"` checkpoints (SURVEY §8(f) row 2;
                                 *     x = fmaf(scale, q, 0.5), corners floor / floor + 1, dense indexing of the levels whose grid
                                 *     fits 2^log2_hashmap_size rows).  The table is [L << log2_hashmap_size, F] in both modes; a
                                 *     dense level's rows sit at the start of its slot (signerf_amd/tcnn_import.py).  UNPINNED. */
} SnHashMlpDesc;

/* Architecture of the nerfacto field + proposal nets (A0). */
typedef struct SnFieldDesc {
    uint32_t struct_size;         /* sizeof(SnFieldDesc) in the caller's header ("ABI evolution" above) */
    SnHashMlpDesc main_field;
    int32_t geo_feat_dim;         /* 15 */
    int32_t hidden_dim_color;     /* 64 */
    int32_t appearance_embed_dim; /* 32 (folded into the colour bias at finalize) */
    int32_t sh_levels;            /* 4 */
    int32_t sh_remap;             /* 0: SH evaluated on (d+1)/2 (torch fallback); 1: on d (tcnn) */
    int32_t num_proposals;        /* 0..SN_MAX_PROPOSALS */
    SnHashMlpDesc proposals[SN_MAX_PROPOSALS];
    float average_init_density;   /* signerf_config.py:35 -> 0.01 */
    float histogram_padding;      /* PDFSampler: 0.01 */
    /* Position -> grid coordinate of every field (appended in r03; zero = nerfacto's default): 0 = SceneContraction(order=inf) then
     * (p + 2) / 4; 1 = NerfactoModelConfig.disable_scene_contraction: SceneBox.get_normalized_positions, (p - aabb[0]) / (aabb[1] -
     * aabb[0]) with the model's scene box below (min xyz, max xyz).  The (0, 1) selector follows in both cases. */
    int32_t disable_scene_contraction;
    float aabb[6];
    /* Memory budget of the DERIVED gather buffers sn_finalize_weights builds beside the uploaded tables (appended in r04; zero = the
     * library defaults, so a zero-initialised descriptor behaves as before).  A viewer holding several models bounds them here instead of
     * through the environment:
     *   dense_levels        how many leading levels of the main grid get a de-hashed copy: 0 = default (11: 1.26 GB for nerfacto's grid),
     *                       -1 = none (the kernels then read the uploaded table: ~14 % slower, no extra memory); any other count -- and
     *                       whatever the per-level cap leaves of it -- is rounded DOWN to a count the main kernel is instantiated for:
     *                       11, 9 (the coefficient-form levels alone: 0.51 GB, ~2 % slower than 11) or none;
     *   dense_copy_cap_mb   per-level size cap of those copies in MB, measured on the plain-row form (R^3 x 8 bytes; the coefficient form of
     *                       the first nine levels takes 4x that): 0 = default (600); the first level above the cap ends the run of copies.
     * The proposal nets' copies (90 + 81 MB) follow dense_levels with their own 100 MB cap.  sn_debug_layout reports what a handle holds. */
    int32_t dense_levels;
    int32_t dense_copy_cap_mb;
    /*   half_grid           1 = also keep the main grid in fp16 STORAGE for SnRenderOpts.precision = 2 (tiny-cuda-nn grids only, ignored
     *                       otherwise): 16-byte quads of the de-hashed levels (2 gathers per level instead of 4) + 4-byte rows of the hashed
     *                       ones, ~1.9 GB for nerfacto's grid.  Without it a precision-2 render reads the uploaded fp32 table (same values:
     *                       every row rounded through fp16 on the fly; ~2x slower).  0 = not built (default). */
    int32_t half_grid;
} SnFieldDesc;

/* Per-call render options (NerfactoModelConfig values that shape one eval render). */
typedef struct SnRenderOpts {
    uint32_t struct_size;                          /* sizeof(SnRenderOpts) in the caller's header ("ABI evolution" above) */
    int32_t num_proposal_iterations;               /* 0 => initial sampler feeds the main field directly */
    int32_t num_proposal_samples[SN_MAX_PROPOSALS]; /* 256, 96 */
    int32_t num_nerf_samples;                      /* 48 (64 in the synthetic benchmark) */
    float near_plane;                              /* collider value used when nears == NULL (eval: 0) */
    float far_plane;                               /* collider value used when fars == NULL (1000) */
    int32_t chunk_rays;                            /* eval_num_rays_per_chunk (signerf_config.py:32); only the
                                                      expected-depth clip bounds depend on it (A17) */
    int32_t precision;                             /* 0: exact fp32 MFMA; 1: split-fp16 (hi+lo) MFMA, fp32 accumulate (fp32-grade, the default);
                                                    * 2 (r04, OPT-IN, grid_mode 1 only): single fp16 operands, fp32 accumulate, fp16 activations
                                                    *   between the layers -- the arithmetic tiny-cuda-nn checkpoints were trained in (README.md:
                                                    *   146,170: `ns-train nerfacto` runs FullyFusedMLP in fp16); applies to the main field, the
                                                    *   proposal nets and the normals kernel keep the split form.  The main grid's table values are
                                                    *   rounded through fp16 once, as the library's inference copy of the parameters is (blend in
                                                    *   fp32); SnFieldDesc.half_grid keeps them in fp16 storage.  NOT fp32-grade: ~1e-3. */
    void* workspace;                               /* device scratch, >= sn_workspace_bytes() */
    size_t workspace_bytes;
    /* Sampler grids, DEVICE pointers, computed by the host shim with the very torch ops nerfstudio uses so
     * that they are bit-identical to the reference's (torch.linspace's vectorised CPU fill has no simple
     * closed form).  NULL => the library falls back to i/N and (k+0.5)/(M+1). */
    const float* initial_spacing_bins;             /* [n0+1] = torch.linspace(0, 1, n0+1), n0 = first level's count */
    const float* pdf_u[SN_MAX_PROPOSALS];          /* pdf_u[k]: [m+1] eval-mode u grid of resampling step k (m = next level's count) */
    /* RGBRenderer's background (NerfactoModelConfig.background_color; appended in r03, zero = the nerfacto default):
     * 0 = "last_sample" (the colour of the ray's last sample), 1 = the constant colour below ("black", "white"; "random" is a training
     * device and composites like black in eval mode).  rgb = sum w c + background (1 - sum w), clamped to [0, 1]. */
    int32_t background_mode;
    float background_rgb[3];
    /* The initial sampler (NerfactoModelConfig.proposal_initial_sampler; appended in r03, zero = the nerfacto default): 0 = "piecewise",
     * UniformLinDispPiecewiseSampler (s(x) = x / 2 below 1, 1 - 1 / (2 x) above); 1 = "uniform", UniformSampler (s(x) = x).  Every
     * spacing bin of the proposal chain maps to a distance through it: t = s^-1(b s(far) + (1 - b) s(near)). */
    int32_t spacing_mode;
    /* March statistics (appended in r05; NULL = none): DEVICE pointer to 3 uint64 counters the render ADDS to (the caller zeroes them):
     * the wave-steps (one sample of each of the 64 rays of an 8x8 tile) that the exact early termination of saturated waves
     * (SN_EARLY_TERM) SKIPPED in [0] the main kernel, [1] proposal level 0, [2] proposal level 1.  A full march is tiles x samples of
     * the level wave-steps (tiles = ceil(W / 8) x ceil(H / 8); H < 8: ceil(W / 64) x H).  One atomic where a wave stops, nothing on
     * the path of a wave that does not.  Diagnostics: bench.py's `trained` leg and tests/test_gpu_trained.py report the skipped
     * fraction with it.  A render with march_stats set runs the COUNTING instantiations of the kernels (the production kernels hold no
     * atomic: one in their exit branch cost the schedule of the whole hash phase, r05) -- same outputs bit for bit, a few per cent slower,
     * and only for the default variant (torch grid, 11 + 5 + 4 de-hashed levels, precision 1, default sampler); SN_ERR_INVALID otherwise. */
    uint64_t* march_stats;
    /* sn_render_normals only (appended in r05; 0 = run the proposal sampler again): nonzero = `workspace` still holds the final sample
     * bins that the PRECEDING sn_render_rays call on this workspace left there -- same rays, frame size, sampler options and proposal
     * counts, nothing written to the workspace in between, the two calls ordered on the device (same stream, or an event) -- so the
     * normals kernel reads them and the proposal kernel is not launched (it is deterministic: the bins are the ones it would write
     * again, bit for bit; 7.5 of 17.5 ms at 1920x1080 with 256 + 96 + 48 samples).  r06: the library keeps, per workspace address, a host-side
     * STAMP of the last call that wrote bins there -- handle, the handle's weights epoch (every sn_upload_weights / sn_finalize_weights
     * advances it), frame size, sample counts, sampler options, the ray / nears / fars / grid pointers -- set by sn_render_rays, cleared
     * by every other call that is handed that workspace; sn_render_normals compares it with its own arguments and returns SN_ERR_STATE on
     * a mismatch (another frame or handle rendered into the workspace since, the weights changed, a different bundle) instead of
     * compositing normals along somebody else's bins.  What the stamp cannot see remains the caller's word: that nothing ELSE wrote to
     * that memory, that the rays behind the pointers are unchanged, and that the two calls are ordered on the device.
     * Ignored when num_proposal_iterations is 0 and by every other entry point. */
    int32_t reuse_final_bins;
} SnRenderOpts;

/* ---- lifetime -------------------------------------------------------------------------- */

/* Creates a context on the current HIP device.  Stands in for constructing the nerfacto
 * Model's field modules (signerf.py:27, NerfactoModel.populate_modules). */
int sn_create(const SnFieldDesc* desc, SnHandle* out);
int sn_destroy(SnHandle h);
/* Last error text of this handle (or of the failed sn_create when h == NULL). */
const char* sn_last_error(SnHandle h);

/* Copies one parameter tensor (host or device pointer) into the library.  `name` is the
 * nerfstudio state-dict key the pipeline filters on (signerf_pipeline.py:93-132):
 *   field.mlp_base.encoder.hash_table                       [L*T, 2]
 *   field.mlp_base.mlp.layers.{0,1}.{weight,bias}
 *   field.mlp_head.layers.{0,1,2}.{weight,bias}
 *   field.mlp_pred_normals.layers.{0,1,2}.{weight,bias}, field.field_head_pred_normals.net.{weight,bias}   (optional)
 *   field.embedding_appearance.mean                         [appearance_embed_dim] (mean over rows, A14)
 *   proposal_networks.{i}.mlp_base.encoder.hash_table       [L*T, 2]
 *   proposal_networks.{i}.mlp_base.mlp.layers.{0,1}.{weight,bias}
 * Stands in for Model.load_state_dict (signerf_pipeline.py:129-131). */
int sn_upload_weights(SnHandle h, const char* name, const void* data, size_t bytes, SnStream stream);
/* Builds the device-side weight images (MFMA operand order, folded appearance bias). */
int sn_finalize_weights(SnHandle h, SnStream stream);

/* ---- row a5: Cameras.generate_rays (datasetgenerator.py:691) ----------------------------- */
/* c2w: 12 floats (3x4 row-major) on the HOST.  Outputs are device pointers; any may be NULL.
 * origins/directions [H,W,3], pixel_area/directions_norm [H,W,1].  If aabb (6 host floats: min xyz,
 * max xyz) is non-NULL, nears/fars [H,W,1] are set from nerfstudio's clamped slab test (A1). */
int sn_generate_rays(const float* c2w, float fx, float fy, float cx, float cy, int32_t height, int32_t width,
                     float* origins, float* directions, float* pixel_area, float* directions_norm,
                     const float* aabb, float* nears, float* fars, SnStream stream);

/* The same call for the cameras the reference takes from the ORIGINAL dataset -- its default source of the generated views
 * (datasetgenerator.py:274-275 `cameras = original_dataset.cameras`, :331 the per-view loop, :356-358 the merge loop;
 * signerf_trainer.py:222 passes `datamanager.train_dataset`): nerfstudio `Cameras` built by its dataparser, with per-camera
 * intrinsics, a `camera_type` and OPENCV `distortion_params` [k1 k2 k3 k4 p1 p2].  nerfstudio un-distorts the three image-plane
 * points (pixel centre, +1 px in x, +1 px in y) with 10 fixed Newton steps of the radial-tangential model (step 0 where
 * |det J| <= 1e-3), then forms the pin-hole direction (u, v, -1) or the fisheye one (u sin t / t, v sin t / t, -cos t), t = |(u, v)|
 * clipped to [0, pi].  With has_distortion = 0 and camera_type = SN_CAMERA_PERSPECTIVE the result is bit-identical to
 * sn_generate_rays.  camera_type = SN_CAMERA_EQUIRECTANGULAR: the spherical mapping below, no un-distortion.
 *   cam     HOST struct.
 *   coords  optional DEVICE [n_coords, 2] image coordinates as (y, x) -- `generate_rays(coords=...)`; NULL = every pixel centre
 *           (row, col) + 0.5 of the height x width image in row-major order (then n_coords is ignored and the outputs are [H,W,.]).
 * Outputs as sn_generate_rays; with coords they are [n_coords, .].  Other camera types return SN_ERR_INVALID. */
#define SN_CAMERA_PERSPECTIVE 1   /* nerfstudio CameraType.PERSPECTIVE.value */
#define SN_CAMERA_FISHEYE 2       /* nerfstudio CameraType.FISHEYE.value */
#define SN_CAMERA_EQUIRECTANGULAR 3 /* nerfstudio CameraType.EQUIRECTANGULAR.value: the viewer's preview camera (signerf/interface/viewer.py:307-319);
                                     * theta = -pi u, phi = pi (1/2 - v), d = (-sin theta sin phi, cos phi, -cos theta sin phi); never un-distorted */
typedef struct SnCameraDesc {
    float c2w[12];           /* 3x4 row-major camera-to-world */
    float fx, fy, cx, cy;
    int32_t height, width;
    int32_t camera_type;     /* SN_CAMERA_* */
    int32_t has_distortion;  /* 0: skip the un-distortion (distortion_params None / all zero / disable_distortion=True) */
    float distortion[6];     /* k1 k2 k3 k4 p1 p2 */
} SnCameraDesc;
int sn_generate_rays_camera(const SnCameraDesc* cam, const float* coords, int64_t n_coords,
                            float* origins, float* directions, float* pixel_area, float* directions_norm,
                            const float* aabb, float* nears, float* fars, SnStream stream);

/* ---- row a4: intersect_with_aabb (signerf/utils/intersection.py:5-56) -------------------- */
/* aabb: 6 host floats (min xyz, max xyz).  nears/fars: [n_rays]. */
int sn_intersect_with_aabb(const float* origins, const float* directions, int64_t n_rays, const float* aabb,
                           float* nears, float* fars, SnStream stream);

/* ---- viewer crop: nerfstudio's intersect_obb, reached from Model.get_outputs_for_camera(camera, obb_box)
 * (signerf/interface/viewer.py:334-336 runs nerfstudio's render thread on the shared model) ----------------------------- */
/* world2box: 12 host floats, the 3x4 row-major inverse of the box pose [R | T]; size: 3 host floats (S).  nears/fars: [n_rays],
 * clamped to [0, 1e10]; a ray that misses gets 1e10 for both. */
int sn_intersect_obb(const float* origins, const float* directions, int64_t n_rays, const float* world2box, const float* size,
                     float* nears, float* fars, SnStream stream);

/* ---- rows a6-a17: Model.get_outputs_for_camera_ray_bundle (datasetgenerator.py:694) ------ */
size_t sn_workspace_bytes(SnHandle h, int32_t height, int32_t width, const SnRenderOpts* opts);
/* origins/directions: [H,W,3]; nears/fars: [H,W,1] or NULL (collider).  Outputs (any may be NULL):
 * rgb [H,W,3], depth [H,W,1] (median), accumulation [H,W,1], expected_depth [H,W,1],
 * prop_depth_i [H,W,1].  Row-major ray order, identical to the reference's chunk loop.
 * Concurrency: calls may come from several host threads and HIP streams, on one handle or several; they are not ordered against
 * each other (see "Conventions"; DESIGN.md §5 has the history of the r01 hazard that once made a chain necessary). */
int sn_render_rays(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars,
                   int32_t height, int32_t width, const SnRenderOpts* opts,
                   float* rgb, float* depth, float* accumulation, float* expected_depth,
                   float* prop_depth_0, float* prop_depth_1, SnStream stream);

/* ---- row a16 / §8(f) row 4: the `predict_normals=True` outputs (signerf_config.py:33) --------------------------- */
/* "normals" (analytic: minus the normalised gradient of the pre-activation density w.r.t. the field's normalised sample
 * location, weight-composited, renormalised, mapped to [0,1]) and "pred_normals" (the field's pred-normal MLP, same
 * rendering), [H,W,3] each, either may be NULL.  A separate launch because DatasetGenerator.render_camera never reads them
 * (datasetgenerator.py:700-701); same rays, options and workspace as sn_render_rays (the proposal sampler is re-run).
 * pred_normals needs field.mlp_pred_normals.layers.{0,1,2}.{weight,bias} and field.field_head_pred_normals.net.{weight,bias}
 * uploaded before sn_finalize_weights, else SN_ERR_STATE. */
int sn_render_normals(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars,
                      int32_t height, int32_t width, const SnRenderOpts* opts, float* normals, float* pred_normals,
                      SnStream stream);

/* SnRenderOpts.precision = 1 (split fp16) is a REQUEST.  sn_finalize_weights conditions every layer of the split-precision MLPs into
 * fp16's range with exact power-of-two scales derived from the uploaded parameters (bounds of the activations over all inputs with
 * |feature| <= max|table row|), which keeps the arithmetic fp32-grade for tables and weights of any magnitude; if a conditioned weight
 * still leaves the fp16 range the handle renders such requests with the exact-fp32 MFMA path instead.  Returns the precision that
 * `requested` resolves to (0 or 1) for kernel 0 = sn_render_rays / sn_field_forward, 1 = sn_render_normals (conditioned the same way
 * since r03: the density MLP's layers as in kernel 0, the pred-normal MLP and the transposed layer of the reverse pass with their own
 * power-of-two scales -- tables initialised at 1e-3 / 1e-4, as real checkpoints are, stay on the split-precision path); -1 on error. */
int sn_effective_precision(SnHandle h, int32_t requested, int32_t kernel);

/* ---- stage-level entry points (used by parity tests).  They run the LITERAL torch-path arithmetic (IEEE divisions in the
 * contraction, floor / ceil corners, the reference's blend association, expf); the fused kernels behind sn_render_rays run the
 * reduced-instruction forms of the same maps (v_rcp_f32 contraction, truncation + fract corners with "ceil = floor + 1", lerp-form
 * blend, v_exp_f32) and read the coarse levels from de-hashed copies -- what THEY fetch is checked through sn_render_rays_debug. */
/* which: -1 main field, i >= 0 proposal net i.  q: [n,3] normalised positions in [0,1).
 * features: [n, L*F] level-major; indices (nullable): [n, L, 8] int32 table rows incl. level offset,
 * corner order as nerfstudio's hashed_0..7 (row a13). */
int sn_hash_encode(SnHandle h, int32_t which, const float* q, int64_t n, float* features, int32_t* indices, SnStream stream);
/* World positions [n,3] (+ directions [n,3] for the main field) -> density [n] and, main field only,
 * rgb [n,3] (rows a9, a14, a15).  rgb/directions may be NULL. */
int sn_field_forward(SnHandle h, int32_t which, const float* positions, const float* directions, int64_t n,
                     int32_t precision, float* density, float* rgb, SnStream stream);
/* The same with the main field's 15 geometry features -- the second value of nerfstudio's `NerfactoField.get_density(ray_samples)`
 * (`base_mlp_out`, which `Field.forward` hands to `get_outputs` as `density_embedding`): geo [n,15], nullable; which must be -1 when
 * it is given.  Stands behind the Field objects' get_density / density_fn / get_outputs / forward (signerf/signerf.py:27 inherits them
 * from NerfactoModel; nerfstudio's samplers and export tools call them). */
int sn_field_forward_geo(SnHandle h, int32_t which, const float* positions, const float* directions, int64_t n,
                         int32_t precision, float* density, float* rgb, float* geo, SnStream stream);
/* Rows a10 + a17 on explicit per-sample inputs: euclid_bins [R,S+1], density [R,S], rgb_samples [R,S,3] ->
 * weights [R,S], rgb [R,3], depth [R], median_index [R] (int32), accumulation [R], expected_depth [R]
 * (clipped to the global [min,max] of the sample mid-points, i.e. one chunk). Outputs may be NULL. */
int sn_composite(const float* euclid_bins, const float* density, const float* rgb_samples, int64_t n_rays, int32_t n_samples,
                 float* weights, float* rgb, float* depth, int32_t* median_index, float* accumulation,
                 float* expected_depth, SnStream stream);
/* Row a11: spacing_bins [R,N+1], weights [R,N] -> new spacing bins [R,M+1] and searchsorted indices [R,M+1] (int32).
 * u: [M+1] device floats (the eval-mode grid). */
int sn_pdf_sample(const float* spacing_bins, const float* weights, int64_t n_rays, int32_t n_in, int32_t n_out,
                  const float* u, float histogram_padding, float* new_bins, int32_t* inds, SnStream stream);

/* ---- test instrumentation of the FUSED kernels (SURVEY.md §8(d) parity gate: "bit-exact: hash corner coords & table indices,
 * PDF searchsorted indices and median-depth index ... count and report ties") -----------------------------------------------
 * sn_render_rays_debug is sn_render_rays with the production kernels instantiated with their DUMP flag: every ray-sample records
 * what it fetched, from the registers that feed the loads.  It exists for the default kernel variants only (torch grid,
 * 11 de-hashed main levels; proposal nets with 5 + 4 de-hashed levels) and returns SN_ERR_INVALID otherwise.  Any pointer may be NULL.
 * A fetch record is 8 uint32 words per (ray, sample, level):
 *   hashed level read from the plain table   words 0..7 = byte offset (row * 8) of the 8 corner rows within the level,
 *                                            nerfstudio corner order 0 ccc, 1 cfc, 2 ffc, 3 fcc, 4 ccf, 5 cff, 6 fff, 7 fcf
 *                                            (x y z; the kernels take the "c" corner as floor + 1: it differs from ceil only where the
 *                                            coordinate is an integer, i.e. where that corner's blend weight is exactly 0);
 *   level read from its de-hashed copy,      words 0..3 = byte offset, within the buffer of copies, of the four 16-byte fetches
 *     plain-row form                         (y1 z1), (y0 z1), (y0 z0), (y1 z0), each holding the x0 and x0 + 1 rows; word 4 = 0xD0000000;
 *     bilinear-coefficient form              words 0, 1 = byte offset of the 32-byte entries of grid points (x0, y0, z0) and (x0, y0, z0 + 1);
 *                                            word 4 = 0xB0000000;
 *   level read from the x-paired tables      words 0..3 = 16-byte entry number of the four fetches (same order), each holding row r
 *                                            and row r ^ (2^(t+1) - 1); word 4 = 0xA0000000 | t.
 * sn_debug_layout / sn_debug_read expose the layout and the contents of those derived buffers so that a test can map every record
 * back to rows of the uploaded hash table and check the copies against it. */
typedef struct SnDebugDump {
    uint32_t* main_fetch;                    /* [H*W][num_nerf_samples][16][8] */
    float* main_q;                           /* [H*W][num_nerf_samples][3] the normalised positions the main kernel hashed */
    int32_t* median_index;                   /* [H*W] index of the median-depth sample (row a17) */
    uint32_t* prop_fetch[SN_MAX_PROPOSALS];  /* [H*W][num_proposal_samples[k]][5][8] */
    float* prop_q[SN_MAX_PROPOSALS];         /* [H*W][num_proposal_samples[k]][3] the normalised positions proposal net k hashed */
    int32_t* pdf_index[SN_MAX_PROPOSALS];    /* [H*W][m_k + 1]: PDFSampler's searchsorted(cdf, u, right) of resampling step k */
} SnDebugDump;
int sn_render_rays_debug(SnHandle h, const float* origins, const float* directions, const float* nears, const float* fars,
                         int32_t height, int32_t width, const SnRenderOpts* opts,
                         float* rgb, float* depth, float* accumulation, float* expected_depth,
                         float* prop_depth_0, float* prop_depth_1, const SnDebugDump* dump, SnStream stream);
typedef struct SnDebugLayout {
    uint32_t struct_size;              /* IN: sizeof(SnDebugLayout) in the caller's header -- the library writes no more than that ("ABI evolution") */
    int32_t n_dense;                   /* leading levels that are read from de-hashed copies */
    int32_t n_bc;                      /* of those, the leading levels stored in bilinear-coefficient form: 32 bytes per grid point (x, y, z) =
                                        * {A, B | C, D} x 2 features, A = v(x,y,z), B = v(x+1,y,z) - A, C = v(x,y+1,z) - A,
                                        * D = (v(x+1,y+1,z) - v(x+1,y,z)) - C; the other copied levels hold plain rows (8 bytes per grid point) */
    uint32_t dense_res[12];            /* R of level l: grid point (x, y, z) sits at entry x + R y + R^2 z */
    uint32_t dense_off[12];            /* byte offset of level l's copy within the buffer */
    uint64_t dense_bytes;              /* size of the buffer of copies */
    uint32_t pair_base[SN_MAX_LEVELS]; /* which >= 0: first 16-byte entry of level l's t = 0 paired table (table t follows at t << log2_T) */
    uint64_t pair_bytes;
    float feature_scale;               /* the copies / paired tables hold table rows TIMES this power of two (range conditioning of the
                                        * split-precision MLPs; the first layer's weights carry its inverse) */
    uint64_t table_bytes;              /* (r04) the uploaded table of field `which` */
    uint64_t handle_bytes;             /* (r04) every device buffer the HANDLE owns: tables, copies, paired tables, weight images */
    uint64_t half_grid_bytes;          /* (r04) the main grid's fp16 storage (SnFieldDesc.half_grid), 0 when not built */
} SnDebugLayout;
/* which: -1 main field, i >= 0 proposal net i. */
int sn_debug_layout(SnHandle h, int32_t which, SnDebugLayout* out);
/* what: 0 = the buffer of de-hashed copies, 1 = the x-paired tables (proposal nets).  dst: device pointer, bytes must match. */
int sn_debug_read(SnHandle h, int32_t which, int32_t what, void* dst, size_t bytes, SnStream stream);
/* The switches of the environment ("Conventions") are read by sn_create and sn_finalize_weights, never by a render call; a test that flips one between two renders
 * of the same handle calls this to have it re-read. */
int sn_debug_reload_env(SnHandle h);

/* (r06) The position maps of the kernels on explicit samples: origins / directions [n,3], starts / ends [n] (device) -> the normalised,
 * selector-multiplied positions q [n,3] as computed by (any pointer may be NULL)
 *   q_strict  the literal torch-path arithmetic (Frustums.get_positions, SceneContraction(inf) with its four IEEE divisions, (p + 2) / 4):
 *             what the stage kernels, the normals kernel and the non-default ("ALT") render instantiations use;
 *   q_exact   the division-free form of the SAME map (v_rcp_f32 + Newton steps + exact residuals; csrc/sn_device.h sn_sample_q_exact)
 *             that the main kernel uses behind the uniform sampler -- must equal q_strict bit for bit (tests/test_gpu_stages.py);
 *   q_fast    the reduced form (FMA positions, one v_rcp_f32; <= 2 ulp) the fused kernels use behind the proposal sampler. */
int sn_debug_sample_positions(const float* origins, const float* directions, const float* starts, const float* ends, int64_t n,
                              float* q_strict, float* q_exact, float* q_fast, SnStream stream);

/* ---- measurement aid (bench.py's issue roofs need the clock the chip actually sustains under the render) ------------------
 * Enqueues eight ONE-WAVE workgroups (one per XCD: the dies of one part run at different clocks under load) that idle for `seconds` (<= 1)
 * of the device's constant-rate wall clock and report how many shader cycles passed meanwhile: out[0] = shader cycles (s_memtime) and
 * out[1] = wall-clock ticks (s_memrealtime), both SUMMED over the eight, out[2] = wall-clock rate in Hz -- out[0] / out[1] * out[2] is the
 * mean shader clock.  out: 3 x uint64 in DEVICE memory (zeroed by the call).  Launch it on a side stream just before the renders to be clocked. */
int sn_clock_probe(uint64_t* out, double seconds, SnStream stream);

/* ---- the mask + condition step after the render, "aabb" masking mode ------------------
 * (signerf/datasetgenerator/datasetgenerator.py:758-818).  Stays on the device: no cv2 round trip (:776-778), no host sync on
 * `torch.sum(visible_mask) > 1e-6` (:770). */
typedef struct SnMaskOpts {
    uint32_t struct_size;            /* sizeof(SnMaskOpts) in the caller's header ("ABI evolution") */
    int32_t inverse_mask;            /* DatasetGeneratorConfig.inverse_mask */
    int32_t dilate_w, dilate_h;      /* mask_dialation, cv2.MORPH_ELLIPSE size; 0 = no dilation; each <= 256 */
    int32_t has_manual_depth;        /* manual_depth is not None */
    double manual_min, manual_max;   /* Python numbers in the reference: (max - min) is formed in double, then cast to fp32 */
    float additional_depth_radius;   /* 0.1 */
} SnMaskOpts;
size_t sn_mask_workspace_bytes(int32_t height, int32_t width);
/* origins/directions [H,W,3], depth [H,W,1] (device); aabb: 6 host floats (min xyz, max xyz).
 * Outputs (device): mask [H,W,1] uint8 (0/1), condition [H,W,1] fp32 (may be NULL: with_condition=False). */
int sn_aabb_mask_condition(const float* origins, const float* directions, const float* depth, int32_t height, int32_t width,
                           const float* aabb, const SnMaskOpts* opts, uint8_t* mask, float* condition, void* workspace,
                           size_t workspace_bytes, SnStream stream);

/* ---- tensor_to_image's numeric part (signerf/utils/image_tensor_converter.py:7-33) -------
 * out[i] = (uint8)(in[i] * 255): a TRUNCATING cast, no rounding, no clamp (numpy astype semantics on x86: float -> int32 ->
 * low 8 bits; NaN -> 0).  in: [n] fp32 device, out: [n] uint8 device. */
int sn_tensor_to_uint8(const float* in, int64_t n, uint8_t* out, SnStream stream);

/* ---- SURVEY §8(f) row 1, second half: the reference-sheet composition ------------------------------------------
 * torch.nn.functional.interpolate(mode="bilinear", align_corners=False) from a [src_h, src_w, C] window to a [dst_h, dst_w, C]
 * window of channel-last device images; windows are given by their first element and their ROW STRIDE in elements, so the same
 * call does "downscale a view and paste it into its cell of the sheet" (signerf/datasetgenerator/datasetgenerator.py:526-539,
 * 634-647) and "cut the edited cell out of the sheet and upscale it" (:577-586, 659).  src_u8 != 0: the source holds uint8
 * 0/1 (the mask of sn_aabb_mask_condition; the reference's mask.float()).  threshold != 0: store (value > 0.5) as 1.0/0.0
 * (mask_scaled, :527).  dst is fp32. */
int sn_resize_bilinear(const void* src, int32_t src_u8, int32_t src_h, int32_t src_w, int64_t src_row_stride, int32_t channels,
                       float* dst, int32_t dst_h, int32_t dst_w, int64_t dst_row_stride, int32_t threshold, SnStream stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGNERF_HIP_H */
