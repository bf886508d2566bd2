/*
 * ucsa_hip.h -- C ABI of libucsa_hip.so, the MI355X (gfx950) implementation of
 * the Semantic-NeRF volume-rendering hot path of ethz-asl/ucsa_neural_rendering.
 *
 * This is the drop-in boundary.  The reference's only native
 * FFI on this path is the pybind11 module `_raymarching`
 * (reference nr4seg/nerf/raymarching/src/bindings.cpp:5-17, signatures
 * raymarching.h:7-18) plus the tiny-cuda-nn Python objects constructed in
 * nr4seg/nerf/network_tcnn_semantics.py:36-100.  Every entry point below names
 * the reference interface it replaces.
 *
 * Conventions (all entry points):
 *   - plain C, no torch / C++ types; pointers are DEVICE pointers unless a
 *     parameter is documented as host;
 *   - the caller owns every buffer; nothing is allocated, nothing syncs;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*);
 *   - return 0 on success, a negative hipError_t on a launch error, or
 *     UCSA_ERR_ARG (-1000 - k) when argument k (0-based) is invalid;
 *   - buffers are contiguous, fp32 unless stated, 16-byte aligned;
 *   - re-entrant.  Process-wide state is limited to: (a) the table of tuning
 *     switches read from the environment ONCE, at the first call that consults
 *     it (INTEGRATION.md "Environment variables": launch shapes and kernel
 *     choices, never results); (b) per caller-stream sets of internal streams /
 *     events created on first use by the calls documented as pipelined
 *     (ucsa_render_view, the merged grid backward).
 */
#ifndef UCSA_HIP_H_
#define UCSA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCSA_VERSION 100 /* major*10000 + minor*100 + patch */
#define UCSA_ERR_ARG (-1000)
#define UCSA_MAX_LEVELS 16

int32_t ucsa_version(void);
/* Human-readable text for a return code (static storage). */
const char* ucsa_error_string(int32_t code);
/* Re-read the tuning switches from the environment (lab tools and tests that
 * flip one inside a process; not for production code, not thread-safe against
 * concurrent calls into the library). */
void ucsa_env_reload(void);

/* ---- hash-grid level table -------------------------------------------------
 * Host-side description of tcnn.Encoding("HashGrid") as configured at
 * reference network_tcnn_semantics.py:34-46.  Filled on the HOST. */
typedef struct ucsa_grid_level {
  float scale;      /* 2^(l*log2 s)*base - 1, float64-evaluated then cast */
  uint32_t res;     /* ceil(scale)+1 */
  uint32_t entries; /* min(roundup8(res^3), 2^log2_hashmap_size) */
  uint32_t offset;  /* first entry of this level in the flat table */
  uint32_t hashed;  /* 1: spatial hash, 0: dense x + y*res + z*res^2 */
} ucsa_grid_level;

typedef struct ucsa_grid {
  uint32_t n_levels;   /* <= UCSA_MAX_LEVELS */
  uint32_t n_features; /* must be 2 */
  uint32_t total_entries;
  float bound; /* positions are mapped (x+bound)/(2*bound) */
  ucsa_grid_level level[UCSA_MAX_LEVELS];
} ucsa_grid;

/* HOST function.  Replaces the tcnn HashGrid constructor
 * (reference network_tcnn_semantics.py:36-46). */
int32_t ucsa_grid_init(ucsa_grid* grid, float bound, uint32_t n_levels,
                       uint32_t log2_hashmap_size, uint32_t base_resolution,
                       double per_level_scale);

/* ---- rays ------------------------------------------------------------------
 * Full-image pinhole rays, row-major pixels, pixel centres at +0.5.
 * Replaces get_rays (reference nr4seg/dataset/ngp_utils.py:28-69) and, with
 * `inds` != NULL, get_rays_train (reference
 * nr4seg/lightning/joint_train_lightning_net.py:108-157).
 *   poses [B,4,4]; intrinsics are HOST scalars (fx,fy,cx,cy);
 *   inds [n] int64 or NULL (then n must be H*W and pixel i is ray i);
 *   out: rays_o, rays_d [B,n,3], norms [B,n]. */
int32_t ucsa_get_rays(const float* poses, uint32_t B, float fx, float fy,
                      float cx, float cy, uint32_t H, uint32_t W,
                      const int64_t* inds, uint32_t n, float* rays_o,
                      float* rays_d, float* norms, void* stream);

/* The pixel indices drawn by get_rays_train (reference
 * joint_train_lightning_net.py:141; int64 [n], duplicates allowed, values in
 * [0, H*W)), reordered tile by tile (tile x tile pixels, row-major inside a
 * tile): the same multiset, neighbours adjacent, which makes the hash-grid
 * backward of the batch about twice as fast.  n <= 8192 (one
 * workgroup sorts in LDS); out may not alias inds. */
int32_t ucsa_tile_order(const int64_t* inds, uint32_t n, uint32_t H, uint32_t W,
                        uint32_t tile, int64_t* out, void* stream);

/* Replaces _backend.near_far_from_aabb (reference
 * nr4seg/nerf/raymarching/src/raymarching.cu:62-126, bindings.cpp:6).
 * aabb is 6 HOST floats (xmin,ymin,zmin,xmax,ymax,zmax). */
int32_t ucsa_near_far_from_aabb(const float* rays_o, const float* rays_d,
                                const float* aabb_host, uint32_t N,
                                float min_near, float* nears, float* fars,
                                void* stream);

/* Coarse sample depths, reference renderer_semantics.py:154-168.
 * t_rand [N,T] or NULL (perturb=False).  out z [N,T]. */
int32_t ucsa_sample_coarse(const float* nears, const float* fars,
                           const float* t_rand, uint32_t N, uint32_t T,
                           float* z, void* stream);

/* ---- field: hash grid + sigma MLP -----------------------------------------
 * Encodes the positions clamp(o + d*z, aabb) of all N*T samples.
 * Replaces tcnn.Encoding.forward as called from density()
 * (reference network_tcnn_semantics.py:133-134) together with the position
 * generation/clip of renderer_semantics.py:171-173.
 *   table [total_entries,2] fp32; out feat [n_levels][N*T][2] (level-major).
 */
int32_t ucsa_hashgrid_encode_rays(const ucsa_grid* grid_host,
                                  const float* table, const float* rays_o,
                                  const float* rays_d, const float* z,
                                  const float* aabb_host, uint32_t N,
                                  uint32_t T, float* feat, void* stream);

/* Same for IMAGE-ORDERED rays: ray r is pixel (r / image_width,
 * r % image_width) of full rows of an image (any number of rows).  The fine
 * levels are gathered one 8x8 pixel tile per wave so that neighbouring pixels
 * share cache lines; the features are bit-identical to
 * ucsa_hashgrid_encode_rays'. */
int32_t ucsa_hashgrid_encode_rays_image(
    const ucsa_grid* grid_host, const float* table, const float* rays_o,
    const float* rays_d, const float* z, const float* aabb_host, uint32_t N,
    uint32_t T, uint32_t image_width, float* feat, void* stream);

/* Same encoder for explicit positions x [M,3] in [-bound,bound] (the
 * reference's network.density(x) entry, network_tcnn_semantics.py:130-135). */
int32_t ucsa_hashgrid_encode_points(const ucsa_grid* grid_host,
                                    const float* table, const float* x,
                                    uint32_t M, float* feat, void* stream);

/* MLP kinds: which of the three tcnn.Network objects
 * (reference network_tcnn_semantics.py:48-58, 74-84, 90-100). */
#define UCSA_MLP_SIGMA 0 /* 32 -> 64 -> 16        (3072 params) */
#define UCSA_MLP_COLOR 1 /* 32 -> 64 -> 64 -> 16  (7168 params) */
#define UCSA_MLP_SEM 2   /* 16 -> 64 -> 48        (4096 params) */

/* Re-lays a tcnn-format flat weight vector (row-major [out,in] matrices back
 * to back, padded shapes) into MFMA A-fragment order.  Must be called after
 * every parameter update, before the kernels below.
 *   packed: same element count as params. */
int32_t ucsa_mlp_pack(int32_t kind, const float* params, float* packed,
                      uint32_t n_classes, void* stream);

/* sigma MLP over level-major features: h [M,16] raw outputs
 * (h[:,0] = log-density, h[:,1:16] = geo_feat), sigma [M] = exp(h[:,0]).
 * Replaces sigma_net + trunc_exp in density()
 * (reference network_tcnn_semantics.py:135-139, activation.py:13). */
int32_t ucsa_sigma_mlp_fwd(const float* feat, const float* packed_sigma,
                           uint32_t M, uint32_t n_levels, float* h,
                           float* sigma, void* stream);


/* ---- hierarchical resampling ----------------------------------------------
 * Coarse weights -> pdf over interior bins -> inverse CDF at u.
 * Replaces renderer_semantics.py:182-207 and sample_pdf :10-46.
 *   z [N,T], sigma [N,T], u [N,t] in [0,1); out new_z [N,t], ASCENDING per
 *   ray (the uniforms are sorted first; the reference only ever consumes
 *   new_z through its sort/merge, :221-222, so the order is unobservable). */
int32_t ucsa_resample(const float* z, const float* sigma, const float* u,
                      uint32_t N, uint32_t T, uint32_t t,
                      float density_scale, float* new_z, void* stream);

/* ---- merge + weights + masked colour/semantics + compositing ---------------
 * Replaces renderer_semantics.py:220-299 and the masked color()/semantics()
 * of network_tcnn_semantics.py:147-207.
 *   z_c/sigma_c/h_c: coarse [N,T], [N,T], [N*T,16]
 *   z_f/sigma_f/h_f: fine   [N,t], [N,t], [N*t,16]   (t may be 0)
 *   out: image [N,3], depth [N], semantics [N,n_classes].
 * Optional training outputs (NULL to skip): merged order `src` [N,S] int32
 * (index into the concatenated [coarse|fine] sample list of the ray),
 * `weights` [N,S] (unmasked weights), S = T+t. */
int32_t ucsa_composite_fwd(const float* rays_d, const float* norms,
                           const float* z_c, const float* sigma_c,
                           const float* h_c, const float* z_f,
                           const float* sigma_f, const float* h_f,
                           const float* packed_color, const float* packed_sem,
                           uint32_t N, uint32_t T, uint32_t t,
                           uint32_t n_classes, float density_scale,
                           float* image, float* depth, float* semantics,
                           int32_t* src, float* weights, void* stream);

/* Scratch bytes the caller must provide for ucsa_render_fwd. */
uint64_t ucsa_render_workspace_bytes(uint32_t N, uint32_t T, uint32_t t,
                                     uint32_t n_levels);

/* Whole forward pass for N rays (the reference's SemanticNeRFRenderer.run,
 * renderer_semantics.py:123-299): near/far, coarse sampling, density,
 * resampling, density, merge, masked colour/semantics, compositing.
 *   t_rand [N,T] or NULL; u [N,t]; ws: workspace of
 *   ucsa_render_workspace_bytes(); packed_*: see ucsa_mlp_pack.
 *   image_width: 0, or the width of the image whose full rows the rays are
 *   (ray r = pixel (r / image_width, r % image_width)): selects the
 *   tile-ordered gather of ucsa_hashgrid_encode_rays_image, same results. */
int32_t ucsa_render_fwd(const ucsa_grid* grid_host, const float* table,
                        const float* packed_sigma, const float* packed_color,
                        const float* packed_sem, const float* rays_o,
                        const float* rays_d, const float* norms,
                        const float* aabb_host, float min_near,
                        const float* t_rand, const float* u, uint32_t N,
                        uint32_t T, uint32_t t, uint32_t n_classes,
                        float density_scale, uint32_t image_width, float* image,
                        float* depth, float* semantics, void* ws, void* stream);

/* The same computation for INFERENCE as two dense kernels (no `src` /
 * `weights` outputs): k_weights_compact (one wave per ray: merge, weights,
 * mask, depth, survivors compacted into a global list) and k_shade_dense /
 * k_shade16 (colour + semantics nets over the survivor list with a
 * software-pipelined operand fetch, per-ray sums in sample order).  Bit-identical outputs to
 * ucsa_composite_fwd; what ucsa_render_fwd uses.  `workspace`:
 * ucsa_composite_infer_workspace_bytes(N, T, t) bytes, caller-owned. */
uint64_t ucsa_composite_infer_workspace_bytes(uint32_t N, uint32_t T, uint32_t t);
int32_t ucsa_composite_infer(const float* rays_d, const float* norms,
                             const float* z_c, const float* sigma_c,
                             const float* h_c, const float* z_f,
                             const float* sigma_f, const float* h_f,
                             const float* packed_color, const float* packed_sem,
                             uint32_t N, uint32_t T, uint32_t t,
                             uint32_t n_classes, float density_scale,
                             float* image, float* depth, float* semantics,
                             void* workspace, void* stream);
/* fp16-MFMA form (weights from ucsa_mlp_pack_f16) */
int32_t ucsa_composite_infer_f16(const float* rays_d, const float* norms,
                                 const float* z_c, const float* sigma_c,
                                 const float* h_c, const float* z_f,
                                 const float* sigma_f, const float* h_f,
                                 const void* packed_color_half,
                                 const void* packed_sem_half, uint32_t N,
                                 uint32_t T, uint32_t t, uint32_t n_classes,
                                 float density_scale, float* image,
                                 float* depth, float* semantics,
                                 void* workspace, void* stream);

/* Pointwise colour / semantics queries: network.color() / network.semantics()
 * (reference network_tcnn_semantics.py:147-207).  dirs [M,3], geo_feat [M,15],
 * mask [M] uint8 or NULL; rgb [M,3] and/or probs [M,n_classes] (either may be
 * NULL); rows with mask == 0 are zero. */
int32_t ucsa_point_shade(const float* dirs, const float* geo_feat,
                         const uint8_t* mask, const float* packed_color,
                         const float* packed_sem, uint32_t M,
                         uint32_t n_classes, float* rgb, float* probs,
                         void* stream);

/* Same, reading the geometry features from the raw sigma-MLP rows h [M,16]
 * (ucsa_sigma_mlp_fwd; slot 0, the log-density, is skipped) so that
 * network.forward(x, d) (reference network_tcnn_semantics.py:102-128) needs
 * no repacking between density() and color()/semantics(). */
int32_t ucsa_point_shade_h(const float* dirs, const float* h,
                           const uint8_t* mask, const float* packed_color,
                           const float* packed_sem, uint32_t M,
                           uint32_t n_classes, float* rgb, float* probs,
                           void* stream);

/* ============================ fp16 hash table ============================== */
/* tiny-cuda-nn stores the hash grid in fp16 (fp32 master copy in the optimizer)
 * and its encoding emits fp16 features.  Optional here: `table_half` = the
 * fp32 table rounded to half (ucsa_cast_f32_to_f16; half2 entries, same level
 * offsets); `feat_half` [L][N*T] half2.  Values are widened on load and
 * interpolated in fp32, so the features equal the fp32 kernels' on the
 * rounded table, rounded to half once (where ucsa_sigma_mlp_fwd_f16 would
 * round them anyway); 4-byte entries let one 16-byte access serve four
 * x-neighbours (5 instead of 6 accesses per sample and hashed level).
 * image_width = 0: ray-ordered samples, else as
 * ucsa_hashgrid_encode_rays_image.  Call site: reference
 * network_tcnn_semantics.py:36-46,133-134. */
int32_t ucsa_cast_f32_to_f16(const float* src, void* dst_half, uint64_t n,
                             void* stream);
int32_t ucsa_hashgrid_encode_rays_h16(const ucsa_grid* grid,
                                      const void* table_half,
                                      const float* rays_o, const float* rays_d,
                                      const float* z, const float* aabb_host,
                                      uint32_t N, uint32_t T,
                                      uint32_t image_width, void* feat_half,
                                      void* stream);
/* fp32 table, fp16 features (for ucsa_sigma_mlp_fwd_f16_h: same h / sigma as
 * ucsa_hashgrid_encode_rays[_image] + ucsa_sigma_mlp_fwd_f16, half the
 * feature traffic) */
int32_t ucsa_hashgrid_encode_rays_hf(const ucsa_grid* grid, const float* table,
                                     const float* rays_o, const float* rays_d,
                                     const float* z, const float* aabb_host,
                                     uint32_t N, uint32_t T,
                                     uint32_t image_width, void* feat_half,
                                     void* stream);
/* ---- depth-ordered encoding of the FINE samples of image-ordered rays ----
 * Replaces, for rays that are the pixels of whole image rows, the
 * `self.encoder(x)` + `self.sigma_net(x)` of the second `density()` call of
 * run() (nr4seg/nerf/renderer_semantics.py:214-226 ->
 * network_tcnn_semantics.py:130-144).  The fine samples of neighbouring rays sit
 * at unrelated depths (importance sampling with independent uniforms), so a wave
 * that holds one sample index of an 8x8 pixel tile shares no cells; dealt to the
 * lanes in DEPTH order the same samples share their cache lines as in the coarse
 * pass.  Three calls, bit-identical h / sigma to ucsa_hashgrid_encode_rays_image
 * + ucsa_sigma_mlp_fwd*:
 *   ucsa_tile_depth_order: counting sort of every tile's 64 x T samples by
 *     depth -> z_sorted [N*T] (depths), pix [N*T] (pixel inside the 8x8 tile),
 *     slot [N*T] (ray-major index r * T + s), tiles back to back.  N must be a
 *     multiple of image_width (whole rows), T <= 1024.
 *   ucsa_hashgrid_encode_sorted[_hf]: features [L][N*T] (float2 / half2) in that
 *     order.
 *   ucsa_sigma_mlp_fwd_scatter: the sigma MLP on such an array, h [N*T,16] and
 *     sigma [N*T] written at the ray-major slots.  mode 0 = f32-input MFMA
 *     (ucsa_mlp_pack), 1 = f16 nets on fp16 features (ucsa_mlp_pack_f16),
 *     2 = bf16x3 (ucsa_mlp_pack_x3), 3 = f16x2 (ucsa_mlp_pack_h2). */
int32_t ucsa_tile_depth_order(const float* z, uint32_t N, uint32_t T,
                              uint32_t image_width, float* z_sorted,
                              uint8_t* pix, uint32_t* slot, void* stream);
/* The same three arrays for COARSE samples without a sort: "sample index, then
 * pixel" (sample s of every ray of an 8x8 tile sits at nearly one depth:
 * renderer_semantics.py:154-173).  The render path uses it for coarse passes of
 * more than 128 samples per ray, where the depth sort costs more than it brings. */
int32_t ucsa_tile_index_order(const float* z, uint32_t N, uint32_t T,
                              uint32_t image_width, float* z_sorted, uint8_t* pix,
                              uint32_t* slot, void* stream);
int32_t ucsa_hashgrid_encode_sorted(const ucsa_grid* grid, const float* table,
                                    const float* rays_o, const float* rays_d,
                                    const float* z_sorted, const uint8_t* pix,
                                    const float* aabb_host, uint32_t N,
                                    uint32_t T, uint32_t image_width,
                                    float* feat, void* stream);
int32_t ucsa_hashgrid_encode_sorted_hf(const ucsa_grid* grid, const float* table,
                                       const float* rays_o, const float* rays_d,
                                       const float* z_sorted, const uint8_t* pix,
                                       const float* aabb_host, uint32_t N,
                                       uint32_t T, uint32_t image_width,
                                       void* feat_half, void* stream);
int32_t ucsa_sigma_mlp_fwd_scatter(int32_t mode, const void* feat,
                                   const void* packed_sigma, uint32_t M,
                                   uint32_t n_levels, const uint32_t* slot,
                                   float* h, float* sigma, void* stream);
/* ucsa_hashgrid_encode_sorted + ucsa_sigma_mlp_fwd_scatter as ONE call in which the
 * sigma MLP encodes levels 0-11 itself (0-7 with UCSA_DENSITY_LEVELS=8) -- a wave owns
 * 64 consecutive samples of the depth order, the level is wave-uniform, the feature pairs
 * go through a wave-private LDS tile and never through HBM -- and reads the remaining
 * levels (12-15: the ones bound by L2 -> L1 line fills) from feat_ws [16][N*T][2]
 * (written here by the per-level depth-ordered encoder): `density()` of a sample
 * batch, reference network_tcnn_semantics.py:130-144, with three quarters of the
 * feature round trip gone.  mode 2 = bf16x3, 3 = f16x2 packs; 16 levels.  The same h / sigma
 * BITS as the two separate calls. */
int32_t ucsa_density_sorted(int32_t mode, const ucsa_grid* grid, const float* table,
                            const float* rays_o, const float* rays_d,
                            const float* z_sorted, const uint8_t* pix,
                            const uint32_t* slot, const float* aabb_host, uint32_t N,
                            uint32_t T, uint32_t image_width, const void* packed_sigma,
                            float* feat_ws, float* h, float* sigma, void* stream);
/* ucsa_sigma_mlp_fwd_f16 on fp16 features (same h / sigma) */
int32_t ucsa_sigma_mlp_fwd_f16_h(const void* feat_half,
                                 const void* packed_sigma_half, uint32_t M,
                                 uint32_t n_levels, float* h, float* sigma,
                                 void* stream);
/* ucsa_render_fwd_f16 reading the fp16 table */
int32_t ucsa_render_fwd_f16_h16(const ucsa_grid* grid, const void* table_half,
                                const void* packed_sigma_half,
                                const void* packed_color_half,
                                const void* packed_sem_half,
                                const float* rays_o, const float* rays_d,
                                const float* norms, const float* aabb_host,
                                float min_near, const float* t_rand,
                                const float* u, uint32_t N, uint32_t T,
                                uint32_t t, uint32_t n_classes,
                                float density_scale, uint32_t image_width,
                                float* image, float* depth, float* semantics,
                                void* workspace, void* stream);

/* ================= bf16x3: fp32-grade nets on the bf16 MFMA pipe ============ */
/* Every fp32 weight and layer input is split exactly into three bf16 terms and
 * each product is accumulated in fp32 from six bf16 partial products
 * (csrc/mfma_mlp_x3.h): ordinary fp32 round-off, NOT bit-identical to the
 * f32-input MFMA of the entry points above, 144 16-cycle MFMAs per 16 samples
 * instead of 176 32-cycle ones.  Same call sites as the fp32 forms
 * (reference network_tcnn_semantics.py:147-207, renderer_semantics.py:238-299).
 * packed_*_x3: ucsa_mlp_pack_x3_bytes(kind, n_classes) bytes from
 * ucsa_mlp_pack_x3. */
uint32_t ucsa_mlp_pack_x3_bytes(int32_t kind, uint32_t n_classes);
int32_t ucsa_mlp_pack_x3(int32_t kind, const float* params, void* packed_x3,
                         uint32_t n_classes, void* stream);
/* density() nets of a sample batch (cf. ucsa_sigma_mlp_fwd) */
int32_t ucsa_sigma_mlp_fwd_x3(const float* feat, const void* packed_sigma_x3,
                              uint32_t M, uint32_t n_levels, float* h,
                              float* sigma, void* stream);
/* ---- "f16x2": the nets on the f16 MFMA pipe with TWO-term operands ----------
 * x ~ f16(x) + f16((x - f16(x)) * 2^11) * 2^-11: 22 significant bits per
 * operand, products accumulated in fp32 from three f16 MFMA passes (bf16x3: six)
 * -- the same 2^-23-per-product error class as bf16x3 and the f32-input MFMA
 * chain (csrc/mfma_mlp_h2.h).  RANGE: that of the reference's own fp16 nets --
 * layer inputs and weights below 65504 in magnitude, hidden activations below
 * 2^20 (ucsa_mlp_pack_h2 scales the first layer by 2^-4 and the last by 2^4,
 * exact for the bias-free ReLU nets).  Values beyond it are not detected (the
 * _x3 entries have fp32's range).
 * Same interfaces as the _x3 entries with packs from ucsa_mlp_pack_h2
 * (ucsa_mlp_pack_h2_bytes bytes).  Replaces what the _x3 entries replace
 * (tcnn.Network forward, reference network_tcnn_semantics.py:135,147-207). */
uint32_t ucsa_mlp_pack_h2_bytes(int32_t kind, uint32_t n_classes);
int32_t ucsa_mlp_pack_h2(int32_t kind, const float* params, void* packed_h2,
                         uint32_t n_classes, void* stream);
/* The same, and *range_bits (uint32 in device memory or in pinned host memory
 * the device can address -- hipHostMalloc / torch pin_memory --, caller-zeroed,
 * accumulated over calls: the waves raise a device word, the last workgroup
 * publishes it with ONE system-scope atomic max) = the largest |value| it converted to f16, as an fp32
 * bit pattern: >= 0x477FE000 (65504.0f) means a weight outside f16x2's range or
 * not finite -- values the kernels would turn into zeros silently
 * (mfma_mlp_h2.h).  The range guard of `nerf: {precision: f16x2}`
 * (network_tcnn_semantics.py `h2_guard`) without a launch of its own. */
int32_t ucsa_mlp_pack_h2_checked(int32_t kind, const float* params, void* packed_h2,
                                 uint32_t n_classes, uint32_t* range_bits,
                                 uint32_t* scratch /* 2 device uint32, zeroed once */,
                                 void* stream);
int32_t ucsa_sigma_mlp_fwd_h2(const float* feat, const void* packed_sigma_h2,
                              uint32_t M, uint32_t n_levels, float* h,
                              float* sigma, void* stream);
int32_t ucsa_composite_infer_h2(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_h2,
    const void* packed_sem_h2, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, void* workspace, void* stream);
/* ucsa_composite_train_fwd_x3 with f16x2 nets (same outputs and aux arrays). */
int32_t ucsa_composite_train_fwd_h2(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_h2,
    const void* packed_sem_h2, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* w, void* workspace, void* stream);
int32_t ucsa_render_fwd_h2(
    const ucsa_grid* grid, const float* table, const void* packed_sigma_h2,
    const void* packed_color_h2, const void* packed_sem_h2, const float* rays_o,
    const float* rays_d, const float* norms, const float* aabb_host,
    float min_near, const float* t_rand, const float* u, uint32_t N, uint32_t T,
    uint32_t t, uint32_t n_classes, float density_scale, uint32_t image_width,
    float* image, float* depth, float* semantics, void* ws, void* stream);

/* A whole batch of rays (e.g. one view) in `chunk`-ray pieces, what the
 * reference's staged `render` loop does (renderer_semantics.py:321-342), as ONE
 * call: software-pipelined over two internal streams (density half of chunk
 * k+1 || shading half of chunk k; events fork from / join to `stream`), two
 * workspaces of ucsa_render_workspace_bytes(min(N, chunk), T, t, L) bytes each
 * (ws1 NULL = serial loop).  mode: 0 = ucsa_render_fwd, 1 = _f16, 2 = _x3,
 * 3 = _f16_h16, 4 = _h2 (their table / packed-weight formats).  Arrays are for all N
 * rays; image_width as in ucsa_render_fwd (needs chunk % (8 * image_width) == 0
 * and N % image_width == 0, else the ray-ordered gather is used -- same
 * results).  Bit-identical to calling ucsa_render_fwd* per chunk. */
int32_t ucsa_render_view(uint32_t mode, const ucsa_grid* grid, const void* table,
                         const void* packed_sigma, const void* packed_color,
                         const void* packed_sem, const float* rays_o,
                         const float* rays_d, const float* norms,
                         const float* aabb_host, float min_near,
                         const float* t_rand, const float* u, uint32_t N,
                         uint32_t T, uint32_t t, uint32_t n_classes,
                         float density_scale, uint32_t image_width,
                         uint32_t chunk, float* image, float* depth,
                         float* semantics, void* ws0, void* ws1, void* stream);

/* run() as one enqueue (cf. ucsa_render_fwd; same workspace) */
int32_t ucsa_render_fwd_x3(const ucsa_grid* grid, const float* table,
                           const void* packed_sigma_x3,
                           const void* packed_color_x3,
                           const void* packed_sem_x3, const float* rays_o,
                           const float* rays_d, const float* norms,
                           const float* aabb_host, float min_near,
                           const float* t_rand, const float* u, uint32_t N,
                           uint32_t T, uint32_t t, uint32_t n_classes,
                           float density_scale, uint32_t image_width,
                           float* image, float* depth, float* semantics,
                           void* workspace, void* stream);
int32_t ucsa_composite_infer_x3(const float* rays_d, const float* norms,
                                const float* z_c, const float* sigma_c,
                                const float* h_c, const float* z_f,
                                const float* sigma_f, const float* h_f,
                                const void* packed_color_x3,
                                const void* packed_sem_x3, uint32_t N,
                                uint32_t T, uint32_t t, uint32_t n_classes,
                                float density_scale, float* image,
                                float* depth, float* semantics,
                                void* workspace, void* stream);

/* Training forward of the same stage on the split pair with the bf16x3 nets:
 * outputs of ucsa_composite_fwd plus its aux arrays src [N, T + t] (int32
 * source index of every depth-sorted sample: < T coarse, else T + fine index)
 * and w [N, T + t] (its weight) for ucsa_composite_bwd.  Reference:
 * renderer_semantics.py:220-299 as called from forward_nerf_train
 * (joint_train_lightning_net.py:167-223). */
int32_t ucsa_composite_train_fwd_x3(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const void* packed_color_x3,
    const void* packed_sem_x3, uint32_t N, uint32_t T, uint32_t t,
    uint32_t n_classes, float density_scale, float* image, float* depth,
    float* semantics, int32_t* src, float* w, void* workspace, void* stream);

/* ======================= fp16-MFMA inference option ======================== */
/* tiny-cuda-nn evaluates the three MLPs with fp16 weights / layer inputs and
 * fp32 accumulation; the entry points below do the same (16x16x32 f16 MFMA)
 * for inference.  Hash-grid, sampling and compositing stay fp32.  The default
 * (and parity) path is the fp32 one above. */
uint32_t ucsa_mlp_pack_f16_halves(int32_t kind, uint32_t n_classes);
int32_t ucsa_mlp_pack_f16(int32_t kind, const float* params, void* packed_half,
                          uint32_t n_classes, void* stream);
int32_t ucsa_sigma_mlp_fwd_f16(const float* feat, const void* packed_sigma_half,
                               uint32_t M, uint32_t n_levels, float* h,
                               float* sigma, void* stream);
int32_t ucsa_composite_fwd_f16(const float* rays_d, const float* norms,
                               const float* z_c, const float* sigma_c,
                               const float* h_c, const float* z_f,
                               const float* sigma_f, const float* h_f,
                               const void* packed_color_half,
                               const void* packed_sem_half, uint32_t N,
                               uint32_t T, uint32_t t, uint32_t n_classes,
                               float density_scale, float* image, float* depth,
                               float* semantics, int32_t* src, float* weights,
                               void* stream);
int32_t ucsa_render_fwd_f16(const ucsa_grid* grid_host, const float* table,
                            const void* packed_sigma_half,
                            const void* packed_color_half,
                            const void* packed_sem_half, const float* rays_o,
                            const float* rays_d, const float* norms,
                            const float* aabb_host, float min_near,
                            const float* t_rand, const float* u, uint32_t N,
                            uint32_t T, uint32_t t, uint32_t n_classes,
                            float density_scale, uint32_t image_width,
                            float* image, float* depth, float* semantics,
                            void* ws, void* stream);

/* ======================= training (backward) ============================== */

/* A fragments of W^T for the dX = W^T dY passes (layout: DESIGN.md).
 * ucsa_mlp_pack_t_size() elements; call after every parameter update. */
uint32_t ucsa_mlp_pack_t_size(int32_t kind, uint32_t n_classes);
int32_t ucsa_mlp_pack_t(int32_t kind, const float* params, float* packed_t,
                        uint32_t n_classes, void* stream);

/* grad[p] = (accumulate ? grad[p] : 0) + sum_w partial[w][p], summed in a
 * fixed order (deterministic alternative to float atomics for dW). */
int32_t ucsa_reduce_partials(const float* partial, uint32_t n_parts,
                             uint32_t n_params, int32_t accumulate,
                             float* grad, void* stream);

/* The same for up to 4 (partial, grad) pairs in ONE launch (the MLP gradients
 * of a training step).  All arrays are HOST arrays of `count` entries; the
 * summation order per parameter is that of ucsa_reduce_partials. */
int32_t ucsa_reduce_partials_multi(uint32_t count, const float* const* partials,
                                   const uint32_t* n_parts,
                                   const uint32_t* n_params,
                                   float* const* grads, int32_t accumulate,
                                   void* stream);

/* Backward of ucsa_sigma_mlp_fwd (autograd of tcnn.Network in density(),
 * reference network_tcnn_semantics.py:135).  d_h [M,16] is the gradient wrt
 * the RAW outputs (slot 0 already multiplied by the trunc_exp backward,
 * reference activation.py:17-21).  Out: d_feat [n_levels][M][2] and per-wave
 * dW partials [ucsa_sigma_mlp_bwd_parts(M)][3072] (tcnn layout). */
uint32_t ucsa_sigma_mlp_bwd_parts(uint32_t M);
int32_t ucsa_sigma_mlp_bwd(const float* feat, const float* d_h,
                           const float* packed_sigma,
                           const float* packed_sigma_t, uint32_t M,
                           uint32_t n_levels, float* d_feat, float* partial,
                           void* stream);
/* The same with the recomputed hidden layer rounded to fp16 before the ReLU
 * gate and dW2 read it: the backward of tiny-cuda-nn's fp16 sigma net
 * (`train_precision: tcnn`; packs of the fp16-rounded parameters, fp16 features
 * widened to fp32).  Reference network_tcnn_semantics.py:48-58,135. */
int32_t ucsa_sigma_mlp_bwd_h16(const float* feat, const float* d_h,
                               const float* packed_sigma,
                               const float* packed_sigma_t, uint32_t M,
                               uint32_t n_levels, float* d_feat, float* partial,
                               void* stream);
/* The same backward with its contractions on the bf16 MFMA pipe as two-term
 * splits (bf16x2, see ucsa_composite_bwd_x2): packed_sigma_x3 from
 * ucsa_mlp_pack_x3(UCSA_MLP_SIGMA), packed_sigma_t_x3 from ucsa_mlp_pack_t_x3
 * (8 fragments).  Same inputs, outputs and partial-slot count. */
int32_t ucsa_sigma_mlp_bwd_x2(const float* feat, const float* d_h,
                              const void* packed_sigma_x3,
                              const void* packed_sigma_t_x3, uint32_t M,
                              uint32_t n_levels, float* d_feat, float* partial,
                              void* stream);

/* ucsa_hashgrid_bwd_rays with 8-byte bin records: the (entry, vx, vy) records
 * of the fine levels carry their value pair as half2 x rec_scale (a power of
 * two > 0) -- half the record traffic of both passes; sums stay fp32.  For
 * the f16 training mode (tiny-cuda-nn accumulates its grid gradient from half2
 * values too).  The workspace is mandatory. */
int32_t ucsa_hashgrid_bwd_rays_h16(const ucsa_grid* grid, const float* rays_o,
                                   const float* rays_d, const float* z,
                                   const float* aabb_host, uint32_t N,
                                   uint32_t T, const float* d_feat,
                                   float* grad_table, void* workspace,
                                   float rec_scale, void* stream);

/* Transposed fp16 fragments for the f16 backward of the colour / semantics
 * nets (kind = UCSA_MLP_COLOR or UCSA_MLP_SEM), and the backward itself:
 * ucsa_composite_bwd with the two nets and their dX contractions on
 * 16x16x32 f16 MFMA (the forward recompute reproduces ucsa_composite_fwd_f16;
 * dW stays fp32).  `f16_scale` (a power of two, e.g. 1024): gradients entering
 * the f16 MFMAs are multiplied by it and every output divided -- tiny-cuda-nn's
 * loss scale; pass 1 when the incoming gradients already carry a GradScaler
 * scale.  Partials: ucsa_composite_bwd_parts_f16(N) slots. */
uint32_t ucsa_mlp_pack_t_f16_halves(int32_t kind, uint32_t n_classes);
int32_t ucsa_mlp_pack_t_f16(int32_t kind, const float* params, void* packed_half,
                            uint32_t n_classes, void* stream);
uint32_t ucsa_composite_bwd_parts_f16(uint32_t N);
int32_t ucsa_composite_bwd_f16(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const void* packed_color_half,
    const void* packed_sem_half, const void* packed_color_t_half,
    const void* packed_sem_t_half, const float* d_image, const float* d_depth,
    const float* d_sem, uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes,
    float density_scale, float f16_scale, float* G, float* d_h_c, float* d_h_f,
    float* partial_color, float* partial_sem, void* stream);

/* The same backward with every contraction of the two nets on the bf16 MFMA
 * pipe as TWO-term splits ("bf16x2": x = bf16(x) + bf16(x - bf16(x)), three
 * partial products per product, 2^-16 relative, fp32 range: no loss scale) --
 * the gradient contractions of `nerf: {train_precision: bf16x3}`, whose forward
 * is ucsa_composite_train_fwd_x3.  Weights: ucsa_mlp_pack_x3 buffers and their
 * transposed form ucsa_mlp_pack_t_x3 (fragments of ucsa_mlp_pack_t_f16, three
 * exact bf16 terms each; ucsa_mlp_pack_t_x3_bytes).  Partials:
 * ucsa_composite_bwd_parts(N) slots.  Runs as the per-net kernel pair only
 * (argument 0 error under UCSA_SHADE_BWD_SPLIT=0). */
uint32_t ucsa_mlp_pack_t_x3_bytes(int32_t kind, uint32_t n_classes);
int32_t ucsa_mlp_pack_t_x3(int32_t kind, const float* params, void* packed_x3,
                           uint32_t n_classes, void* stream);
int32_t ucsa_composite_bwd_x2(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const void* packed_color_x3, const void* packed_sem_x3,
    const void* packed_color_t_x3, const void* packed_sem_t_x3,
    const float* d_image, const float* d_depth, const float* d_sem, uint32_t N,
    uint32_t T, uint32_t t, uint32_t n_classes, float density_scale, float* G,
    float* d_h_c, float* d_h_f, float* partial_color, float* partial_sem,
    void* stream);

/* Backward of ucsa_hashgrid_encode_rays: adds into grad_table
 * [total_entries,2] (caller zeroes it).  Autograd of tcnn.Encoding.
 * With a workspace of ucsa_hashgrid_bwd_workspace_bytes() the binned two-pass
 * algorithm is used (no global float atomics); workspace == NULL selects the
 * direct-atomics kernels. */
uint64_t ucsa_hashgrid_bwd_workspace_bytes(uint32_t N, uint32_t T,
                                           uint32_t n_levels);
int32_t ucsa_hashgrid_bwd_rays(const ucsa_grid* grid_host, const float* rays_o,
                               const float* rays_d, const float* z,
                               const float* aabb_host, uint32_t N, uint32_t T,
                               const float* d_feat, float* grad_table,
                               void* workspace, void* stream);
/* Both density passes of a training step (the coarse samples z_c [N,Tc] and the
 * resampled ones z_f [N,Tf], renderer_semantics.py:154-236) in ONE call: every
 * ray's Tc + Tf samples are walked in SORTED depth order -- src [N,Tc+Tf] int32
 * as ucsa_composite_fwd / _train_fwd_x3 wrote it (e < Tc: coarse sample e, else
 * fine sample e - Tc).  On the coarse levels the fine samples fall into cells
 * the coarse samples of the same ray already visit: the merged walk combines
 * them into the same runs, so the fine pass adds almost no accumulator traffic;
 * half the launches.  d_feat_c [L][N*Tc][2], d_feat_f [L][N*Tf][2]; workspace
 * ucsa_hashgrid_bwd_workspace_bytes(N, Tc + Tf, L) (required).  The gradient
 * of the two single-pass calls up to the order of fp32 additions. */
int32_t ucsa_hashgrid_bwd_rays_merged(
    const ucsa_grid* grid, const float* rays_o, const float* rays_d,
    const float* z_c, const float* z_f, const int32_t* src, const float* aabb_host,
    uint32_t N, uint32_t Tc, uint32_t Tf, const float* d_feat_c,
    const float* d_feat_f, float* grad_table, void* workspace, void* stream);

/* Deterministic form of ucsa_hashgrid_bwd_rays (debug / reproducibility mode,
 * `UCSA_DETERMINISTIC=1`; SURVEY 5 "race detection": the reference only seeds,
 * scripts/train_joint.py:48 `seed_everything`; this is what
 * torch.use_deterministic_algorithms would ask of its backward,
 * joint_train_lightning_net.py:509-513): every contribution w * d_feat as 64-bit fixed point (2^-44 units) added
 * with integer atomics -- the sum does not depend on the order of the additions,
 * so two runs of a step give the same bits.  `fix`: int64
 * [ucsa_hashgrid_bwd_det_workspace_bytes / 8], zeroed by the caller, accumulates
 * over any number of _det calls (both density passes); _finish adds it into
 * grad_table (NaN everywhere if a contribution was non-finite or >= 2^18, or if
 * an entry's running sum left int64 -- each addition checks its own overflow). */
uint64_t ucsa_hashgrid_bwd_det_workspace_bytes(const ucsa_grid* grid);
int32_t ucsa_hashgrid_bwd_rays_det(const ucsa_grid* grid, const float* rays_o,
                                   const float* rays_d, const float* z,
                                   const float* aabb_host, uint32_t N, uint32_t T,
                                   const float* d_feat, void* fix, void* stream);
int32_t ucsa_hashgrid_bwd_det_finish(const ucsa_grid* grid, const void* fix,
                                     float* grad_table, void* stream);

/* ucsa_hashgrid_bwd_rays / ucsa_hashgrid_bwd_rays_merged with PACKED bin
 * records: one 64-bit word = entry index inside its bin (L bits) | vx | vy,
 * each value the fp32 rounded to nearest-even to its top min(32, (64 - L) / 2)
 * bits -- 26 bits (2^-18 relative) for the 2^19-entry levels of the
 * reference's grid; since round 6 two such words travel as ONE 16-byte record
 * per x-pair of corners (the corners x and x + 1 of a cell share a bin: half the
 * records, counters and staging slots).  Half the record bytes of both passes;
 * the sums stay fp32; non-finite values stay non-finite.  Meant for training modes whose MLP
 * backward is itself a two-term bf16 split (2^-16): the Python host uses them
 * for bwd_precision "bf16x2".  Same arguments; the workspace is mandatory.
 * (No reference counterpart: tiny-cuda-nn scatters with atomics,
 * nr4seg/nerf/network_tcnn_semantics.py:36-46 is the call site.) */
int32_t ucsa_hashgrid_bwd_rays_p64(const ucsa_grid* grid, const float* rays_o,
                                   const float* rays_d, const float* z,
                                   const float* aabb_host, uint32_t N,
                                   uint32_t T, const float* d_feat,
                                   float* grad_table, void* workspace,
                                   void* stream);
int32_t ucsa_hashgrid_bwd_rays_merged_p64(
    const ucsa_grid* grid, const float* rays_o, const float* rays_d,
    const float* z_c, const float* z_f, const int32_t* src, const float* aabb_host,
    uint32_t N, uint32_t Tc, uint32_t Tf, const float* d_feat_c,
    const float* d_feat_f, float* grad_table, void* workspace, void* stream);

/* ---- the differentiable render of a training step as TWO calls --------------
 * SURVEY 8(b)'s ucsa_render_fused_fwd / ucsa_render_fused_bwd.  Replaces
 * SemanticNeRFRenderer.run under autograd (reference
 * nr4seg/nerf/renderer_semantics.py:123-299 as driven by
 * nr4seg/lightning/joint_train_lightning_net.py:473-513): rows a2-a9 for N rays
 * with T coarse + t fine samples, forward keeping what the backward needs, and
 * the backward down to the four flat parameter gradients.  Arithmetic of the
 * default training mode: bf16x3 nets forward (fp32-grade), bf16x2 contractions
 * and 8-byte packed bin records backward.  Both only SEQUENCE the per-stage
 * entry points above on `stream` (same launches, order and arguments as the
 * Python host issues one by one: bit-identical results); they never allocate or
 * synchronise.  All buffers are the caller's, device memory, fp32 unless noted. */
typedef struct ucsa_train_buffers {
  float* z_c;     /* [N,T]  coarse depths */
  float* feat_c;  /* [L][N*T][2]  their hash-grid features */
  float* h_c;     /* [N*T,16]  raw sigma-net outputs */
  float* sigma_c; /* [N,T] */
  float* z_f;     /* [N,t]  resampled depths (t > 0) */
  float* feat_f;  /* [L][N*t][2] */
  float* h_f;     /* [N*t,16] */
  float* sigma_f; /* [N,t] */
  int32_t* src;   /* [N,T+t]  sorted order of the merged samples */
  float* weights; /* [N,T+t]  compositing weights in that order */
} ucsa_train_buffers;

typedef struct ucsa_train_packs {
  const void* sigma_x3;   /* ucsa_mlp_pack_x3(UCSA_MLP_SIGMA) */
  const void* color_x3;   /* ... UCSA_MLP_COLOR */
  const void* sem_x3;     /* ... UCSA_MLP_SEM */
  const void* sigma_t_x3; /* ucsa_mlp_pack_t_x3: backward only (may be NULL forward) */
  const void* color_t_x3;
  const void* sem_t_x3;
  const void* sigma_h2;   /* ucsa_mlp_pack_h2: when all three are non-NULL the FORWARD */
  const void* color_h2;   /* runs the nets as f16x2 (three MFMA passes per product    */
  const void* sem_h2;     /* instead of six; same fp32-grade error) -- may be NULL     */
} ucsa_train_packs;

uint64_t ucsa_render_fused_fwd_workspace_bytes(uint32_t N, uint32_t T, uint32_t t);
/* t_rand [N,T] or NULL (no perturbation), u [N,t] (t > 0).  Writes `out`
 * (every member; the *_f ones only when t > 0), image [N,3], depth [N],
 * semantics [N,n_classes]. */
int32_t ucsa_render_fused_fwd(
    const ucsa_grid* grid, const float* table, const ucsa_train_packs* packs,
    const float* rays_o, const float* rays_d, const float* norms,
    const float* aabb_host, float min_near, const float* t_rand, const float* u,
    uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes, float density_scale,
    const ucsa_train_buffers* out, float* image, float* depth, float* semantics,
    void* workspace, void* stream);

uint64_t ucsa_render_fused_bwd_workspace_bytes(uint32_t N, uint32_t T, uint32_t t,
                                               uint32_t n_classes,
                                               uint32_t n_levels);
/* `saved` as ucsa_render_fused_fwd left it.  grad_table [total_entries,2] is
 * ADDED to (zero it first); grad_sigma [3072], grad_color [7168] and grad_sem
 * [1024 + 1024 * ceil(n_classes / 16)] are overwritten (layouts of the flat
 * tcnn parameter tensors, ucsa_mlp_pack). */
int32_t ucsa_render_fused_bwd(
    const ucsa_grid* grid, const ucsa_train_packs* packs, const float* rays_o,
    const float* rays_d, const float* norms, const float* aabb_host,
    const ucsa_train_buffers* saved, const float* d_image, const float* d_depth,
    const float* d_sem, uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes,
    float density_scale, float* grad_table, float* grad_sigma, float* grad_color,
    float* grad_sem, void* workspace, void* stream);

/* ucsa_hashgrid_bwd_rays_merged with the half2 x rec_scale records of
 * ucsa_hashgrid_bwd_rays_h16 (training modes fp16 / tcnn). */
int32_t ucsa_hashgrid_bwd_rays_merged_h16(
    const ucsa_grid* grid, const float* rays_o, const float* rays_d,
    const float* z_c, const float* z_f, const int32_t* src, const float* aabb_host,
    uint32_t N, uint32_t Tc, uint32_t Tf, const float* d_feat_c,
    const float* d_feat_f, float* grad_table, void* workspace, float rec_scale,
    void* stream);

/* Backward of ucsa_hashgrid_encode_points (x [M,3] explicit points; workspace
 * of ucsa_hashgrid_bwd_workspace_bytes(M, 1, n_levels) or NULL). */
int32_t ucsa_hashgrid_bwd_points(const ucsa_grid* grid_host, const float* x,
                                 uint32_t M, const float* d_feat,
                                 float* grad_table, void* workspace,
                                 void* stream);

/* Backward of ucsa_composite_fwd (autograd of renderer_semantics.py:238-299
 * and the masked color()/semantics()).  Needs the forward's `src` and
 * `weights`.  Out: d_h_c [N*T,16], d_h_f [N*t,16] (all 16 slots; slot 0 is
 * d/d(log-density)), scratch G [N,S], per-wave partials
 * [ucsa_composite_bwd_parts(N)][7168] and [..][1024 + 1024*ceil(C/16)]. */
uint32_t ucsa_composite_bwd_parts(uint32_t N);
int32_t ucsa_composite_bwd(
    const float* rays_d, const float* norms, const float* z_c,
    const float* sigma_c, const float* h_c, const float* z_f,
    const float* sigma_f, const float* h_f, const int32_t* src,
    const float* weights, const float* packed_color, const float* packed_sem,
    const float* packed_color_t, const float* packed_sem_t,
    const float* d_image, const float* d_depth, const float* d_sem, uint32_t N,
    uint32_t T, uint32_t t, uint32_t n_classes, float density_scale, float* G,
    float* d_h_c, float* d_h_f, float* partial_color, float* partial_sem,
    void* stream);

/* torch.optim.Adam step as configured at reference
 * joint_train_lightning_net.py:897-919 (L2 weight decay in the gradient,
 * non-AMSGrad).  `step` is 1-based; grads are multiplied by inv_grad_scale
 * first (GradScaler unscale, :509-513). */
int32_t ucsa_adam_step(float* params, const float* grads, float* exp_avg,
                       float* exp_avg_sq, uint64_t n, uint32_t step, float lr,
                       float beta1, float beta2, float eps, float weight_decay,
                       float inv_grad_scale, void* stream);

/* The same step under torch.amp.GradScaler(enabled=True) (reference
 * joint_train_lightning_net.py:46,:509-513) without a host read-back: the
 * scale and the found-inf flag are DEVICE scalars (what GradScaler hands to
 * optimizers with _step_supports_amp_scaling).  found_inf[0] != 0: nothing is
 * written and the step does not count -- the bias corrections use
 * step - skipped[0].  Call ucsa_adam_count_skipped once per optimizer step
 * after the last tensor (skipped[0] += found_inf[0] != 0). */
int32_t ucsa_adam_step_scaled(float* params, const float* grads, float* exp_avg,
                              float* exp_avg_sq, uint64_t n, uint32_t step,
                              float lr, float beta1, float beta2, float eps,
                              float weight_decay, const float* grad_scale,
                              const float* found_inf, const uint32_t* skipped,
                              void* stream);
int32_t ucsa_adam_count_skipped(const float* found_inf, uint32_t* skipped,
                                void* stream);

/* ======================= losses / post-processing / metric ================ */

/* Scratch floats needed by the reductions below for n rays / pixels. */
uint32_t ucsa_loss_partial_floats(uint32_t n);

/* NeRF losses of reference joint_train_lightning_net.py:180-223 with the
 * weighting of :44-45,:503-507, and (when d_* != NULL) the gradient of
 * grad_scale * total wrt the rendered outputs.
 *   rgb,gt_rgb [N,3]; sem [N,C] composited probabilities; depth, gt_depth [N];
 *   labels [N] int64 (-1 ignored).  stats[8] (device) = {loss_color,
 *   loss_semantics (NaN = the reference's None), loss_depth, n_invalid_sem,
 *   n_valid_depth, total, loss_semantics with None as 0, 0}. */
int32_t ucsa_nerf_loss(const float* rgb, const float* sem, const float* depth,
                       const float* gt_rgb, const int64_t* labels,
                       const float* gt_depth, uint32_t N, uint32_t C,
                       float one_m_to_scene_uom, float w_sem, float w_depth,
                       float grad_scale, float* stats, float* d_rgb,
                       float* d_sem, float* d_depth, float* partial,
                       void* stream);

/* Backward of the loss node of reference joint_train_lightning_net.py:497-513
 * (`total.backward()` / `scaler.scale(total).backward()`) in one launch, from
 * the gradients ucsa_nerf_loss wrote with the same w_sem / w_depth:
 *   out_rgb = d_rgb * (g_total + g_color),  out_sem = d_sem * (g_total + g_sem / w_sem),
 *   out_depth = d_depth * (g_total + g_depth / w_depth)
 * (out_* may alias d_* for an in-place update; the autograd node writes fresh
 * buffers so that a second backward through a retained graph sees the stored
 * gradients unscaled -- ADVICE r4)
 * where g_* are DEVICE scalars (the cotangents autograd hands back: of the
 * total, and of the single terms when a caller combined them itself); NULL = 0.
 * Replaces the ~10 elementwise launches torch's autograd spends on
 * `lc + 0.04 ls + 0.1 ld` and its backward. */
int32_t ucsa_nerf_loss_apply(const float* d_rgb, const float* d_sem,
                             const float* d_depth, float* out_rgb,
                             float* out_sem, float* out_depth,
                             uint32_t N, uint32_t C, const float* g_total,
                             const float* g_color, const float* g_sem,
                             const float* g_depth, float w_sem, float w_depth,
                             void* stream);

/* reference joint_train_lightning_net.py:246-251: rows summing to 0 become
 * uniform, normalise, argmax.  normalised may be NULL. */
int32_t ucsa_semantic_postproc(const float* sem, uint32_t N, uint32_t C,
                               float* normalised, int64_t* argmax,
                               void* stream);

/* Segmentation tail on NCHW logits [B,C,P]: softmax (prob, may be NULL),
 * argmax (may be NULL) and, when labels != NULL, the reference's
 * CrossEntropyLoss(ignore_index=-1, reduction="none")(softmax(out), label)
 * .mean() (double softmax; :37-38,:163,:456-458) plus grad_scale * dLoss/dlogits
 * (d_logits may be NULL). */
int32_t ucsa_seg_tail(const float* logits, const int64_t* labels, uint32_t B,
                      uint32_t C, uint32_t P, float grad_scale, float* prob,
                      int64_t* argmax, float* loss, float* d_logits,
                      float* partial, void* stream);

/* ============ fused BatchNorm2d (+ residual) (+ ReLU), channels-last ======
 * The memory-bound passes between DeepLabV3's convolutions (row a14;
 * reference nr4seg/network/deeplabv3.py:6-19 -> torchvision's bottlenecks:
 * conv -> BatchNorm2d -> ReLU and conv -> BatchNorm2d -> (+ identity) -> ReLU;
 * replaces torch.nn.functional.batch_norm + add + relu and their backward
 * kernels).  Activations are NHWC, i.e. row-major [M = N*H*W][C], C % 4 == 0,
 * 16-byte aligned; dtype 0 = fp32, 1 = bf16 (statistics, gamma / beta and all
 * sums are fp32 / double either way).
 *
 * forward:  y = relu?( (x - mean) * invstd * gamma + beta (+ residual) )
 *   training != 0: batch statistics (biased variance); running_mean /
 *     running_var (may both be NULL) updated as torch.nn.BatchNorm2d does
 *     (momentum, unbiased variance); save_mean / save_invstd [C] written for
 *     the backward.
 *   training == 0: the running statistics normalise; save_* untouched.
 * backward: g = dy * (y > 0) when relu (y = the forward's output), else dy;
 *   dresidual (may be NULL) = g; dbeta = sum g; dgamma = sum g * xhat;
 *   dx = gamma * invstd * (g - dbeta / M - xhat * dgamma / M)   (training mode).
 * workspace: ucsa_bn_workspace_bytes(M, C) bytes of caller-owned scratch.
 * C <= 4096. */
uint64_t ucsa_bn_workspace_bytes(uint32_t M, uint32_t C);
int32_t ucsa_bn_act_fwd(const void* x, const void* residual, const float* gamma,
                        const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps,
                        uint32_t M, uint32_t C, int32_t relu, int32_t training,
                        int32_t dtype, void* y, float* save_mean,
                        float* save_invstd, void* workspace, void* stream);
int32_t ucsa_bn_act_bwd(const void* dy, const void* x, const void* y,
                        const float* gamma, const float* save_mean,
                        const float* save_invstd, uint32_t M, uint32_t C,
                        int32_t relu, int32_t dtype, void* dx, void* dresidual,
                        float* dgamma, float* dbeta, void* workspace,
                        void* stream);

/* Confusion matrix, rows = truth, truth == -1 dropped (reference
 * nr4seg/utils/metrics.py:31-46); adds into cm [C,C] int64. */
int32_t ucsa_confusion_matrix(const int64_t* preds, const int64_t* truths,
                              uint64_t n, uint32_t C, int64_t* cm,
                              void* stream);

/* ============ occupancy-grid ray marching ===============
 * The `_raymarching` functions that are dormant in the reference
 * (bindings.cpp:8-16, raymarching.h:9-18).  Same argument meaning as the
 * reference's host functions; at::Tensor -> device pointer.  Where the CUDA
 * kernels hand out output slots with atomicAdd (arrival order), these use
 * prefix sums over the ray index: deterministic, and one of the orders the
 * reference can produce.  density_grid is [C,H,H,H] fp32. */

/* Scratch for ucsa_march_rays_train over N rays. */
uint64_t ucsa_march_workspace_bytes(uint32_t N);

/* Replaces march_rays_train (reference raymarching.cu:138-307,:490-498;
 * wrapper raymarching.py:54-163).  xyzs, dirs [M,3], deltas [M,2] (dt, and
 * t - last_t for depth), rays [N,3] int32 = (ray id, first point, n points);
 * counter[2] int32 += (points, rays), its old values are the bases.  A ray
 * whose span would reach M is counted but not written.  perturb != 0 jitters
 * the start by MIN_STEPSIZE * pcg32(n).next_float(). */
int32_t ucsa_march_rays_train(const float* rays_o, const float* rays_d,
                              const float* density_grid, float mean_density,
                              float bound, float dt_gamma, uint32_t N,
                              uint32_t C, uint32_t H, uint32_t M,
                              const float* nears, const float* fars,
                              float* xyzs, float* dirs, float* deltas,
                              int32_t* rays, int32_t* counter, uint32_t perturb,
                              void* workspace, void* stream);

/* Replaces composite_rays_train_forward (reference raymarching.cu:318-394,
 * :501-509) and, with n_sem > 0, the disabled
 * composite_rays_train_semantics_forward (raymarching.h:12,
 * raymarching.py:249-309): local_sem [M,n_sem] is composited with the same
 * weights into semantics [N,n_sem].  Outputs are indexed by ray id. */
int32_t ucsa_composite_rays_train_fwd(const float* sigmas, const float* rgbs,
                                      const float* local_sem,
                                      const float* deltas, const int32_t* rays,
                                      uint32_t M, uint32_t N, uint32_t n_sem,
                                      float* weights_sum, float* depth,
                                      float* image, float* semantics,
                                      void* stream);

/* Replaces composite_rays_train_backward (reference raymarching.cu:408-487,
 * :512-520).  Rows of grad_sigmas / grad_rgbs / grad_local_sem that belong to
 * no composited ray are left untouched (the caller zero-fills,
 * raymarching.py:225-226).  The semantic weights are detached, as on the live
 * path (reference renderer_semantics.py:268-271). */
int32_t ucsa_composite_rays_train_bwd(
    const float* grad_weights_sum, const float* grad_image,
    const float* grad_semantics, const float* sigmas, const float* rgbs,
    const float* deltas, const int32_t* rays, const float* weights_sum,
    const float* image, uint32_t M, uint32_t N, uint32_t n_sem,
    float* grad_sigmas, float* grad_rgbs, float* grad_local_sem, void* stream);

/* Replaces march_rays (reference raymarching.cu:528-634,:637-643; wrapper
 * raymarching.py:367-449): up to n_step points for each of the first n_alive
 * entries of rays_alive / rays_t into rows [n*n_step, (n+1)*n_step); rows
 * past a ray's last point are left untouched (zero-filled by the caller). */
int32_t ucsa_march_rays(uint32_t n_alive, uint32_t n_step,
                        const int32_t* rays_alive, const float* rays_t,
                        const float* rays_o, const float* rays_d, float bound,
                        float dt_gamma, uint32_t C, uint32_t H,
                        const float* density_grid, float mean_density,
                        const float* nears, const float* fars, float* xyzs,
                        float* dirs, float* deltas, uint32_t perturb,
                        void* stream);

/* Replaces composite_rays (reference raymarching.cu:647-729,:732-738, unbound
 * in bindings.cpp; wrapper raymarching.py:455-501) and, with n_sem > 0,
 * composite_rays_semantics (raymarching.py:507-555).  In place on
 * weights_sum/depth/image/semantics (by ray id) and rays_t (by slot; -1 marks
 * a ray that stopped: zero delta, or transmittance below 1e-4). */
int32_t ucsa_composite_rays(uint32_t n_alive, uint32_t n_step,
                            const int32_t* rays_alive, float* rays_t,
                            const float* sigmas, const float* rgbs,
                            const float* local_sem, const float* deltas,
                            uint32_t n_sem, float* weights_sum, float* depth,
                            float* image, float* semantics, void* stream);

/* Scratch for ucsa_compact_rays over n_alive slots. */
uint64_t ucsa_compact_workspace_bytes(uint32_t n_alive);

/* Replaces compact_rays (reference raymarching.cu:838-864; wrapper
 * raymarching.py:561-592): slots with rays_t_old >= 0 are copied, in order,
 * to rays_alive / rays_t starting at alive_counter[0], which is advanced. */
int32_t ucsa_compact_rays(uint32_t n_alive, int32_t* rays_alive,
                          const int32_t* rays_alive_old, float* rays_t,
                          const float* rays_t_old, int32_t* alive_counter,
                          void* workspace, void* stream);

/* ---- training through the marcher ---------------------------------------------
 * Forward / backward of one training batch sampled by ucsa_march_rays_train
 * (rays [N,3] = ray id, first point, count; M points).  What the reference's
 * parent code did with march_rays_train -> network -> composite_rays_train,
 * fused like the inference path: sigma / h [M,16] come from the sigma MLP,
 * the colour and semantics nets run inside, only on samples with w > w_min.
 *   rays rows must be in point order (row n's span starts where row n-1's
 *   ends, as ucsa_march_rays_train writes them): a wave streams the points
 *   of its rows as one contiguous range.
 *   fwd: weights_sum, depth (= sum w*t, t measured from the ray origin:
 *        nears[ray] + the accumulated deltas), image, semantics by ray id --
 *        ADDED to zero-filled buffers; w_out, t_out [M] (zero-filled) keep
 *        each sample's weight and ray parameter for the backward.  No early
 *        stop (raymarching.cu:363 is commented out in the reference as well);
 *        spans reaching M are skipped (:338).
 *   bwd: d_image [N,3], d_depth [N] (wrt depth/norms[ray]), d_sem [N,C] ->
 *        d_h [M,16] (slot 0: d/d log-density incl. the trunc_exp backward),
 *        scratch G [M], per-wave dW partials
 *        [ucsa_composite_bwd_parts(N)][7168] and [..][1024+1024*ceil(C/16)].
 *        Semantic weights are detached (reference renderer_semantics.py:270).
 *        Unlike the reference's composite_rays_train backward
 *        (raymarching.py:209) the depth gradient IS propagated, as the live
 *        path does. */
int32_t ucsa_march_train_fwd(
    const int32_t* rays, uint32_t N, uint32_t M, const float* nears,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* packed_color, const float* packed_sem,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth,
    float* image, float* semantics, float* w_out, float* t_out, void* stream);

int32_t ucsa_march_train_bwd(
    const int32_t* rays, uint32_t N, uint32_t M, const float* rays_d,
    const float* norms, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* w_all, const float* t_all,
    const float* packed_color, const float* packed_sem,
    const float* packed_color_t, const float* packed_sem_t, uint32_t n_classes,
    float w_min, const float* d_image, const float* d_depth, const float* d_sem,
    float* G, float* d_h, float* partial_color, float* partial_sem,
    void* stream);

/* ---- density-grid maintenance -----------------------------------------------
 * No counterpart in the reference's FFI: the reference keeps the state
 * (density_grid [cascade,128,128,128], mean_density, iter_density:
 * renderer_semantics.py:91-103,111-121) that march_rays* read, but ships no
 * code that fills it.  These two entries do, for cuda_ray=True. */

/* One point per cell of cascade `cascade` of an [H,H,H] grid over
 * [-b,b]^3, b = min(2^cascade, bound): the cell centre plus, when seed != 0, a
 * pcg32(cell, seed) jitter inside the cell.  xyz [H^3,3], cell order
 * x*H*H + y*H + z (the marcher's lookup order, raymarching.cu:204). */
int32_t ucsa_density_grid_points(uint32_t cascade, uint32_t H, float bound,
                                 uint32_t seed, float* xyz, void* stream);

uint64_t ucsa_density_grid_workspace_bytes(void);

/* density_grid[i] = max(density_grid[i]*decay, fresh[i]*fresh_scale) where
 * both are >= 0 (negative marks "never update"); *mean_density (device) =
 * mean(max(density_grid, 0)).  n = cascade*H^3. */
int32_t ucsa_density_grid_update(float* density_grid, const float* fresh,
                                 uint64_t n, float decay, float fresh_scale,
                                 float* mean_density, void* workspace,
                                 void* stream);

/* ---- segmented marching: the inference loop without per-step host round trips
 * No counterpart in the reference's FFI.  The reference API above pads every
 * alive ray to n_step rows and needs the survivor count on the host after
 * every iteration; these entries march up to `cap` samples per alive ray into
 * an exact-size buffer, composite them with the same early-termination rule
 * (raymarching.cu:693-706) one wave per ray, and keep the alive count in
 * device memory.  n_cap = host upper bound of the alive count (launch size);
 * n_alive_dev = device int32 holding the actual count, or NULL (= n_cap).
 * Slot n of rays_alive / rays_t / span is alive ray n of this round. */
uint64_t ucsa_march_segment_workspace_bytes(uint32_t n_cap);

/* Count the samples of each alive ray (<= cap) from rays_t[n] on;
 * span[n] = (first point, count) int32 [n_cap,2]; workspace uint32[0] = total
 * points, [1] = alive count used (device; read them back to size the point
 * buffers).  perturb != 0 jitters the start by MIN_STEPSIZE*pcg32(ray,
 * perturb) -- pass it in the first round only, and to _write as well. */
int32_t ucsa_march_segment_count(uint32_t n_cap, const int32_t* n_alive_dev,
                                 uint32_t cap, const int32_t* rays_alive,
                                 const float* rays_t, const float* rays_o,
                                 const float* rays_d, float bound,
                                 float dt_gamma, uint32_t C, uint32_t H,
                                 const float* density_grid, float mean_density,
                                 const float* fars, uint32_t perturb,
                                 int32_t* span, void* workspace, float* stage,
                                 void* stream);

/* `stage` (optional, ucsa_march_segment_stage_bytes(n_cap, cap) bytes): when
 * given to _count, the samples are kept in slot-major staging rows during the
 * counting march and _write (same `stage`, same `cap`) only copies them into
 * place -- one march per round instead of two.
 * Write the counted samples: xyzs, dirs [M,3], deltas [M,2] as march_rays. */
uint64_t ucsa_march_segment_stage_bytes(uint32_t n_cap, uint32_t cap);
int32_t ucsa_march_segment_write(uint32_t n_cap, const int32_t* n_alive_dev,
                                 const int32_t* rays_alive, const float* rays_t,
                                 const float* rays_o, const float* rays_d,
                                 float bound, float dt_gamma, uint32_t C,
                                 uint32_t H, const float* density_grid,
                                 float mean_density, const float* fars,
                                 uint32_t perturb, const int32_t* span,
                                 float* xyzs, float* dirs, float* deltas,
                                 const float* stage, uint32_t cap,
                                 void* stream);

/* Composite each slot's span onto weights_sum/depth/image/semantics (by ray
 * id, accumulated in place; sigma is multiplied by sigma_scale) and set
 * rays_t[n] = -1 if the ray is finished (transmittance fell below 1e-4, or the
 * span is shorter than cap), else the ray parameter after its last sample. */
int32_t ucsa_march_segment_composite(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* sigmas, float sigma_scale, const float* rgbs,
    const float* local_sem, const float* deltas, uint32_t n_sem,
    float* weights_sum, float* depth, float* image, float* semantics,
    void* stream);

/* Fused alternative to ucsa_point_shade_h + ucsa_march_segment_composite: the
 * weights (same early-stop rule) are formed first, the samples with
 * w > w_min are compacted with wave ballots, only those go through the colour
 * and semantics nets (MFMA, weights in LDS; h [M,16] = raw sigma-MLP rows,
 * rays_d [N,3] by ray id), and the per-ray sums are added in place.  w_min = 0
 * is the reference's composite exactly; 1e-4 is the mask of the live path
 * (reference renderer_semantics.py:249-250; depth is masked with it, as
 * there).  weights_sum always takes every sample. */
int32_t ucsa_march_segment_shade(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const float* packed_color, const float* packed_sem,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth,
    float* image, float* semantics, void* stream);

/* Same with the fp16-MFMA nets (packed by ucsa_mlp_pack_f16). */
int32_t ucsa_march_segment_shade_f16(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const void* packed_color_half,
    const void* packed_sem_half, uint32_t n_classes, float w_min,
    float* weights_sum, float* depth, float* image, float* semantics,
    void* stream);

/* Same with the f16x2 nets (packed by ucsa_mlp_pack_h2: two f16 terms per operand,
 * three MFMA passes per product -- the fp32-grade arithmetic of ucsa_render_fwd_h2;
 * round 6). */
int32_t ucsa_march_segment_shade_h2(
    uint32_t n_cap, const int32_t* n_alive_dev, uint32_t cap,
    const int32_t* rays_alive, float* rays_t, const int32_t* span,
    const float* rays_d, const float* sigmas, float sigma_scale, const float* h,
    const float* deltas, const void* packed_color_h2, const void* packed_sem_h2,
    uint32_t n_classes, float w_min, float* weights_sum, float* depth,
    float* image, float* semantics, void* stream);

/* Stable compaction of the slots with rays_t_old >= 0 into rays_alive/rays_t;
 * n_alive_out[0] (device, != n_alive_dev) = number of survivors. */
int32_t ucsa_march_segment_compact(uint32_t n_cap, const int32_t* n_alive_dev,
                                   int32_t* rays_alive,
                                   const int32_t* rays_alive_old, float* rays_t,
                                   const float* rays_t_old,
                                   int32_t* n_alive_out, void* workspace,
                                   void* stream);

/* ---- rendered-image augmentation of the joint step -------
 * Replaces data_aug (reference joint_train_lightning_net.py:259-302; the
 * transforms of :89-101 are torchvision 0.12.0 tensor ops, ~40 kernels per
 * image).  Every random draw is an argument. */
typedef struct ucsa_aug_params {
  int32_t order[4];  /* ColorJitter's drawn permutation of 0 brightness,
                        1 contrast, 2 saturation, 3 hue */
  float brightness, contrast, saturation, hue; /* the four drawn factors */
  float angle_deg;   /* tvf.rotate angle (counter-clockwise, degrees) */
  int32_t flip;      /* != 0: tvf.hflip */
  int32_t crop_i, crop_j; /* RandomCrop.get_params top / left */
} ucsa_aug_params;

uint64_t ucsa_augment_workspace_bytes(uint32_t B, uint32_t H, uint32_t W);

/* img [B,3,H,W] in [0,1], label [B,H,W] int64 (-1 unknown) or NULL;
 * params_host: B structs on the HOST.  out_img [B,3,oh,ow], out_label
 * [B,oh,ow]: jitter -> rotate (bilinear, resp. nearest on label+1, fill 0) ->
 * crop to (oh,ow) at (crop_i,crop_j) -> flip. */
int32_t ucsa_augment(const float* img, const int64_t* label, uint32_t B,
                     uint32_t H, uint32_t W, const ucsa_aug_params* params_host,
                     uint32_t oh, uint32_t ow, float* out_img,
                     int64_t* out_label, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UCSA_HIP_H_ */
