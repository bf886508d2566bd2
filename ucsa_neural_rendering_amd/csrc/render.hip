// ucsa_render_fwd: the reference's SemanticNeRFRenderer.run()
// (nr4seg/nerf/renderer_semantics.py:123-299) as one enqueue of the staged
// kernels, with every intermediate in a caller-provided workspace.
// Also: ucsa_version / ucsa_error_string.
#include <cstdio>
#include <cstdlib>
#include <mutex>

#include "ucsa_common.h"

#include <cassert>
#include <cstring>

// ---------------------------------------------------------------------------
// Environment switches: ONE table, snapshotted once per process (VERDICT r5 item
// 10: they used to be getenv() calls on every C call).  None of them changes a
// result; INTEGRATION.md lists what each selects.
// ---------------------------------------------------------------------------
namespace {
const char* const kEnvNames[] = {
    // launch shapes / kernel choices of the render path
    "UCSA_SHADE_VARIANT", "UCSA_SPLIT_COMPOSITE", "UCSA_ENC_SORTED", "UCSA_ENC_ML",
    "UCSA_DENSITY_FUSED", "UCSA_DENSITY_LEVELS",
    "UCSA_ENC_SORTED_ML", "UCSA_ENC_SORTED_LEAN",
    // ... of the training path
    "UCSA_SHADE_BWD_SPLIT", "UCSA_BWD_OVERLAP", "UCSA_BWD_BIN_SCALE",
    // lab only (tools/encode_*.py, tools/coresident_exp.py)
    "UCSA_ENC_ORDER", "UCSA_ENC_SIMPLE", "UCSA_ENC_SIMPLE_H", "UCSA_ENC_SIMPLE_RAYS",
    "UCSA_ENC_LDS_PAD", "UCSA_SORT_BINS", "UCSA_SORT_EXACT"};
constexpr int kEnvCount = (int)(sizeof(kEnvNames) / sizeof(kEnvNames[0]));
struct EnvTable {
  char value[kEnvCount][64];
  bool set[kEnvCount];
  void load() {
    for (int i = 0; i < kEnvCount; ++i) {
      const char* v = getenv(kEnvNames[i]);
      set[i] = v && *v && strlen(v) < sizeof(value[i]);
      if (set[i]) strcpy(value[i], v);
    }
  }
};
std::mutex env_mu;
EnvTable* env_table() {
  static EnvTable t = []() { EnvTable x; x.load(); return x; }();   // once, thread-safe
  return &t;
}
}  // namespace

const char* ucsa_getenv(const char* name) {
  EnvTable* t = env_table();
  for (int i = 0; i < kEnvCount; ++i)
    if (strcmp(name, kEnvNames[i]) == 0) return t->set[i] ? t->value[i] : nullptr;
  assert(!"ucsa_getenv: a name that is not in kEnvNames");
  return nullptr;
}

// lab tools and tests that flip a switch inside one process (tools/bwd_switches_ab.py,
// tests of the alternative kernels); not for production code, not thread-safe
// against concurrent calls into the library
extern "C" void ucsa_env_reload(void) {
  std::lock_guard<std::mutex> lk(env_mu);
  env_table()->load();
}

namespace {
struct Ws {
  float *nears, *fars, *z_c, *z_f, *feat, *h_c, *sigma_c, *h_f, *sigma_f;
  // depth order of the fine samples (hashgrid_sorted.hip)
  float* zs_sorted;
  uint32_t* slot;
  uint8_t* pix;
  void* cmp;  // ucsa_composite_infer's survivor lists
  uint64_t bytes;
};

inline uint64_t align256(uint64_t b) { return (b + 255ull) & ~255ull; }

Ws carve(void* base, uint32_t N, uint32_t T, uint32_t t, uint32_t L) {
  Ws w;
  uint64_t off = 0;
  char* p = (char*)base;
  auto take = [&](uint64_t n_floats) {
    float* r = (float*)(p + off);
    off += align256(n_floats * 4);
    return r;
  };
  const uint64_t Mc = (uint64_t)N * T, Mf = (uint64_t)N * t;
  const uint64_t Mmax = Mc > Mf ? Mc : Mf;
  w.nears = take(N);
  w.fars = take(N);
  w.z_c = take(Mc);
  w.z_f = take(Mf ? Mf : 1);
  w.feat = take(Mmax * 2 * L);
  w.h_c = take(Mc * 16);
  w.sigma_c = take(Mc);
  w.h_f = take(Mf ? Mf * 16 : 1);
  w.sigma_f = take(Mf ? Mf : 1);
  const uint64_t Ms = Mmax;   // (UCSA_ENC_SORTED=2 orders the coarse pass too)
  w.zs_sorted = take(Ms);
  w.slot = (uint32_t*)take(Ms);
  w.pix = (uint8_t*)take((Ms + 3) / 4);
  w.cmp = take(ucsa_composite_infer_workspace_bytes(N, T, t) / 4 + 64);
  w.bytes = off;
  return w;
}
}  // namespace

extern "C" uint64_t ucsa_render_workspace_bytes(uint32_t N, uint32_t T,
                                                uint32_t t,
                                                uint32_t n_levels) {
  return carve(nullptr, N, T, t, n_levels).bytes;
}

#define UCSA_TRY(expr)          \
  do {                          \
    int32_t rc_ = (expr);       \
    if (rc_ != 0) return rc_;   \
  } while (0)

// image_width > 0: image-ordered rays -> tile-ordered gather (same features)
static int32_t encode(const ucsa_grid* grid, const void* table_any,
                      bool table_half, const float* rays_o, const float* rays_d,
                      const float* z, const float* aabb_host, uint32_t N,
                      uint32_t T, uint32_t image_width, float* feat,
                      void* stream) {
  if (table_half)
    return ucsa_hashgrid_encode_rays_h16(grid, table_any, rays_o, rays_d, z,
                                         aabb_host, N, T, image_width, feat,
                                         stream);
  const float* table = (const float*)table_any;
  if (image_width)
    return ucsa_hashgrid_encode_rays_image(grid, table, rays_o, rays_d, z,
                                           aabb_host, N, T, image_width, feat,
                                           stream);
  return ucsa_hashgrid_encode_rays(grid, table, rays_o, rays_d, z, aabb_host, N,
                                   T, feat, stream);
}

#ifndef UCSA_ENC_SORTED_DEFAULT
#define UCSA_ENC_SORTED_DEFAULT 2
#endif

static bool density_fused() {
  const char* v = ucsa_getenv("UCSA_DENSITY_FUSED");
  return !(v && v[0] == '0');
}

// Which composite: the f32-input MFMA is bound by the matrix pipe itself (352
// MFMAs per 32 samples = 1.62 ms per 61 440-ray chunk at 100 % of the pipe)
// and the fused k_composite (2.25 ms) beats the split pair there (2.5-2.8 ms);
// the split pair wins once the nets run on the 16-bit MFMA pipe (f16, bf16x3).
// UCSA_SPLIT_COMPOSITE=0/1 overrides (bf16x3 exists as the split pair only).
static bool split_composite(int prec) {
  if (prec >= 2) return true;
  const char* v = ucsa_getenv("UCSA_SPLIT_COMPOSITE");
  if (v && (v[0] == '0' || v[0] == '1')) return v[0] == '1';
  return prec == 1;
}

// prec: 0 = f32-input MFMA nets (packed by ucsa_mlp_pack), 1 = f16 MFMA
// (ucsa_mlp_pack_f16), 2 = bf16x3 (ucsa_mlp_pack_x3), 3 = f16x2
// (ucsa_mlp_pack_h2)
// (table_half is only offered together with the f16 nets: its encoder emits
// fp16 features, which only the f16 sigma MLP reads)
// stage: bit 0 = density half (near/far .. sigma of the fine samples, into the
// workspace), bit 1 = shading half (weights, compaction, colour / semantics
// nets, compositing, from the workspace); 3 = the whole run().
static int32_t render_impl(int prec, const ucsa_grid* grid,
                           const void* table_any, bool table_half,
                           const void* packed_sigma, const void* packed_color,
                           const void* packed_sem, const float* rays_o,
                           const float* rays_d, const float* norms,
                           const float* aabb_host, float min_near,
                           const float* t_rand, const float* u, uint32_t N,
                           uint32_t T, uint32_t t, uint32_t n_classes,
                           float density_scale, uint32_t image_width,
                           float* image, float* depth, float* semantics,
                           void* ws, void* stream, uint32_t stage = 3u) {
  UCSA_CHECK_ARG(grid, 0);
  UCSA_CHECK_ARG(ws, 21);
  UCSA_CHECK_ARG(t == 0 || u, 11);
  if (N == 0) return 0;
  const Ws w = carve(ws, N, T, t, grid->n_levels);
  const float* table = table_half ? nullptr : (const float*)table_any;
  if (stage & 1u) {
  UCSA_TRY(ucsa_near_far_from_aabb(rays_o, rays_d, aabb_host, N, min_near,
                                   w.nears, w.fars, stream));
  UCSA_TRY(ucsa_sample_coarse(w.nears, w.fars, t_rand, N, T, w.z_c, stream));
  // the depth-ordered path (hashgrid_sorted.hip) for image-ordered rays.
  // UCSA_ENC_SORTED = 0 off, 1 the fine pass only, 2 (default since round 6) both
  // passes -- the coarse one through the per-tile depth sort up to 128 samples per ray
  // and in "sample index, then pixel" order beyond (ucsa_tile_index_order), and only
  // where the fused kernel takes it (bf16x3 / f16x2 nets) --, 3 both passes always
  // through the sort, 4 like 2 with long coarse passes image-ordered.  Measured: with
  // levels 0-11 encoded inside the sigma MLP (density_sorted.hip) the coarse pass gains
  // more from the fused kernel than its order costs -- cfg2 view 16.18 ms (fine pass
  // only, unfused) / 15.89 (fine fused) / 15.66 (both fused, 8 levels inside) / 15.01
  // (12 levels inside); at 96 samples the index order measured 15.75 against 15.66 with
  // the sort (a tile's depth slabs are tighter than its equal-index sample sets); at the
  // reference's native 256 + 256 samples the sort (144 KiB of LDS per tile) costs more
  // than it brings (cfg3's joint step 178.7 ms sorted) while the index order wins
  // (171.8 ms with that pass image-ordered -> 168.0).  Same h / sigma bits whatever the
  // mode (profiles/r06_density_fused_ab.txt).
  const char* es = ucsa_getenv("UCSA_ENC_SORTED");
  const int sorted_mode = es && es[0] >= '0' && es[0] <= '4' ? es[0] - '0' : UCSA_ENC_SORTED_DEFAULT;
  // encode + sigma MLP of one pass (z [N,n] -> h, sigma)
  auto density = [&](const float* z, uint32_t n, float* h, float* sigma) -> int32_t {
    const bool fine = z == w.z_f;
    // a coarse pass of more than 128 samples per ray: "sample index, then pixel" order
    // instead of the depth sort (mode 2, default); mode 3: always the sort; mode 4:
    // such a pass stays image-ordered (the rule before ucsa_tile_index_order came back)
    const bool coarse_long = !fine && n > 128u;
    if (image_width && !table_half && n <= 1024u &&
        N % image_width == 0 && grid->n_levels == 16 &&
        (fine ? sorted_mode >= 1
              // (the coarse pass only where the fused kernel takes it: with the staged
              // encoder + sigma-MLP pair -- fp32 / fp16 nets -- an ordered coarse pass is
              // slower than the image-ordered tiled one, 16.70 against 16.18 ms per view)
              : (sorted_mode == 3 || (sorted_mode >= 2 && prec >= 2 && density_fused()))) &&
        !(coarse_long && sorted_mode == 4)) {
      if (coarse_long && sorted_mode == 2)
        UCSA_TRY(ucsa_tile_index_order(z, N, n, image_width, w.zs_sorted, w.pix,
                                       w.slot, stream));
      else
        UCSA_TRY(ucsa_tile_depth_order(z, N, n, image_width, w.zs_sorted, w.pix,
                                       w.slot, stream));
      // bf16x3 / f16x2 nets: levels 0-11 are encoded INSIDE the sigma MLP (their
      // features never travel through HBM: density_sorted.hip; same h / sigma
      // bits; UCSA_DENSITY_FUSED=0 keeps the staged pair for A/B runs)
      if (prec >= 2 && density_fused())
        return ucsa_density_sorted(prec, grid, table, rays_o, rays_d, w.zs_sorted,
                                   w.pix, w.slot, aabb_host, N, n, image_width,
                                   packed_sigma, w.feat, h, sigma, stream);
      if (prec == 1)
        UCSA_TRY(ucsa_hashgrid_encode_sorted_hf(grid, table, rays_o, rays_d,
                                                w.zs_sorted, w.pix, aabb_host, N, n,
                                                image_width, w.feat, stream));
      else
        UCSA_TRY(ucsa_hashgrid_encode_sorted(grid, table, rays_o, rays_d,
                                             w.zs_sorted, w.pix, aabb_host, N, n,
                                             image_width, w.feat, stream));
      return ucsa_sigma_mlp_fwd_scatter(prec, w.feat, packed_sigma, N * n,
                                        grid->n_levels, w.slot, h, sigma, stream);
    }
    if (prec == 1 && !table_half) {  // f16 nets: fp16 features at the source
      UCSA_TRY(ucsa_hashgrid_encode_rays_hf(grid, table, rays_o, rays_d, z,
                                            aabb_host, N, n, image_width, w.feat,
                                            stream));
      return ucsa_sigma_mlp_fwd_f16_h(w.feat, packed_sigma, N * n,
                                      grid->n_levels, h, sigma, stream);
    }
    UCSA_TRY(encode(grid, table_any, table_half, rays_o, rays_d, z, aabb_host, N,
                    n, image_width, w.feat, stream));
    if (prec == 0)
      return ucsa_sigma_mlp_fwd(w.feat, (const float*)packed_sigma, N * n,
                                grid->n_levels, h, sigma, stream);
    if (prec == 1 && table_half)  // the h16 encoder wrote fp16 features
      return ucsa_sigma_mlp_fwd_f16_h(w.feat, packed_sigma, N * n,
                                      grid->n_levels, h, sigma, stream);
    if (prec == 3)
      return ucsa_sigma_mlp_fwd_h2(w.feat, packed_sigma, N * n, grid->n_levels, h,
                                   sigma, stream);
    return ucsa_sigma_mlp_fwd_x3(w.feat, packed_sigma, N * n, grid->n_levels, h,
                                 sigma, stream);
  };
  UCSA_TRY(density(w.z_c, T, w.h_c, w.sigma_c));
  if (t > 0) {
    UCSA_TRY(ucsa_resample(w.z_c, w.sigma_c, u, N, T, t, density_scale, w.z_f,
                           stream));
    UCSA_TRY(density(w.z_f, t, w.h_f, w.sigma_f));
  }
  }  // stage & 1
  if (!(stage & 2u)) return 0;
  if (split_composite(prec)) {
    if (prec == 0)
      return ucsa_composite_infer(rays_d, norms, w.z_c, w.sigma_c, w.h_c, w.z_f,
                                  w.sigma_f, w.h_f, (const float*)packed_color,
                                  (const float*)packed_sem, N, T, t, n_classes,
                                  density_scale, image, depth, semantics, w.cmp,
                                  stream);
    const auto infer = prec == 3 ? ucsa_composite_infer_h2
                       : prec == 2 ? ucsa_composite_infer_x3 : ucsa_composite_infer_f16;
    return infer(rays_d, norms, w.z_c, w.sigma_c, w.h_c, w.z_f, w.sigma_f, w.h_f,
                 packed_color, packed_sem, N, T, t, n_classes, density_scale,
                 image, depth, semantics, w.cmp, stream);
  }
  if (prec == 1)
    return ucsa_composite_fwd_f16(rays_d, norms, w.z_c, w.sigma_c, w.h_c, w.z_f,
                                  w.sigma_f, w.h_f, packed_color, packed_sem, N,
                                  T, t, n_classes, density_scale, image, depth,
                                  semantics, nullptr, nullptr, stream);
  return ucsa_composite_fwd(rays_d, norms, w.z_c, w.sigma_c, w.h_c, w.z_f,
                            w.sigma_f, w.h_f, (const float*)packed_color,
                            (const float*)packed_sem, N, T, t, n_classes,
                            density_scale, image, depth, semantics, nullptr,
                            nullptr, stream);
}

extern "C" int32_t ucsa_render_fwd(
    const ucsa_grid* grid, const float* table, const float* packed_sigma,
    const float* packed_color, const float* packed_sem, const float* rays_o,
    const float* rays_d, const float* norms, const float* aabb_host,
    float min_near, const float* t_rand, const float* u, uint32_t N,
    uint32_t T, uint32_t t, uint32_t n_classes, float density_scale,
    uint32_t image_width, float* image, float* depth, float* semantics,
    void* ws, void* stream) {
  return render_impl(0, grid, table, false, packed_sigma, packed_color, packed_sem,
                     rays_o, rays_d, norms, aabb_host, min_near, t_rand, u, N, T,
                     t, n_classes, density_scale, image_width, image, depth,
                     semantics, ws, stream);
}

extern "C" int32_t ucsa_render_fwd_f16(
    const ucsa_grid* grid, const float* table, const void* packed_sigma_half,
    const void* packed_color_half, const void* packed_sem_half,
    const float* rays_o, const float* rays_d, const float* norms,
    const float* aabb_host, float min_near, const float* t_rand, const float* u,
    uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes, float density_scale,
    uint32_t image_width, float* image, float* depth, float* semantics,
    void* ws, void* stream) {
  return render_impl(1, grid, table, false, packed_sigma_half, packed_color_half,
                     packed_sem_half, rays_o, rays_d, norms, aabb_host, min_near,
                     t_rand, u, N, T, t, n_classes, density_scale, image_width,
                     image, depth, semantics, ws, stream);
}

// fp16 nets AND fp16 table (what tiny-cuda-nn stores and computes with)
extern "C" int32_t ucsa_render_fwd_f16_h16(
    const ucsa_grid* grid, const void* table_half, const void* packed_sigma_half,
    const void* packed_color_half, const void* packed_sem_half,
    const float* rays_o, const float* rays_d, const float* norms,
    const float* aabb_host, float min_near, const float* t_rand, const float* u,
    uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes, float density_scale,
    uint32_t image_width, float* image, float* depth, float* semantics,
    void* ws, void* stream) {
  UCSA_CHECK_ARG(table_half, 1);
  return render_impl(1, grid, table_half, true, packed_sigma_half,
                     packed_color_half, packed_sem_half, rays_o, rays_d, norms,
                     aabb_host, min_near, t_rand, u, N, T, t, n_classes,
                     density_scale, image_width, image, depth, semantics, ws,
                     stream);
}

extern "C" int32_t ucsa_render_fwd_x3(
    const ucsa_grid* grid, const float* table, const void* packed_sigma_x3,
    const void* packed_color_x3, const void* packed_sem_x3, const float* rays_o,
    const float* rays_d, const float* norms, const float* aabb_host,
    float min_near, const float* t_rand, const float* u, uint32_t N, uint32_t T,
    uint32_t t, uint32_t n_classes, float density_scale, uint32_t image_width,
    float* image, float* depth, float* semantics, void* ws, void* stream) {
  return render_impl(2, grid, table, false, packed_sigma_x3, packed_color_x3,
                     packed_sem_x3, rays_o, rays_d, norms, aabb_host, min_near,
                     t_rand, u, N, T, t, n_classes, density_scale, image_width,
                     image, depth, semantics, ws, stream);
}

extern "C" int32_t ucsa_render_fwd_h2(
    const ucsa_grid* grid, const float* table, const void* packed_sigma_h2,
    const void* packed_color_h2, const void* packed_sem_h2, const float* rays_o,
    const float* rays_d, const float* norms, const float* aabb_host,
    float min_near, const float* t_rand, const float* u, uint32_t N, uint32_t T,
    uint32_t t, uint32_t n_classes, float density_scale, uint32_t image_width,
    float* image, float* depth, float* semantics, void* ws, void* stream) {
  return render_impl(3, grid, table, false, packed_sigma_h2, packed_color_h2,
                     packed_sem_h2, rays_o, rays_d, norms, aabb_host, min_near,
                     t_rand, u, N, T, t, n_classes, density_scale, image_width,
                     image, depth, semantics, ws, stream);
}

// ---------------------------------------------------------------------------
// ucsa_render_view: a whole batch of rays (a view) in chunks, software
// pipelined over two internal streams -- the density half of chunk k+1 runs
// while the shading half of chunk k does (two workspaces, events per chunk).
// The two halves bind on different things (the encoder on the per-CU L1's
// line look-ups, the shader on MFMA + VALU issue) but share the SIMDs' issue
// ports, so the overlap is partial: 14.4 -> 15.6 M rays/s on the cfg2 view
// (profiles/r04_coresident.txt; shading stream at the higher priority).
// Results are bit-identical to the serial loop.  One (dens, shade, events) set
// per caller stream, as in hashgrid_bwd.hip.
// ---------------------------------------------------------------------------
namespace {
struct RenderPipe {
  hipStream_t caller = nullptr;
  int dev = -1;
  hipStream_t dens = nullptr, shade = nullptr;
  hipEvent_t fork = nullptr, d_done[2] = {nullptr, nullptr},
             s_done[2] = {nullptr, nullptr}, join_d = nullptr, join_s = nullptr;
  bool ok = false, used = false;
};

RenderPipe* render_pipe(hipStream_t caller) {
  static std::mutex mu;
  static RenderPipe slots[32];
  std::lock_guard<std::mutex> lk(mu);
  int dev = -1;
  hipDevice_t sdev;
  if (caller && hipStreamGetDevice(caller, &sdev) == hipSuccess) dev = (int)sdev;
  else if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  RenderPipe* free_slot = nullptr;
  for (RenderPipe& p : slots) {
    if (p.used && p.caller == caller && p.dev == dev) return p.ok ? &p : nullptr;
    if (!p.used && !free_slot) free_slot = &p;
  }
  if (!free_slot) return nullptr;
  RenderPipe& p = *free_slot;
  p.used = true;
  p.caller = caller;
  p.dev = dev;
  int cur = -1;
  const bool sw = hipGetDevice(&cur) == hipSuccess && cur != dev;
  if (sw && hipSetDevice(dev) != hipSuccess) return nullptr;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  auto ev = [](hipEvent_t* e) {
    return hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  };
  p.ok = hipStreamCreateWithPriority(&p.dens, hipStreamNonBlocking, least) == hipSuccess &&
         hipStreamCreateWithPriority(&p.shade, hipStreamNonBlocking, greatest) == hipSuccess &&
         ev(&p.fork) && ev(&p.d_done[0]) && ev(&p.d_done[1]) && ev(&p.s_done[0]) &&
         ev(&p.s_done[1]) && ev(&p.join_d) && ev(&p.join_s);
  if (sw) (void)hipSetDevice(cur);
  return p.ok ? &p : nullptr;
}

#define UCSA_HIP_TRY(expr)                      \
  do {                                          \
    hipError_t e_ = (expr);                     \
    if (e_ != hipSuccess) return -(int32_t)e_;  \
  } while (0)
}  // namespace

extern "C" int32_t ucsa_render_view(
    uint32_t mode, const ucsa_grid* grid, const void* table,
    const void* packed_sigma, const void* packed_color, const void* packed_sem,
    const float* rays_o, const float* rays_d, const float* norms,
    const float* aabb_host, float min_near, const float* t_rand, const float* u,
    uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes, float density_scale,
    uint32_t image_width, uint32_t chunk, float* image, float* depth,
    float* semantics, void* ws0, void* ws1, void* stream) {
  UCSA_CHECK_ARG(mode <= 4, 0);
  UCSA_CHECK_ARG(grid, 1);
  UCSA_CHECK_ARG(chunk >= 1, 19);
  UCSA_CHECK_ARG(ws0, 23);
  if (N == 0) return 0;
  const int prec = mode == 3 ? 1 : (mode == 4 ? 3 : (int)mode);
  const bool table_half = mode == 3;
  if (image_width && (N % image_width != 0 || chunk % (8 * image_width) != 0))
    image_width = 0;  // not whole 8-row bands: ray-ordered gather (same results)
  auto part = [&](uint32_t head, uint32_t n, void* ws, void* s, uint32_t stage) {
    return render_impl(prec, grid, table, table_half, packed_sigma, packed_color,
                       packed_sem, rays_o + 3ull * head, rays_d + 3ull * head,
                       norms + head, aabb_host, min_near,
                       t_rand ? t_rand + (uint64_t)head * T : nullptr,
                       u ? u + (uint64_t)head * t : nullptr, n, T, t, n_classes,
                       density_scale, image_width, image + 3ull * head, depth + head,
                       semantics + (uint64_t)head * n_classes, ws, s, stage);
  };
  const uint32_t n_chunks = (N + chunk - 1) / chunk;
  RenderPipe* p = (ws1 && n_chunks >= 2) ? render_pipe((hipStream_t)stream) : nullptr;
  if (!p) {  // one chunk, no second workspace, or no streams: the serial loop
    for (uint32_t head = 0; head < N; head += chunk)
      UCSA_TRY(part(head, N - head < chunk ? N - head : chunk, ws0, stream, 3u));
    return 0;
  }
  void* ws[2] = {ws0, ws1};
  UCSA_HIP_TRY(hipEventRecord(p->fork, (hipStream_t)stream));
  UCSA_HIP_TRY(hipStreamWaitEvent(p->dens, p->fork, 0));
  UCSA_HIP_TRY(hipStreamWaitEvent(p->shade, p->fork, 0));
  int32_t rc = 0;
  uint32_t k = 0;
  for (uint32_t head = 0; head < N && rc == 0; head += chunk, ++k) {
    const uint32_t n = N - head < chunk ? N - head : chunk, b = k & 1u;
    // the shading half of chunk k-2 has finished reading this workspace
    if (k >= 2 && hipStreamWaitEvent(p->dens, p->s_done[b], 0) != hipSuccess) rc = -1;
    if (rc == 0) rc = part(head, n, ws[b], p->dens, 1u);
    if (rc == 0 && hipEventRecord(p->d_done[b], p->dens) != hipSuccess) rc = -1;
    if (rc == 0 && hipStreamWaitEvent(p->shade, p->d_done[b], 0) != hipSuccess) rc = -1;
    if (rc == 0) rc = part(head, n, ws[b], p->shade, 2u);
    if (rc == 0 && hipEventRecord(p->s_done[b], p->shade) != hipSuccess) rc = -1;
  }
  // always join: the caller's stream must not run ahead of what was enqueued
  const hipError_t j1 = hipEventRecord(p->join_d, p->dens);
  const hipError_t j2 = hipStreamWaitEvent((hipStream_t)stream, p->join_d, 0);
  const hipError_t j3 = hipEventRecord(p->join_s, p->shade);
  const hipError_t j4 = hipStreamWaitEvent((hipStream_t)stream, p->join_s, 0);
  if (rc == 0 && (j1 != hipSuccess || j2 != hipSuccess || j3 != hipSuccess || j4 != hipSuccess))
    rc = -(int32_t)hipErrorUnknown;
  return rc;
}

extern "C" int32_t ucsa_version(void) { return UCSA_VERSION; }

extern "C" const char* ucsa_error_string(int32_t code) {
  static thread_local char buf[96];
  if (code == 0) return "ok";
  if (code <= UCSA_ERR_ARG) {
    snprintf(buf, sizeof(buf), "invalid argument #%d", UCSA_ERR_ARG - code);
    return buf;
  }
  return hipGetErrorString((hipError_t)(-code));
}
