// The differentiable render of a training step as TWO calls (SURVEY 8b:
// ucsa_render_fused_fwd / ucsa_render_fused_bwd): rows a2-a9 of SURVEY 8a for a
// batch of rays, forward with the activations the backward needs, and the
// backward down to the four flat parameter gradients.
//
// Reference: SemanticNeRFRenderer.run (nr4seg/nerf/renderer_semantics.py:123-299)
// under autograd, as JointTrainLightningNet.training_step_nerf drives it
// (nr4seg/lightning/joint_train_lightning_net.py:473-513).
//
// Arithmetic: the default training mode of the Python host (train_precision
// "bf16x3", bwd_precision "bf16x2"): forward nets as three-term bf16 splits
// (fp32-grade), backward contractions as two-term splits, 8-byte packed records
// in the hash-grid backward.  These two functions only SEQUENCE the per-stage
// entry points of this library on `stream` -- the same launches, in the same
// order, with the same arguments as nerf/network_tcnn_semantics.py's _RenderFn
// issues one by one, so results are bit-identical to that path
// (tests/test_gpu_backward.py::test_fused_train_calls_*).  No allocation, no
// synchronisation; every buffer is the caller's.
#include "ucsa_common.h"

#define UCSA_TRY(expr)          \
  do {                          \
    int32_t rc_ = (expr);       \
    if (rc_ != 0) return rc_;   \
  } while (0)

int32_t ucsa_reduce_partials_chained(
    uint32_t count, const float* const* partials, const uint32_t* n_parts,
    const float* const* partials2, const uint32_t* n_parts2,
    const uint32_t* n_params, float* const* grads, int32_t accumulate, void* stream);

namespace {

inline uint64_t al256(uint64_t b) { return (b + 255ull) & ~255ull; }

struct Carver {
  char* p;
  uint64_t used = 0;
  uint64_t n = 0;
  explicit Carver(void* base) : p((char*)base) {}
  template <typename T>
  T* take(uint64_t count) {
    T* r = p ? (T*)(p + used) : nullptr;
    // + an odd number of 256-byte lines: the arrays have power-of-two sizes, and
    // arrays a power of two apart map element i of each to the same HBM channel
    used += al256(count * sizeof(T)) + 256ull * (2 * (++n) + 1);
    return r;
  }
};

inline uint32_t sem_params(uint32_t n_classes) {
  return 1024u + 1024u * ((n_classes + 15u) / 16u);
}

struct FwdWs {
  float *nears, *fars;
  void* cmp;
  uint64_t bytes;
};
FwdWs fwd_ws(void* base, uint32_t N, uint32_t T, uint32_t t) {
  Carver c(base);
  FwdWs w;
  w.nears = c.take<float>(N);
  w.fars = c.take<float>(N);
  w.cmp = c.take<char>(ucsa_composite_infer_workspace_bytes(N, T, t));
  w.bytes = c.used;
  return w;
}

struct BwdWs {
  float *G, *d_h_c, *d_h_f, *pc, *ps, *d_feat_c, *d_feat_f, *psig, *psig_f;
  void* bins;
  uint32_t parts_c, parts_sc, parts_sf;
  uint64_t bytes;
};
BwdWs bwd_ws(void* base, uint32_t N, uint32_t T, uint32_t t, uint32_t C, uint32_t L) {
  Carver c(base);
  BwdWs w;
  const uint64_t Mc = (uint64_t)N * T, Mf = (uint64_t)N * t;
  w.parts_c = ucsa_composite_bwd_parts(N);
  w.parts_sc = ucsa_sigma_mlp_bwd_parts((uint32_t)Mc);
  w.parts_sf = t ? ucsa_sigma_mlp_bwd_parts((uint32_t)Mf) : 0u;
  w.G = c.take<float>(Mc + Mf);
  w.d_h_c = c.take<float>(Mc * 16);
  w.d_h_f = c.take<float>(Mf * 16);
  w.pc = c.take<float>((uint64_t)w.parts_c * 7168u);
  w.ps = c.take<float>((uint64_t)w.parts_c * sem_params(C));
  w.d_feat_c = c.take<float>((uint64_t)L * Mc * 2);
  w.d_feat_f = c.take<float>((uint64_t)L * Mf * 2);
  w.psig = c.take<float>((uint64_t)w.parts_sc * 3072u);
  w.psig_f = c.take<float>((uint64_t)w.parts_sf * 3072u);
  w.bins = c.take<char>(ucsa_hashgrid_bwd_workspace_bytes(N, T + t, L));
  w.bytes = c.used;
  return w;
}

}  // namespace

extern "C" uint64_t ucsa_render_fused_fwd_workspace_bytes(uint32_t N, uint32_t T,
                                                          uint32_t t) {
  return fwd_ws(nullptr, N, T, t).bytes;
}

extern "C" int32_t ucsa_render_fused_fwd(
    const ucsa_grid* grid, const float* table, const ucsa_train_packs* packs,
    const float* rays_o, const float* rays_d, const float* norms,
    const float* aabb_host, float min_near, const float* t_rand, const float* u,
    uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes, float density_scale,
    const ucsa_train_buffers* out, float* image, float* depth, float* semantics,
    void* workspace, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(table, 1);
  UCSA_CHECK_ARG(packs && ((packs->sigma_x3 && packs->color_x3 && packs->sem_x3) ||
                           (packs->sigma_h2 && packs->color_h2 && packs->sem_h2)), 2);
  UCSA_CHECK_ARG(rays_o && rays_d && norms, 3);
  UCSA_CHECK_ARG(aabb_host, 6);
  UCSA_CHECK_ARG(T >= 1 && (uint64_t)N * (T + t) < 0x80000000ull, 11);
  UCSA_CHECK_ARG(t == 0 || u, 9);
  UCSA_CHECK_ARG(out && out->z_c && out->feat_c && out->h_c && out->sigma_c &&
                     out->src && out->weights, 15);
  UCSA_CHECK_ARG(t == 0 || (out->z_f && out->feat_f && out->h_f && out->sigma_f), 15);
  UCSA_CHECK_ARG(image && depth && semantics, 16);
  if (N == 0) return 0;
  UCSA_CHECK_ARG(workspace, 19);
  const FwdWs w = fwd_ws(workspace, N, T, t);
  const uint32_t L = grid->n_levels;
  UCSA_TRY(ucsa_near_far_from_aabb(rays_o, rays_d, aabb_host, N, min_near, w.nears,
                                   w.fars, stream));
  UCSA_TRY(ucsa_sample_coarse(w.nears, w.fars, t_rand, N, T, out->z_c, stream));
  UCSA_TRY(ucsa_hashgrid_encode_rays(grid, table, rays_o, rays_d, out->z_c, aabb_host,
                                     N, T, out->feat_c, stream));
  const bool h2 = packs->sigma_h2 && packs->color_h2 && packs->sem_h2;
  if (h2)
    UCSA_TRY(ucsa_sigma_mlp_fwd_h2(out->feat_c, packs->sigma_h2, N * T, L, out->h_c,
                                   out->sigma_c, stream));
  else
    UCSA_TRY(ucsa_sigma_mlp_fwd_x3(out->feat_c, packs->sigma_x3, N * T, L, out->h_c,
                                   out->sigma_c, stream));
  if (t > 0) {
    UCSA_TRY(ucsa_resample(out->z_c, out->sigma_c, u, N, T, t, density_scale, out->z_f,
                           stream));
    UCSA_TRY(ucsa_hashgrid_encode_rays(grid, table, rays_o, rays_d, out->z_f,
                                       aabb_host, N, t, out->feat_f, stream));
    if (h2)
      UCSA_TRY(ucsa_sigma_mlp_fwd_h2(out->feat_f, packs->sigma_h2, N * t, L, out->h_f,
                                     out->sigma_f, stream));
    else
      UCSA_TRY(ucsa_sigma_mlp_fwd_x3(out->feat_f, packs->sigma_x3, N * t, L, out->h_f,
                                     out->sigma_f, stream));
  }
  if (h2)
    return ucsa_composite_train_fwd_h2(
        rays_d, norms, out->z_c, out->sigma_c, out->h_c, t ? out->z_f : nullptr,
        t ? out->sigma_f : nullptr, t ? out->h_f : nullptr, packs->color_h2,
        packs->sem_h2, N, T, t, n_classes, density_scale, image, depth, semantics,
        out->src, out->weights, w.cmp, stream);
  return ucsa_composite_train_fwd_x3(
      rays_d, norms, out->z_c, out->sigma_c, out->h_c, t ? out->z_f : nullptr,
      t ? out->sigma_f : nullptr, t ? out->h_f : nullptr, packs->color_x3,
      packs->sem_x3, N, T, t, n_classes, density_scale, image, depth, semantics,
      out->src, out->weights, w.cmp, stream);
}

extern "C" uint64_t ucsa_render_fused_bwd_workspace_bytes(uint32_t N, uint32_t T,
                                                          uint32_t t,
                                                          uint32_t n_classes,
                                                          uint32_t n_levels) {
  return bwd_ws(nullptr, N, T, t, n_classes, n_levels).bytes;
}

extern "C" int32_t ucsa_render_fused_bwd(
    const ucsa_grid* grid, const ucsa_train_packs* packs, const float* rays_o,
    const float* rays_d, const float* norms, const float* aabb_host,
    const ucsa_train_buffers* saved, const float* d_image, const float* d_depth,
    const float* d_sem, uint32_t N, uint32_t T, uint32_t t, uint32_t n_classes,
    float density_scale, float* grad_table, float* grad_sigma, float* grad_color,
    float* grad_sem, void* workspace, void* stream) {
  UCSA_CHECK_ARG(grid && grid->n_features == 2 && grid->n_levels > 0 &&
                     grid->n_levels <= UCSA_MAX_LEVELS, 0);
  UCSA_CHECK_ARG(packs && packs->sigma_x3 && packs->color_x3 && packs->sem_x3 &&
                     packs->sigma_t_x3 && packs->color_t_x3 && packs->sem_t_x3, 1);
  UCSA_CHECK_ARG(rays_o && rays_d && norms, 2);
  UCSA_CHECK_ARG(aabb_host, 5);
  UCSA_CHECK_ARG(saved && saved->z_c && saved->feat_c && saved->h_c &&
                     saved->sigma_c && saved->src && saved->weights, 6);
  UCSA_CHECK_ARG(t == 0 || (saved->z_f && saved->feat_f && saved->h_f && saved->sigma_f), 6);
  UCSA_CHECK_ARG(d_image && d_depth && d_sem, 7);
  UCSA_CHECK_ARG(T >= 1 && (uint64_t)N * (T + t) < 0x80000000ull, 11);
  UCSA_CHECK_ARG(grad_table && grad_sigma && grad_color && grad_sem, 15);
  if (N == 0) return 0;
  UCSA_CHECK_ARG(workspace, 19);
  const uint32_t L = grid->n_levels;
  const BwdWs w = bwd_ws(workspace, N, T, t, n_classes, L);
  UCSA_TRY(ucsa_composite_bwd_x2(
      rays_d, norms, saved->z_c, saved->sigma_c, saved->h_c, t ? saved->z_f : nullptr,
      t ? saved->sigma_f : nullptr, t ? saved->h_f : nullptr, saved->src,
      saved->weights, packs->color_x3, packs->sem_x3, packs->color_t_x3,
      packs->sem_t_x3, d_image, d_depth, d_sem, N, T, t, n_classes, density_scale,
      w.G, w.d_h_c, t ? w.d_h_f : nullptr, w.pc, w.ps, stream));
  UCSA_TRY(ucsa_sigma_mlp_bwd_x2(saved->feat_c, w.d_h_c, packs->sigma_x3,
                                 packs->sigma_t_x3, N * T, L, w.d_feat_c, w.psig,
                                 stream));
  if (t > 0)
    UCSA_TRY(ucsa_sigma_mlp_bwd_x2(saved->feat_f, w.d_h_f, packs->sigma_x3,
                                   packs->sigma_t_x3, N * t, L, w.d_feat_f, w.psig_f,
                                   stream));
  {
    // the three nets' partial gradients in ONE launch (they were four of ~18 us):
    // per parameter the same additions in the same order as
    // ucsa_reduce_partials(colour), (semantics), (sigma coarse), (sigma fine, accumulate)
    const float* p1[3] = {w.pc, w.ps, w.psig};
    const float* p2[3] = {nullptr, nullptr, t ? w.psig_f : nullptr};
    const uint32_t n1[3] = {w.parts_c, w.parts_c, w.parts_sc};
    const uint32_t n2[3] = {0u, 0u, w.parts_sf};
    const uint32_t np[3] = {7168u, sem_params(n_classes), 3072u};
    float* gr[3] = {grad_color, grad_sem, grad_sigma};
    UCSA_TRY(ucsa_reduce_partials_chained(3, p1, n1, p2, n2, np, gr, 0, stream));
  }
  if (t == 0)
    return ucsa_hashgrid_bwd_rays_p64(grid, rays_o, rays_d, saved->z_c, aabb_host, N,
                                      T, w.d_feat_c, grad_table, w.bins, stream);
  return ucsa_hashgrid_bwd_rays_merged_p64(
      grid, rays_o, rays_d, saved->z_c, saved->z_f, saved->src, aabb_host, N, T, t,
      w.d_feat_c, w.d_feat_f, grad_table, w.bins, stream);
}
