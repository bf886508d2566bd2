// Internal helpers shared by the HIP translation units of libucsa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ucsa_hip.h"

#define UCSA_WAVE 64

// The library's environment switches ("Environment variables"):
// every one of them is read ONCE per process, at the first call that asks for
// any (render.hip), never per call -- a launch shape / kernel choice, never a
// result.  ucsa_env_reload() (lab tools and tests only) re-reads them.
// Returns nullptr when unset or empty; asking for a name that is not in the
// table is a programming error (returns nullptr, asserts in debug builds).
const char* ucsa_getenv(const char* name);

// negative hipError_t on failure, 0 otherwise
static inline int32_t ucsa_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : -(int32_t)e;
}

#define UCSA_CHECK_ARG(cond, k) \
  do {                          \
    if (!(cond)) return UCSA_ERR_ARG - (k); \
  } while (0)

// hipGetLastError() is sticky per host thread and shared with every other
// user of the HIP runtime in the process (PyTorch): drop whatever is pending
// before a launch so ucsa_launch_status() reports OUR launch only.
#define UCSA_CLEAR_ERR() ((void)hipGetLastError())

static inline uint32_t ucsa_div_up(uint64_t a, uint32_t b) {
  return (uint32_t)((a + b - 1) / b);
}

// Device copy of the level table (passed by value as a kernel argument).
struct GridDev {
  uint32_t n_levels;
  float bound;
  float scale[UCSA_MAX_LEVELS];
  uint32_t res[UCSA_MAX_LEVELS];
  uint32_t entries[UCSA_MAX_LEVELS];
  uint32_t offset[UCSA_MAX_LEVELS];
  uint32_t hashed[UCSA_MAX_LEVELS];
};

static inline GridDev ucsa_grid_dev(const ucsa_grid* g) {
  GridDev d;
  d.n_levels = g->n_levels;
  d.bound = g->bound;
  for (uint32_t l = 0; l < UCSA_MAX_LEVELS; ++l) {
    const ucsa_grid_level& lv = g->level[l < g->n_levels ? l : 0];
    d.scale[l] = lv.scale;
    d.res[l] = lv.res;
    d.entries[l] = lv.entries;
    d.offset[l] = lv.offset;
    d.hashed[l] = lv.hashed;
  }
  return d;
}

struct Aabb {
  float lo[3];
  float hi[3];
};

static inline Aabb ucsa_aabb(const float* a) {
  Aabb b;
  for (int i = 0; i < 3; ++i) {
    b.lo[i] = a[i];
    b.hi[i] = a[i + 3];
  }
  return b;
}
