// fc_internal.h -- glue between the C ABI (fc_api.cpp) and the per-tile kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include "conv1d_fused.hpp"
#include "spectrum1d.hpp"

namespace fc {

struct TileImpl {
  int T, P, S, NT;
  int lseq;     // complex LDS slots per sequence
  hipError_t (*conv1d)(int cib, const Conv1dArgs& a, int grid, size_t lds, hipStream_t st);
  hipError_t (*spec1d)(const Spec1dArgs& a, int grid, size_t lds, hipStream_t st);
};

#define FC_DECLARE_TILE(P, S) const TileImpl* get_tile_P##P##_S##S();
FC_DECLARE_TILE(8, 1)
FC_DECLARE_TILE(16, 1)
FC_DECLARE_TILE(16, 2)
FC_DECLARE_TILE(32, 1)
FC_DECLARE_TILE(32, 2)
FC_DECLARE_TILE(32, 4)

}  // namespace fc
