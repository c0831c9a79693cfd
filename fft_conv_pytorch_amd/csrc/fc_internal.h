// fc_internal.h -- glue between the C ABI (fc_api.cpp) and the per-tile kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include "conv1d_fused.hpp"
#include "conv1d_pers.hpp"
#include "conv1d_wide.hpp"
#include "nd_passes.hpp"
#include "dense1d.hpp"
#include "spectrum1d.hpp"
#include "wgrad1d.hpp"
#include "planes3d.hpp"

namespace fc {

struct TileImpl {
  int T, P, S, NT;
  int lseq;     // complex LDS slots per sequence (fused 1-D kernel)
  int lseqp;    // sequence stride used by the N-d passes
  int nseq_c;   // sequences per workgroup of the c2c passes (rows passes: 2*nseq_r rows)
  int nseq_r;
  hipError_t (*conv1d)(int cib, const Conv1dArgs& a, int grid, size_t lds, hipStream_t st);
  hipError_t (*spec1d)(const Spec1dArgs& a, int grid, size_t lds, hipStream_t st);
  hipError_t (*rows_r2c)(const RowsR2CArgs& a, hipStream_t st);
  hipError_t (*c2c_fwd)(const C2CArgs& a, hipStream_t st);
  hipError_t (*c2c_inv)(const C2CArgs& a, hipStream_t st);
  hipError_t (*rows_c2r)(const RowsC2RArgs& a, hipStream_t st);
  hipError_t (*fusedc)(int cib, const FusedCArgs& a, hipStream_t st);   // hipErrorInvalidValue if cib unsupported
  int fusedc_max_cib;
  // persistent fused 1-D kernel (CIB = 8): nb = batch items per workgroup; returns hipErrorInvalidValue if not built
  hipError_t (*conv1d_pers)(int nb, const Conv1dPersArgs& a, int grid, hipStream_t st);
  int pers_nb[2];        // supported nb values (0 = none)
  size_t pers_lds[2];    // LDS bytes for each
  int pers_nt[2];
  // batch-sharing kernel for > 8 input channels per group (conv1d_wide.hpp); wide_nb = 0 when not built
  hipError_t (*conv1d_wide)(const Conv1dPersArgs& a, int grid, hipStream_t st);
  int wide_nb;
  size_t wide_lds;
  // 1-D weight gradient (wgrad1d.hpp), built for the 1024-point tile only (else null)
  hipError_t (*wgrad1d)(const WGradArgs& a, int grid, hipStream_t st);
  hipError_t (*wgrad1d_diag)(const WGradArgs& a, int grid, hipStream_t st);   // depthwise blocks of 8 channels
  int wgrad_nb;          // items per iteration of that kernel
  // many-channel 1-D pipeline (dense1d.hpp), built for the 1024-point tile only (else null): which = 0 forward
  // transforms, 1 per-bin GEMM (MFMA), 2 inverse transforms
  hipError_t (*dense)(int which, const DenseArgs& a, hipStream_t st);
  hipError_t (*dense_spec)(const DenseSpecArgs& a, hipStream_t st);
  // plane-major 3-D pipeline (planes3d.hpp), built for the 64-point tile only (else null)
  hipError_t (*planes_fwd)(const PlaneFwdArgs& a, int n_images, hipStream_t st);
  hipError_t (*colz)(const ColZArgs& a, hipStream_t st);
  hipError_t (*planes_inv)(const PlaneInvArgs& a, int n_images, hipStream_t st);
};

#define FC_DECLARE_TILE(P, S) const TileImpl* get_tile_P##P##_S##S();
FC_DECLARE_TILE(8, 1)
FC_DECLARE_TILE(8, 2)
FC_DECLARE_TILE(16, 1)
FC_DECLARE_TILE(16, 2)
FC_DECLARE_TILE(32, 1)
FC_DECLARE_TILE(32, 2)
FC_DECLARE_TILE(32, 4)

}  // namespace fc
