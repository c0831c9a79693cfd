// tile_inst.hip -- instantiates every kernel for ONE tile geometry (P, S).
// Built once per geometry with -DFC_P=.. -DFC_S=.. -DFC_NT=.. (see Makefile) so the
// geometries compile in parallel.
#include "fc_internal.h"

#ifndef FC_P
#error "compile with -DFC_P=<points per thread> -DFC_S=<lane split> -DFC_NT=<threads>"
#endif

namespace fc {
namespace {

// Opt in to > 64 KiB of dynamic LDS (gfx950: 160 KiB per workgroup).  Done once per
// kernel with the full 160 KiB so nothing but the launch happens on later calls
// (launches may be under HIP-graph capture).
template <class K>
hipError_t ensure_lds(K kernel, size_t lds, bool* done) {
  if (lds <= 64 * 1024 || *done) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) *done = true;
  return e;
}

template <int CIB>
hipError_t launch_conv1d(const Conv1dArgs& a, int grid, size_t lds, hipStream_t st) {
  auto k = conv1d_fused_kernel<FC_P, FC_S, CIB, FC_NT>;
  static bool done = false;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(FC_NT), lds, st, a);
  return hipGetLastError();
}

hipError_t conv1d_dispatch(int cib, const Conv1dArgs& a, int grid, size_t lds, hipStream_t st) {
  switch (cib) {
    case 2: return launch_conv1d<2>(a, grid, lds, st);
    case 4: return launch_conv1d<4>(a, grid, lds, st);
    case 8: return launch_conv1d<8>(a, grid, lds, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t spec1d_dispatch(const Spec1dArgs& a, int grid, size_t lds, hipStream_t st) {
  auto k = spectrum1d_kernel<FC_P, FC_S, FC_NT>;
  static bool done = false;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(FC_NT), lds, st, a);
  return hipGetLastError();
}

}  // namespace

#define FC_CAT_(a, b, c, d) a##b##c##d
#define FC_CAT(a, b, c, d) FC_CAT_(a, b, c, d)
const TileImpl* FC_CAT(get_tile_P, FC_P, _S, FC_S)() {
  static const TileImpl impl = {Geo<FC_P, FC_S>::T, FC_P, FC_S, FC_NT, Geo<FC_P, FC_S>::LSEQ,
                                conv1d_dispatch, spec1d_dispatch};
  return &impl;
}

}  // namespace fc
