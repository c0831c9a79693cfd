// tile_inst.hip -- instantiates every kernel for ONE tile geometry (P, S).
// Built once per geometry with -DFC_P=.. -DFC_S=.. -DFC_NT=.. (see Makefile) so the
// geometries compile in parallel.
#include "fc_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

#ifndef FC_P
#error "compile with -DFC_P=<points per thread> -DFC_S=<lane split> -DFC_NT=<threads of the fused 1-D kernel>"
#endif

namespace fc {
namespace {

using GG = Geo<FC_P, FC_S>;
constexpr int kT = GG::T;
constexpr int kNSEQ_C = (8192 / kT) > 16 ? 16 : ((8192 / kT) < 2 ? 2 : (8192 / kT));
#ifndef FC_ROWS_DIV
#define FC_ROWS_DIV 2
#endif
constexpr int kNSEQ_R = (kNSEQ_C / FC_ROWS_DIV) < 1 ? 1 : (kNSEQ_C / FC_ROWS_DIV);
constexpr int kLSEQP = SeqLayout<GG>::LSEQP;
constexpr int kFusedMaxCib = (8 * kLSEQP * 8 <= 160 * 1024 && 8 * GG::TS <= 1024) ? 8 : 4;

// Opt in to > 64 KiB of dynamic LDS (gfx950: 160 KiB per workgroup).  Done once per kernel AND device with
// the full 160 KiB so nothing but the launch happens on later calls (launches may be under HIP-graph
// capture).  The attribute belongs to the function on the current device, so the "done" state is a bit per
// device ordinal; the flag is atomic because plans are shared between threads (a duplicate
// hipFuncSetAttribute from two racing first calls is harmless).
struct LdsOptIn {
  std::atomic<unsigned long long> mask{0};
};
template <class K>
hipError_t ensure_lds(K kernel, size_t lds, LdsOptIn* done) {
  if (lds <= 64 * 1024) return hipSuccess;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && (done->mask.load(std::memory_order_acquire) >> dev & 1ull)) return hipSuccess;
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess && tracked) done->mask.fetch_or(1ull << dev, std::memory_order_release);
  return e;
}

template <int CIB>
hipError_t launch_conv1d(const Conv1dArgs& a, int grid, size_t lds, hipStream_t st) {
  auto k = conv1d_fused_kernel<FC_P, FC_S, CIB, FC_NT>;
  static LdsOptIn done;
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
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(FC_NT), lds, st, a);
  return hipGetLastError();
}

hipError_t rows_r2c_dispatch(const RowsR2CArgs& a, hipStream_t st) {
  constexpr int NT = kNSEQ_R * GG::TS;
  auto k = rows_r2c_kernel<FC_P, FC_S, kNSEQ_R, NT>;
  const size_t lds = (size_t)kNSEQ_R * kLSEQP * sizeof(float2);
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  const long long nyb = (a.NY + 2 * kNSEQ_R - 1) / (2 * kNSEQ_R);
  const long long grid = (long long)a.NA * a.NC * a.nxt * nyb;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  auto b = a;
  b.d_nyb = make_fastdiv((unsigned)nyb); b.d_nxt = make_fastdiv((unsigned)a.nxt); b.d_nc = make_fastdiv((unsigned)a.NC);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, st, b);
  return hipGetLastError();
}

template <bool INV>
hipError_t c2c_dispatch(const C2CArgs& a, hipStream_t st) {
  constexpr int NT = kNSEQ_C * GG::TS;
  const size_t lds = (size_t)kNSEQ_C * kLSEQP * sizeof(float2);
  const long long nbb = (a.NB + kNSEQ_C - 1) / kNSEQ_C;
  const long long grid = (long long)a.NA * a.NC * nbb;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  static LdsOptIn done;
  C2CArgs b = a;
  b.d_nbb = make_fastdiv((unsigned)nbb); b.d_nc = make_fastdiv((unsigned)a.NC);
  if (INV) {
    auto k = c2c_inv_kernel<FC_P, FC_S, kNSEQ_C, NT>;
    hipError_t e = ensure_lds(k, lds, &done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, st, b);
  } else {
    auto k = c2c_fwd_kernel<FC_P, FC_S, kNSEQ_C, NT>;
    hipError_t e = ensure_lds(k, lds, &done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, st, b);
  }
  return hipGetLastError();
}

hipError_t rows_c2r_dispatch(const RowsC2RArgs& a, hipStream_t st) {
  constexpr int NT = kNSEQ_R * GG::TS;
  auto k = rows_c2r_kernel<FC_P, FC_S, kNSEQ_R, NT>;
  const size_t lds = (size_t)kNSEQ_R * kLSEQP * sizeof(float2);
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  const long long nyb = (a.NY + 2 * kNSEQ_R - 1) / (2 * kNSEQ_R);
  const long long grid = (long long)a.NA * a.NC * a.nxt * nyb;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  auto b = a;
  b.d_nyb = make_fastdiv((unsigned)nyb); b.d_nxt = make_fastdiv((unsigned)a.nxt); b.d_nc = make_fastdiv((unsigned)a.NC);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, st, b);
  return hipGetLastError();
}

// fused column pass: NB batch items per workgroup share the spectrum loads.  NB is the largest of
// {4, 2, 1} that fits the batch, 1024 threads, the 160 KiB of LDS and 32 spectrum values per mix
// thread, and still leaves two workgroups per CU's worth of grid.
template <int CIB, int NB>
constexpr bool fusedc_fits() {
  return CIB <= kFusedMaxCib && NB * CIB * GG::TS <= 1024 && NB * CIB <= 32 &&
         (size_t)NB * CIB * kLSEQP * sizeof(float2) <= 160 * 1024;
}
template <int CIB, int NB>
hipError_t launch_fusedc_nb(const FusedCArgs& a, hipStream_t st) {
  if constexpr (!fusedc_fits<CIB, NB>()) {
    return hipErrorInvalidValue;
  } else {
    constexpr int NT = NB * CIB * GG::TS;
    auto k = fusedc_kernel<FC_P, FC_S, CIB, NB, NT>;
    const size_t lds = (size_t)(a.accumulate ? 2 : 1) * NB * CIB * kLSEQP * sizeof(float2);
    static LdsOptIn done;
    hipError_t e = ensure_lds(k, lds, &done);
    if (e != hipSuccess) return e;
    const long long nbb = (a.B + NB - 1) / NB;
    const long long grid = nbb * a.ntiles * a.n_ochunks * a.G * ((a.ncol + 7) / 8) * 8;
    if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
    FusedCArgs b = a;
    b.d_nbb = make_fastdiv((unsigned)nbb); b.d_ncb = make_fastdiv((unsigned)((a.ncol + 7) / 8));
    b.d_g = make_fastdiv((unsigned)a.G); b.d_noc = make_fastdiv((unsigned)a.n_ochunks);
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds, st, b);
    return hipGetLastError();
  }
}
template <int CIB>
hipError_t launch_fusedc(const FusedCArgs& a, hipStream_t st) {
  auto ok = [&](int nb, bool fits) {
    if (!fits || nb > a.B) return false;
    const size_t lds = (size_t)(a.accumulate ? 2 : 1) * nb * CIB * kLSEQP * sizeof(float2);
    const long long grid = (long long)((a.B + nb - 1) / nb) * a.ntiles * a.n_ochunks * a.G * a.ncol;
    return lds <= 160 * 1024 && (nb == 1 || grid >= 512);
  };
  static const int force = getenv("FFTCONV_FUSEDC_NB") ? atoi(getenv("FFTCONV_FUSEDC_NB")) : 0;   // tuning knob
  if (force != 1) {
    if ((force == 0 || force == 4) && ok(4, fusedc_fits<CIB, 4>())) return launch_fusedc_nb<CIB, 4>(a, st);
    if ((force == 0 || force == 2) && ok(2, fusedc_fits<CIB, 2>())) return launch_fusedc_nb<CIB, 2>(a, st);
  }
  return launch_fusedc_nb<CIB, 1>(a, st);
}

hipError_t fusedc_dispatch(int cib, const FusedCArgs& a, hipStream_t st) {
  switch (cib) {
    case 2: return launch_fusedc<2>(a, st);
    case 4: return launch_fusedc<4>(a, st);
    case 8: return launch_fusedc<8>(a, st);
    default: return hipErrorInvalidValue;
  }
}

// persistent kernel: only for geometries whose twiddle table + NB*4 sequences fit in LDS
#if FC_P == 32 && FC_S == 2
constexpr int kPersNb0 = 1, kPersNb1 = 2;
#elif FC_P == 32 && FC_S == 1
constexpr int kPersNb0 = 2, kPersNb1 = 4;
#else
constexpr int kPersNb0 = 0, kPersNb1 = 0;
#endif
constexpr size_t pers_lds_bytes(int nb) { return ((size_t)FC_P * GG::N2 + (size_t)nb * 4 * GG::LSEQ) * sizeof(float2); }

// batch-sharing kernel builds: PHASES (dilation as phases) x DIAG (depthwise blocks); the plain one keeps its
// immediate offsets and is the only one the headline configuration runs
template <int NB, bool PHASES, bool DIAG, bool SEG = false, bool PH2 = false, bool STAMPS = false, bool PH4 = false>
hipError_t launch_pers_variant(const Conv1dPersArgs& a, int grid, hipStream_t st) {
  constexpr int NT = NB * 4 * GG::TS;
  const size_t lds = pers_lds_bytes(NB);
  auto k = conv1d_pers_kernel<FC_P, FC_S, 8, NB, NT, PHASES, 2, DIAG, SEG, PH2, STAMPS, PH4>;
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
  return hipGetLastError();
}

template <int NB>
hipError_t launch_pers(const Conv1dPersArgs& a, int grid, hipStream_t st) {
  if constexpr (NB == 0) {
    return hipErrorInvalidValue;
  } else {
    const bool ph = a.c.ph > 1, dg = a.c.diag != 0;
    if (a.c.segmented) return dg ? launch_pers_variant<NB, false, true, true>(a, grid, st)
                                 : launch_pers_variant<NB, false, false, true>(a, grid, st);
    if (ph && dg) return launch_pers_variant<NB, true, true>(a, grid, st);
    const bool stamped = a.c.stamps != nullptr;       // profiling builds exist for the plain and the phase kernels only
#if FC_S == 1
    if constexpr (NB == 4) {
      if (ph && a.c.ph2 == 2 && !stamped) return launch_pers_variant<NB, true, false, false, false, false, true>(a, grid, st);
    }
    if (ph && a.c.ph2) return stamped ? launch_pers_variant<NB, true, false, false, true, true>(a, grid, st)
                                      : launch_pers_variant<NB, true, false, false, true>(a, grid, st);
#endif
    if (ph) return stamped ? launch_pers_variant<NB, true, false, false, false, true>(a, grid, st)
                           : launch_pers_variant<NB, true, false>(a, grid, st);
    if (dg) return launch_pers_variant<NB, false, true>(a, grid, st);
    return stamped ? launch_pers_variant<NB, false, false, false, false, true>(a, grid, st)
                   : launch_pers_variant<NB, false, false>(a, grid, st);
  }
}

hipError_t pers_dispatch(int nb, const Conv1dPersArgs& a, int grid, hipStream_t st) {
  if (nb != 0 && nb == kPersNb0) return launch_pers<kPersNb0>(a, grid, st);
  if (nb != 0 && nb == kPersNb1) return launch_pers<kPersNb1>(a, grid, st);
  return hipErrorInvalidValue;
}

#if FC_P == 32 && (FC_S == 1 || FC_S == 2)
constexpr int kWideNb = 2;
hipError_t wide_dispatch(const Conv1dPersArgs& a, int grid, hipStream_t st) {
  constexpr int NT = kWideNb * 4 * GG::TS;
  auto k = conv1d_wide_kernel<FC_P, FC_S, kWideNb, NT>;
  const size_t lds = pers_lds_bytes(kWideNb);
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
  return hipGetLastError();
}
#else
constexpr int kWideNb = 0;
hipError_t wide_dispatch(const Conv1dPersArgs&, int, hipStream_t) { return hipErrorInvalidValue; }
#endif

#if FC_P == 32 && FC_S == 1
constexpr int kWgradNb = 2;
hipError_t wgrad_dispatch(const WGradArgs& a, int grid, hipStream_t st) {
  constexpr int NT = kWgradNb * 4 * GG::TS;
  auto k = wgrad1d_kernel<FC_P, FC_S, kWgradNb, NT>;
  const size_t lds = ((size_t)FC_P * GG::N2 + (size_t)kWgradNb * 4 * GG::LSEQ) * sizeof(float2);
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
  return hipGetLastError();
}
hipError_t wgrad_diag_dispatch(const WGradArgs& a, int grid, hipStream_t st) {
  constexpr int NT = 8 * GG::TS;
  auto k = wgrad1d_diag_kernel<FC_P, FC_S, NT>;
  const size_t lds = ((size_t)FC_P * GG::N2 + (size_t)8 * GG::LSEQ) * sizeof(float2);
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
  return hipGetLastError();
}
#else
constexpr int kWgradNb = 0;
#endif

#if FC_P == 32 && (FC_S == 1 || FC_S == 2)
// many-channel pipeline (1024 and 2048 tiles): channel pairs per workgroup in the two transform kernels
#ifndef FC_DENSE_NSEQ
#define FC_DENSE_NSEQ 8      // measured 64->64, B 8, L 16384, k 129 (1024 tile): forward 21.1 / inverse 33.6 us at 8 (256 threads,
#endif                       // two workgroups per CU), 24.0 / 35.3 us at 16 (512 threads, one per CU); 2048 tile: 8 x 64 threads
constexpr int kDenseNseq = FC_DENSE_NSEQ;
hipError_t dense_dispatch(int which, const DenseArgs& a, hipStream_t st) {
  constexpr int NT = kDenseNseq * GG::TS;
  const size_t lds_fft = (size_t)kDenseNseq * kLSEQP * sizeof(float2);
  if (which == 0 || which == 2) {
    const int nch = which == 0 ? a.Kc : a.Nc;
    const long long ncb = (nch / 2 + kDenseNseq - 1) / kDenseNseq;
    const long long grid = (long long)a.mcount * ncb * a.G;
    if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
    static LdsOptIn done_f, done_i;
    if (which == 0) {
      auto k = dense_fwd_kernel<FC_P, FC_S, kDenseNseq, NT>;
      hipError_t e = ensure_lds(k, lds_fft, &done_f);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds_fft, st, a);
    } else {
      auto k = dense_inv_kernel<FC_P, FC_S, kDenseNseq, NT>;
      hipError_t e = ensure_lds(k, lds_fft, &done_i);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NT), lds_fft, st, a);
    }
    return hipGetLastError();
  }
  // GEMM: the widest column block that the output channels fill (8 channels per wave), the smallest K chunk that
  // covers the input channels (or 64 and several chunks)
  const int nf = kT / 2 + 1;
  const int nct = a.Nc >= 64 ? 8 : (a.Nc >= 32 ? 4 : 2);
  const int k2n = a.Kc <= 16 ? 4 : (a.Kc <= 32 ? 8 : 16);
  const int mbr = (a.mcount <= 64 && nct >= 4) ? 32 : 128;                    // rows per unit
  const size_t lds = 2 * (size_t)mbr * std::max(8 * k2n + 4, 16 * nct + 4) * sizeof(float);   // two panels (each also holds an output block)
  const long long nmb = (a.mcount + mbr - 1) / mbr, nnb = (a.Nc + 8 * nct - 1) / (8 * nct);
  const long long units = nmb * nnb * nf * a.G;
  // persistent: one workgroup per CU (two with the small unit)
  const long long grid = std::min<long long>(units, (long long)std::max(1, a.cus) * (mbr == 32 ? 2 : 1));
  if (grid <= 0 || units > 0x7fffffffffffLL) return hipErrorInvalidValue;
#define FC_DENSE_GEMM(N, K, MBR)                                                 \
  if (nct == N && k2n == K && mbr == MBR) {                                      \
    static LdsOptIn done;                                                        \
    auto k = dense_gemm_kernel<N, K, MBR>;                                       \
    hipError_t e = ensure_lds(k, lds, &done);                                    \
    if (e != hipSuccess) return e;                                               \
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, st, a, nf);      \
    return hipGetLastError();                                                    \
  }
  FC_DENSE_GEMM(8, 16, 128) FC_DENSE_GEMM(8, 8, 128) FC_DENSE_GEMM(8, 4, 128)
  FC_DENSE_GEMM(4, 16, 128) FC_DENSE_GEMM(4, 8, 128) FC_DENSE_GEMM(4, 4, 128)
  FC_DENSE_GEMM(2, 16, 128) FC_DENSE_GEMM(2, 8, 128) FC_DENSE_GEMM(2, 4, 128)
  FC_DENSE_GEMM(8, 16, 32) FC_DENSE_GEMM(8, 8, 32) FC_DENSE_GEMM(8, 4, 32)
  FC_DENSE_GEMM(4, 16, 32) FC_DENSE_GEMM(4, 8, 32) FC_DENSE_GEMM(4, 4, 32)
#undef FC_DENSE_GEMM
  return hipErrorInvalidValue;
}
hipError_t dense_spec_dispatch(const DenseSpecArgs& a, hipStream_t st) {
  const size_t total = (size_t)a.G * (a.T / 2 + 1) * a.Kc * a.Nc;
  const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(dense_spec_kernel<0>, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}
#endif

#if FC_P == 8 && FC_S == 1
// plane-major 3-D pipeline: 64-point transforms on all three axes
hipError_t planes_fwd_dispatch(const PlaneFwdArgs& a, int n_images, hipStream_t st) {
  PlaneFwdArgs b = a;
  if (b.nxt < 1) b.nxt = 1;
  if (b.nyt < 1) b.nyt = 1;
  const long long ntile = (long long)b.nxt * b.nyt;
  const long long grid = (long long)n_images * a.NZ * ntile;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  b.d_nz = make_fastdiv((unsigned)a.NZ); b.d_nt = make_fastdiv((unsigned)ntile); b.d_nx = make_fastdiv((unsigned)b.nxt);
  hipLaunchKernelGGL(planes_fwd_kernel<kPlNT>, dim3((unsigned)grid), dim3(kPlNT), 0, st, b);
  return hipGetLastError();
}
hipError_t planes_inv_dispatch(const PlaneInvArgs& a, int n_images, hipStream_t st) {
  PlaneInvArgs b = a;
  if (b.nxt < 1) b.nxt = 1;
  if (b.nyt < 1) b.nyt = 1;
  if (b.nxt * b.nyt == 1) { b.Vx = a.NVx > 0 ? a.NVx : 1; b.Vy = a.NVy > 0 ? a.NVy : 1; }    // (one tile: the whole window)
  const long long ntile = (long long)b.nxt * b.nyt;
  const long long grid = (long long)n_images * a.NZo * ntile;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  b.d_nz = make_fastdiv((unsigned)a.NZo); b.d_nt = make_fastdiv((unsigned)ntile); b.d_nx = make_fastdiv((unsigned)b.nxt);
  hipLaunchKernelGGL(planes_inv_kernel<kPlNT>, dim3((unsigned)grid), dim3(kPlNT), 0, st, b);
  return hipGetLastError();
}
template <int NB, bool STAMPS, int DIAG, int NCOLC>
hipError_t launch_colz_n(const ColZArgs& a, hipStream_t st) {
  constexpr int RING = NB == 4 ? 3 : 2;
  auto k = colz_kernel<NB, RING, STAMPS, DIAG, NCOLC>;
  const size_t lds = colz_lds_bytes(NB);
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  if (a.ncol < 16 || a.ncol % 16 || (NCOLC > 0 && a.ncol != NCOLC)) return hipErrorInvalidValue;
  const long long nbp = (a.B + NB - 1) / NB;
  const long long per = (a.ncol / 16 + 7) / 8;                  // column blocks per XCD
  const long long grid = nbp * a.ntiles * a.n_ochunks * a.G * per * 8;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  ColZArgs b = a;
  b.d_nbp = make_fastdiv((unsigned)nbp); b.d_ntiles = make_fastdiv((unsigned)a.ntiles);
  b.d_per = make_fastdiv((unsigned)per); b.d_g = make_fastdiv((unsigned)a.G);
  if (b.hcol <= 0) b.hcol = a.ncol;
  if (b.hcol % 16 || a.ncol % b.hcol) return hipErrorInvalidValue;
  b.d_hcol = make_fastdiv((unsigned)b.hcol);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(NB * 128), lds, st, b);
  return hipGetLastError();
}
// 3-D pipeline: the column count is the compile-time kPlCols; 2-D pipeline (ncol = Tx/2): the run-time build (no stamps / diag)
template <int NB, bool STAMPS = false, int DIAG = 0>
hipError_t launch_colz(const ColZArgs& a, hipStream_t st) {
  if (a.ncol == kPlCols) return launch_colz_n<NB, STAMPS, DIAG, kPlCols>(a, st);
  if constexpr (DIAG == 0) return launch_colz_n<NB, STAMPS, 0, 0>(a, st);
  else return hipErrorInvalidValue;
}
template <bool STAMPS = false>
hipError_t launch_colz2(const ColZArgs& a, hipStream_t st) {
  auto k = colz2_kernel<5, STAMPS>;
  const size_t lds = colz2_lds_bytes();
  static LdsOptIn done;
  hipError_t e = ensure_lds(k, lds, &done);
  if (e != hipSuccess) return e;
  const long long nbp = (a.B + 1) / 2;
  const long long grid = nbp * a.ntiles * a.n_ochunks * a.G * (kPlCols / 16);
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, st, a);
  return hipGetLastError();
}
// batch items per workgroup: 2 (two workgroups per CU).  4 (one 512-thread workgroup per CU, half the spectrum traffic,
// FFTCONV_COLZ_NB=4) measured slower at cfgC: mix 28.4 against 21.2 us per workgroup -- its inputs no longer stay in
// registers across the output channels and the mix waits on LDS round trips instead (profiles/r03_experiments.md)
hipError_t colz_dispatch(const ColZArgs& a, hipStream_t st) {
  static const int force = getenv("FFTCONV_COLZ_NB") ? atoi(getenv("FFTCONV_COLZ_NB")) : 0;
  // two threads per sequence (colz2_kernel) for batches of two and more; FFTCONV_COLZ_NB = 1 / 2 / 4 forces the
  // thread-per-sequence builds (A/B runs, tests)
  // (the pair build measured no faster at cfgC -- 112.9 against 108.3 us per forward, profiles/r03_experiments.md block 5 --
  // and stays behind FFTCONV_COLZ_PAIRS=1)
  static const int pairs = getenv("FFTCONV_COLZ_PAIRS") ? atoi(getenv("FFTCONV_COLZ_PAIRS")) : 0;
  if (pairs && force == 0 && a.B >= 2) return a.stamps ? launch_colz2<true>(a, st) : launch_colz2<false>(a, st);
  const int nb = (force == 1 || force == 2 || force == 4) ? force : (a.B >= 2 ? 2 : 1);
  static const int diag = getenv("FFTCONV_COLZ_DIAG") ? atoi(getenv("FFTCONV_COLZ_DIAG")) : 0;    // experiments (wrong results)
  if (a.stamps && nb == 2 && diag == 1) return launch_colz<2, true, 1>(a, st);
  if (a.stamps && nb == 2 && diag == 2) return launch_colz<2, true, 2>(a, st);
  if (a.stamps) return nb == 4 ? launch_colz<4, true>(a, st) : (nb == 2 ? launch_colz<2, true>(a, st) : launch_colz<1, true>(a, st));
  return nb == 4 ? launch_colz<4>(a, st) : (nb == 2 ? launch_colz<2>(a, st) : launch_colz<1>(a, st));
}
#endif

}  // namespace

#define FC_CAT_(a, b, c, d) a##b##c##d
#define FC_CAT(a, b, c, d) FC_CAT_(a, b, c, d)
const TileImpl* FC_CAT(get_tile_P, FC_P, _S, FC_S)() {
  static const TileImpl impl = {kT, FC_P, FC_S, FC_NT, GG::LSEQ, kLSEQP, kNSEQ_C, kNSEQ_R,
                                conv1d_dispatch, spec1d_dispatch, rows_r2c_dispatch, c2c_dispatch<false>,
                                c2c_dispatch<true>, rows_c2r_dispatch, fusedc_dispatch, kFusedMaxCib,
                                pers_dispatch, {kPersNb0, kPersNb1},
                                {kPersNb0 ? pers_lds_bytes(kPersNb0) : 0, kPersNb1 ? pers_lds_bytes(kPersNb1) : 0},
                                {kPersNb0 * 4 * GG::TS, kPersNb1 * 4 * GG::TS},
                                wide_dispatch, kWideNb, kWideNb ? pers_lds_bytes(kWideNb) : 0,
#if FC_P == 32 && FC_S == 1
                                wgrad_dispatch, wgrad_diag_dispatch,
#else
                                nullptr, nullptr,
#endif
                                kWgradNb,
#if FC_P == 32 && (FC_S == 1 || FC_S == 2)
                                dense_dispatch, dense_spec_dispatch,
#else
                                nullptr, nullptr,
#endif
#if FC_P == 8 && FC_S == 1
                                planes_fwd_dispatch, colz_dispatch, planes_inv_dispatch
#else
                                nullptr, nullptr, nullptr
#endif
  };
  return &impl;
}

}  // namespace fc
