// fc_api.cpp -- C ABI of libfftconv_amd.so (see include/fftconv_amd.h).
//
// Host-side planning for the forward FFT convolution: hyper-parameter checks,
// tile choice, twiddle tables, launch geometry.  The arithmetic of
// /root/reference/fft_conv_pytorch/functional.py:44-47,66,76-82 that runs on the
// host lives here; everything else is in the HIP kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "direct_f64.h"
#include "fft_f64.h"
#include "fc_internal.h"
#include "fftconv_amd.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define FC_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) return fail(FC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Plan creation allocates and uploads device tables (twiddles, work lists) with synchronous calls: it must run OUTSIDE
// stream capture -- run the call once before capturing; every later call of the same shape only launches kernels on the
// caller's stream and is capture-safe (tests/test_gpu_round3.py).  A capture error gets that hint instead of a bare code.
#define FC_HIP_SETUP(expr)                                                               \
  do {                                                                                   \
    hipError_t e_ = (expr);                                                              \
    if (e_ != hipSuccess) {                                                              \
      const char* name_ = hipGetErrorName(e_);                                           \
      (void)hipGetLastError();                                                           \
      if (name_ && std::strstr(name_, "Capture"))                                        \
        return fail(FC_ERR_HIP, "%s: %s -- plans cannot be created while a stream is being captured: run this call once "  \
                    "before the capture (plan creation allocates device tables; later calls only launch kernels)", #expr, \
                    hipGetErrorString(e_));                                              \
      return fail(FC_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                   \
    }                                                                                    \
  } while (0)

const fc::TileImpl* const* all_tiles(int* n) {
  static const fc::TileImpl* tiles[] = {fc::get_tile_P8_S1(),  fc::get_tile_P8_S2(),  fc::get_tile_P16_S1(), fc::get_tile_P16_S2(),
                                        fc::get_tile_P32_S1(), fc::get_tile_P32_S2(), fc::get_tile_P32_S4()};   // T = 64 .. 4096
  *n = (int)(sizeof tiles / sizeof tiles[0]);
  return tiles;
}

const fc::TileImpl* find_tile(int T) {
  int n;
  auto t = all_tiles(&n);
  for (int i = 0; i < n; ++i)
    if (t[i]->T == T) return t[i];
  return nullptr;
}

// Device twiddle tables, shared by every plan of the same tile geometry and device.
struct Twiddles {
  fc::f2* twA = nullptr;  // [P][N2]  exp(-2 pi i n2 k1 / T)
  fc::f2* twB = nullptr;  // [S][P]   exp(-2 pi i r k / N2)
};
std::mutex g_tw_mutex;
std::map<std::pair<int, int>, Twiddles> g_tw;  // (device, T)

int get_twiddles(const fc::TileImpl* t, Twiddles* out) {
  int dev = 0;
  FC_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto key = std::make_pair(dev, t->T);
  auto it = g_tw.find(key);
  if (it != g_tw.end()) { *out = it->second; return FC_OK; }
  const int P = t->P, S = t->S, N2 = P * S, T = t->T;
  std::vector<fc::f2> a((size_t)P * N2), b((size_t)S * P);
  const double tau = 6.283185307179586476925286766559;
  for (int k1 = 0; k1 < P; ++k1)
    for (int n2 = 0; n2 < N2; ++n2) {
      const double ang = -tau * (double)((long long)k1 * n2 % T) / (double)T;
      a[(size_t)k1 * N2 + n2] = fc::f2{(float)std::cos(ang), (float)std::sin(ang)};
    }
  for (int r = 0; r < S; ++r)
    for (int k = 0; k < P; ++k) {
      const double ang = -tau * (double)(r * k) / (double)N2;
      b[(size_t)r * P + k] = fc::f2{(float)std::cos(ang), (float)std::sin(ang)};
    }
  Twiddles tw;
  FC_HIP_SETUP(hipMalloc(&tw.twA, a.size() * sizeof(fc::f2)));
  FC_HIP_SETUP(hipMalloc(&tw.twB, b.size() * sizeof(fc::f2)));
  FC_HIP_SETUP(hipMemcpy(tw.twA, a.data(), a.size() * sizeof(fc::f2), hipMemcpyHostToDevice));
  FC_HIP_SETUP(hipMemcpy(tw.twB, b.data(), b.size() * sizeof(fc::f2), hipMemcpyHostToDevice));
  g_tw[key] = tw;
  *out = tw;
  return FC_OK;
}

// Twiddle tables of an already prepared (device, tile) pair: never allocates (hot-call side of get_twiddles).
int find_twiddles(const fc::TileImpl* t, Twiddles* out) {
  int dev = 0;
  FC_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_tw_mutex);
  auto it = g_tw.find(std::make_pair(dev, t->T));
  if (it == g_tw.end())
    return fail(FC_ERR_INVALID, "device tables for the %d-point tile are not prepared on device %d "
                "(call fc_wgrad1d_slices on this device first)", t->T, dev);
  *out = it->second;
  return FC_OK;
}

// CU count of the current device, queried once per device.
int current_device_cus(int* cus_out) {
  static std::mutex m;
  static std::map<int, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  std::lock_guard<std::mutex> lock(m);
  auto it = cache.find(dev);
  if (it == cache.end()) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
    it = cache.emplace(dev, cus).first;
  }
  *cus_out = it->second;
  return 1;
}

int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

}  // namespace

struct fc_plan {
  fc_desc d;
  int nd;
  int64_t out_sp[3];
  int64_t kd[3];      // dilated kernel extent per axis
  // channel blocking
  int Cig, Cog, CB, cob, Cig_pad, Cog_pad, n_ochunks, accumulate;
  // last-axis (fused) tiling
  const fc::TileImpl* tile;
  Twiddles tw;
  int V, ntiles, Lfull;
  size_t lds_conv, lds_spec;
  size_t spectrum_bytes, workspace_bytes;
  // ---- N-d (2-D / 3-D): axis 0 = fused (outermost), axis nd-1 = rows (x), middle axis only in 3-D
  const fc::TileImpl* tx;     // rows (last axis), full-length FFT
  const fc::TileImpl* tm;     // middle axis (3-D), full-length FFT
  Twiddles twx, twm;
  int Sp[3], Lf[3];           // padded extent / stride-1 output extent per axis
  int need[3];                // shortest cyclic length that yields all Lf outputs exactly (<= Sp: zero padding absorbs the wrap)
  int padl[3], up[3], ostride[3];   // left pad in grid coordinates, source spread step, output decimation
  int Fx;                     // Tx/2
  int nxt, Vx, Fxt;           // overlap-save tiles along the rows axis (nxt = 1: one full-length transform), valid
                              // stride-1 samples per tile, bin columns per plane = nxt * Fx
  int nyt, Vy;                // the same for the middle axis of a 3-D problem (one c2c launch per tile)
  int nd_cob, nd_Cog_pad;
  int f64_T, f64_V, f64_ntiles, f64_cob;   // float64 1-D FFT path (fft_f64.hip): tile, valid samples, tiles per row, out-chunk; 0 = direct kernel
  int planes;                 // 1: 3-D plane-major three-launch pipeline (planes3d.hpp) instead of the five separable passes;
                              // 2: 2-D with the same thread-per-sequence column pass between row passes that keep the rows as they are
  size_t ws_a, ws_b;          // fc::f2 counts of the two workspace regions
  // ---- persistent fused 1-D kernel (fast path)
  int pers_nb;                // batch items per workgroup (0 = fast path not used)
  int pers_nb_choice;         // planner's pick for this plan (0 = general kernel)
  int pers_grid, pers_items;
  int chunk_launches;         // general kernel launched once per input chunk, later chunks add into y (see plan_1d)
  int wide;                   // > 8 input channels per group on the batch-sharing work list (conv1d_wide.hpp)
  int dense;                  // >= 16 channels per group on both sides: spectra through HBM + MFMA contraction (dense1d.hpp)
  int dense_mslab;            // rows (batch x tiles) per slab of that pipeline's workspace
  int dense_cus;              // CUs of the plan's device (grid of its persistent GEMM)
  size_t dense_pers_bytes;    // its kernel spectrum before the bin-major re-layout (scratch of fc_transform_kernel)
  int nseg, seg_taps;         // 1-D: the kernel runs in nseg segments of seg_taps taps (1 = whole kernel)
  int64_t kd_plan;            // dilated extent the tiles are planned for (of one segment)
  size_t seg_spectrum_bytes;  // kernel-spectrum bytes of one segment
  int diag;                   // depthwise (groups == Cin == Cout, multiple of 8): 8-channel blocks, per-channel mix
  int bd_gs;                  // groups of 2 or 4 channels regrouped into block-diagonal 8 x 8 blocks (0 = off)
  int G;                      // channel groups as the 1-D kernels see them (C/8 blocks for a depthwise plan)
  int slot_tiles;             // work-item slots = consecutive tiles of one batch item (else consecutive batch items)
  int ph;                     // dilation run as this many phases of a virtual batch (batch-sharing kernel), else 1
  int ph2;                    // the phases run in pairs (conv1d_pers.hpp PH2)
  fc::WorkItem* d_items;
  // ---- weight-gradient plan of an N-d convolution (fc_wgrad_nd): the convolution with batch and channels exchanged;
  // the tensors keep the caller's layout (ImgMap in the row passes)
  int swap;
  int64_t sw_B, sw_Cig, sw_Cog, sw_g;      // of the ORIGINAL convolution
};

// the original convolution behind a weight-gradient plan: batch, channels per group, groups, taps to keep per axis
struct WgradSwap {
  int64_t B, Cig, Cog, g, keep[3];
};

extern "C" {

int fc_version(void) { return FC_ABI_VERSION; }

const char* fc_last_error(void) { return g_err.c_str(); }

static int plan_1d_persistent(fc_plan* p);
static void set_channel_layout(fc_plan* p, int G, int Cig, int Cog);
static int choose_fast_path(fc_plan* p, int* tile_out);
static bool fast_path_eligible(const fc_plan* p);

static void set_channel_layout(fc_plan* p, int G, int Cig, int Cog) {
  p->G = G; p->Cig = Cig; p->Cog = Cog;
  const int cmax = std::max(Cig, Cog);
  p->CB = cmax <= 2 ? 2 : (cmax <= 4 ? 4 : 8);
  p->Cig_pad = (int)round_up(Cig, p->CB);
  p->cob = std::min(p->CB, (int)round_up(Cog, 2));
  p->Cog_pad = (int)round_up(Cog, p->cob);
  p->n_ochunks = p->Cog_pad / p->cob;
  p->accumulate = p->Cig_pad > p->CB;
}

static int plan_1d_inner(fc_plan* p);

// Depthwise rows (groups == Cin == Cout >= 5, stride 1) run on the batch-sharing kernel as blocks of
// 8 channels with a per-channel mix; when that kernel cannot take the shape the generic grouped plan is used.
static int plan_1d(fc_plan* p) {
  const fc_desc& d = p->d;
  const char* env = getenv("FFTCONV_DIAG");
  const bool want = !env || atoi(env) != 0;
  p->diag = 0;
  p->bd_gs = 0;
  {
    // groups of 2 or 4 channels (in == out per group): 8 / gs of them form one dense 8 x 8 block whose
    // cross-group spectrum entries are zero -- the batch-sharing kernel runs it as is
    const int64_t gs = d.in_channels / d.groups;
    if (want && (gs == 2 || gs == 4) && d.out_channels / d.groups == gs && d.groups % (8 / gs) == 0 && d.stride[0] == 1 &&
        !(d.tile_hint && !getenv("FFTCONV_PERS"))) {
      p->bd_gs = (int)gs;
      set_channel_layout(p, (int)(d.groups * gs / 8), 8, 8);
      const int rc = plan_1d_inner(p);
      if (rc == FC_OK && p->pers_nb != 0) return FC_OK;
      if (p->d_items) { (void)hipFree(p->d_items); p->d_items = nullptr; }
      p->bd_gs = 0;
      set_channel_layout(p, (int)d.groups, (int)(d.in_channels / d.groups), (int)(d.out_channels / d.groups));
    }
  }
  if (want && d.groups == d.in_channels && d.groups == d.out_channels && d.groups >= 5 && d.stride[0] == 1 &&
      !(d.tile_hint && !getenv("FFTCONV_PERS"))) {
    p->diag = 1;                                       // (the last block may be partly empty: the kernel masks it)
    set_channel_layout(p, (int)((d.groups + 7) / 8), 8, 8);
    const int rc = plan_1d_inner(p);
    if (rc == FC_OK && p->pers_nb != 0) return FC_OK;
    if (p->d_items) { (void)hipFree(p->d_items); p->d_items = nullptr; }
    p->diag = 0;
    set_channel_layout(p, (int)d.groups, (int)(d.in_channels / d.groups), (int)(d.out_channels / d.groups));
  }
  return plan_1d_inner(p);
}

static int plan_1d_inner(fc_plan* p) {
  const fc_desc& d = p->d;
  // Long kernels run in segments of taps: segment j is the convolution with taps [j*Ks, (j+1)*Ks) read
  // j*Ks*dilation samples further into the row, later segments add into y.  This lifts the 4096-point tile
  // limit on the dilated extent and keeps 8-channel shapes on the batch-sharing kernel beyond its 2048 tile.
  p->nseg = 1; p->seg_taps = (int)d.kernel[0]; p->kd_plan = p->kd[0];
  {
    const bool sharing_shape = p->CB == 8 && p->Cog % 8 == 0 && d.stride[0] == 1;
    const bool want_seg = p->kd[0] > 4096 || (sharing_shape && p->kd[0] > 1537 && !d.tile_hint);
    if (want_seg) {
      const int64_t ks = std::max<int64_t>(1, 1024 / d.dilation[0] + 1);       // (ks - 1) * dilation + 1 <= 1025
      p->seg_taps = (int)std::min<int64_t>(ks, d.kernel[0]);
      p->nseg = (int)((d.kernel[0] + p->seg_taps - 1) / p->seg_taps);
      p->kd_plan = (int64_t)(p->seg_taps - 1) * d.dilation[0] + 1;
    }
  }
  const int64_t L = d.spatial[0], Kd = p->kd_plan;
  const int64_t Lfull = p->Lf[0];
  p->Lfull = (int)Lfull;
  if (L * (int64_t)std::max(p->Cig, 1) * 4 >= (int64_t)1 << 32)
    return fail(FC_ERR_UNSUPPORTED, "1-D signal too long for 32-bit buffer offsets (Cin/groups * L * 4 must be < 4 GiB)");

  const int NPI = p->CB / 2;
  const size_t lds_cap = 160 * 1024;
  const fc::TileImpl* best = nullptr;
  double best_cost = 0;
  int ntl;
  auto tiles = all_tiles(&ntl);
  int forced_tile = d.tile_hint;
  p->pers_nb_choice = 0;
  p->ph = 1;
  p->wide = 0;
  p->dense = 0;
  {
    // 16 or more channels per group on BOTH sides, stride 1, kernel within the 1024 tile: transforms and contraction
    // in separate launches, the contraction as one real GEMM per frequency bin on the matrix pipe (dense1d.hpp).
    // FFTCONV_DENSE=0 keeps the fused kernels (A/B runs).
    const char* env = getenv("FFTCONV_DENSE");
    const int want_dense = env ? atoi(env) : 1;
    // Measured against the fused kernels (scripts/dense_check.py, us): 128->96 M = 30 rows 58 / 175; 32->32 M = 144
    // 50 / 84; 64->64 M = 152 91 / 150; but (first build) 24->40 M = 18 44 / 33, 16->24 x 2 groups M = 26 41 / 27,
    // 16->16 M = 2 31 / 23: three launches need work to amortise -- at least 32 channels a side and 64 K row-channel
    // products.  FFTCONV_DENSE=2 forces the pipeline for every shape it can run (tests).
    const int64_t Kd_d = p->kd[0];
    const int dT = Kd_d <= 769 ? 1024 : 2048;             // (at least a quarter of the tile valid)
    const fc::TileImpl* dt = find_tile(dT);
    const int64_t Vd = std::max<int64_t>(1, dT + 1 - Kd_d);
    const int64_t Md = d.batch * ((p->Lf[0] + Vd - 1) / Vd);
    const bool pays = want_dense == 2 || (p->Cig >= 32 && p->Cog >= 32 && Md * p->Cig * p->Cog >= 65536);
    if (want_dense && pays && dt && dt->dense && p->nseg == 1 && p->CB == 8 && p->Cig >= 16 && p->Cog >= 16 && d.stride[0] == 1 &&
        p->up[0] == 1 && !p->diag && !p->bd_gs && Kd <= 1537 && (!forced_tile || forced_tile == dT) &&
        (int64_t)p->Cig * d.spatial[0] * 4 < ((int64_t)1 << 32)) {
      // 32-bit offsets of the pipeline: dense_inv marks dead stores with bit 31 of an offset into one group's output rows
      // (Cog * Lout * 4 bytes), and the slab resources / bin strides are 32-bit too (a slab row block of at least 128
      // rows x NF bins x max(Kc, Nc) channels).  Shapes beyond either limit stay with the fused kernels.
      const int64_t NFd = dT / 2 + 1;
      const int64_t rows_min = std::min<int64_t>(Md, 128);
      const bool out_ok = (int64_t)p->Cog * p->out_sp[0] * 4 < ((int64_t)1 << 31);
      const bool slab_ok = NFd * rows_min * std::max(p->Cig_pad, p->Cog_pad) * 8 < ((int64_t)1 << 32);
      if (out_ok && slab_ok) {
        p->dense = 1;
        forced_tile = dT;
      }
    }
  }
  if (!p->dense) {
    // more than 8 input channels per group, whole out-chunks, stride 1: the register-accumulating
    // batch-sharing kernel (1024 or 2048 tile, whichever keeps at least a quarter of the tile valid).
    // Short kernels stay with the general kernel and its small tiles (measured: 16->16, k = 33, L = 4096:
    // 24.6 us there vs 29.1 us here; k = 129 ... 1025: 2.0-2.5x faster here).
    const char* env = getenv("FFTCONV_WIDE");
    const int want_wide = env ? atoi(env) : 1;
    if (want_wide && p->nseg == 1 && p->CB == 8 && p->accumulate && p->Cog % 8 == 0 && d.stride[0] == 1 && d.batch >= 2 &&
        ((int64_t)d.in_channels * 3 + p->Cig) * d.spatial[0] * 4 < ((int64_t)1 << 32)) {
      const int wt = Kd < 97 ? 0 : (Kd <= 768 ? 1024 : (Kd <= 1536 ? 2048 : 0));
      if (wt && (!forced_tile || forced_tile == wt) && find_tile(wt) && find_tile(wt)->wide_nb) {
        p->wide = 1;
        forced_tile = wt;
      }
    }
  }
  if (p->wide || p->dense) {
    // tile fixed above
  } else if (!forced_tile) {
    int rc = choose_fast_path(p, &forced_tile);
    if (rc != FC_OK) return rc;
  } else if (d.dilation[0] > 1 && fast_path_eligible(p) && getenv("FFTCONV_PERS") && atoi(getenv("FFTCONV_PERS")) > 0) {
    p->ph = (int)d.dilation[0];           // explicit tile + explicit flavour: dilation as phases
  }
  // dilation as phases: the kernel seen by a tile is the undilated one, rows are 1/ph as long
  const int64_t Kd_t = p->ph > 1 ? d.kernel[0] : Kd;
  const int64_t Lfull_t = p->ph > 1 ? (Lfull + p->ph - 1) / p->ph : Lfull;
  p->chunk_launches = 0;
  for (int attempt = 0; attempt < 2 && !best; ++attempt) {
  if (attempt == 1) {
    // No tile holds the kernel together with the second (running-sum) LDS region of a multi-chunk plan:
    // launch the general kernel once per input chunk instead, chunks after the first adding into y.
    if (!p->accumulate || p->wide || p->dense || forced_tile) break;
    p->accumulate = 0;
    p->chunk_launches = 1;
  }
  for (int i = 0; i < ntl; ++i) {
    const fc::TileImpl* t = tiles[i];
    if (forced_tile && t->T != forced_tile) continue;
    if (t->T < Kd_t) continue;
    const size_t lds = p->wide ? t->wide_lds : (p->dense ? 0 : (size_t)(p->accumulate ? 2 : 1) * NPI * t->lseq * sizeof(fc::f2));
    if (lds > lds_cap) continue;
    if (t->NT / (t->P * t->S) < NPI) continue;
    const int64_t V = t->T - Kd_t + 1;
    const int64_t nt = (Lfull_t + V - 1) / V;
    // work model: FFT passes + channel mix per tile; the largest tile runs one
    // workgroup per CU (LDS), which costs latency hiding
    double cost = (double)nt * t->T * (2.0 * std::log2((double)t->T) + 4.0 + p->CB);
    // measured (cfgD, MI355X): one 512-thread workgroup per CU and the four-lane split cost the
    // 4096 tile ~1.7x per sample of tile; it only wins when the kernel is nearly as long as 2048
    if (lds > 80 * 1024) cost *= 1.7;
    if (!best || cost < best_cost) { best = t; best_cost = cost; }
  }
  }
  if (!best) {
    if (d.tile_hint) return fail(FC_ERR_INVALID, "tile_hint %d is not usable for this problem", d.tile_hint);
    if (Kd <= 4096)
      return fail(FC_ERR_UNSUPPORTED, "dilated kernel extent %lld needs the 4096-point tile, which cannot hold the running "
                  "sums of more than 8 input channels per group (%d here)", (long long)Kd, p->Cig);
    return fail(FC_ERR_UNSUPPORTED, "dilated kernel extent %lld exceeds the largest FFT tile (4096)", (long long)Kd);
  }
  p->tile = best;
  p->V = (int)(best->T - Kd_t + 1);
  p->ntiles = (int)((Lfull_t + p->V - 1) / p->V);
  p->lds_conv = (size_t)(p->accumulate ? 2 : 1) * NPI * best->lseq * sizeof(fc::f2);
  p->lds_spec = (size_t)(best->NT / (best->P * best->S)) * best->lseq * sizeof(fc::f2);
  const size_t per_group = (size_t)p->Cog_pad * (p->Cig_pad / 2) * (best->T / 2) * sizeof(fc::f4);
  if (per_group >= ((size_t)1 << 32))
    return fail(FC_ERR_UNSUPPORTED, "kernel spectrum of one group exceeds 4 GiB");
  p->seg_spectrum_bytes = p->diag ? (size_t)(round_up(d.in_channels, 8) / 2) * (best->T / 2) * sizeof(fc::f4) : per_group * (size_t)p->G;
  p->spectrum_bytes = p->seg_spectrum_bytes * (size_t)p->nseg;
  p->workspace_bytes = 0;
  int rc = get_twiddles(best, &p->tw);
  if (rc != FC_OK) return rc;
  if (p->dense) {
    // spectrum: bin-major complex matrices; workspace: one slab of X and Y rows (<= 192 MiB), or the fused-layout
    // spectrum while the kernel is being transformed
    const size_t NF = (size_t)best->T / 2 + 1;
    p->dense_pers_bytes = p->spectrum_bytes;
    p->spectrum_bytes = (size_t)p->G * NF * p->Cig_pad * p->Cog_pad * sizeof(fc::f2);
    p->seg_spectrum_bytes = p->spectrum_bytes;
    const int64_t M = d.batch * (int64_t)p->ntiles;
    const size_t row_bytes = (size_t)p->G * NF * (size_t)(p->Cig_pad + p->Cog_pad) * sizeof(fc::f2);
    int64_t slab = (int64_t)(((size_t)192 << 20) / row_bytes) / 128 * 128;
    slab = std::max<int64_t>(128, slab);
    // (32-bit offsets inside a slab: NF * rows * channels * 8 bytes per side; plan_1d_inner admitted the shape for 128 rows)
    while (slab > 128 && (int64_t)NF * slab * std::max(p->Cig_pad, p->Cog_pad) * 8 >= ((int64_t)1 << 32)) slab -= 128;
    if (const char* e = getenv("FFTCONV_DENSE_SLAB")) slab = std::max(1, atoi(e));     // testing knob: rows per slab
    p->dense_mslab = (int)std::min<int64_t>(M, slab);
    p->dense_cus = 256;
    if (!current_device_cus(&p->dense_cus)) return fail(FC_ERR_HIP, "cannot query the current device");
    p->workspace_bytes = std::max(row_bytes * (size_t)p->dense_mslab, p->dense_pers_bytes);
    p->pers_nb = 0; p->d_items = nullptr; p->pers_items = 0; p->pers_grid = 0;
    return FC_OK;
  }
  rc = plan_1d_persistent(p);
  if (rc == FC_OK && p->ph > 1 && p->pers_nb == 0)
    return fail(FC_ERR_INVALID, "internal: phase plan without the batch-sharing kernel");
  return rc;
}


static bool fast_path_eligible(const fc_plan* p) {
  const fc_desc& d = p->d;
  if (p->CB != 8 || p->accumulate || p->chunk_launches || p->Cog % 8 != 0 || d.stride[0] != 1) return false;   // (a transposed plan with stride 1 is a padded correlation: same kernel)
  if (((int64_t)d.in_channels * 3 + p->Cig) * d.spatial[0] * 4 >= ((int64_t)1 << 32)) return false;
  return true;
}

// Joint choice of FFT tile and kernel flavour for 8-channel chunks.  Measured per-workgroup times on
// MI355X (us, phase_profile.py, cfgA-like rows): the general kernel at 2048 / 1024 and the
// batch-sharing kernel at (2048, nb 2), (1024, nb 2), (1024, nb 4); estimated launch time =
// residency rounds x time per workgroup.  Small problems are decided by the rounds, large ones by
// outputs per microsecond.
static int choose_fast_path(fc_plan* p, int* tile_out) {
  const fc_desc& d = p->d;
  const char* env = getenv("FFTCONV_PERS");
  const int want = env ? atoi(env) : -1;            // -1 auto, 0 general kernel only, n force nb = n
  const bool fast_ok = want != 0 && fast_path_eligible(p);
  int cus = 256;
  if (!current_device_cus(&cus)) return fail(FC_ERR_HIP, "cannot query the current device");
  const int64_t per_item_units = (int64_t)p->n_ochunks * p->G;
  // {tile, batch items per workgroup (0 = general kernel), resident workgroups per CU, us per workgroup}
  // Launch-time model (round 3, `profiles/r03_planner_sweep.jsonl`: every candidate forced in turn on 12 shapes):
  //   general kernel        est = rounds x t_item, one item per workgroup (t_item: a full round, launch included)
  //   batch-sharing kernel  workgroups run up to two items back to back (grid as plan_1d_persistent builds it);
  //                         est = kLaunchUs + sum over waves of workgroups of (items per workgroup x t(occupancy)),
  //                         t(occ) between t_alone (one workgroup on its CU) and t_item (CU full): a 256-thread
  //                         workgroup alone on a CU runs an item in 10.5 us, beside a second one in 14.2
  // The round-2 table priced a batch-sharing workgroup at 15-19.5 us whatever it ran beside and however many items it
  // took: 15-25 % regret wherever fewer workgroups than slots exist or the grid spills into a second wave.
  struct Cand { int T, nb, wgs_per_cu; double t_item, t_alone; };
  const Cand cands[] = {{256, 0, 8, 11.7, 0}, {512, 0, 6, 17.0, 0}, {1024, 0, 4, 25.5, 0}, {2048, 0, 2, 28.9, 0},
                        {2048, 2, 1, 17.3, 17.3}, {2048, 1, 2, 26.0, 13.4}, {1024, 2, 2, 14.2, 10.5}, {1024, 4, 1, 13.3, 13.3}};
  const double kLaunchUs = 4.0;
  double best = 0;
  int best_T = 0, best_nb = 0, best_ph = 1;
  bool best_tiles = false;
  // second round: dilation d as d phases of a virtual batch B*d against the undilated kernel
  const int rounds = (fast_ok && d.dilation[0] > 1 && p->nseg == 1) ? 2 : 1;
  for (int round = 0; round < rounds; ++round) {
    const int ph = round ? (int)d.dilation[0] : 1;
    const int64_t Kd = round ? d.kernel[0] : p->kd_plan;
    const int64_t Lfull = (p->Lf[0] + ph - 1) / ph;
    const int64_t B = d.batch * ph;
    for (const Cand& c : cands) {
      if ((round || p->diag) && c.nb == 0) continue;  // only the batch-sharing kernel knows phases / depthwise blocks
      if (c.T < Kd || p->accumulate) continue;
      if (c.nb != 0 && !fast_ok) continue;
      if (want > 0 && c.nb != want) continue;
      const int64_t V = c.T - Kd + 1;
      if (V * 4 < c.T) continue;                      // less than a quarter of the tile useful: leave to the cost model
      const int64_t nt = (Lfull + V - 1) / V;
      // fewer batch items than slots: the slots of a work item become consecutive TILES of one batch item
      // (they share the spectrum just the same); measured 1.3-1.6x on batch-1 rows of 2^20 samples
      const bool by_tiles = c.nb > B;
      if (by_tiles && nt < c.nb) continue;
      const int64_t groups_of = by_tiles ? B * ((nt + c.nb - 1) / c.nb) : ((B + std::max(c.nb, 1) - 1) / std::max(c.nb, 1)) * nt;
      const int64_t items = groups_of * per_item_units;
      const int64_t slots = (int64_t)cus * c.wgs_per_cu;
      double est;
      if (c.nb == 0) {
        est = (double)((items + slots - 1) / slots) * c.t_item;
      } else {
        const int64_t grid = std::max<int64_t>((items + 1) / 2, std::min<int64_t>(items, slots));
        const double ipw = (double)items / (double)grid;                 // 1 .. 2 items per workgroup
        auto t_occ = [&](int64_t wgs) {                                  // per item, `wgs` workgroups spread over the CUs
          const int64_t occ = std::min<int64_t>(c.wgs_per_cu, (wgs + cus - 1) / cus);
          return c.wgs_per_cu > 1 ? c.t_alone + (c.t_item - c.t_alone) * (double)(occ - 1) / (double)(c.wgs_per_cu - 1) : c.t_item;
        };
        const int64_t full = grid / slots, rem = grid % slots;
        // (the makespan of a wave is its slowest workgroup: whole items)
        est = kLaunchUs + (double)full * std::ceil(ipw) * c.t_item + (rem ? std::ceil(ipw) * t_occ(rem) : 0.0);
      }
      if (best_T == 0 || est < best) { best = est; best_T = c.T; best_nb = c.nb; best_ph = ph; best_tiles = by_tiles; }
    }
  }
  if (best_T == 0) return FC_OK;                    // general planner (cost model) decides
  *tile_out = best_T;
  p->pers_nb_choice = best_nb;
  p->ph = best_ph;
  p->slot_tiles = best_tiles ? 1 : 0;
  return FC_OK;
}

// Work list of the persistent fused kernel: items of up to NB batch items that share (tile, group,
// out-chunk), largest first; workgroup w takes items w, w+grid, ...  One workgroup per LDS slot.
static int plan_1d_persistent(fc_plan* p) {
  p->pers_nb = 0; p->d_items = nullptr; p->pers_items = 0; p->pers_grid = 0;
  const fc_desc& d = p->d;
  if (!p->wide && !fast_path_eligible(p)) return FC_OK;
  const fc::TileImpl* t = p->tile;
  int cus = 256;
  if (!current_device_cus(&cus)) return fail(FC_ERR_HIP, "cannot query the current device");
  const int64_t B = d.batch * p->ph;                 // virtual batch (dilation phases)
  int nb = p->pers_nb_choice;
  if (d.tile_hint && !p->wide) {                    // explicit tile: FFTCONV_PERS picks the flavour (default general)
    const char* env = getenv("FFTCONV_PERS");
    nb = env ? atoi(env) : 0;
  }
  int wgs_per_cu;
  if (p->wide) {
    nb = t->wide_nb;
    wgs_per_cu = std::max(1, (int)((160 * 1024) / t->wide_lds));
  } else {
    if (nb != t->pers_nb[0] && nb != t->pers_nb[1]) nb = 0;
    if (nb == 0) return FC_OK;
    const int slot = nb == t->pers_nb[0] ? 0 : 1;
    wgs_per_cu = std::max(1, (int)((160 * 1024) / t->pers_lds[slot]));
  }
  // Items: up to nb batch items that share (tile, group, out-chunk), full items first.  When the last
  // residency round would fill less than half of the CUs, its items are split in two so the tail
  // spreads over twice as many CUs (cfgA: 336 pairs on 256 CUs -> 256 pairs + 160 singles).
  std::vector<fc::WorkItem> items;
  const int nfull = (int)(B / nb), rem = (int)(B % nb);
  const int64_t slots = (int64_t)cus * wgs_per_cu;
  // border tiles (staged, slower loads) are issued first so they never form the tail of the launch
  std::vector<int> tile_order;
  {
    const int T = t->T, V = p->V;
    for (int pass = 0; pass < 2; ++pass)
      for (int tile = 0; tile < p->ntiles; ++tile) {
        const int64_t pos = (int64_t)tile * V * p->ph - p->padl[0];
        const bool interior = p->up[0] == 1 && pos >= 0 && pos + (int64_t)(T - 1) * p->ph + p->ph <= d.spatial[0];
        if ((pass == 0) == !interior) tile_order.push_back(tile);
      }
  }
  {
    const char* env = getenv("FFTCONV_SLOTS");      // "tiles" / "batch": overrides the planner's choice
    if (env) p->slot_tiles = env[0] == 't' ? 1 : 0;
    if (p->wide) p->slot_tiles = 0;
  }
  auto is_border = [&](int tile) {
    const int64_t pos = (int64_t)tile * p->V * p->ph - p->padl[0];
    return !(p->up[0] == 1 && pos >= 0 && pos + (int64_t)(t->T - 1) * p->ph + p->ph <= d.spatial[0]);
  };
  if (p->slot_tiles) {
    // slots = consecutive tiles of one (virtual) batch item: chunks that touch a border tile go first
    for (int pass = 0; pass < 2; ++pass)
      for (int64_t vb = 0; vb < B; ++vb)
        for (int goc = 0; goc < p->n_ochunks * p->G; ++goc)
          for (int t0 = 0; t0 < p->ntiles; t0 += nb) {
            const int n = std::min(nb, p->ntiles - t0);
            bool border = false;
            for (int k = 0; k < n; ++k) border |= is_border(t0 + k);
            if ((pass == 0) == border) items.push_back({(int)vb, n, t0, goc});
          }
  } else {
  for (int tile : tile_order)
    for (int goc = 0; goc < p->n_ochunks * p->G; ++goc)
      for (int c = 0; c < nfull; ++c) items.push_back({c * nb, nb, tile, goc});
  }
  // phase quads (conv1d_pers.hpp PH4): four phases, a multiple of 4 of them per batch item, one full 8 x 8 channel block;
  // FFTCONV_PH2 = 0 / 1 keeps single phases / pairs (A/B runs, tests).  Decided before the tail split: a quad item cannot
  // be halved (a wave owns all four phases of its channel).
  const char* ph_env = getenv("FFTCONV_PH2");
  const int ph_want = ph_env ? atoi(ph_env) : 2;
  const bool ph_base = p->ph > 1 && !p->slot_tiles && !p->diag && !p->wide && p->nseg == 1 && nb >= 2 && t->S == 1;
  const bool quads = ph_want >= 2 && ph_base && p->ph % 4 == 0 && nb == 4 && p->Cig == 8 && p->cob == 8 && p->Cog % 8 == 0 && !p->bd_gs;
  if (nb >= 2 && (int64_t)items.size() > slots && !quads) {
    const int64_t tail = (int64_t)items.size() % slots;
    if (tail > 0 && tail <= slots / 2) {
      std::vector<fc::WorkItem> split;
      for (int64_t k = (int64_t)items.size() - tail; k < (int64_t)items.size(); ++k) {
        const fc::WorkItem w = items[k];
        const int h = w.nbc / 2;
        if (h == 0) { split.push_back(w); continue; }
        split.push_back({w.b0, h, w.tile, w.goc});
        if (p->slot_tiles) split.push_back({w.b0, w.nbc - h, w.tile + h, w.goc});
        else split.push_back({w.b0 + h, w.nbc - h, w.tile, w.goc});
      }
      items.resize(items.size() - tail);
      items.insert(items.end(), split.begin(), split.end());
    }
  }
  if (rem && !p->slot_tiles)
    for (int tile = 0; tile < p->ntiles; ++tile)
      for (int goc = 0; goc < p->n_ochunks * p->G; ++goc) items.push_back({nfull * nb, rem, tile, goc});
  if (items.size() > 0x7fffffffu) return FC_OK;
  p->pers_items = (int)items.size();
  // up to two items per workgroup (the second one's input is prefetched): item i and i + grid
  p->pers_grid = (int)std::max<int64_t>((p->pers_items + 1) / 2, std::min<int64_t>(p->pers_items, slots));
  if (p->wide) p->pers_grid = p->pers_items;         // one item per workgroup
  FC_HIP_SETUP(hipMalloc(&p->d_items, items.size() * sizeof(fc::WorkItem)));
  FC_HIP_SETUP(hipMemcpy(p->d_items, items.data(), items.size() * sizeof(fc::WorkItem), hipMemcpyHostToDevice));
  p->pers_nb = nb;
  // phases in pairs: an even number of phases, slots = batch items (so slots 2j, 2j+1 are neighbouring phases of one
  // batch item), plain dense-block kernel on a P*P tile; quads (above) take precedence
  p->ph2 = quads ? 2 : ((ph_want != 0 && ph_base && p->ph % 2 == 0) ? 1 : 0);
  return FC_OK;
}

static const fc::TileImpl* smallest_tile_at_least(int64_t n) {
  int ntl;
  auto tiles = all_tiles(&ntl);
  const fc::TileImpl* best = nullptr;
  for (int i = 0; i < ntl; ++i)
    if (tiles[i]->T >= n && (!best || tiles[i]->T < best->T)) best = tiles[i];
  return best;
}

static int plan_nd(fc_plan* p) {
  const fc_desc& d = p->d;
  const int nd = p->nd;
  // rows axis: one full-length transform when the padded row fits the largest FFT, overlap-save tiles otherwise
  // (the reference has no size limit: functional.py:66-70); middle axis (3-D): full-length transform
  p->nxt = 1;
  p->Vx = p->Lf[nd - 1];
  p->tx = smallest_tile_at_least(p->need[nd - 1]);
  {
    const char* env = getenv("FFTCONV_XTILE");       // testing knob: force x tiles of this length (where the kernel fits)
    const int64_t kdx = p->kd[nd - 1];
    int forced = env ? atoi(env) : 0;
    if (forced && (!find_tile(forced) || find_tile(forced)->T < kdx)) forced = 0;
    if (!p->tx || forced) {
      const fc::TileImpl* t = forced ? find_tile(forced) : find_tile(kdx <= 1025 ? 2048 : 4096);
      if (!t || t->T < kdx)
        return fail(FC_ERR_UNSUPPORTED, "dilated kernel extent %lld along the last axis exceeds the largest FFT (4096)", (long long)kdx);
      p->tx = t;
      p->Vx = (int)(t->T - kdx + 1);
      p->nxt = (int)((p->Lf[nd - 1] + p->Vx - 1) / p->Vx);
    } else if (!env && !p->swap) {
      // A row just past a power of two ('same' padding on a power-of-two image: 518 samples -> a 1024-point transform, and
      // twice the bin columns for every pass behind it) is cheaper in overlap-save tiles of a quarter of that length: the
      // points per row decide (measured, scripts/experiments/sweep_same_xtile.py: B16 512^2 k7 'same' 499 us with one
      // 1024-point transform, 281 us in 128-point tiles, 342 in 256-point ones; B8 1024^2 k5 1,209 / 537).  Taken when it
      // saves at least 15 % of the points; tiles keep at least half of themselves and are at least 64 long.
      const int64_t single = p->tx->T;
      int64_t best_pts = single;
      const fc::TileImpl* best_t = nullptr;
      int ntl2;
      auto tl = all_tiles(&ntl2);
      for (int i = 0; i < ntl2; ++i) {
        const fc::TileImpl* t = tl[i];
        if (t->T < 64 || t->T >= single || t->T < 2 * kdx) continue;
        const int64_t V = t->T - kdx + 1, n = (p->Lf[nd - 1] + V - 1) / V, pts = n * t->T;
        if (pts * 100 <= single * 85 && pts < best_pts) { best_pts = pts; best_t = t; }
      }
      if (best_t) {
        p->tx = best_t;
        p->Vx = (int)(best_t->T - kdx + 1);
        p->nxt = (int)((p->Lf[nd - 1] + p->Vx - 1) / p->Vx);
      }
    }
  }
  p->Fx = p->tx->T / 2;        // odd-frequency bins along the rows axis (nd_passes.hpp, rows_r2c)
  p->Fxt = p->nxt * p->Fx;
  {
    // the row passes address one (bin column, row) block per workgroup with 32-bit byte offsets
    const int64_t rows = std::max<int64_t>(p->Sp[nd - 2], p->out_sp[nd - 2]);
    if ((int64_t)p->Fx * rows * 8 >= ((int64_t)1 << 31))
      return fail(FC_ERR_UNSUPPORTED, "%lld rows of %d-point transforms along the last axis exceed the 2 GiB a block of bin "
                  "columns may span (split the second-to-last axis)", (long long)rows, p->tx->T);
    // ... and the rows_c2r output stores a workgroup's (at most 32) output rows the same way
    if (p->out_sp[nd - 1] * 4 * 32 >= ((int64_t)1 << 31))
      return fail(FC_ERR_UNSUPPORTED, "output rows of %lld samples exceed the 2 GiB a workgroup's block of rows may span "
                  "(2-D / 3-D plans; 1-D rows have no such limit)", (long long)p->out_sp[nd - 1]);
  }
  p->tm = nullptr;
  p->nyt = 1;
  p->Vy = nd == 3 ? p->Lf[1] : 0;
  if (nd == 3) {
    p->tm = smallest_tile_at_least(p->need[1]);
    const char* env = getenv("FFTCONV_YTILE");       // testing knob: force middle-axis tiles of this length
    const int64_t kdy = p->kd[1];
    int forced = env ? atoi(env) : 0;
    if (forced && (!find_tile(forced) || find_tile(forced)->T < kdy)) forced = 0;
    if (!p->tm || forced) {
      const fc::TileImpl* t = forced ? find_tile(forced) : find_tile(kdy <= 1025 ? 2048 : 4096);
      if (!t || t->T < kdy)
        return fail(FC_ERR_UNSUPPORTED, "dilated kernel extent %lld along the middle axis exceeds the largest FFT (4096)", (long long)kdy);
      p->tm = t;
      p->Vy = (int)(t->T - kdy + 1);
      p->nyt = (int)((p->Lf[1] + p->Vy - 1) / p->Vy);
    } else if (!env && !p->swap) {
      // the same choice as on the rows axis: overlap-save tiles (one c2c launch per tile) where they save >= 15 % of the points
      const int64_t single = p->tm->T;
      int64_t best_pts = single;
      const fc::TileImpl* best_t = nullptr;
      int ntl2;
      auto tl = all_tiles(&ntl2);
      for (int i = 0; i < ntl2; ++i) {
        const fc::TileImpl* t = tl[i];
        if (t->T < 64 || t->T >= single || t->T < 2 * kdy) continue;
        const int64_t V = t->T - kdy + 1, n = (p->Lf[1] + V - 1) / V, pts = n * t->T;
        if (n <= 8 && pts * 100 <= single * 85 && pts < best_pts) { best_pts = pts; best_t = t; }
      }
      if (best_t) {
        p->tm = best_t;
        p->Vy = (int)(best_t->T - kdy + 1);
        p->nyt = (int)((p->Lf[1] + p->Vy - 1) / p->Vy);
      }
    }
    if (p->nyt == 1) p->Sp[1] = std::min(p->Sp[1], p->tm->T);    // (rows past the transform are zero padding: not produced)
  }
  // 3-D planes larger than 64 x 64 after padding: cut into overlap-save tiles of 64 x 64 so that the plane-major pipeline
  // (below) still applies -- each tile is one workgroup of planes_fwd / planes_inv and one block of 2048 columns of colz.
  // Measured against the separable passes with the planner's own x / y tiles (profiles/r03_experiments.md block 12).
  if (nd == 3 && !getenv("FFTCONV_XTILE") && !getenv("FFTCONV_YTILE") && !p->swap && !d.tile_hint) {
    const char* pl = getenv("FFTCONV_PLANES");
    const fc::TileImpl* t64 = find_tile(64);
    const bool wide = p->tx->T > 64 || p->tm->T > 64 || p->nxt > 1 || p->nyt > 1;
    if ((!pl || atoi(pl) != 0) && t64 && t64->colz && wide && p->CB == 8 && !p->accumulate && p->kd[0] <= 33 && p->kd[1] <= 33 &&
        p->kd[2] <= 33) {
      const int64_t Vx = 64 - p->kd[2] + 1, Vy = 64 - p->kd[1] + 1;
      const int64_t nx = p->need[2] <= 64 ? 1 : (p->Lf[2] + Vx - 1) / Vx, ny = p->need[1] <= 64 ? 1 : (p->Lf[1] + Vy - 1) / Vy;
      // (taken while the tiles hold at most 1.5x the points of the separable plan's own transforms: at equal points the
      //  pipeline measured 1.3-2.0x faster -- 64^3 k3 'same' 663 -> 507 us, 128^3 k5 'same' 772 -> 455, 200^3 k5 1,571 -> 787 --
      //  at 2.25x, 128^3 k9 unpadded against single 128-point transforms, 14 % slower)
      const int64_t sep_pts = (int64_t)p->nxt * p->tx->T * p->nyt * p->tm->T;
      if (nx * ny <= 36 && nx * ny * 4096 * 2 <= sep_pts * 3) {
        p->tx = t64; p->tm = t64;
        p->nxt = (int)nx; p->Vx = nx == 1 ? p->Lf[2] : (int)Vx;
        p->nyt = (int)ny; p->Vy = ny == 1 ? p->Lf[1] : (int)Vy;
        p->Fx = 32; p->Fxt = p->nxt * p->Fx;
        if (p->nyt == 1) p->Sp[1] = std::min(p->Sp[1], 64);
      }
    }
  }
  // channel blocking of the fused (complex) pass: one sequence per channel
  p->nd_cob = std::min(p->CB, p->Cog);
  p->nd_Cog_pad = (int)round_up(p->Cog, p->nd_cob);
  // Plane-major 3-D pipeline (planes3d.hpp): padded (y, x) planes within one 64 x 64 transform, 8-channel chunks with a
  // single input chunk, and a 64-point z tile (taken whenever the z kernel leaves at least half of it valid).
  // FFTCONV_PLANES=0 keeps the separable passes (A/B runs, tests).
  bool planes_ok = false;
  {
    const char* env = getenv("FFTCONV_PLANES");
    const fc::TileImpl* t64 = find_tile(64);
    planes_ok = (!env || atoi(env) != 0) && !p->swap && nd == 3 && t64 && t64->colz && p->tx->T == 64 && p->tm->T == 64 &&
                (int64_t)p->nxt * p->nyt <= 36 && (p->nxt == 1 || p->kd[2] <= 33) && (p->nyt == 1 || p->kd[1] <= 33) &&
                p->CB == 8 && !p->accumulate && p->kd[0] <= 33 && (!d.tile_hint || d.tile_hint == 64) &&
                (int64_t)2 * std::max(d.in_channels, d.out_channels) * std::max<int64_t>(p->Sp[0], p->out_sp[0]) * p->nxt * p->nyt < 65536;   // (32-bit offsets below 2 GiB per workgroup)
    // 2-D: the same column pass (one thread per 64-point sequence along y, lanes over neighbouring bin columns) between
    // row passes that keep the rows as they are -- taken under the same conditions on the y kernel and the channel blocks
    // Measured (scripts/experiments/time_rows2d.py, profiles/r03_experiments.md block 10): 5-13 % faster than the LDS column
    // pass on large images with y kernels up to ~25 taps (B16 512^2 k3..k23, B2 1024^2 k7), level at k31 (a 64-point tile then
    // keeps 34 samples), 8-10 % slower on small problems (B4 256^2) -- taken from 2^20 intermediate samples per channel and
    // 25 dilated taps down (33 where the LDS pass would need several tiles); FFTCONV_PLANES=2 takes it wherever it is possible (tests), 0 never.
    if (nd == 2) {
      const int knob = env ? atoi(env) : 1;
      // (26-33 taps: level with ONE 512-point tile of the LDS column pass -- cfgB -- but ahead of several of them:
      //  B16 512^2 k31 'same' 341 us against 439)
      const bool big = (int64_t)d.batch * p->Sp[0] * p->Fxt >= ((int64_t)1 << 20) && (p->kd[0] <= 25 || p->need[0] > 512);
      planes_ok = knob != 0 && (big || knob == 2) && !p->swap && t64 && t64->colz && p->CB == 8 && !p->accumulate &&
                  p->kd[0] <= 33 && (!d.tile_hint || d.tile_hint == 64) && p->Fx % 16 == 0 &&
                  (int64_t)4 * std::max(d.in_channels, d.out_channels) * std::max<int64_t>(p->Sp[0], p->out_sp[0]) * p->Fxt * 8 < ((int64_t)1 << 31);
    }
  }
  // overlap-save tiles along the outermost axis
  const int64_t Kd = p->kd[0], Lfull = p->Lf[0];
  const size_t lds_cap = 160 * 1024;
  const fc::TileImpl* best = nullptr;
  double best_cost = 0;
  int ntl;
  auto tiles = all_tiles(&ntl);
  for (int i = 0; i < ntl; ++i) {
    const fc::TileImpl* t = tiles[i];
    if (d.tile_hint && t->T != d.tile_hint) continue;
    if (planes_ok && t->T != 64) continue;
    if (t->T < Kd || p->CB > t->fusedc_max_cib) continue;
    const size_t lds = (size_t)(p->accumulate ? 2 : 1) * p->CB * t->lseqp * sizeof(fc::f2);
    if (lds > lds_cap) continue;
    const int64_t V = t->T - Kd + 1;
    const int64_t nt = t->T >= p->need[0] ? 1 : (Lfull + V - 1) / V;     // (one tile when the zero padding absorbs the wrap)
    double cost = (double)nt * t->T * (2.0 * std::log2((double)t->T) + 4.0 + 2.0 * p->CB);
    if (lds > 80 * 1024) cost *= 1.25;
    if (!best || cost < best_cost) { best = t; best_cost = cost; }
  }
  if (!best) {
    if (d.tile_hint) return fail(FC_ERR_INVALID, "tile_hint %d is not usable for this problem", d.tile_hint);
    return fail(FC_ERR_UNSUPPORTED, "no FFT tile fits the outermost axis (dilated kernel extent %lld, %d channels per chunk)",
                (long long)Kd, p->CB);
  }
  p->tile = best;
  if (best->T >= p->need[0]) {
    // the whole axis in one cyclic tile: every one of its Lfull outputs is kept, padded positions past the tile are zero
    p->V = (int)std::max<int64_t>(best->T - Kd + 1, Lfull);
    p->ntiles = 1;
    p->Sp[0] = std::min(p->Sp[0], best->T);
  } else {
    p->V = (int)(best->T - Kd + 1);
    p->ntiles = (int)((Lfull + p->V - 1) / p->V);
  }
  p->Lfull = (int)Lfull;
  p->lds_conv = (size_t)(p->accumulate ? 2 : 1) * p->CB * best->lseqp * sizeof(fc::f2);

  const size_t B = (size_t)d.batch, Ci = (size_t)d.in_channels, Co = (size_t)d.out_channels;
  const size_t Fx = (size_t)p->Fx;          // bin columns of the kernel spectrum (one x tile)
  const size_t Fs = (size_t)p->Fxt;         // bin columns of the signal side (all x tiles)
  size_t ncol, a_sig, b_sig, a_w, b_w;
  if (nd == 2) {
    ncol = Fx;
    a_sig = B * Ci * Fs * p->Sp[0];                       // S1[(b,ci)][xt,fx][yp]
    b_sig = B * Co * Fs * (size_t)p->out_sp[0];           // O1[(b,co)][xt,fx][y_out]
    a_w = Co * p->Cig * Fx * (size_t)p->kd[0];            // S1w[(o,i)][fx][y<Kd]
    b_w = 0;
  } else {
    const size_t Ty = (size_t)p->tm->T, Tys = Ty * (size_t)p->nyt;  // kernel side / signal side (all middle-axis tiles)
    ncol = Fx * Ty;
    a_sig = std::max(B * Ci * p->Sp[0] * Fs * p->Sp[1],            // S1[(b,ci)][zp][xt,fx][yp]
                     B * Co * Fs * Tys * (size_t)p->out_sp[0]);     // O2[(b,co)][xt,fx][yt,fy][z_out]
    b_sig = std::max(B * Ci * Fs * Tys * p->Sp[0],                  // S2[(b,ci)][xt,fx][yt,fy][zp]
                     B * Co * (size_t)p->out_sp[0] * Fs * (size_t)p->out_sp[1]);   // O1[(b,co)][z_out][xt,fx][y_out]
    a_w = Co * p->Cig * (size_t)p->kd[0] * Fx * (size_t)p->kd[1];
    b_w = Co * p->Cig * Fx * Ty * (size_t)p->kd[0];
  }
  p->planes = (planes_ok && best->T == 64) ? (nd == 3 ? 1 : 2) : 0;
  if (p->planes == 1) {
    const size_t ntile = (size_t)p->nxt * p->nyt;
    a_sig = B * Ci * (size_t)p->Sp[0] * ntile * fc::kPlCols;              // S[(b,ci)][zp][tile][col]
    b_sig = B * Co * (size_t)p->out_sp[0] * ntile * fc::kPlCols;          // O[(b,co)][z_out][tile][col]
  }
  p->ws_a = std::max(a_sig, a_w);
  p->ws_b = std::max(b_sig, b_w);
  p->workspace_bytes = (p->ws_a + p->ws_b) * sizeof(fc::f2);
  p->spectrum_bytes = (size_t)d.groups * p->nd_Cog_pad * (p->Cig_pad / 2) * ncol * best->T * sizeof(fc::f4);
  int rc = get_twiddles(best, &p->tw);
  if (rc == FC_OK) rc = get_twiddles(p->tx, &p->twx);
  if (rc == FC_OK && p->tm) rc = get_twiddles(p->tm, &p->twm);
  return rc;
}

static int plan_create_impl(const fc_desc* desc, const WgradSwap* sw, fc_plan** out_plan);

int fc_plan_create(const fc_desc* desc, fc_plan** out_plan) { return plan_create_impl(desc, nullptr, out_plan); }

static int plan_create_impl(const fc_desc* desc, const WgradSwap* sw, fc_plan** out_plan) {
  if (!desc || !out_plan) return fail(FC_ERR_INVALID, "null argument");
  (void)hipGetLastError();   // a stale sticky error of an earlier, unrelated call (e.g. an invalidated capture) is not this call's
  *out_plan = nullptr;
  const fc_desc& d = *desc;
  if (d.ndim < 1 || d.ndim > 3) return fail(FC_ERR_INVALID, "ndim must be 1, 2 or 3 (got %d)", d.ndim);
  if (d.dtype != FC_F32 && d.dtype != FC_F64) return fail(FC_ERR_UNSUPPORTED, "dtype must be FC_F32 or FC_F64");
  if (d.batch < 1 || d.in_channels < 1 || d.out_channels < 1 || d.groups < 1)
    return fail(FC_ERR_INVALID, "batch, channels and groups must be positive");
  if (d.in_channels % d.groups || d.out_channels % d.groups)
    return fail(FC_ERR_INVALID, "in_channels (%lld) and out_channels (%lld) must be divisible by groups (%lld)",
                (long long)d.in_channels, (long long)d.out_channels, (long long)d.groups);
  if (d.padding_mode < 0 || d.padding_mode > 3) return fail(FC_ERR_INVALID, "unknown padding_mode %d", d.padding_mode);

  fc_plan* p = new fc_plan();
  std::memset(p, 0, sizeof *p);
  p->d = d;
  p->nd = d.ndim;
  if (sw) { p->swap = 1; p->sw_B = sw->B; p->sw_Cig = sw->Cig; p->sw_Cog = sw->Cog; p->sw_g = sw->g; }
  if (d.transposed && d.padding_mode != FC_PAD_CONSTANT) {
    delete p;
    return fail(FC_ERR_INVALID, "a transposed plan supports zero padding only");
  }
  // A cyclic transform of length T >= Sp holds the whole padded axis.  With ZERO padding a shorter one does: output n reads
  // the padded positions n .. n+kd-1; those past T wrap to the head of the tile, and the result is unchanged when both the
  // true sample (position >= T) and the one wrapped in (position - T) are zero -- all data inside the tile (T >= padl + size)
  // and the wrapped range inside the left padding (T >= size + padr).  The input gradient of an unpadded convolution is the
  // case that matters: its padded axis is size + 2(kd-1) = out + kd - 1 long, just past the power of two the image has, and
  // out itself is enough (cfgB dX: 512 instead of 1024-point rows and one 512-point column tile instead of several).
  const char* zw_env = getenv("FFTCONV_ZEROWRAP");
  const bool zero_wrap = !zw_env || atoi(zw_env) != 0;
  auto set_need = [&](int i) {
    int64_t need = p->Sp[i];
    const int64_t size_eff = (d.spatial[i] - 1) * p->up[i] + 1;
    const int64_t padr = p->Sp[i] - p->padl[i] - size_eff;
    if (zero_wrap && d.padding_mode == FC_PAD_CONSTANT && p->padl[i] >= 0 && padr >= 0)
      need = std::min<int64_t>(need, std::max<int64_t>(std::max<int64_t>(p->Lf[i], p->kd[i]), std::max<int64_t>(p->padl[i] + size_eff, size_eff + padr)));
    p->need[i] = (int)need;
  };
  for (int i = 0; i < d.ndim; ++i) {
    if (d.spatial[i] < 1 || d.kernel[i] < 1 || d.stride[i] < 1 || d.dilation[i] < 1 || d.padding[i] < 0 ||
        (d.transposed && d.output_padding[i] < 0)) {
      delete p;
      return fail(FC_ERR_INVALID, "axis %d: spatial/kernel/stride/dilation must be >= 1 and padding >= 0", i);
    }
    p->kd[i] = (d.kernel[i] - 1) * d.dilation[i] + 1;
    if (d.transposed) {
      // functional.py:126-154: spread by the stride, full correlation, keep out samples from `padding`
      const int64_t out = (d.spatial[i] - 1) * d.stride[i] - 2 * d.padding[i] + p->kd[i] - 1 + d.output_padding[i] + 1;
      if (out < 1) {
        delete p;
        return fail(FC_ERR_INVALID, "axis %d: transposed output extent %lld is not positive", i, (long long)out);
      }
      p->out_sp[i] = out;
      p->padl[i] = (int)(p->kd[i] - 1 - d.padding[i]);
      p->up[i] = (int)d.stride[i];
      p->ostride[i] = 1;
      p->Sp[i] = (int)(out + p->kd[i] - 1);
      p->Lf[i] = (int)out;
      set_need(i);
      continue;
    }
    const int64_t span = d.spatial[i] + 2 * d.padding[i] - p->kd[i];
    if (span < 0) {
      delete p;
      return fail(FC_ERR_INVALID, "axis %d: dilated kernel extent %lld is larger than the padded input %lld", i,
                  (long long)p->kd[i], (long long)(d.spatial[i] + 2 * d.padding[i]));
    }
    p->out_sp[i] = span / d.stride[i] + 1;
    p->padl[i] = (int)d.padding[i];
    p->up[i] = 1;
    p->ostride[i] = (int)d.stride[i];
    p->Sp[i] = (int)(d.spatial[i] + 2 * d.padding[i]);
    p->Lf[i] = (int)(span + 1);
    if (sw && p->out_sp[i] > sw->keep[i]) {
      // weight gradient: only the first k lags are taps of dW (the input samples a strided convolution never reached
      // would give more); the passes then read no further than those lags need
      p->out_sp[i] = sw->keep[i];
      p->Lf[i] = (int)((sw->keep[i] - 1) * d.stride[i] + 1);
      p->Sp[i] = (int)(p->Lf[i] + p->kd[i] - 1);
    }
    set_need(i);
    if (d.padding_mode == FC_PAD_REFLECT && d.padding[i] >= d.spatial[i]) {
      delete p;
      return fail(FC_ERR_INVALID, "axis %d: reflect padding (%lld) must be smaller than the input size (%lld)", i,
                  (long long)d.padding[i], (long long)d.spatial[i]);
    }
    if (d.padding_mode == FC_PAD_CIRCULAR && d.padding[i] > d.spatial[i]) {
      delete p;
      return fail(FC_ERR_INVALID, "axis %d: circular padding (%lld) must not exceed the input size (%lld)", i,
                  (long long)d.padding[i], (long long)d.spatial[i]);
    }
  }
  set_channel_layout(p, (int)d.groups, (int)(d.in_channels / d.groups), (int)(d.out_channels / d.groups));

  int rc;
  if (d.dtype == FC_F64) {
    // float64: direct time-domain kernel (direct_f64.hip); the "kernel spectrum" is the weight tensor itself
    size_t nw = (size_t)(d.transposed ? d.in_channels : d.out_channels) * (size_t)((d.transposed ? d.out_channels : d.in_channels) / d.groups);
    for (int i = 0; i < d.ndim; ++i) nw *= (size_t)d.kernel[i];
    p->spectrum_bytes = nw * sizeof(double);
    p->workspace_bytes = 0;
    p->tile = nullptr;
    rc = FC_OK;
    // 1-D, not transposed, at least 16 taps, dilated extent within half of a 2048-point tile: the FFT path in double
    // precision (fft_f64.hip).  FFTCONV_F64_FFT=0 keeps the direct kernel (A/B runs, tests).
    const char* env = getenv("FFTCONV_F64_FFT");
    if ((!env || atoi(env) != 0) && d.ndim == 1 && !d.transposed && d.kernel[0] >= 16 && p->kd[0] <= 1025 &&
        (int64_t)d.batch * d.groups * ((p->out_sp[0] + 255) / 256) < 0x40000000) {
      int T = 256;
      while (T < 2 * p->kd[0] && T < 2048) T *= 2;
      // a longer tile wastes less of itself on the overlap; taken while the launch still has two workgroups per CU
      while (T < 2048) {
        const int64_t V2 = 2 * T - p->kd[0] + 1, tiles2 = (p->Lf[0] + V2 - 1) / V2;
        if (d.batch * d.groups * ((p->Cog + 7) / 8) * tiles2 < 512) break;
        T *= 2;
      }
      p->f64_T = T;
      p->f64_V = (int)(T - p->kd[0] + 1);
      p->f64_ntiles = (int)((p->Lf[0] + p->f64_V - 1) / p->f64_V);
      // output channels per workgroup: 8, fewer (even) while the launch would leave CUs idle -- a workgroup's life is its
      // transforms in a row (Cig/2 forward + cob/2 inverse), so small launches gain from more, shorter workgroups
      int cob = 8;
      while (cob > 2 && d.batch * d.groups * ((p->Cog + cob - 1) / cob) * p->f64_ntiles < 256) cob /= 2;   // (fewer workgroups than CUs)
      p->f64_cob = std::min(cob, std::max(p->Cog, 1));
      p->Lfull = p->Lf[0];
      p->spectrum_bytes = (size_t)d.out_channels * p->Cig * T * 2 * sizeof(double);
    }
  } else if (d.ndim == 1) rc = plan_1d(p);
  else rc = plan_nd(p);
  if (rc != FC_OK) { delete p; return rc; }
  *out_plan = p;
  return FC_OK;
}

void fc_plan_destroy(fc_plan* plan) {
  if (!plan) return;
  if (plan->d_items) (void)hipFree(plan->d_items);
  delete plan;
}

int fc_output_shape(const fc_plan* plan, int64_t out_spatial[3]) {
  if (!plan || !out_spatial) return fail(FC_ERR_INVALID, "null argument");
  for (int i = 0; i < 3; ++i) out_spatial[i] = i < plan->nd ? plan->out_sp[i] : 1;
  return FC_OK;
}

size_t fc_kernel_spectrum_bytes(const fc_plan* plan) { return plan ? plan->spectrum_bytes : 0; }
size_t fc_workspace_bytes(const fc_plan* plan) { return plan ? plan->workspace_bytes : 0; }
int fc_plan_tile(const fc_plan* plan) { return plan ? (plan->tile ? plan->tile->T : plan->f64_T) : 0; }

// ---- 1-D weight gradient
namespace {
struct WgradGeom {
  const fc::TileImpl* t;
  int kd_seg, seg_taps, nseg, V, ntiles, nob, nib, Cig, Cog, n_items, ipw, slices, pad, diag;
};
int wgrad_geometry(const fc_desc& d, WgradGeom* g) {
  if (d.ndim != 1 || d.dtype != FC_F32 || d.transposed || d.stride[0] < 1 || d.stride[0] > 64 || d.groups < 1) return 0;
  if (d.batch < 1 || d.in_channels % d.groups || d.out_channels % d.groups) return 0;
  const int64_t Cig = d.in_channels / d.groups, Cog = d.out_channels / d.groups;
  if (Cig > 64 || Cog > 64) return 0;      // every 4 x 4 channel block repeats the transforms of its rows: beyond this the plan path wins
  const int64_t kd = (d.kernel[0] - 1) * d.dilation[0] + 1;
  const fc::TileImpl* t = find_tile(1024);
  if (!t || !t->wgrad1d || d.padding[0] < 0 || d.dilation[0] > 512) return 0;
  if (d.spatial[0] + 2 * d.padding[0] - kd < 0) return 0;
  const int64_t Lout = (d.spatial[0] + 2 * d.padding[0] - kd) / d.stride[0] + 1;
  const int64_t Lext = (Lout - 1) * d.stride[0] + 1;       // the gradient row spread over the stride's grid
  if (d.padding_mode == FC_PAD_REFLECT && d.padding[0] >= d.spatial[0]) return 0;
  if (d.padding_mode == FC_PAD_CIRCULAR && d.padding[0] > d.spatial[0]) return 0;
  if ((int64_t)d.batch * d.in_channels * d.spatial[0] * 4 >= ((int64_t)1 << 32) ||
      (int64_t)d.batch * d.out_channels * Lout * 4 >= ((int64_t)1 << 32)) return 0;
  // the lags of one launch fit half a tile; longer kernels run in segments of taps (x read further in)
  const int64_t ks = std::min<int64_t>(d.kernel[0], kd <= 768 ? d.kernel[0] : 512 / d.dilation[0] + 1);
  const int64_t kd_seg = (ks - 1) * d.dilation[0] + 1;
  const int64_t nseg = (d.kernel[0] + ks - 1) / ks;
  if (nseg > 64) return 0;
  const int64_t V = (t->T - kd_seg + 1) / d.stride[0] * d.stride[0];      // tiles start on the stride's grid
  if (V < 1) return 0;
  const int64_t ntiles = (Lext + V - 1) / V, n_items = (int64_t)d.batch * ntiles;
  if (n_items > 0x3fffffff) return 0;
  int cus = 256;
  if (!current_device_cus(&cus)) return 0;
  const char* diag_env = getenv("FFTCONV_DIAG");        // read per call, like the plan-creation knobs (not frozen at first use)
  const bool diag_on = !diag_env || atoi(diag_env) != 0;
  g->diag = diag_on && d.groups == d.in_channels && d.groups == d.out_channels && d.groups % 8 == 0 &&
            t->wgrad1d_diag != nullptr;
  const int nb = g->diag ? 1 : t->wgrad_nb;
  g->t = t; g->kd_seg = (int)kd_seg; g->seg_taps = (int)ks; g->nseg = (int)nseg; g->V = (int)V; g->ntiles = (int)ntiles;
  g->Cig = (int)Cig; g->Cog = (int)Cog;
  g->nob = (int)(Cog + 3) / 4; g->nib = (int)(Cig + 3) / 4; g->n_items = (int)n_items; g->pad = (int)d.padding[0];
  const int64_t types = g->diag ? d.groups / 8 : (int64_t)d.groups * g->nob * g->nib;
  // two workgroups per CU; every slice costs one inverse transform and one partial result, so a slice
  // gets at least 4 iterations of work
  int64_t slices = std::max<int64_t>(1, (2 * (int64_t)cus + types - 1) / types);
  slices = std::min<int64_t>(slices, std::max<int64_t>(1, n_items / (4 * nb)));
  int64_t ipw = (n_items + slices - 1) / slices;
  ipw = (ipw + nb - 1) / nb * nb;
  g->ipw = (int)ipw;
  g->slices = (int)((n_items + ipw - 1) / ipw);
  return 1;
}
}  // namespace

int fc_wgrad1d_slices(const fc_desc* desc) {
  if (!desc) return 0;
  WgradGeom g;
  if (!wgrad_geometry(*desc, &g)) return 0;
  Twiddles tw;                                   // first use on this device: build the tables here, not in the launch
  if (get_twiddles(g.t, &tw) != FC_OK) return 0;
  return g.slices;
}

int fc_wgrad1d_db_supported(const fc_desc* desc) {
  if (!desc) return 0;
  WgradGeom g;
  return wgrad_geometry(*desc, &g) && !g.diag;
}

int fc_wgrad1d(const fc_desc* desc, const float* x, const float* dy, float* partial, int slices, void* hip_stream) {
  return fc_wgrad1d_db(desc, x, dy, partial, nullptr, 0, slices, hip_stream);
}

int fc_wgrad1d_db(const fc_desc* desc, const float* x, const float* dy, float* partial, float* db_partial,
                  long long slice_stride, int slices, void* hip_stream) {
  if (!desc || !x || !dy || !partial) return fail(FC_ERR_INVALID, "null argument");
  (void)hipGetLastError();   // a stale sticky error of an earlier, unrelated call (e.g. an invalidated capture) is not this call's
  WgradGeom g;
  if (!wgrad_geometry(*desc, &g)) return fail(FC_ERR_UNSUPPORTED, "fc_wgrad1d does not cover this shape");
  if (db_partial && g.diag) return fail(FC_ERR_UNSUPPORTED, "the depthwise weight-gradient kernel has no bias-gradient output "
                                        "(ask fc_wgrad1d_db_supported first)");
  {
    const long long dense = (long long)desc->out_channels * (desc->in_channels / desc->groups) * desc->kernel[0];
    if (slice_stride == 0) slice_stride = dense;
    if (slice_stride < dense) return fail(FC_ERR_INVALID, "slice_stride %lld is smaller than one partial tensor (%lld floats)", slice_stride, dense);
  }
  if (slices != g.slices) return fail(FC_ERR_INVALID, "partial holds %d slices, the plan needs %d", slices, g.slices);
  Twiddles tw;
  int rc = find_twiddles(g.t, &tw);
  if (rc != FC_OK) return rc;
  const fc_desc& d = *desc;
  fc::WGradArgs a;
  a.x = x; a.dy = dy; a.part = partial; a.twA = tw.twA; a.twB = tw.twB;
  a.B = (int)d.batch; a.Cin = (int)d.in_channels; a.Cout = (int)d.out_channels;
  a.G = g.diag ? (int)(d.groups / 8) : (int)d.groups;
  a.Cig = g.Cig; a.Cog = g.Cog; a.L = (int)d.spatial[0]; a.pad = g.pad; a.pad_mode = d.padding_mode;
  const int64_t kd = (d.kernel[0] - 1) * d.dilation[0] + 1;
  a.Lout = (int)((d.spatial[0] + 2 * d.padding[0] - kd) / d.stride[0] + 1);
  a.stride = (int)d.stride[0]; a.Lext = (a.Lout - 1) * a.stride + 1;
  a.dil = (int)d.dilation[0]; a.V = g.V; a.ntiles = g.ntiles;
  a.n_items = g.n_items; a.items_per_slice = g.ipw; a.nob = g.nob; a.nib = g.nib;
  a.scale = 1.0f / (4.0f * (float)g.t->T);
  a.Krow = (int)d.kernel[0];
  a.part_stride = slice_stride;
  const int64_t grid = g.diag ? (int64_t)g.slices * (d.groups / 8) : (int64_t)g.slices * d.groups * g.nob * g.nib;
  if (grid > 0x7fffffff) return fail(FC_ERR_UNSUPPORTED, "grid too large");
  for (int j = 0; j < g.nseg; ++j) {
    a.tap0 = j * g.seg_taps;
    a.K = std::min(g.seg_taps, (int)d.kernel[0] - a.tap0);
    a.pos_shift = a.tap0 * (int)d.dilation[0];
    a.dbpart = j == 0 ? db_partial : nullptr;      // every segment sees all of dY: the bias gradient comes from the first
    if (g.diag) FC_HIP(g.t->wgrad1d_diag(a, (int)grid, (hipStream_t)hip_stream));
    else FC_HIP(g.t->wgrad1d(a, (int)grid, (hipStream_t)hip_stream));
  }
  return FC_OK;
}

int fc_plan_layout(const fc_plan* plan, int32_t layout[8]) {
  if (!plan || !layout) return fail(FC_ERR_INVALID, "null argument");
  const fc_plan& p = *plan;
  layout[0] = p.tile ? p.tile->T : 0;
  layout[1] = p.ph; layout[2] = p.nseg; layout[3] = p.seg_taps;
  layout[4] = p.diag; layout[5] = p.bd_gs; layout[6] = p.dense ? 2 : p.wide; layout[7] = p.pers_nb;
  if (p.nd != 1) {   // N-d: the spectrum is laid out over the row / middle-axis transform lengths too
    layout[1] = p.tx ? p.tx->T : 0; layout[2] = p.tm ? p.tm->T : 0; layout[3] = p.nd_cob;   // (x tiles share one kernel spectrum)
    layout[4] = layout[5] = layout[6] = 0;
    layout[7] = p.planes;      // 1 / 2: the thread-per-sequence column pass (3-D plane-major / 2-D); same spectrum bytes either way
  }
  return FC_OK;
}

long long fc_debug_grid(const fc_plan* plan) {
  if (!plan || plan->d.dtype != FC_F32) return 0;
  if (plan->planes) {   // upper bound of colz's grid (one batch item per workgroup)
    const long long ncol = plan->planes == 1 ? (long long)fc::kPlCols * plan->nxt * plan->nyt : plan->Fxt;
    return (long long)plan->d.batch * plan->ntiles * (plan->nd_Cog_pad / plan->nd_cob) * plan->d.groups * ((ncol / 16 + 7) / 8) * 8;
  }
  if (plan->nd != 1) {   // upper bound of the fused column pass's grid (one batch item per workgroup)
    const long long ncol = plan->nd == 2 ? plan->Fxt : (long long)plan->Fxt * plan->tm->T * plan->nyt;
    return (long long)plan->d.batch * plan->ntiles * (plan->nd_Cog_pad / plan->nd_cob) * plan->d.groups * ((ncol + 7) / 8) * 8;
  }
  if (plan->pers_nb) return (long long)plan->pers_items * 16;   // one record per wave (up to 16) of every work item
  return (long long)plan->d.batch * plan->ntiles * plan->n_ochunks * plan->G;
}

static void fill_f64_args(const fc_plan& p, fc::FftF64Args* a) {
  a->B = (int)p.d.batch; a->Cin = (int)p.d.in_channels; a->Cout = (int)p.d.out_channels; a->G = (int)p.d.groups;
  a->Cig = p.Cig; a->Cog = p.Cog;
  a->L = (int)p.d.spatial[0]; a->pad = (int)p.d.padding[0]; a->pad_mode = p.d.padding_mode;
  a->K = (int)p.d.kernel[0]; a->dil = (int)p.d.dilation[0]; a->stride = (int)p.d.stride[0];
  a->T = p.f64_T; a->V = p.f64_V; a->ntiles = p.f64_ntiles; a->Lfull = p.Lf[0]; a->Lout = (int)p.out_sp[0];
  a->cob = p.f64_cob; a->n_ochunks = (p.Cog + p.f64_cob - 1) / p.f64_cob;
}

int fc_transform_kernel(const fc_plan* plan, const float* weight, void* w_hat, void* workspace, void* hip_stream) {
  if (!plan || !weight || !w_hat) return fail(FC_ERR_INVALID, "null argument");
  (void)hipGetLastError();   // a stale sticky error of an earlier, unrelated call (e.g. an invalidated capture) is not this call's
  (void)workspace;
  hipStream_t st = (hipStream_t)hip_stream;
  const fc_plan& p = *plan;
  if (p.d.dtype == FC_F64 && p.f64_T) {
    fc::FftF64Args a{};
    fill_f64_args(p, &a);
    a.w = (const double*)weight; a.wspec = (double2*)w_hat;
    FC_HIP(fc::launch_fft_f64(0, a, st));
    return FC_OK;
  }
  if (p.d.dtype == FC_F64) {
    FC_HIP(hipMemcpyAsync(w_hat, weight, p.spectrum_bytes, hipMemcpyDeviceToDevice, st));
    return FC_OK;
  }
  if (p.nd == 1) {
    fc::Spec1dArgs a;
    a.w = weight;
    a.wspec = (fc::f4*)w_hat;
    a.twA = p.tw.twA;
    a.twB = p.tw.twB;
    a.G = p.G; a.Cig = p.Cig; a.Cog = p.Cog; a.Cig_pad = p.Cig_pad; a.Cog_pad = p.Cog_pad;
    if (p.diag) {   // depthwise: (C, 1, K) read as one output row over C inputs -> [C/2 pairs][T/2] float4
      a.G = 1; a.Cog = 1; a.Cog_pad = 1; a.Cig = (int)p.d.in_channels; a.Cig_pad = (int)round_up(p.d.in_channels, 8);
    }
    a.gs = p.bd_gs;
    a.dil = p.ph > 1 ? 1 : (int)p.d.dilation[0];
    a.nseq = a.G * a.Cog_pad * (a.Cig_pad / 2);
    a.transposed = p.d.transposed;
    a.Krow = (int)p.d.kernel[0];
    {
      const unsigned long long wb = 4ull * (unsigned long long)(p.d.transposed ? p.d.in_channels : p.d.out_channels) *
                                    (unsigned long long)((p.d.transposed ? p.d.out_channels : p.d.in_channels) / p.d.groups) * (unsigned long long)p.d.kernel[0];
      a.w_bytes = wb < 0x7F000000ull ? (unsigned)wb : 0u;   // (dead offsets are bit 31 minus at most a few KB: they must stay outside)
    }
    const int per_wg = p.tile->NT / (p.tile->P * p.tile->S);
    const int grid = (a.nseq + per_wg - 1) / per_wg;
    if (p.dense) {
      // transform into the scratch area in the fused kernels' layout, then re-lay bin-major for the GEMM
      if (!workspace) return fail(FC_ERR_INVALID, "workspace is NULL but %zu bytes are required", p.workspace_bytes);
      a.k0 = 0; a.K = (int)p.d.kernel[0]; a.wspec = (fc::f4*)workspace;
      FC_HIP(p.tile->spec1d(a, grid, p.lds_spec, st));
      fc::DenseSpecArgs ds;
      ds.wspec = (const fc::f4*)workspace; ds.Hd = (fc::f2*)w_hat; ds.G = p.G; ds.Kc = p.Cig_pad; ds.Nc = p.Cog_pad; ds.T = p.tile->T;
      FC_HIP(p.tile->dense_spec(ds, st));
      return FC_OK;
    }
    for (int j = 0; j < p.nseg; ++j) {
      a.k0 = j * p.seg_taps;
      a.K = std::min(p.seg_taps, (int)p.d.kernel[0] - a.k0);
      a.wspec = (fc::f4*)((char*)w_hat + (size_t)j * p.seg_spectrum_bytes);
      FC_HIP(p.tile->spec1d(a, grid, p.lds_spec, st));
    }
    return FC_OK;
  }
  // ---- 2-D / 3-D: the separable passes, fed from the dilated taps
  if (p.workspace_bytes && !workspace) return fail(FC_ERR_INVALID, "workspace is NULL but %zu bytes are required", p.workspace_bytes);
  fc::f2* wsA = (fc::f2*)workspace;
  fc::f2* wsB = wsA + p.ws_a;
  const int nd = p.nd;
  const int Co = (int)p.d.out_channels;
  // phantom channels (padding of the channel counts up to the chunk size) must read as zero; without any, every
  // entry of the spectrum is written by the passes below and the fill (6 us per call on a 2-D training step) is skipped
  if (p.Cig_pad != p.Cig || p.nd_Cog_pad != p.Cog) FC_HIP(hipMemsetAsync(w_hat, 0, p.spectrum_bytes, st));
  fc::RowsR2CArgs r{};
  r.src = weight; r.dst = wsA; r.twA = p.twx.twA; r.twB = p.twx.twB; r.from_kernel = 1;
  r.kx = (int)p.d.kernel[nd - 1]; r.dx = (int)p.d.dilation[nd - 1];
  r.ky = (int)p.d.kernel[nd - 2]; r.dy = (int)p.d.dilation[nd - 2];
  r.kz = nd == 3 ? (int)p.d.kernel[0] : 1; r.dz = nd == 3 ? (int)p.d.dilation[0] : 1;
  r.NA = Co * p.Cig; r.NC = nd == 3 ? (int)p.kd[0] : 1; r.NY = (int)p.kd[nd - 2]; r.NYa = r.NY;
  r.SZ = r.kz; r.SY = r.ky; r.SX = r.kx; r.Fx = p.Fx;
  r.nxt = 1; r.Vx = 0;                       // the kernel sits in the first x tile
  r.transposed = p.d.transposed; r.Cig = p.Cig; r.Cog = p.Cog;
  if (p.swap) {   // "kernel" = the output gradient (B, g*Cog, *Lout), read as ((g, o), b): image o_all*B + b sits at b*(g*Cog) + o_all
    r.im.on = 1; r.im.n1 = 1; r.im.n2 = (int)p.sw_B; r.im.s0 = 1; r.im.s1 = 0; r.im.s2 = p.sw_g * p.sw_Cog;
    // a tensor as large as the signal: read it through the signal's index maps (taps spread by the dilation = a source
    // spread over a grid of that step, nothing in front), which have the unrolled zero-padding path the tap loop lacks
    r.from_kernel = 0;
    auto tmap = [&](int64_t taps, int64_t dil) { fc::AxisMap m; m.size = (int)taps; m.pad = 0; m.mode = FC_PAD_CONSTANT; m.up = (int)dil; return m; };
    r.mx = tmap(r.kx, r.dx); r.my = tmap(r.ky, r.dy); r.mz = tmap(r.kz, r.dz);
    const unsigned long long bytes = 4ull * (unsigned long long)r.NA * r.SZ * r.SY * r.SX;
    r.src_bytes = bytes < 0xFFFFFFFFull ? (unsigned)bytes : 0u;
  }
  FC_HIP(p.tx->rows_r2c(r, st));
  const float norm = 1.0f / ((float)p.tx->T * (float)p.tile->T * (nd == 3 ? (float)p.tm->T : 1.0f));
  fc::C2CArgs c{};
  c.Cig = p.Cig; c.Cog = p.Cog; c.Cig_pad = p.Cig_pad; c.Cog_pad = p.nd_Cog_pad; c.scale = norm;
  c.NV = 0; c.stride = 1; c.noff = 0;
  if (nd == 2) {
    // S1w[(o,i)][fx][y<Kd] -> wspec[..][fx][fy]
    c.src = wsA; c.dst = (fc::f2*)w_hat; c.twA = p.tw.twA; c.twB = p.tw.twB;
    c.NA = Co * p.Cig; c.NC = 1; c.NB = p.Fx; c.NLEN = (int)p.kd[0];
    c.sa = (long long)p.Fx * r.NYa; c.sc = 0; c.sb = r.NYa; c.store_mode = 1;
    FC_HIP(p.tile->c2c_fwd(c, st));
  } else {
    const int Ty = p.tm->T, Kz = (int)p.kd[0];
    // S1w[(o,i)][z<Kdz][fx][y<Kdy] -> S2w[(o,i)][fx][fy][z<Kdz]
    c.src = wsA; c.dst = wsB; c.twA = p.twm.twA; c.twB = p.twm.twB;
    c.NA = Co * p.Cig; c.NC = p.Fx; c.NB = Kz; c.NLEN = (int)p.kd[1];
    c.sa = (long long)Kz * p.Fx * r.NYa; c.sb = (long long)p.Fx * r.NYa; c.sc = r.NYa;
    c.ta = (long long)p.Fx * Ty * Kz; c.tc = (long long)Ty * Kz; c.tf = Kz; c.store_mode = 0;
    FC_HIP(p.tm->c2c_fwd(c, st));
    // S2w[(o,i)][(fx,fy)][z<Kdz] -> wspec[..][(fx,fy)][fz]
    c.src = wsB; c.dst = (fc::f2*)w_hat; c.twA = p.tw.twA; c.twB = p.tw.twB;
    c.NC = 1; c.NB = p.Fx * Ty; c.NLEN = Kz;
    c.sa = (long long)p.Fx * Ty * Kz; c.sc = 0; c.sb = Kz; c.store_mode = 1;
    FC_HIP(p.tile->c2c_fwd(c, st));
  }
  return FC_OK;
}

int fc_forward(const fc_plan* plan, const float* x, const void* w_hat, const float* bias, float* y, void* workspace,
               void* hip_stream) {
  return fc_forward_stamped(plan, x, w_hat, bias, y, workspace, hip_stream, nullptr);
}

int fc_forward_stamped(const fc_plan* plan, const float* x, const void* w_hat, const float* bias, float* y,
                       void* workspace, void* hip_stream, void* stamps) {
  if (!plan || !x || !w_hat || !y) return fail(FC_ERR_INVALID, "null argument");
  (void)hipGetLastError();   // a stale sticky error of an earlier, unrelated call (e.g. an invalidated capture) is not this call's
  (void)workspace;
  hipStream_t st = (hipStream_t)hip_stream;
  const fc_plan& p = *plan;
  if (p.d.has_bias && !bias) return fail(FC_ERR_INVALID, "plan was created with has_bias=1 but bias is NULL");
  if (p.d.dtype == FC_F64 && p.f64_T) {
    if (stamps) return fail(FC_ERR_UNSUPPORTED, "no timestamp hook in the float64 kernels");
    fc::FftF64Args a{};
    fill_f64_args(p, &a);
    a.x = (const double*)x; a.wspec = (double2*)const_cast<void*>(w_hat); a.bias = p.d.has_bias ? (const double*)bias : nullptr;
    a.y = (double*)y;
    FC_HIP(fc::launch_fft_f64(1, a, st));
    return FC_OK;
  }
  if (p.d.dtype == FC_F64) {
    if (stamps) return fail(FC_ERR_UNSUPPORTED, "no timestamp hook in the float64 kernel");
    fc::DirectF64Args a{};
    a.x = (const double*)x; a.w = (const double*)w_hat; a.bias = p.d.has_bias ? (const double*)bias : nullptr; a.y = (double*)y;
    a.B = (int)p.d.batch; a.Cin = (int)p.d.in_channels; a.Cout = (int)p.d.out_channels; a.G = (int)p.d.groups;
    a.pad_mode = p.d.padding_mode; a.transposed = p.d.transposed;
    for (int i = 0; i < 3; ++i) {          // axes right-aligned: leading axes of extent 1 for 1-D / 2-D
      const int ax = i - (3 - p.nd);
      const bool live = ax >= 0;
      a.S[i] = live ? (int)p.d.spatial[ax] : 1; a.K[i] = live ? (int)p.d.kernel[ax] : 1; a.O[i] = live ? (int)p.out_sp[ax] : 1;
      a.stride[i] = live ? (int)p.d.stride[ax] : 1; a.pad[i] = live ? (int)p.d.padding[ax] : 0; a.dil[i] = live ? (int)p.d.dilation[ax] : 1;
    }
    FC_HIP(fc::launch_direct_f64(a, st));
    return FC_OK;
  }
  if (p.nd == 1 && p.dense) {
    if (!workspace) return fail(FC_ERR_INVALID, "workspace is NULL but %zu bytes are required", p.workspace_bytes);
    fc::DenseArgs a{};
    const size_t NF = (size_t)p.tile->T / 2 + 1;
    a.x = x; a.y = y; a.bias = p.d.has_bias ? bias : nullptr; a.Hd = (const fc::f2*)w_hat;
    a.twA = p.tw.twA; a.twB = p.tw.twB;
    a.B = (int)p.d.batch; a.Cin = (int)p.d.in_channels; a.Cout = (int)p.d.out_channels; a.G = p.G;
    a.Cig = p.Cig; a.Cog = p.Cog; a.Kc = p.Cig_pad; a.Nc = p.Cog_pad;
    a.L = (int)p.d.spatial[0]; a.pad = p.padl[0]; a.pad_mode = p.d.padding_mode;
    a.V = p.V; a.ntiles = p.ntiles; a.Lfull = p.Lfull; a.Lout = (int)p.out_sp[0];
    a.cus = p.dense_cus;
    const int64_t M = p.d.batch * (int64_t)p.ntiles;
    for (int64_t m0 = 0; m0 < M; m0 += p.dense_mslab) {
      a.m0 = (int)m0; a.mcount = (int)std::min<int64_t>(p.dense_mslab, M - m0);
      a.X = (fc::f2*)workspace;
      a.Y = a.X + (size_t)p.G * NF * (size_t)a.mcount * (size_t)a.Kc;
      FC_HIP(p.tile->dense(0, a, st));
      FC_HIP(p.tile->dense(1, a, st));
      FC_HIP(p.tile->dense(2, a, st));
    }
    return FC_OK;
  }
  if (p.nd == 1) {
    fc::Conv1dArgs a;
    a.x = x; a.wspec = (const fc::f4*)w_hat; a.bias = p.d.has_bias ? bias : nullptr; a.y = y;
    a.twA = p.tw.twA; a.twB = p.tw.twB;
    a.B = (int)p.d.batch; a.Cin = (int)p.d.in_channels; a.Cout = (int)p.d.out_channels; a.G = p.G;
    a.Cig = p.Cig; a.Cog = p.Cog; a.Cig_pad = p.Cig_pad; a.Cog_pad = p.Cog_pad; a.cob = p.cob; a.n_ochunks = p.n_ochunks;
    a.L = (int)p.d.spatial[0]; a.pad = p.padl[0]; a.pad_mode = p.d.padding_mode; a.up = p.up[0]; a.ph = p.ph; a.slot_tiles = p.slot_tiles; a.diag = p.diag;
    a.ph2 = p.ph2;
    a.Kd = (int)p.kd[0]; a.V = p.V; a.ntiles = p.ntiles; a.Lfull = p.Lfull; a.Lout = (int)p.out_sp[0];
    a.stride = p.ostride[0]; a.accumulate = p.accumulate;
    a.ic_begin = 0; a.ic_end = p.Cig_pad / p.CB; a.add_out = 0;
    a.stamps = (unsigned long long*)stamps;
    a.segmented = p.nseg > 1; a.pos_shift = 0;
    if (p.pers_nb) {
      for (int j = 0; j < p.nseg; ++j) {
        fc::Conv1dPersArgs pa;
        a.pos_shift = j * p.seg_taps * (int)p.d.dilation[0];
        a.wspec = (const fc::f4*)((const char*)w_hat + (size_t)j * p.seg_spectrum_bytes);
        a.add_out = j > 0;
        if (j > 0) a.bias = nullptr;
        pa.c = a; pa.items = p.d_items; pa.n_items = p.pers_items;
        if (p.wide) FC_HIP(p.tile->conv1d_wide(pa, p.pers_grid, st));
        else FC_HIP(p.tile->conv1d_pers(p.pers_nb, pa, p.pers_grid, st));
      }
      return FC_OK;
    }
    const int64_t grid = (int64_t)a.B * a.ntiles * a.n_ochunks * a.G;
    if (grid > 0x7fffffff) return fail(FC_ERR_UNSUPPORTED, "grid too large");
    const int n_ichunks = p.Cig_pad / p.CB;
    if (p.chunk_launches) {
      for (int ic = 0; ic < n_ichunks; ++ic) {
        a.ic_begin = ic; a.ic_end = ic + 1; a.add_out = ic > 0;
        if (ic > 0) a.bias = nullptr;
        FC_HIP(p.tile->conv1d(p.CB, a, (int)grid, p.lds_conv, st));
      }
      return FC_OK;
    }
    for (int j = 0; j < p.nseg; ++j) {
      a.pos_shift = j * p.seg_taps * (int)p.d.dilation[0];
      a.wspec = (const fc::f4*)((const char*)w_hat + (size_t)j * p.seg_spectrum_bytes);
      a.ic_begin = 0; a.ic_end = n_ichunks; a.add_out = j > 0;
      if (j > 0) a.bias = nullptr;
      FC_HIP(p.tile->conv1d(p.CB, a, (int)grid, p.lds_conv, st));
    }
    return FC_OK;
  }
  // ---- 2-D / 3-D
  if (p.workspace_bytes && !workspace) return fail(FC_ERR_INVALID, "workspace is NULL but %zu bytes are required", p.workspace_bytes);
  fc::f2* wsA = (fc::f2*)workspace;
  fc::f2* wsB = wsA + p.ws_a;
  const int nd = p.nd;
  const int B = (int)p.d.batch, Ci = (int)p.d.in_channels, Co = (int)p.d.out_channels;
  auto amap = [&](int ax) { fc::AxisMap m; m.size = (int)p.d.spatial[ax]; m.pad = p.padl[ax]; m.mode = p.d.padding_mode; m.up = p.up[ax]; return m; };
  if (p.planes == 1) {
    // x (B,Ci,Z,Y,X) -> S[(b,ci)][zp][col] -> O[(b,co)][z_out][col] -> y; col = fx*64 + fy
    fc::PlaneFwdArgs f1{};
    f1.src = x; f1.dst = wsA; f1.twA = p.twx.twA; f1.twB = p.twx.twB;
    f1.mx = amap(2); f1.my = amap(1); f1.mz = amap(0);
    f1.SZ = (int)p.d.spatial[0]; f1.SY = (int)p.d.spatial[1]; f1.SX = (int)p.d.spatial[2]; f1.NZ = p.Sp[0];
    f1.nxt = p.nxt; f1.nyt = p.nyt; f1.Vx = p.Vx; f1.Vy = p.Vy;
    FC_HIP(p.tile->planes_fwd(f1, B * Ci, st));
    fc::ColZArgs cz{};
    cz.src = wsA; cz.wspec = (const fc::f4*)w_hat; cz.dst = wsB;
    cz.B = B; cz.Cin = Ci; cz.Cout = Co; cz.G = (int)p.d.groups; cz.Cig = p.Cig; cz.Cog = p.Cog; cz.Cog_pad = p.nd_Cog_pad;
    cz.cob = p.nd_cob; cz.n_ochunks = p.nd_Cog_pad / p.nd_cob;
    cz.NZ = p.Sp[0]; cz.NZo = (int)p.out_sp[0];
    cz.V = p.V; cz.ntiles = p.ntiles; cz.Lfull = p.Lfull; cz.stride = p.ostride[0];
    cz.ncol = fc::kPlCols * p.nxt * p.nyt; cz.hcol = fc::kPlCols;      // (the tiles of a plane share the spectrum's 2048 columns)
    cz.stamps = (unsigned long long*)stamps;        // profiling build of the column pass (scripts/phase_profile_nd.py)
    FC_HIP(p.tile->colz(cz, st));
    fc::PlaneInvArgs f3{};
    f3.src = wsB; f3.dst = y; f3.bias = p.d.has_bias ? bias : nullptr; f3.twA = p.twx.twA; f3.twB = p.twx.twB;
    f3.NZo = (int)p.out_sp[0]; f3.Cout = Co;
    f3.NVy = p.Lf[1]; f3.sy = p.ostride[1]; f3.Yo = (int)p.out_sp[1];
    f3.NVx = p.Lf[2]; f3.sx = p.ostride[2]; f3.Xo = (int)p.out_sp[2];
    f3.nxt = p.nxt; f3.nyt = p.nyt; f3.Vx = p.Vx; f3.Vy = p.Vy;
    FC_HIP(p.tile->planes_inv(f3, B * Co, st));
    return FC_OK;
  }
  fc::RowsR2CArgs r{};
  r.src = x; r.dst = wsA; r.twA = p.twx.twA; r.twB = p.twx.twB; r.from_kernel = 0;
  r.mx = amap(nd - 1); r.my = amap(nd - 2);
  // a one-plane padded z axis still goes through its map: a transposed plan can crop its only source plane away
  if (nd == 3) r.mz = amap(0);
  else { r.mz.size = 1; r.mz.pad = 0; r.mz.mode = FC_PAD_CONSTANT; r.mz.up = 1; }
  r.kx = r.ky = r.kz = r.dx = r.dy = r.dz = 1; r.transposed = 0; r.Cig = p.Cig; r.Cog = p.Cog;
  r.NA = B * Ci; r.NC = nd == 3 ? p.Sp[0] : 1; r.NY = p.Sp[nd - 2]; r.NYa = r.NY;
  r.SZ = nd == 3 ? (int)p.d.spatial[0] : 1; r.SY = (int)p.d.spatial[nd - 2]; r.SX = (int)p.d.spatial[nd - 1]; r.Fx = p.Fx;
  r.nxt = p.nxt; r.Vx = p.Vx;
  const int Fs = p.Fxt;                       // signal-side bin columns per plane (all x tiles)
  {
    const unsigned long long bytes = 4ull * (unsigned long long)B * Ci * r.SZ * r.SY * r.SX;
    r.src_bytes = bytes < 0xFFFFFFFFull ? (unsigned)bytes : 0u;
  }
  r.rowmajor = p.planes == 2;
  if (p.swap) {   // signal = x (B, g*Cig, *S) read as (i, (g, b)): image (i*g + gi)*B + b sits at b*(g*Cig) + gi*Cig + i
    r.im.on = 1; r.im.n1 = (int)p.sw_g; r.im.n2 = (int)p.sw_B; r.im.s0 = 1; r.im.s1 = p.sw_Cig; r.im.s2 = p.sw_g * p.sw_Cig;
  }
  FC_HIP(p.tx->rows_r2c(r, st));

  fc::FusedCArgs f{};
  f.wspec = (const fc::f4*)w_hat; f.twA = p.tw.twA; f.twB = p.tw.twB;
  f.B = B; f.Cin = Ci; f.Cout = Co; f.G = (int)p.d.groups; f.Cig = p.Cig; f.Cog = p.Cog;
  f.Cig_pad = p.Cig_pad; f.Cog_pad = p.nd_Cog_pad; f.cob = p.nd_cob; f.n_ochunks = p.nd_Cog_pad / p.nd_cob;
  f.Kd = (int)p.kd[0]; f.V = p.V; f.ntiles = p.ntiles; f.Lfull = p.Lfull; f.NVo = (int)p.out_sp[0];
  f.stride = p.ostride[0]; f.accumulate = p.accumulate; f.NLEN = p.Sp[0];
  f.stamps = (unsigned long long*)stamps;

  fc::RowsC2RArgs o{};
  o.dst = y; o.bias = p.d.has_bias ? bias : nullptr; o.twA = p.twx.twA; o.twB = p.twx.twB;
  o.NA = B * Co; o.Fx = p.Fx; o.Cout = Co; o.nxt = p.nxt; o.Vx = p.Vx;
  f.wfx = p.Fx; f.wty = nd == 3 ? p.tm->T : 1; f.wrep = nd == 3 ? p.nyt : 1; f.wncol = f.wfx * f.wty;
  o.NV = p.Lf[nd - 1]; o.stride = p.ostride[nd - 1]; o.Xo = (int)p.out_sp[nd - 1];
  o.NY = (int)p.out_sp[nd - 2]; o.NYa = o.NY;
  if (p.swap) {   // output (i, (g, o), *k) written as dW ((g, o), i, *k): image i*(g*Cog) + o_all goes to o_all*Cig + i
    o.im.on = 1; o.im.n1 = 1; o.im.n2 = (int)(p.sw_g * p.sw_Cog); o.im.s0 = 1; o.im.s1 = 0; o.im.s2 = p.sw_Cig;
  }

  if (nd == 2 && p.planes == 2) {
    // x (B,Ci,Y,X) -> S[(b,ci)][yp][fx] (rows_r2c above, rows as they are) -> O[(b,co)][y_out][fx] -> y
    fc::ColZArgs cz{};
    cz.src = wsA; cz.wspec = (const fc::f4*)w_hat; cz.dst = wsB;
    cz.B = B; cz.Cin = Ci; cz.Cout = Co; cz.G = (int)p.d.groups; cz.Cig = p.Cig; cz.Cog = p.Cog; cz.Cog_pad = p.nd_Cog_pad;
    cz.cob = p.nd_cob; cz.n_ochunks = p.nd_Cog_pad / p.nd_cob;
    cz.NZ = p.Sp[0]; cz.NZo = (int)p.out_sp[0];
    cz.V = p.V; cz.ntiles = p.ntiles; cz.Lfull = p.Lfull; cz.stride = p.ostride[0];
    cz.ncol = Fs; cz.hcol = p.Fx;        // (all x tiles of the signal; they share the Tx/2 spectrum columns)
    cz.stamps = (unsigned long long*)stamps;
    FC_HIP(p.tile->colz(cz, st));
    o.src = wsB; o.NC = 1; o.rowmajor = 1;
    FC_HIP(p.tx->rows_c2r(o, st));
  } else if (nd == 2) {
    f.src = wsA; f.dst = wsB; f.ncol = Fs;
    FC_HIP(p.tile->fusedc(p.CB, f, st));
    o.src = wsB; o.NC = 1;
    FC_HIP(p.tx->rows_c2r(o, st));
  } else {
    const int Ty = p.tm->T, Szp = p.Sp[0], Syp = p.Sp[1], Lzo = (int)p.out_sp[0], Lyo = (int)p.out_sp[1];
    fc::C2CArgs c{};
    c.scale = 1.f; c.store_mode = 0; c.twA = p.twm.twA; c.twB = p.twm.twB;
    // S1[(b,ci)][zp][fx][yp] -> S2[(b,ci)][fx][yt,fy][zp]   (one launch per middle-axis tile yt)
    const int nyt = p.nyt, Vy = p.Vy;
    const long long Tys = (long long)nyt * Ty;
    c.NA = B * Ci; c.NC = Fs; c.NB = Szp;
    c.sa = (long long)Szp * Fs * Syp; c.sb = (long long)Fs * Syp; c.sc = Syp;
    c.ta = (long long)Fs * Tys * Szp; c.tc = Tys * Szp; c.tf = Szp;
    c.NV = 0; c.stride = 1; c.noff = 0;
    for (int yt = 0; yt < nyt; ++yt) {
      c.src = wsA + (size_t)yt * Vy; c.dst = wsB + (size_t)yt * Ty * Szp;
      c.NLEN = std::min(Ty, Syp - yt * Vy);
      FC_HIP(p.tm->c2c_fwd(c, st));
    }
    f.src = wsB; f.dst = wsA; f.ncol = (int)(Fs * Tys);
    FC_HIP(p.tile->fusedc(p.CB, f, st));
    // O2[(b,co)][fx][yt,fy][z_out] -> O1[(b,co)][z_out][fx][y_out]
    c.NA = B * Co; c.NC = Fs; c.NB = Lzo;
    c.sa = (long long)Fs * Tys * Lzo; c.sc = Tys * Lzo; c.sb = Lzo;
    c.ta = (long long)Lzo * Fs * Lyo; c.tb = (long long)Fs * Lyo; c.tc = Lyo;
    c.stride = p.ostride[1];
    for (int yt = 0; yt < nyt; ++yt) {
      c.src = wsA + (size_t)yt * Ty * Lzo; c.dst = wsB;
      c.noff = yt * Vy; c.NV = std::min(Vy, p.Lf[1] - yt * Vy);
      FC_HIP(p.tm->c2c_inv(c, st));
    }
    o.src = wsB; o.NC = Lzo;
    FC_HIP(p.tx->rows_c2r(o, st));
  }
  return FC_OK;
}

// ---- N-d weight gradient (the reference's dW comes from autograd through its
// rfftn / einsum / irfftn graph, tests/test_functional.py:111-117): dW[(g,o)][i][k] = sum_b sum_t dY[b][(g,o)][t] *
// Xp[b][(g,i)][t*s + k*d] is the convolution of signal' = X with batch and channels exchanged against kernel' = dY,
// stride' = dilation, dilation' = stride, of which the first k lags per axis are kept.  The plan is that convolution;
// both tensors are read, and dW written, in the caller's layout (ImgMap) -- no transposed copies, no crop afterwards.
int fc_wgrad_nd_plan_create(const fc_desc* conv, fc_plan** out_plan) {
  if (!conv || !out_plan) return fail(FC_ERR_INVALID, "null argument");
  *out_plan = nullptr;
  const fc_desc& c = *conv;
  if (c.ndim < 2 || c.ndim > 3) return fail(FC_ERR_UNSUPPORTED, "fc_wgrad_nd covers 2-D and 3-D convolutions (1-D: fc_wgrad1d)");
  if (c.dtype != FC_F32 || c.transposed) return fail(FC_ERR_UNSUPPORTED, "fc_wgrad_nd: float32, not transposed");
  if (c.batch < 1 || c.in_channels < 1 || c.out_channels < 1 || c.groups < 1 || c.in_channels % c.groups || c.out_channels % c.groups)
    return fail(FC_ERR_INVALID, "batch, channels and groups must be positive and the channels divisible by groups");
  WgradSwap sw;
  sw.B = c.batch; sw.g = c.groups; sw.Cig = c.in_channels / c.groups; sw.Cog = c.out_channels / c.groups;
  fc_desc d = c;
  d.batch = sw.Cig; d.in_channels = sw.g * sw.B; d.out_channels = sw.g * sw.Cog; d.groups = sw.g;
  d.has_bias = 0; d.tile_hint = 0;
  for (int i = 0; i < 3; ++i) sw.keep[i] = 1;
  for (int i = 0; i < c.ndim; ++i) {
    if (c.spatial[i] < 1 || c.kernel[i] < 1 || c.stride[i] < 1 || c.dilation[i] < 1 || c.padding[i] < 0)
      return fail(FC_ERR_INVALID, "axis %d: spatial/kernel/stride/dilation must be >= 1 and padding >= 0", i);
    const int64_t kd = (c.kernel[i] - 1) * c.dilation[i] + 1;
    const int64_t span = c.spatial[i] + 2 * c.padding[i] - kd;
    if (span < 0) return fail(FC_ERR_INVALID, "axis %d: dilated kernel extent %lld is larger than the padded input", i, (long long)kd);
    d.kernel[i] = span / c.stride[i] + 1;          // taps of kernel' = output extent of the convolution
    d.stride[i] = c.dilation[i];
    d.dilation[i] = c.stride[i];
    sw.keep[i] = c.kernel[i];
  }
  return plan_create_impl(&d, &sw, out_plan);
}

int fc_wgrad_nd(const fc_plan* plan, const float* x, const float* dy, float* dw, void* spectrum, void* workspace, void* hip_stream) {
  if (!plan || !x || !dy || !dw || !spectrum) return fail(FC_ERR_INVALID, "null argument");
  if (!plan->swap) return fail(FC_ERR_INVALID, "not a weight-gradient plan (fc_wgrad_nd_plan_create)");
  int rc = fc_transform_kernel(plan, dy, spectrum, workspace, hip_stream);
  if (rc != FC_OK) return rc;
  return fc_forward(plan, x, spectrum, nullptr, dw, workspace, hip_stream);
}

}  // extern "C"
