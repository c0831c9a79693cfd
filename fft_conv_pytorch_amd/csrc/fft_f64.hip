// fft_f64.hip -- float64 tensors through an FFT path (the reference is dtype-agnostic:
// functional.py:19-89 runs complex128 FFTs when handed float64 tensors).
//
// Round 2 served float64 callers with a direct time-domain kernel only (direct_f64.hip, O(outputs x Cin/g x taps)).
// This file is the same 1-D algorithm as the fp32 kernels -- overlap-save tiles, forward transform, per-bin channel
// contraction against the pre-transformed kernel, inverse transform, valid window + stride + bias -- written in plain
// double precision: a Stockham radix-2 transform in LDS (natural order in and out, one butterfly per thread and stage,
// twiddles from a table the workgroup builds with sincospi), two real channels per complex sequence, the output
// channels' spectra accumulated in registers over the input channels (a thread owns bins t and t + T/2).  It is not
// tuned like the packed-fp32 engine (fft_engine.hpp is written on v_pk_*_f32 pairs and 8-byte LDS slots); it exists so
// that float64 results come from the same transform-domain arithmetic as the reference's at FFT cost.  2-D / 3-D and
// transposed float64 plans keep the direct kernel.
#include <hip/hip_runtime.h>

#include <atomic>

#include "fft_f64.h"

namespace fc {
namespace {

__device__ __forceinline__ double2 cmul_d(double2 a, double2 b) {
  return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x));
}

__device__ __forceinline__ int src_index_d(int pos, int size, int pad, int mode) {   // unpadded coordinate or -1 (zero)
  if ((unsigned)pos < (unsigned)size) return pos;
  if (pos < -pad || pos >= size + pad || mode == 0) return -1;
  if (mode == 1) return pos < 0 ? -pos : 2 * (size - 1) - pos;      // reflect
  if (mode == 2) return pos < 0 ? 0 : size - 1;                     // replicate
  return pos < 0 ? pos + size : pos - size;                          // circular
}

// Stockham radix-2, T points, T/2 threads, natural order in (buffer `a`) and out (returned pointer: a or b).
// DIR = -1 forward, +1 inverse (unnormalised).  tw[k] = exp(-2 pi i k / T), k < T/2.
template <int DIR>
__device__ __forceinline__ double2* fft_stockham(double2* a, double2* b, const double2* tw, int T, int t) {
  const int half = T >> 1;
  for (int ns = 1; ns < T; ns <<= 1) {
    const int k = t & (ns - 1);
    const double2 u = a[t];
    double2 v = a[t + half];
    double2 w = tw[k * (half / ns)];
    if (DIR > 0) w.y = -w.y;
    v = cmul_d(v, w);
    const int j = ((t - k) << 1) + k;
    b[j] = make_double2(u.x + v.x, u.y + v.y);
    b[j + ns] = make_double2(u.x - v.x, u.y - v.y);
    __syncthreads();
    double2* s = a; a = b; b = s;
  }
  return a;
}

__device__ __forceinline__ void build_table(double2* tw, int T, int t) {
  double s, c;
  sincospi(-2.0 * (double)t / (double)T, &s, &c);
  tw[t] = make_double2(c, s);
}

// ---- kernel transform: one (o, i) pair per workgroup
__global__ __launch_bounds__(1024) void spectrum_f64_kernel(const FftF64Args a) {
  extern __shared__ __attribute__((aligned(16))) double2 lds64[];
  const int T = a.T, t = threadIdx.x;
  double2* bufA = lds64;
  double2* bufB = lds64 + T;
  double2* tw = lds64 + 2 * T;
  const int oi = blockIdx.x;                       // o_all * Cig + i
  const double* wrow = a.w + (size_t)oi * a.K;
  build_table(tw, T, t);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int n = t + h * (T >> 1);
    const int tap = n / a.dil;
    bufA[n] = make_double2((tap * a.dil == n && tap < a.K) ? wrow[tap] : 0.0, 0.0);
  }
  __syncthreads();
  double2* r = fft_stockham<-1>(bufA, bufB, tw, T, t);
  const double sc = 1.0 / (double)T;
  double2* out = a.wspec + (size_t)oi * T;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int f = t + h * (T >> 1);
    out[f] = make_double2(r[f].x * sc, -r[f].y * sc);      // conj: cross-correlation, like rfftn(kernel).conj()
  }
}

// ---- forward: workgroup = (batch b, group g, out-chunk oc, tile)
__global__ __launch_bounds__(1024) void conv1d_f64_kernel(const FftF64Args a) {
  extern __shared__ __attribute__((aligned(16))) double2 lds64[];
  const int T = a.T, half = T >> 1, t = threadIdx.x;
  double2* bufA = lds64;
  double2* bufB = lds64 + T;
  double2* tw = lds64 + 2 * T;
  int id = blockIdx.x;
  const int tile = id % a.ntiles; id /= a.ntiles;
  const int oc = id % a.n_ochunks; id /= a.n_ochunks;
  const int g = id % a.G;
  const int b = id / a.G;
  build_table(tw, T, t);
  const int pos0 = tile * a.V - a.pad;                 // source position of the tile's first sample
  const int nout = min(a.cob, a.Cog - oc * a.cob);
  double2 acc[8][2];
#pragma unroll
  for (int o = 0; o < 8; ++o) acc[o][0] = acc[o][1] = make_double2(0.0, 0.0);
  // two real input channels ride one complex transform (z = x_a + i x_b; X_a[f] = (Z[f] + conj Z[T-f]) / 2,
  // X_b[f] = (Z[f] - conj Z[T-f]) / 2i), two output channels one inverse transform
  for (int i = 0; i < a.Cig; i += 2) {
    const bool two = i + 1 < a.Cig;
    const double* xa = a.x + ((size_t)b * a.Cin + (size_t)g * a.Cig + i) * a.L;
    const double* xb = xa + a.L;
    __syncthreads();                                   // (table built / previous pair's spectrum consumed)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = t + h * half;
      const int q = src_index_d(pos0 + n, a.L, a.pad, a.pad_mode);
      bufA[n] = make_double2(q >= 0 ? xa[q] : 0.0, (q >= 0 && two) ? xb[q] : 0.0);
    }
    __syncthreads();
    const double2* Z = fft_stockham<-1>(bufA, bufB, tw, T, t);
    double2 xs[2][2];                                  // [bin t / t + half][channel a / b]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f = t + h * half;
      const double2 zf = Z[f], zg = Z[(T - f) & (T - 1)];
      xs[h][0] = make_double2(0.5 * (zf.x + zg.x), 0.5 * (zf.y - zg.y));
      xs[h][1] = make_double2(0.5 * (zf.y + zg.y), 0.5 * (zg.x - zf.x));
    }
    const double2* hrow = a.wspec + (((size_t)g * a.Cog + (size_t)oc * a.cob) * a.Cig + i) * T;
#pragma unroll
    for (int o = 0; o < 8; ++o)
      if (o < nout) {
        const double2* hp = hrow + (size_t)o * a.Cig * T;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int f = t + h * half;
          double2 p = cmul_d(xs[h][0], hp[f]);
          if (two) { const double2 p2 = cmul_d(xs[h][1], hp[T + f]); p.x += p2.x; p.y += p2.y; }
          acc[o][h].x += p.x; acc[o][h].y += p.y;
        }
      }
  }
  const int t0 = tile * a.V;
  const int limit = min(a.V, a.Lfull - t0);
#pragma unroll
  for (int o = 0; o < 8; o += 2) {
    if (o >= nout) break;                              // uniform
    const bool two = o + 1 < nout;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double2 ya = acc[o][h], yb = two ? acc[o + 1][h] : make_double2(0.0, 0.0);
      bufA[t + h * half] = make_double2(ya.x - yb.y, ya.y + yb.x);      // Y_a + i Y_b
    }
    __syncthreads();
    const double2* Y = fft_stockham<+1>(bufA, bufB, tw, T, t);
    const int co = g * a.Cog + oc * a.cob + o;
    const double bias_a = a.bias ? a.bias[co] : 0.0, bias_b = (a.bias && two) ? a.bias[co + 1] : 0.0;
    double* yrow = a.y + ((size_t)b * a.Cout + co) * a.Lout;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int n = t + h * half;
      const int pos = t0 + n, idx = pos / a.stride;
      if (n < limit && idx * a.stride == pos) {
        yrow[idx] = Y[n].x + bias_a;
        if (two) yrow[a.Lout + idx] = Y[n].y + bias_b;
      }
    }
  }
}

}  // namespace

hipError_t launch_fft_f64(int which, const FftF64Args& a, hipStream_t st) {
  const size_t lds = fft_f64_lds_bytes(a.T);
  const int nt = a.T / 2;
  if (nt < 64 || nt > 1024 || lds > 160 * 1024) return hipErrorInvalidValue;
  // > 64 KiB of dynamic LDS (the 2048-point tile: 80 KiB) needs the opt-in, once per kernel AND device
  static std::atomic<unsigned long long> opted[2];
  auto kernel = which == 0 ? spectrum_f64_kernel : conv1d_f64_kernel;
  if (lds > 64 * 1024) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !(opted[which].load(std::memory_order_acquire) >> dev & 1ull)) {
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (e != hipSuccess) return e;
      if (tracked) opted[which].fetch_or(1ull << dev, std::memory_order_release);
    }
  }
  long long grid = which == 0 ? (long long)a.Cout * a.Cig : (long long)a.B * a.G * a.n_ochunks * a.ntiles;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(nt), lds, st, a);
  return hipGetLastError();
}

}  // namespace fc
