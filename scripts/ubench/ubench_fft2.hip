// ubench_fft2.hip -- does 4 waves per SIMD with 16 points per thread overlap LDS and VALU better than
// 2 waves per SIMD with 32 points per thread?  Same data volume per workgroup (16 sequences of 1024 points,
// 128 KB of LDS), one "forward transform" per repetition:
//   P = 32, S = 1, 512 threads : fft32, twiddle (62 pk), exchange, fft32, write natural       (the shipped geometry)
//   P = 16, S = 4, 1024 threads: fft16, twiddle (30 pk), exchange, fft16 + lane split (DPP), write natural
// Twiddles come from LDS as in the persistent kernel.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fft_conv_pytorch_amd/csrc/fft_engine.hpp"

using namespace fc;

template <int P, int S, int NT, int MODE>
__global__ __launch_bounds__(NT) void k(const f2* __restrict__ in, f2* __restrict__ out, const f2* __restrict__ twA,
                                        const f2* __restrict__ twBp, unsigned long long* cyc, int R) {
  using G = Geo<P, S>;
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  constexpr int TWN = P * G::N2;
  f2* twl = lds;
  f2* zbuf = lds + TWN;
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  f2* zseq = zbuf + sq * G::LSEQ;
  const BufRsrc twB = make_rsrc(twBp, (unsigned)(S * P * 8));
  for (int i = tid; i < TWN; i += NT) twl[i] = twA[i];
  f2 v[P];
#pragma unroll
  for (int i = 0; i < P; ++i) v[i] = in[(size_t)i * 1024 + (tid & 1023)];
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; ++r) {
    if constexpr (MODE == 0) {          // full forward transform
      passA_fft_twiddle_store_lds<G, -1>(v, zseq, tseq, twl);
      seq_sync<G>();
      passB_load<G>(v, zseq, tseq);
      seq_sync<G>();
      const int j = passB_compute<G, -1>(v, tseq, twB);
      const int k1 = tseq >> G::LGS;
      f2* dst = zseq + G::nat(k1 + P * P * j);
#pragma unroll
      for (int kk = 0; kk < P; ++kk) dst[P * kk] = v[kk];
      seq_sync<G>();
      nat_load<G>(v, zseq, tseq);       // (stands in for the mix reading the bins back)
      seq_sync<G>();
    } else {                            // VALU part only: no LDS traffic at all
      fft_regs<P, -1>(v);
#pragma unroll
      for (int kk = 1; kk < P; ++kk) v[kk] = cmul(v[kk], v[0]);
      const int j = passB_compute<G, -1>(v, tseq, twB);
      v[0].x += (float)j;
    }
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = v[i] * mk2(0.03125f, 0.03125f);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f2 acc = v[0];
#pragma unroll
  for (int i = 1; i < P; ++i) acc = acc + v[i];
  out[(size_t)blockIdx.x * NT + tid] = acc;
  if ((tid & 63) == 0) cyc[(size_t)blockIdx.x * (NT / 64) + tid / 64] = t1 - t0;
}

template <int P, int S, int NT, int MODE>
void run(int R, const f2* in, f2* out, const f2* twA, const f2* twB, unsigned long long* cyc, const char* name) {
  using G = Geo<P, S>;
  const int grid = 256;
  const size_t lds = ((size_t)P * G::N2 + (size_t)(NT / G::TS) * G::LSEQ) * sizeof(f2);
  auto kern = k<P, S, NT, MODE>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) { printf("attr failed\n"); return; }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  kern<<<grid, NT, lds>>>(in, out, twA, twB, cyc, R);
  hipEventRecord(e0);
  kern<<<grid, NT, lds>>>(in, out, twA, twB, cyc, R);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  if (hipGetLastError() != hipSuccess) { printf("%s: launch failed\n", name); return; }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)grid * NT / 64);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("%-44s NT=%4d lds=%6zu B  kernel %7.1f us  = %.3f us per repetition (16 sequences x 1024 points), wave cycles/rep %.0f\n",
         name, NT, lds, ms * 1e3, ms * 1e3 / R, (double)h[h.size() / 2] / R);
}

int main() {
  f2 *in, *out, *twA32, *twA16, *twB;
  unsigned long long* cyc;
  hipMalloc(&in, 32 * 1024 * sizeof(f2));
  hipMalloc(&out, 256 * 1024 * sizeof(f2));
  hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  hipMalloc(&twA32, 1024 * sizeof(f2)); hipMalloc(&twA16, 1024 * sizeof(f2)); hipMalloc(&twB, 64 * sizeof(f2));
  std::vector<f2> h(32 * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = f2{(float)(rand() % 1000) / 1000.f - 0.5f, (float)(rand() % 1000) / 1000.f - 0.5f};
  hipMemcpy(in, h.data(), h.size() * sizeof(f2), hipMemcpyHostToDevice);
  std::vector<f2> t(1024, f2{0.6f, 0.8f});
  hipMemcpy(twA32, t.data(), 1024 * sizeof(f2), hipMemcpyHostToDevice);
  hipMemcpy(twA16, t.data(), 1024 * sizeof(f2), hipMemcpyHostToDevice);
  hipMemcpy(twB, t.data(), 64 * sizeof(f2), hipMemcpyHostToDevice);
  const int R = 64;
  run<32, 1, 512, 0>(R, in, out, twA32, twB, cyc, "P=32 S=1 forward transform (LDS + VALU)");
  run<32, 1, 512, 1>(R, in, out, twA32, twB, cyc, "P=32 S=1 VALU part only");
  run<16, 4, 1024, 0>(R, in, out, twA16, twB, cyc, "P=16 S=4 forward transform (LDS + VALU)");
  run<16, 4, 1024, 1>(R, in, out, twA16, twB, cyc, "P=16 S=4 VALU part only");
  run<32, 1, 256, 0>(R, in, out, twA32, twB, cyc, "P=32 S=1 8 sequences, 1 wave per SIMD");
  run<16, 4, 512, 0>(R, in, out, twA16, twB, cyc, "P=16 S=4 8 sequences, 2 waves per SIMD");
  return 0;
}
