// ubench_fft.hip -- calibration of the fused kernel's building blocks on one MI355X (diagnostic, not product):
//   mode 0: 32-point register FFT only (v_pk_* stream as fft_engine.hpp emits it), R repetitions
//   mode 1: + pass-A style LDS exchange (32 ds_write_b64 + 32 strided ds_read_b64 per repetition)
//   mode 2: LDS exchange only
//   mode 3: independent v_pk_fma_f32 chains (8 accumulators) -- the issue-rate yardstick
//   mode 4: dependent v_pk_fma_f32 chain (1 accumulator)
// Grid = 256 workgroups of 64 * 4 * W threads (W waves per SIMD).  Per-wave shader cycles come from s_memtime
// around the loop; prints median cycles per repetition and per v_pk instruction.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fft_conv_pytorch_amd/csrc/fft_engine.hpp"

using namespace fc;

template <int MODE>
__global__ __launch_bounds__(1024) void k(const f2* __restrict__ in, f2* __restrict__ out, unsigned long long* cyc, int R) {
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const int tid = threadIdx.x;
  f2 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = in[(size_t)i * 1024 + (tid & 1023)];
  // per-thread LDS column as in pass A: row stride 33 complex (padded), sequence of 32 threads
  f2* seq = lds + (tid / 32) * (32 * 33);
  const int n2 = tid % 32;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; ++r) {
    if constexpr (MODE == 0 || MODE == 1) fft_regs<32, -1>(v);
    if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
      for (int k1 = 0; k1 < 32; ++k1) seq[k1 * 33 + n2] = v[k1];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      lds_read_strided<32, 1>(v, seq + n2 * 33);
      lds_arrive(v);
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (MODE == 3) {
#pragma unroll
      for (int j = 0; j < 24; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pkfma(v[8 + i], v[16 + i], v[i]);
    }
    if constexpr (MODE == 4) {
#pragma unroll
      for (int j = 0; j < 192; ++j) v[0] = pkfma(v[8], v[16], v[0]);
    }
    if constexpr (MODE == 0) {
      // keep magnitudes bounded without changing the instruction mix much: one scale per repetition
      v[0] = v[0] * mk2(0.03125f, 0.03125f);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f2 acc = v[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) acc = acc + v[i];
  out[(size_t)blockIdx.x * blockDim.x + tid] = acc;
  if ((tid & 63) == 0) cyc[(size_t)blockIdx.x * (blockDim.x / 64) + tid / 64] = t1 - t0;
}

template <int MODE>
void run(int W, int R, const f2* in, f2* out, unsigned long long* cyc, int pk_per_rep, const char* name) {
  const int nt = 64 * 4 * W, grid = 256;
  const size_t lds = (size_t)(nt / 32) * 32 * 33 * sizeof(f2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<grid, nt, lds>>>(in, out, cyc, R);
  hipEventRecord(e0);
  k<MODE><<<grid, nt, lds>>>(in, out, cyc, R);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h((size_t)grid * nt / 64);
  hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double med = (double)h[h.size() / 2] / R;
  // SIMD-level: W waves share one SIMD, so SIMD cycles per pk instruction = med / (W * pk_per_rep) ... per wave med / pk
  printf("%-28s W=%d  lds=%6zu B  kernel %.1f us  cycles/rep/wave %.0f", name, W, lds, ms * 1e3, med);
  if (pk_per_rep) printf("  cycles per v_pk per wave %.2f  (SIMD cycles per v_pk %.2f)", med / pk_per_rep, med / pk_per_rep / W);
  printf("  clock %.2f GHz\n", (double)h[h.size() / 2] / (ms * 1e3) / 1e3);
}

int main() {
  f2 *in, *out;
  unsigned long long* cyc;
  hipMalloc(&in, 32 * 1024 * sizeof(f2));
  hipMalloc(&out, 256 * 1024 * sizeof(f2));
  hipMalloc(&cyc, 256 * 16 * sizeof(unsigned long long));
  std::vector<f2> h(32 * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = f2{(float)(rand() % 1000) / 1000.f - 0.5f, (float)(rand() % 1000) / 1000.f - 0.5f};
  hipMemcpy(in, h.data(), h.size() * sizeof(f2), hipMemcpyHostToDevice);
  const int R = 64;
  for (int W : {1, 2, 4}) {
    run<3>(W, R, in, out, cyc, 192, "independent pk_fma x192");
    run<4>(W, R, in, out, cyc, 192, "dependent pk_fma x192");
    run<0>(W, R, in, out, cyc, 195, "fft32 regs (194 pk + 1)");
    if ((size_t)(64 * 4 * W / 32) * 32 * 33 * 8 <= 160 * 1024) {
      run<2>(W, R, in, out, cyc, 0, "lds exchange only (32w+32r)");
      run<1>(W, R, in, out, cyc, 194, "fft32 + lds exchange");
    }
  }
  return 0;
}
