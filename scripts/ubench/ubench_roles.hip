// ubench_roles.hip -- do LDS traffic and VALU work of the two waves of a SIMD overlap?  512-thread workgroups
// (2 waves per SIMD).  Each half of the workgroup (waves 0-3 / waves 4-7; SIMD partners are waves w and w+4)
// gets a role: 'L' = pass-A style LDS exchange only (32 ds_write_b64 + 32 ds_read_b64 per repetition),
// 'V' = 32-point register FFT only (194 v_pk), 'B' = both back to back, '-' = idle.
// Prints the median shader cycles per repetition of each half.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fft_conv_pytorch_amd/csrc/fft_engine.hpp"

using namespace fc;

__device__ __forceinline__ void do_lds(f2 (&v)[32], f2* seq, int n2) {
#pragma unroll
  for (int k1 = 0; k1 < 32; ++k1) seq[k1 * 33 + n2] = v[k1];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  lds_read_strided<32, 1>(v, seq + n2 * 33);
  lds_arrive(v);
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void do_valu(f2 (&v)[32]) {
  fft_regs<32, -1>(v);
  v[0] = v[0] * mk2(0.03125f, 0.03125f);
}

// One wave per SIMD, two independent sequence sets per thread, software-pipelined: the LDS exchange of one set is
// in flight while the other set's butterflies issue (LDS instructions complete asynchronously; a wave only stalls
// at its s_waitcnt).  fft32 split in two halves to cover both LDS round trips of an exchange.
template <int DIR>
__device__ __forceinline__ void fft_first_half(f2 (&t)[32], const f2 (&v)[32]) {   // bit reversal + stages len 2, 4, 8
#pragma unroll
  for (int i = 0; i < 32; ++i) t[bitrev(i, 5)] = v[i];
#pragma unroll
  for (int len = 2; len <= 8; len <<= 1)
#pragma unroll
    for (int blk = 0; blk < 32; blk += len)
#pragma unroll
      for (int j = 0; j < len / 2; ++j) bfly<DIR>(t[blk + j], t[blk + j + len / 2], j * (64 / len));
}
template <int DIR>
__device__ __forceinline__ void fft_second_half(f2 (&t)[32]) {                    // stages len 16, 32
#pragma unroll
  for (int len = 16; len <= 32; len <<= 1)
#pragma unroll
    for (int blk = 0; blk < 32; blk += len)
#pragma unroll
      for (int j = 0; j < len / 2; ++j) bfly<DIR>(t[blk + j], t[blk + j + len / 2], j * (64 / len));
}
template <int K1>
__device__ __forceinline__ void issue_write_one(unsigned addr, f2 val) {
  asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(addr), "v"(val), "n"(K1 * 33 * 8) : "memory");
}
template <int K1 = 0>
__device__ __forceinline__ void issue_writes_from(const f2 (&v)[32], unsigned addr) {
  if constexpr (K1 < 32) {
    issue_write_one<K1>(addr, v[K1]);
    issue_writes_from<K1 + 1>(v, addr);
  }
}
__device__ __forceinline__ void issue_writes(const f2 (&v)[32], f2* seq, int n2) { issue_writes_from<0>(v, lds_off(seq + n2)); }
__device__ __forceinline__ void wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); }

__global__ __launch_bounds__(256, 1) void kp(const f2* __restrict__ in, f2* __restrict__ out, unsigned long long* cyc, int R) {
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const int tid = threadIdx.x;
  f2 a[32], b[32], ta[32], tb[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) { a[i] = in[(size_t)i * 1024 + tid]; b[i] = in[(size_t)i * 1024 + tid + 256]; }
  f2* seqa = lds + (tid / 32) * (32 * 33);
  f2* seqb = seqa + 8 * (32 * 33);
  const int n2 = tid % 32;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; ++r) {
    // set b: exchange; set a: butterflies
    issue_writes(b, seqb, n2);
    fft_first_half<-1>(ta, a);
    __builtin_amdgcn_sched_barrier(0);
    wait_lds();
    lds_read_strided<32, 1>(b, seqb + n2 * 33);
    __builtin_amdgcn_sched_barrier(0);
    fft_second_half<-1>(ta);
    ta[0] = ta[0] * mk2(0.03125f, 0.03125f);
    __builtin_amdgcn_sched_barrier(0);
    lds_arrive(b);
    __builtin_amdgcn_wave_barrier();
    // set a: exchange; set b: butterflies
    issue_writes(ta, seqa, n2);
    fft_first_half<-1>(tb, b);
    __builtin_amdgcn_sched_barrier(0);
    wait_lds();
    lds_read_strided<32, 1>(a, seqa + n2 * 33);
    __builtin_amdgcn_sched_barrier(0);
    fft_second_half<-1>(tb);
    tb[0] = tb[0] * mk2(0.03125f, 0.03125f);
    __builtin_amdgcn_sched_barrier(0);
    lds_arrive(a);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 32; ++i) b[i] = tb[i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f2 acc = a[0] + b[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) acc = acc + a[i] + b[i];
  out[(size_t)blockIdx.x * 256 + tid] = acc;
  if ((tid & 63) == 0) cyc[(size_t)blockIdx.x * 8 + tid / 64] = t1 - t0;
}

// 'BB' with the repetitions written out RU times (straight-line code as in the product kernel, which executes every
// instruction once per work item) inside an outer loop: tests whether instruction supply limits such a stream.
template <int RU>
__global__ __launch_bounds__(512, 2) void ku(const f2* __restrict__ in, f2* __restrict__ out, unsigned long long* cyc, int R) {
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const int tid = threadIdx.x;
  f2 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = in[(size_t)i * 1024 + tid];
  f2* seq = lds + (tid / 32) * (32 * 33);
  const int n2 = tid % 32;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; r += RU) {
#pragma unroll
    for (int u = 0; u < RU; ++u) { do_valu(v); do_lds(v, seq, n2); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f2 acc = v[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) acc = acc + v[i];
  out[(size_t)blockIdx.x * 512 + tid] = acc;
  if ((tid & 63) == 0) cyc[(size_t)blockIdx.x * 8 + tid / 64] = t1 - t0;
}

__global__ __launch_bounds__(512, 2) void k(const f2* __restrict__ in, f2* __restrict__ out, unsigned long long* cyc, int R, int role0, int role1,
                                            int prio) {
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const int tid = threadIdx.x;
  f2 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = in[(size_t)i * 1024 + tid];
  f2* seq = lds + (tid / 32) * (32 * 33);
  const int n2 = tid % 32;
  const int role = tid < 256 ? role0 : role1;
  if (prio == 1 && tid < 256) __builtin_amdgcn_s_setprio(1);
  if (prio == 2 && tid >= 256) __builtin_amdgcn_s_setprio(1);
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 'L') { for (int r = 0; r < R; ++r) do_lds(v, seq, n2); }
  else if (role == 'V') { for (int r = 0; r < R; ++r) do_valu(v); }
  else if (role == 'B') { for (int r = 0; r < R; ++r) { do_valu(v); do_lds(v, seq, n2); } }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f2 acc = v[0];
#pragma unroll
  for (int i = 1; i < 32; ++i) acc = acc + v[i];
  out[(size_t)blockIdx.x * 512 + tid] = acc;
  if ((tid & 63) == 0) cyc[(size_t)blockIdx.x * 8 + tid / 64] = t1 - t0;
}

int main() {
  f2 *in, *out;
  unsigned long long* cyc;
  hipMalloc(&in, 32 * 1024 * sizeof(f2));
  hipMalloc(&out, 256 * 512 * sizeof(f2));
  hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long));
  std::vector<f2> h(32 * 1024);
  for (size_t i = 0; i < h.size(); ++i) h[i] = f2{(float)(rand() % 1000) / 1000.f - 0.5f, (float)(rand() % 1000) / 1000.f - 0.5f};
  hipMemcpy(in, h.data(), h.size() * sizeof(f2), hipMemcpyHostToDevice);
  const size_t lds = 16 * 32 * 33 * sizeof(f2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const int R = 64;
  const char* combos[] = {"L-", "-L", "V-", "-V", "LL", "VV", "LV", "VL", "BB", "B-"};
  for (int prio = 0; prio < 3; ++prio)
    for (const char* c : combos) {
      if (prio && !(c[0] != '-' && c[1] != '-')) continue;
      k<<<256, 512, lds>>>(in, out, cyc, R, c[0], c[1], prio);
      k<<<256, 512, lds>>>(in, out, cyc, R, c[0], c[1], prio);
      hipDeviceSynchronize();
      std::vector<unsigned long long> hc(256 * 8);
      hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
      std::vector<unsigned long long> h0, h1;
      for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? h0 : h1).push_back(hc[b * 8 + w]);
      std::sort(h0.begin(), h0.end()); std::sort(h1.begin(), h1.end());
      printf("roles %s prio %d : half0 %6.0f cycles/rep   half1 %6.0f cycles/rep\n", c, prio, (double)h0[h0.size() / 2] / R, (double)h1[h1.size() / 2] / R);
    }
  auto run_u = [&](auto kern, int ru) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    kern<<<256, 512, lds>>>(in, out, cyc, R);
    kern<<<256, 512, lds>>>(in, out, cyc, R);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(256 * 8);
    hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> h0, h1;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? h0 : h1).push_back(hc[b * 8 + w]);
    std::sort(h0.begin(), h0.end()); std::sort(h1.begin(), h1.end());
    printf("BB, body written out %2d times: half0 %6.0f cycles/rep   half1 %6.0f cycles/rep\n", ru, (double)h0[h0.size() / 2] / R, (double)h1[h1.size() / 2] / R);
  };
  run_u(ku<1>, 1); run_u(ku<4>, 4); run_u(ku<16>, 16); run_u(ku<32>, 32); run_u(ku<64>, 64);
  {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kp), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    kp<<<256, 256, lds>>>(in, out, cyc, R);
    kp<<<256, 256, lds>>>(in, out, cyc, R);
    hipDeviceSynchronize();
    std::vector<unsigned long long> hc(256 * 8);
    hipMemcpy(hc.data(), cyc, hc.size() * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> h0;
    for (int b = 0; b < 256; ++b)
      for (int w = 0; w < 4; ++w) h0.push_back(hc[b * 8 + w]);
    std::sort(h0.begin(), h0.end());
    printf("pipelined, one wave per SIMD, two sets per thread: %6.0f cycles per repetition (= 2 FFTs + 2 exchanges per thread; "
           "same work as one BB repetition of BOTH halves)\n", (double)h0[h0.size() / 2] / R);
  }
  return 0;
}
