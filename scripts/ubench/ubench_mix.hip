// ubench_mix.hip -- the channel contraction ("mix") of one work item of the batch-sharing kernel (4 batch items,
// 8 -> 8 channels, 1024-point tile: 512 bin pairs, 16 sequences of 1024 complex values resident in LDS), as
//   VALU : the shipped formulation (per thread one bin pair: untangle the packed real spectra, 64 complex MACs per
//          batch item against the spectrum streamed from L2 as float4, re-tangle, write back in place)
//   MFMA : v_mfma_f32_4x4x1_16B_f32 -- 16 bin pairs per wave instruction, rows = the 4 batch items, columns = 4 of the
//          16 real outputs, 16 rank-1 steps over the 16 real inputs; untangle and re-tangle are folded into a
//          pre-computed real 16 x 16 matrix per bin pair (the B operand, 1 KiB per bin pair instead of 512 B)
// Both read the same LDS image and the results are compared.  R back-to-back mixes per launch, 256 workgroups
// (one per CU) of 512 threads, spectrum L2-resident as in the real kernel.  Diagnostic, not product code.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../fft_conv_pytorch_amd/csrc/fft_engine.hpp"

using namespace fc;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int T = 1024, NB = 4, NPI = 4, NT = 512, LSEQ = 32 * 33;   // Geo<32,1>: LSEQ = N1 * RS = 32 * 33
__device__ __host__ constexpr int nat(int f) { return f; }           // S = 1: natural layout has no padding

// ------------------------------------------------------------------ VALU formulation (as conv1d_pers.hpp)
__global__ __launch_bounds__(NT, 2) void mix_valu(const f2* __restrict__ zin, const f4* __restrict__ wspec, f2* __restrict__ zout,
                                                  unsigned long long* cyc, int R) {
  extern __shared__ __attribute__((aligned(16))) f2 zbuf[];
  const int tid = threadIdx.x;
  for (int i = tid; i < NB * NPI * LSEQ; i += NT) zbuf[i] = zin[i];
  __syncthreads();
  const BufRsrc wg = make_rsrc(wspec, (unsigned)(8 * NPI * (T / 2) * 16));
  const unsigned ostride = (unsigned)NPI * (T / 2) * 16u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; ++r) {
    const int f = tid, fm = (T - f) & (T - 1);
    f2 xe[NB][NPI], xo[NB][NPI];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int p = 0; p < NPI; ++p) {
        const f2 zf = zbuf[(b * NPI + p) * LSEQ + nat(f)], zg = zbuf[(b * NPI + p) * LSEQ + nat(fm)];
        xe[b][p] = add_conj(zf, zg);
        xo[b][p] = sub_conj_divi(zf, zg);
      }
#pragma unroll
    for (int q = 0; q < NPI; ++q) {
      f4 wc[2 * NPI];
#pragma unroll
      for (int p = 0; p < NPI; ++p) {
        wc[2 * p] = buf_load_f32x4(wg, (unsigned)tid * 16u, (2 * q) * ostride + p * (T / 2) * 16);
        wc[2 * p + 1] = buf_load_f32x4(wg, (unsigned)tid * 16u, (2 * q + 1) * ostride + p * (T / 2) * 16);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        f2 ya = mk2(0.f, 0.f), yb = mk2(0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          cmac(ya, xe[b][p], wc[2 * p].xy); cmac(ya, xo[b][p], wc[2 * p].zw);
          cmac(yb, xe[b][p], wc[2 * p + 1].xy); cmac(yb, xo[b][p], wc[2 * p + 1].zw);
        }
        if (f != 0) {
          f2* zb = zbuf + (b * NPI + q) * LSEQ;
          zb[nat(f)] = add_pi(ya, yb);
          zb[nat(fm)] = conj_add_iconj(ya, yb);
        }
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = tid; i < NB * NPI * LSEQ; i += NT) zout[(size_t)blockIdx.x * NB * NPI * LSEQ + i] = zbuf[i];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// ------------------------------------------------------------------ MFMA formulation
// mexp: [group of 16 bin pairs (32)][i = 4q + k/4 (16)][lane (64)] float4 = rank-1 steps k = 4(i%4) .. +3 of output
// column c = 4q + (lane & 3) at bin pair 16*group + (lane >> 2): every wave instruction reads 1 KiB contiguous.
// Lane l of a wave: block (= bin pair of the group) l >> 2, row / column index l & 3.
__global__ __launch_bounds__(NT, 2) void mix_mfma(const f2* __restrict__ zin, const float* __restrict__ mexp, f2* __restrict__ zout,
                                                  unsigned long long* cyc, int R) {
  extern __shared__ __attribute__((aligned(16))) f2 zbuf[];
  const int tid = threadIdx.x;
  for (int i = tid; i < NB * NPI * LSEQ; i += NT) zbuf[i] = zin[i];
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, blk = lane >> 2, ij = lane & 3;
  const BufRsrc mg = make_rsrc(mexp, (unsigned)((T / 2) * 256 * 4));
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < R; ++r) {
#pragma unroll 1
    for (int grp = wave; grp < (T / 2) / 16; grp += NT / 64) {
      const int f = grp * 16 + blk, fm = (T - f) & (T - 1);
      // B operand: 64 consecutive floats of this lane's (bin pair, j)
      f4 bw[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) bw[i] = buf_load_f32x4(mg, (unsigned)(grp * 16 * 64 + lane) * 16u, i * 64 * 16);   // coalesced: 1 KiB per wave instruction
      // A operand: the 16 real inputs of (bin pair f, batch item ij): Z_p[f], Z_p[T-f], p = 0..3
      f2 za[NPI], zb[NPI];
#pragma unroll
      for (int p = 0; p < NPI; ++p) {
        za[p] = zbuf[(ij * NPI + p) * LSEQ + nat(f)];
        zb[p] = zbuf[(ij * NPI + p) * LSEQ + nat(fm)];
      }
      f32x4 acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          const f4 w = bw[q * 4 + p];          // k = 4p .. 4p+3 of column block q
          acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(za[p].x, w.x, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(za[p].y, w.y, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(zb[p].x, w.z, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(zb[p].y, w.w, acc[q], 0, 0, 0);
        }
      }
      // D: register i = batch item, this lane's column j = ij: {Re Y[f], Im Y[f], Re Y[T-f], Im Y[T-f]} of output pair q
      if (f != 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            float* dst = reinterpret_cast<float*>(zbuf + (b * NPI + q) * LSEQ + nat((ij & 2) ? fm : f)) + (ij & 1);
            *dst = acc[q][b];
          }
      }
    }
    __syncthreads();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  for (int i = tid; i < NB * NPI * LSEQ; i += NT) zout[(size_t)blockIdx.x * NB * NPI * LSEQ + i] = zbuf[i];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

// layout probe of v_mfma_f32_4x4x1_16B_f32: a = lane + 1, b = 100 * (lane + 1); D[i] of lane l should be a(block, i) * b(block, j = l & 3)
__global__ void probe(float* out) {
  const int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32((float)(lane + 1), 100.f * (lane + 1), c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = c[i];
}

int main() {
  const int grid = 256, R = 32;
  const size_t nz = (size_t)NB * NPI * LSEQ;
  std::vector<f2> hz(nz);
  srand(1);
  for (auto& v : hz) v = f2{(float)(rand() % 2001 - 1000) / 1000.f, (float)(rand() % 2001 - 1000) / 1000.f};
  // spectrum H[o][i][f] (complex), shipped layout wspec[o][p][f] = {H(o,2p)[f], H(o,2p+1)[f]}
  std::vector<std::complex<double>> H((size_t)8 * 8 * (T / 2));
  for (auto& h : H) h = {(rand() % 2001 - 1000) / 4000.0, (rand() % 2001 - 1000) / 4000.0};
  std::vector<f4> hw((size_t)8 * NPI * (T / 2));
  for (int o = 0; o < 8; ++o)
    for (int p = 0; p < NPI; ++p)
      for (int f = 0; f < T / 2; ++f) {
        const auto a = H[((size_t)o * 8 + 2 * p) * (T / 2) + f], b = H[((size_t)o * 8 + 2 * p + 1) * (T / 2) + f];
        hw[((size_t)o * NPI + p) * (T / 2) + f] = f4{(float)a.real(), (float)a.imag(), (float)b.real(), (float)b.imag()};
      }
  // Expanded real matrix per bin pair, derived numerically from the VALU formulation's arithmetic (linear in the
  // 16 real inputs): column c of M = response to the c-th unit input.  The shipped mix computes, per output pair q:
  //   xe_p = zf_p + conj(zg_p), xo_p = (zf_p - conj(zg_p)) / i   [add_conj, sub_conj_divi]
  //   ya = sum_p xe_p * Ha_p.xy + xo_p * Ha_p.zw ; yb likewise with the odd output's row
  //   out[f] = ya + i*yb ; out[T-f] = conj(ya) + i*conj(yb)
  std::vector<float> hm((size_t)(T / 2) * 256);
  for (int f = 0; f < T / 2; ++f)
    for (int k = 0; k < 16; ++k) {
      std::complex<double> zf[4] = {}, zg[4] = {};
      const int p0 = k >> 2, part = k & 3;
      if (part == 0) zf[p0] = {1, 0}; else if (part == 1) zf[p0] = {0, 1}; else if (part == 2) zg[p0] = {1, 0}; else zg[p0] = {0, 1};
      for (int q = 0; q < 4; ++q) {
        std::complex<double> ya = 0, yb = 0;
        for (int p = 0; p < 4; ++p) {
          const auto xe = zf[p] + std::conj(zg[p]);
          const auto xo = (zf[p] - std::conj(zg[p])) / std::complex<double>(0, 1);
          ya += xe * H[((size_t)(2 * q) * 8 + 2 * p) * (T / 2) + f] + xo * H[((size_t)(2 * q) * 8 + 2 * p + 1) * (T / 2) + f];
          yb += xe * H[((size_t)(2 * q + 1) * 8 + 2 * p) * (T / 2) + f] + xo * H[((size_t)(2 * q + 1) * 8 + 2 * p + 1) * (T / 2) + f];
        }
        const auto of = ya + std::complex<double>(0, 1) * yb, om = std::conj(ya) + std::complex<double>(0, 1) * std::conj(yb);
        const double col[4] = {of.real(), of.imag(), om.real(), om.imag()};
        for (int j = 0; j < 4; ++j) {
          const int grp = f / 16, lane = (f % 16) * 4 + j, i = q * 4 + k / 4;
          hm[(((size_t)grp * 16 + i) * 64 + lane) * 4 + (k % 4)] = (float)col[j];
        }
      }
    }
  f2 *dz, *dout;
  f4* dw;
  float *dm, *dprobe;
  unsigned long long* dcyc;
  hipMalloc(&dz, nz * sizeof(f2)); hipMalloc(&dout, (size_t)grid * nz * sizeof(f2));
  hipMalloc(&dw, hw.size() * sizeof(f4)); hipMalloc(&dm, hm.size() * sizeof(float)); hipMalloc(&dcyc, grid * 8); hipMalloc(&dprobe, 256 * 4);
  hipMemcpy(dz, hz.data(), nz * sizeof(f2), hipMemcpyHostToDevice);
  hipMemcpy(dw, hw.data(), hw.size() * sizeof(f4), hipMemcpyHostToDevice);
  hipMemcpy(dm, hm.data(), hm.size() * sizeof(float), hipMemcpyHostToDevice);
  // ---- layout probe
  probe<<<1, 64>>>(dprobe);
  std::vector<float> hp(256);
  hipMemcpy(hp.data(), dprobe, 1024, hipMemcpyDeviceToHost);
  bool ok = true;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 4; ++i) {
      const int blk = l >> 2, j = l & 3;
      const float want = (float)(blk * 4 + i + 1) * 100.f * (blk * 4 + j + 1);
      if (hp[l * 4 + i] != want) ok = false;
    }
  printf("v_mfma_f32_4x4x1_16B_f32 layout (block = lane>>2, A row = lane&3, B col = lane&3, D[reg i][lane j]): %s\n", ok ? "confirmed" : "DIFFERENT");
  if (!ok) { for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, hp[l * 4], hp[l * 4 + 1], hp[l * 4 + 2], hp[l * 4 + 3]); }
  const size_t lds = nz * sizeof(f2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_valu), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipFuncSetAttribute(reinterpret_cast<const void*>(mix_mfma), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // ---- parity at R = 1
  std::vector<f2> o1(nz), o2(nz);
  mix_valu<<<grid, NT, lds>>>(dz, dw, dout, dcyc, 1);
  hipMemcpy(o1.data(), dout, nz * sizeof(f2), hipMemcpyDeviceToHost);
  mix_mfma<<<grid, NT, lds>>>(dz, dm, dout, dcyc, 1);
  hipMemcpy(o2.data(), dout, nz * sizeof(f2), hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  for (size_t i = 0; i < nz; ++i) {
    scale = std::max(scale, (double)std::max(std::fabs(o1[i].x), std::fabs(o1[i].y)));
    worst = std::max(worst, (double)std::max(std::fabs(o1[i].x - o2[i].x), std::fabs(o1[i].y - o2[i].y)));
  }
  printf("parity MFMA vs VALU mix: max abs diff %.3e on values up to %.3f (rel %.2e)\n", worst, scale, worst / scale);
  // ---- timing
  for (int which = 0; which < 2; ++which) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (which == 0) mix_valu<<<grid, NT, lds>>>(dz, dw, dout, dcyc, R);
      else mix_mfma<<<grid, NT, lds>>>(dz, dm, dout, dcyc, R);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(grid);
    hipMemcpy(hc.data(), dcyc, grid * 8, hipMemcpyDeviceToHost);
    std::sort(hc.begin(), hc.end());
    printf("%s mix: %.2f us per work item (kernel %.1f us for %d mixes), %.0f shader cycles per mix, spectrum bytes per mix %d KiB\n",
           which == 0 ? "VALU" : "MFMA", ms * 1e3 / R, ms * 1e3, R, (double)hc[grid / 2] / R, which == 0 ? 256 : 512);
  }
  return 0;
}
