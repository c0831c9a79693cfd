// conv1d_wide.hpp -- batch-sharing fused 1-D FFT convolution for MORE than 8 input channels per group.
//
// Same contract, work list and spectrum layout as conv1d_pers.hpp.  The input channels of a group come in
// chunks of 8 (4 packed sequences per batch slot); for every chunk the workgroup transforms the NB x 4
// sequences and contracts them against the chunk's slice of the kernel spectrum, ACCUMULATING the 8 output
// channels of its out-chunk in registers (a thread owns BP bin pairs x NB batch slots x 4 output pairs);
// after the last chunk the sums are re-packed into LDS once and inverse-transformed.  Compared with the
// general kernel (conv1d_fused.hpp) the spectrum loads are shared by NB batch items and the running sums
// never travel through LDS.  One work item per workgroup; two workgroups per CU cover each other's
// load latency (NB = 2), so there is no register prefetch here.
#pragma once
#include "conv1d_pers.hpp"

namespace fc {

template <int P, int S, int NB, int NT>
__global__ __launch_bounds__(NT, 2) void conv1d_wide_kernel(const Conv1dPersArgs pa) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int CIB = 8, NPI = 4;
  constexpr int NSEQ = NB * NPI;
  static_assert(NT == NSEQ * G::TS, "one thread slot per point group of every sequence");
  static_assert(G::TS <= 64 && (NPI * G::TS) % 64 == 0, "a batch slot is a whole number of wavefronts");
  static_assert((T / 2) % NT == 0, "bin pairs divide evenly over the threads");
  static_assert(NB * CIB * 2 <= 64, "the self-paired bins are handled by one wave");
  constexpr int BP = (T / 2) / NT;
  static_assert(BP * NB <= 4, "register budget of the running sums");
  constexpr int TWN = P * G::N2;
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const Conv1dArgs& a = pa.c;

  const int tid = threadIdx.x;
  const int sq = tid / G::TS, tseq = tid % G::TS;
  const int nb = __builtin_amdgcn_readfirstlane(tid / (NPI * G::TS)), pr = sq % NPI;
  f2* twl = lds;
  f2* zbuf = lds + TWN;
  f2* zseq = zbuf + sq * G::LSEQ;

  const WorkItem wi = pa.items[blockIdx.x];
  const int g = wi.goc / a.n_ochunks, oc = wi.goc % a.n_ochunks;
  const bool act_in = nb < wi.nbc;
  const PadMap pm = make_padmap(a.pad_mode, a.L);
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const size_t wgroup = (size_t)a.Cog_pad * (a.Cig_pad / 2) * (T / 2);   // float4 per group
  const BufRsrc wg = make_rsrc(a.wspec + (size_t)g * wgroup, (unsigned)(wgroup * 16));
  const float* xbase = a.x + ((size_t)wi.b0 * a.Cin + (size_t)g * a.Cig) * a.L;
  const BufRsrc xg = make_rsrc(xbase, (unsigned)(((size_t)(wi.nbc - 1) * a.Cin + a.Cig) * a.L * 4));
  const int tile_pos = wi.tile * a.V - a.pad;
  const bool interior = (tile_pos >= 0) && (tile_pos + T <= a.L);
  const int n_ichunks = a.Cig_pad / CIB;
  const unsigned ostride = (unsigned)(a.Cig_pad / 2) * (T / 2) * 16u;    // bytes between output channels
  const unsigned wbase = (unsigned)(oc * a.cob) * ostride;
  const int cg0 = g * a.Cog + oc * a.cob + 2 * pr;
  const float bias0 = a.bias ? a.bias[cg0] : 0.f;
  const float bias1 = a.bias ? a.bias[cg0 + 1] : 0.f;

  copy_table_to_lds<TWN, NT>(twl, a.twA, tid);

  // running sums: [bin pair][batch slot][output pair] x {even, odd output channel}
  f2 ya[BP][NB][NPI], yb[BP][NB][NPI];
#pragma unroll
  for (int m = 0; m < BP; ++m)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int q = 0; q < NPI; ++q) { ya[m][b][q] = mk2(0.f, 0.f); yb[m][b][q] = mk2(0.f, 0.f); }
  // self-paired bins 0 and T/2: lane (batch b, output o, bin) of wave 0 owns one real output
  const int sb_b = tid / (2 * CIB), sb_o = (tid >> 1) % CIB, sb_f = (tid & 1) ? T / 2 : 0;
  const bool sb_act = tid < NB * CIB * 2 && sb_b < wi.nbc;
  float sb_acc = 0.f;
  __syncthreads();

#pragma unroll 1
  for (int ic = 0; ic < n_ichunks; ++ic) {
    // ------------------------------------------------ this chunk's samples -> forward FFT (wave-local)
    if (act_in) {
      f2 v[P];
      const int ci0 = ic * CIB + 2 * pr;
      const bool has0 = ci0 < a.Cig, has1 = ci0 + 1 < a.Cig;
      const unsigned ro0 = ((unsigned)nb * (unsigned)a.Cin + (unsigned)ci0) * (unsigned)a.L * 4u;
      const unsigned ro1 = ro0 + (unsigned)a.L * 4u;
      if (interior && has1) {
        const unsigned v0 = ro0 + (unsigned)(tile_pos + tseq) * 4u, v1 = ro1 + (unsigned)(tile_pos + tseq) * 4u;
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          v[n1].x = buf_load_f32(xg, v0, G::N2 * n1 * 4);
          v[n1].y = buf_load_f32(xg, v1, G::N2 * n1 * 4);
        }
      } else {
        // The per-sample padded offsets do not depend on the chunk: left alone, hipcc hoists all 2 * P of them
        // out of the chunk loop and keeps them live beside the running sums (42 spilled VGPRs).  The opaque
        // copy ties them to this iteration.
        int tpos = tile_pos;
        asm volatile("" : "+s"(tpos));
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          const int pos = tpos + G::N2 * n1 + tseq;
          v[n1].x = buf_load_f32(xg, padded_offset(ro0, pos, a.L, a.pad, pm, has0), 0);
          v[n1].y = buf_load_f32(xg, padded_offset(ro1, pos, a.L, a.pad, pm, has1), 0);
        }
      }
      passA_fft_twiddle_store_lds_lowreg<G, -1>(v, zseq, tseq, twl);
      seq_sync<G>();
      passB_load<G>(v, zseq, tseq);
      seq_sync<G>();
      const int j = passB_compute<G, -1>(v, tseq, twB);
      const int k1 = tseq >> G::LGS;
      f2* dst = zseq + G::nat(k1 + P * P * j);
#pragma unroll
      for (int k = 0; k < P; ++k) dst[P * k] = v[k];
    }
    // ------------------------------------------------ contract against this chunk's slice of the spectrum
    const unsigned cbase = wbase + (unsigned)(ic * NPI) * (T / 2) * 16u;
    f4 sbw[NPI];
    if (sb_act) {
#pragma unroll
      for (int p = 0; p < NPI; ++p) sbw[p] = buf_load_f32x4(wg, (unsigned)sb_o * ostride, cbase + p * (T / 2) * 16);
    }
    auto issue = [&](int m, int q, f4 (&dst)[2 * NPI]) {
      const unsigned vo = (unsigned)(tid + m * NT) * 16u;
      const unsigned sa = cbase + (unsigned)(2 * q) * ostride, sb = sa + ostride;
#pragma unroll
      for (int p = 0; p < NPI; ++p) {
        dst[2 * p] = buf_load_f32x4(wg, vo, sa + p * (T / 2) * 16);
        dst[2 * p + 1] = buf_load_f32x4(wg, vo, sb + p * (T / 2) * 16);
      }
    };
    f4 wA[2 * NPI], wB[2 * NPI];
    issue(0, 0, wA);
    issue(0, 1, wB);
    __syncthreads();
    if (sb_act) {
#pragma unroll
      for (int p = 0; p < NPI; ++p) {
        const f2 z = zbuf[(sb_b * NPI + p) * G::LSEQ + G::nat(sb_f)];
        sb_acc = fmaf(2.f * z.x, (tid & 1) ? sbw[p].y : sbw[p].x, sb_acc);
        sb_acc = fmaf(2.f * z.y, (tid & 1) ? sbw[p].w : sbw[p].z, sb_acc);
      }
    }
    static_for<0, BP>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const int f = tid + m * NT;
      const int fm = (T - f) & (T - 1);
      f2 xe[NB][NPI], xo[NB][NPI];
      const unsigned af = lds_off(zbuf + G::nat(f)), ag = lds_off(zbuf + G::nat(fm));
      static_for<0, NB>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        const unsigned bf = af + b * NPI * G::LSEQ * 8, bg = ag + b * NPI * G::LSEQ * 8;
        static_for<0, NPI>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          xe[b][p] = lds_rd_far<p * G::LSEQ * 8>(bf);
          xo[b][p] = lds_rd_far<p * G::LSEQ * 8>(bg);
        });
      });
#pragma unroll
      for (int b = 0; b < NB; ++b) { lds_arrive(xe[b]); lds_arrive(xo[b]); }
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          const f2 zf = xe[b][p], zg = xo[b][p];
          xe[b][p] = add_conj(zf, zg);
          xo[b][p] = sub_conj_divi(zf, zg);
        }
      auto contract = [&](int q, const f4 (&wc)[2 * NPI]) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
          for (int p = 0; p < NPI; ++p) {
            const f4 ha = wc[2 * p], hb = wc[2 * p + 1];
            cmac(ya[m][b][q], xe[b][p], ha.xy); cmac(ya[m][b][q], xo[b][p], ha.zw);
            cmac(yb[m][b][q], xe[b][p], hb.xy); cmac(yb[m][b][q], xo[b][p], hb.zw);
          }
        }
      };
#pragma unroll
      for (int q = 0; q < NPI; q += 2) {
        contract(q, wA);
        if (q + 2 < NPI) issue(m, q + 2, wA);
        else if (m + 1 < BP) issue(m + 1, 0, wA);
        contract(q + 1, wB);
        if (q + 3 < NPI) issue(m, q + 3, wB);
        else if (m + 1 < BP) issue(m + 1, 1, wB);
      }
    });
    __syncthreads();         // every read of this chunk's spectra is done before the next chunk overwrites them
  }

  // -------------------------------------------------- sums -> LDS (two output channels per packed sequence)
#pragma unroll
  for (int m = 0; m < BP; ++m) {
    const int f = tid + m * NT;
    const int fm = (T - f) & (T - 1);
    if (f != 0) {
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (b < wi.nbc) {
#pragma unroll
          for (int q = 0; q < NPI; ++q) {
            f2* zb = zbuf + (b * NPI + q) * G::LSEQ;
            zb[G::nat(f)] = add_pi(ya[m][b][q], yb[m][b][q]);
            zb[G::nat(fm)] = conj_add_iconj(ya[m][b][q], yb[m][b][q]);
          }
        }
    }
  }
  if (sb_act) {              // bins 0 and T/2 belong to these lanes alone
    float* dstf = reinterpret_cast<float*>(zbuf + (sb_b * NPI + (sb_o >> 1)) * G::LSEQ + G::nat(sb_f)) + (sb_o & 1);
    *dstf = sb_acc;
  }
  __syncthreads();
  // -------------------------------------------------- inverse FFT + store
  if (act_in) {
    f2 v[P];
    nat_load<G>(v, zseq, tseq);
    seq_sync<G>();
    passA_fft_twiddle_store_lds<G, +1>(v, zseq, tseq, twl);
    seq_sync<G>();
    passB_load<G>(v, zseq, tseq);
    const int j = passB_compute<G, +1>(v, tseq, twB);
    const int o1 = tseq >> G::LGS;
    const int t0 = wi.tile * a.V;
    const int limit = min(a.V, a.Lfull - t0);
    const int nbase = o1 + P * P * j;
    float* y0 = a.y + ((size_t)(wi.b0 + nb) * a.Cout + cg0) * a.Lout + t0 + nbase;
    float* y1 = y0 + a.Lout;
#pragma unroll
    for (int k = 0; k < P; ++k)
      if (nbase + P * k < limit) { y0[P * k] = v[k].x + bias0; y1[P * k] = v[k].y + bias1; }
  }
}

}  // namespace fc
