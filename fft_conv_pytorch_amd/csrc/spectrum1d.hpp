// spectrum1d.hpp -- kernel (weight) transform for the fused 1-D path.
//
// Replaces rows a2 + a6 of SURVEY 8a (functional.py:49-57 and :71): dilation is a
// scatter of the taps to positions k*d, the zero padding to the tile length T is
// implicit, two real kernels (o, 2ip), (o, 2ip+1) ride one complex FFT, and the
// result is stored conjugated, scaled by 1/(2T) (inverse-FFT norm and the 1/2 of
// the spectrum unpacking) in the layout the fused kernel's mix step streams:
//   wspec[((g*Cog_pad + o)*(Cig_pad/2) + ip)*(T/2) + f] = {H(o,2ip)[f], H(o,2ip+1)[f]}
// with H[.][0] = {Re H[0], Re H[T/2]} (both bins are real).  Phantom channels
// (padding up to the chunk size) get zeros.
#pragma once
#include "fft_engine.hpp"

namespace fc {

struct Spec1dArgs {
  const float* w;      // (Cout, Cig, K)
  f4* wspec;
  const f2* twA;
  const f2* twB;
  int G, Cig, Cog, Cig_pad, Cog_pad, K, dil, nseq;
  int gs;              // > 0: the 8 x 8 blocks are block-diagonal, built from groups of gs channels (2 or 4) of a
                       // (C, gs, K) weight tensor; entries that cross a group stay zero
  int Krow, k0;        // taps per weight row and first tap of this kernel segment (K = taps of the segment)
  int transposed;      // kernel is (Cin, Cout/g, K): swap in/out inside the group and flip the taps
  unsigned w_bytes;    // size of the weight tensor when it is below 2 GiB (branch-free buffer loads), else 0
};

template <int P, int S, int NT>
__global__ __launch_bounds__(NT) void spectrum1d_kernel(const Spec1dArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int SEQ_PER_WG = NT / G::TS;
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const int tid = threadIdx.x;
  const int sl = tid / G::TS, tseq = tid % G::TS;
  const int seq = blockIdx.x * SEQ_PER_WG + sl;
  const bool act = seq < a.nseq;
  f2* z = lds + sl * G::LSEQ;
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));

  const int npi = a.Cig_pad / 2;
  const int ip = act ? seq % npi : 0;
  const int o = act ? (seq / npi) % a.Cog_pad : 0;
  const int g = act ? seq / (npi * a.Cog_pad) : 0;
  {
    f2 v[P];
    bool has0 = act && o < a.Cog && 2 * ip < a.Cig;
    bool has1 = act && o < a.Cog && 2 * ip + 1 < a.Cig;
    const float* w0 = a.transposed ? a.w + ((size_t)(g * a.Cig + 2 * ip) * a.Cog + o) * a.Krow
                                   : a.w + ((size_t)(g * a.Cog + o) * a.Cig + 2 * ip) * a.Krow;
    const float* w1 = w0 + (a.transposed ? (size_t)a.Cog * a.Krow : (size_t)a.Krow);
    if (a.gs > 0) {
      // block g holds 8 / gs original groups; (o, input pair ip) is live only inside one of them
      const int gl = o / a.gs, ol = o % a.gs, il = (2 * ip) % a.gs;
      const bool same = (2 * ip) / a.gs == gl;
      has0 = has0 && same; has1 = has1 && same;
      const size_t grp = (size_t)g * (8 / a.gs) + gl;                  // original group
      w0 = a.transposed ? a.w + ((grp * a.gs + il) * a.gs + ol) * a.Krow     // (Cin, gs, K): [input][output in group]
                        : a.w + ((grp * a.gs + ol) * a.gs + il) * a.Krow;    // (Cout, gs, K): [output][input in group]
      w1 = w0 + (a.transposed ? (size_t)a.gs * a.Krow : (size_t)a.Krow);
      if (!same) { w0 = a.w; w1 = a.w; }
    }
    if (a.w_bytes != 0) {
      // every tap of the thread requested at once through a buffer resource over the weight tensor: a missing tap (zero
      // padding, dilation gap, phantom channel) is an offset outside it.  Written as `hit ? w[ts] : 0` each of the 2*P
      // loads sat in its own branch behind a full wait -- 64 memory latencies in a row were the 10 us of this kernel.
      const BufRsrc wr = make_rsrc(a.w, a.w_bytes);
      const unsigned o0 = (unsigned)((w0 - a.w) * 4), o1 = (unsigned)((w1 - a.w) * 4);
      if (a.dil == 1) {
        // undilated (the usual case): tap = position, no emulated integer division per tap (2*P of them were ~half of this
        // kernel's instructions); the tap order of a transposed plan is a sign and a constant
        const int sgn = a.transposed ? -1 : 1, t00 = a.transposed ? a.Krow - 1 - a.k0 : a.k0;
        const unsigned b0 = has0 ? o0 + (unsigned)((t00 + sgn * tseq) * 4) : 0x80000000u;
        const unsigned b1 = has1 ? o1 + (unsigned)((t00 + sgn * tseq) * 4) : 0x80000000u;
        const int step = sgn * G::N2 * 4;
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          const bool hit = G::N2 * n1 + tseq < a.K;
          v[n1].x = buf_load_f32(wr, hit ? b0 + (unsigned)(step * n1) : 0x80000000u, 0);
          v[n1].y = buf_load_f32(wr, hit ? b1 + (unsigned)(step * n1) : 0x80000000u, 0);
        }
      } else {
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          const int n = G::N2 * n1 + tseq;
          const int tap = n / a.dil;
          const bool hit = (tap * a.dil == n) && tap < a.K;
          const int ts = a.transposed ? a.Krow - 1 - (a.k0 + tap) : a.k0 + tap;
          v[n1].x = buf_load_f32(wr, (hit && has0) ? o0 + (unsigned)ts * 4u : 0x80000000u, 0);
          v[n1].y = buf_load_f32(wr, (hit && has1) ? o1 + (unsigned)ts * 4u : 0x80000000u, 0);
        }
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int n = G::N2 * n1 + tseq;
        const int tap = n / a.dil;
        const bool hit = (tap * a.dil == n) && tap < a.K;
        const int ts = a.transposed ? a.Krow - 1 - (a.k0 + tap) : a.k0 + tap;
        v[n1].x = (hit && has0) ? w0[ts] : 0.f;
        v[n1].y = (hit && has1) ? w1[ts] : 0.f;
      }
    }
    passA_fft_twiddle_store<G, -1>(v, z, tseq, twA);
  }
  seq_sync<G>();
  {
    f2 v[P];
    passB_load<G>(v, z, tseq);
    seq_sync<G>();
    const int j = passB_compute<G, -1>(v, tseq, twB);
    const int k1 = tseq >> G::LGS;
    f2* dst = z + G::nat(k1 + P * P * j);
#pragma unroll
    for (int k = 0; k < P; ++k) dst[P * k] = v[k];
  }
  seq_sync<G>();
  if (!act) return;
  const float sc = 0.5f / (float)T;
  f4* out = a.wspec + (size_t)seq * (T / 2);
  // (all bins of this thread read from LDS first, then unpacked and stored: as a loop of read -> unpack -> store every
  // iteration paid the LDS latency, 16 in a row on the 1024 tile, in a kernel whose whole life is ~8 us)
  constexpr int NIT = (T / 2) / G::TS;
  static_assert((T / 2) % G::TS == 0, "whole rounds of bins per thread");
  f2 zf[NIT], zg[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int f = tseq + it * G::TS;
    zf[it] = lds_rd<0>(lds_off(z + G::nat(f)));                       // (asm reads: hipcc would sink them back to their uses)
    zg[it] = lds_rd<0>(lds_off(z + G::nat((T - f) & (T - 1))));
  }
  f2 zh = lds_rd<0>(lds_off(z + G::nat(T / 2)));
  lds_arrive(zf); lds_arrive(zg);
  asm volatile("" : "+v"(zh));
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int f = tseq + it * G::TS;
    f4 h;
    if (f == 0) {
      h.x = zf[it].x * sc; h.y = zh.x * sc; h.z = zf[it].y * sc; h.w = zh.y * sc;
    } else {
      // W_a = (Zf + conj(Zg))/2, W_b = (Zf - conj(Zg))/(2i); H = conj(W)/(2T)
      const float h2 = 0.5f * sc;
      h.x = (zf[it].x + zg[it].x) * h2; h.y = -(zf[it].y - zg[it].y) * h2; h.z = (zf[it].y + zg[it].y) * h2; h.w = -(zg[it].x - zf[it].x) * h2;
    }
    out[f] = h;
  }
}

}  // namespace fc
