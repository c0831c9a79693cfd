// wgrad1d.hpp -- weight gradient of the 1-D convolution.
//
//   dW[o][i][k] = sum_b sum_t dY[b][o][t] * Xp[b][i][t*s + k*d]        (Xp = padded input, s = stride)
//
// (a strided convolution's gradient is read spread over a grid of step s -- sample t sits at position t*s of the
// tile, zeros between -- so everything below is the stride-1 scheme on that zero-stuffed row of (Lout-1)*s + 1 samples)
//
// is a correlation whose lags 0..kd-1 are wanted, summed over the batch and over the whole row.  In the
// frequency domain the sum commutes with the transform: per overlap-save tile (V samples of dY against
// T = V + kd - 1 samples of Xp starting at the same position) the cross-spectrum conj(DY[f]) * X[f] is
// ACCUMULATED over all (batch item, tile) items in registers, and only one inverse transform per (o, i)
// is run at the end.  No kernel-spectrum buffer, no per-call spectrum transform, no role-swapped copies.
//
// One workgroup owns a 4 x 4 block of (output, input) channels and a slice of the items; per item it
// transforms 2 + 2 packed sequences (two real channels per complex sequence, as in the forward
// kernel), NB items at a time.  In the accumulation step a thread owns BP bin pairs (f, T-f) and the 16
// cross-spectra of the block at those bins.  The self-paired bins 0 and T/2 (real) share one complex
// accumulator per (o, i): .x = bin 0, .y = bin T/2.  Partial results (one per slice) are summed by the caller.
#pragma once
#include "conv1d_fused.hpp"

namespace fc {

struct WGradArgs {
  const float* x;      // (B, Cin, L)
  const float* dy;     // (B, Cout, Lout)
  float* part;         // [slices][Cout][Cig][K]
  const f2* twA;       // pass-A table of the tile
  const f2* twB;
  int B, Cin, Cout, G, Cig, Cog;
  int L, pad, pad_mode, Lout;
  int stride, Lext;                // stride of the convolution, extent of the zero-stuffed gradient row (Lout-1)*stride + 1
  int K, dil, V, ntiles;           // taps of THIS segment, dilation, gradient positions per tile (a multiple of stride), tiles per row
  int Krow, tap0;                  // taps per weight row, first tap of this segment (kernels longer than one
  int pos_shift;                   //   tile's lag window run in segments: x is read tap0*dil samples further in)
  int n_items, items_per_slice;    // (b, tile) items in all / per slice (a multiple of NB)
  int nob, nib;                    // 4-channel blocks per group: outputs, inputs
  float scale;                     // 1 / (4 T)
  long long part_stride;           // floats between two slices of `part` (>= Cout*Cig*Krow)
  float* dbpart;                   // optional: bias gradient partials, dbpart[slice*part_stride + channel] (dense kernel,
                                   // first tap segment only): db[o] = sum of dY = bin 0 of the gradient spectra this
                                   // kernel forms anyway -- the separate reduction over dY (a 33 MB read at cfgA) goes away
};

// acc += conj(y) * x
__device__ __forceinline__ void cmacc(f2& acc, f2 y, f2 x) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(y), "v"(x));
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[0,1,0]" : "+v"(acc) : "v"(y), "v"(x));
}

template <int P, int S, int NB, int NT>
__global__ __launch_bounds__(NT, 2) void wgrad1d_kernel(const WGradArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int NSEQ = NB * 4;              // per item: 2 input pairs + 2 output-gradient pairs
  static_assert(NT == NSEQ * G::TS, "one thread slot per point group of every sequence");
  static_assert(G::TS <= 64 && (2 * G::TS) % 64 == 0, "a sequence pair (same role) is a whole number of wavefronts");
  static_assert((T / 2) % NT == 0, "bin pairs divide evenly over the threads");
  constexpr int BP = (T / 2) / NT;
  constexpr int TWN = P * G::N2;
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  f2* twl = lds;
  f2* zbuf = lds + TWN;                     // [NSEQ][LSEQ]: item slot nb: X pair 0, X pair 1, dY pair 0, dY pair 1

  const int tid = threadIdx.x;
  const int sq = tid / G::TS, tseq = tid % G::TS;
  const int nb = __builtin_amdgcn_readfirstlane(sq / 4);                 // item slot (wave-uniform)
  const int role = __builtin_amdgcn_readfirstlane((sq % 4) >> 1);        // 0: input pair, 1: gradient pair
  const int pr = sq & 1;                                                 // which pair of the role
  f2* zseq = zbuf + sq * G::LSEQ;

  int id = blockIdx.x;
  const int ib = id % a.nib; id /= a.nib;
  const int ob = id % a.nob; id /= a.nob;
  const int g = id % a.G;
  const int slice = id / a.G;
  const int item_lo = slice * a.items_per_slice;
  const int item_hi = min(item_lo + a.items_per_slice, a.n_items);

  const PadMap pm = make_padmap(a.pad_mode, a.L);
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const BufRsrc xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.Cin * a.L * 4));
  const BufRsrc yr = make_rsrc(a.dy, (unsigned)((size_t)a.B * a.Cout * a.Lout * 4));
  copy_table_to_lds<TWN, NT>(twl, a.twA, tid);

  f2 acc[BP][4][4];                          // [bin pair][o][i]
  float dbacc[4] = {0.f, 0.f, 0.f, 0.f};     // (thread 0 only: the owner of bin 0)
#pragma unroll
  for (int m = 0; m < BP; ++m)
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[m][o][i] = mk2(0.f, 0.f);
  __syncthreads();

#pragma unroll 1
  for (int base = item_lo; base < item_hi; base += NB) {
    // ------------------------------------------------ load this slot's two rows of the item's tile
    const int item = base + nb;
    const bool act = item < item_hi;
    if (act) {
      const int b = item / a.ntiles, tile = item - b * a.ntiles;
      f2 v[P];
      if (role == 0) {
        const int ci0 = ib * 4 + 2 * pr;
        const bool has0 = ci0 < a.Cig, has1 = ci0 + 1 < a.Cig;
        const unsigned ro0 = ((unsigned)b * (unsigned)a.Cin + (unsigned)(g * a.Cig + ci0)) * (unsigned)a.L * 4u;
        const unsigned ro1 = ro0 + (unsigned)a.L * 4u;
        const int tile_pos = tile * a.V - a.pad + a.pos_shift;
        if (tile_pos >= 0 && tile_pos + T <= a.L && has1) {
          const unsigned v0 = ro0 + (unsigned)(tile_pos + tseq) * 4u, v1 = ro1 + (unsigned)(tile_pos + tseq) * 4u;
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) {
            v[n1].x = buf_load_f32(xr, v0, G::N2 * n1 * 4);
            v[n1].y = buf_load_f32(xr, v1, G::N2 * n1 * 4);
          }
        } else {
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) {
            const int pos = tile_pos + G::N2 * n1 + tseq;
            v[n1].x = buf_load_f32(xr, padded_offset(ro0, pos, a.L, a.pad, pm, has0), 0);
            v[n1].y = buf_load_f32(xr, padded_offset(ro1, pos, a.L, a.pad, pm, has1), 0);
          }
        }
      } else {
        const int co0 = ob * 4 + 2 * pr;
        const bool has0 = co0 < a.Cog, has1 = co0 + 1 < a.Cog;
        const unsigned ro0 = ((unsigned)b * (unsigned)a.Cout + (unsigned)(g * a.Cog + co0)) * (unsigned)a.Lout * 4u;
        const unsigned ro1 = ro0 + (unsigned)a.Lout * 4u;
        const int t0 = tile * a.V;
        const int limit = min(a.V, a.Lext - t0);        // gradient positions of this tile (zero beyond)
        if (a.stride == 1) {
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) {
            const int n = G::N2 * n1 + tseq;
            const bool in = n < limit;
            v[n1].x = buf_load_f32(yr, (in && has0) ? ro0 + (unsigned)(t0 + n) * 4u : 0xFFFFFFFFu, 0);
            v[n1].y = buf_load_f32(yr, (in && has1) ? ro1 + (unsigned)(t0 + n) * 4u : 0xFFFFFFFFu, 0);
          }
        } else {
          const int q0 = t0 / a.stride;                  // (t0 is a multiple of the stride)
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) {
            const int n = G::N2 * n1 + tseq;
            const int nq = n / a.stride;
            const bool in = n < limit && nq * a.stride == n;
            v[n1].x = buf_load_f32(yr, (in && has0) ? ro0 + (unsigned)(q0 + nq) * 4u : 0xFFFFFFFFu, 0);
            v[n1].y = buf_load_f32(yr, (in && has1) ? ro1 + (unsigned)(q0 + nq) * 4u : 0xFFFFFFFFu, 0);
          }
        }
      }
      // ---------------------------------------------- forward FFT (wave-local)
      passA_fft_twiddle_store_lds<G, -1>(v, zseq, tseq, twl);
      seq_sync<G>();
      passB_load<G>(v, zseq, tseq);
      seq_sync<G>();
      const int j = passB_compute<G, -1>(v, tseq, twB);
      const int k1 = tseq >> G::LGS;
      f2* dst = zseq + G::nat(k1 + P * P * j);
#pragma unroll
      for (int k = 0; k < P; ++k) dst[P * k] = v[k];
    }
    __syncthreads();
    // ------------------------------------------------ accumulate the cross-spectra of the 4 x 4 block
#pragma unroll
    for (int m = 0; m < BP; ++m) {
      const int f = tid + m * NT;
      const int fm = (T - f) & (T - 1);
      const unsigned af = lds_off(zbuf + G::nat(f)), ag = lds_off(zbuf + G::nat(fm));
      const unsigned ah = lds_off(zbuf + G::nat(T / 2));
#pragma unroll
      for (int s = 0; s < NB; ++s) {
        if (base + s < item_hi) {            // uniform
          f2 zf[4], zg[4];
          static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            zf[q] = lds_rd_far<(0 + q) * G::LSEQ * 8>(af + s * 4 * G::LSEQ * 8);
            zg[q] = lds_rd_far<(0 + q) * G::LSEQ * 8>(ag + s * 4 * G::LSEQ * 8);
          });
          lds_arrive(zf);
          lds_arrive(zg);
          f2 xs[4], ys[4];                   // 2 * spectrum of input channels 0..3 / gradient channels 0..3 at bin f
          xs[0] = add_conj(zf[0], zg[0]); xs[1] = sub_conj_divi(zf[0], zg[0]);
          xs[2] = add_conj(zf[1], zg[1]); xs[3] = sub_conj_divi(zf[1], zg[1]);
          ys[0] = add_conj(zf[2], zg[2]); ys[1] = sub_conj_divi(zf[2], zg[2]);
          ys[2] = add_conj(zf[3], zg[3]); ys[3] = sub_conj_divi(zf[3], zg[3]);
          if (f != 0) {
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
              for (int i = 0; i < 4; ++i) cmacc(acc[m][o][i], ys[o], xs[i]);
          } else {
            // bins 0 and T/2 are real: (2 Re, 2 Im) of Z[0] and Z[T/2] give the two channels of a pair
            f2 zh[4];
            static_for<0, 4>([&](auto qc) {
              constexpr int q = decltype(qc)::value;
              zh[q] = lds_rd_far<q * G::LSEQ * 8>(ah + s * 4 * G::LSEQ * 8);
            });
            lds_arrive(zh);
            f2 x2[4], y2[4];                 // (bin 0, bin T/2) of every channel
            x2[0] = mk2(2.f * zf[0].x, 2.f * zh[0].x); x2[1] = mk2(2.f * zf[0].y, 2.f * zh[0].y);
            x2[2] = mk2(2.f * zf[1].x, 2.f * zh[1].x); x2[3] = mk2(2.f * zf[1].y, 2.f * zh[1].y);
            y2[0] = mk2(2.f * zf[2].x, 2.f * zh[2].x); y2[1] = mk2(2.f * zf[2].y, 2.f * zh[2].y);
            y2[2] = mk2(2.f * zf[3].x, 2.f * zh[3].x); y2[3] = mk2(2.f * zf[3].y, 2.f * zh[3].y);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[m][o][i] = pkfma(y2[o], x2[i], acc[m][o][i]);
#pragma unroll
            for (int o = 0; o < 4; ++o) dbacc[o] += y2[o].x;     // 2 * (sum of this tile's gradient samples)
          }
        }
      }
    }
    __syncthreads();
  }

  // -------------------------------------------------- inverse: 16 cross-spectra = 8 packed sequences, NSEQ at a time
  const int ci_base = g * a.Cig + ib * 4, co_base = g * a.Cog + ob * 4;
  if (a.dbpart != nullptr && ib == 0 && tid == 0) {
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (ob * 4 + o < a.Cog) a.dbpart[(size_t)slice * a.part_stride + co_base + o] = 0.5f * dbacc[o];
  }
  constexpr int ROUNDS = (8 + NSEQ - 1) / NSEQ;
#pragma unroll
  for (int r = 0; r < ROUNDS; ++r) {
    // sequence slot q of this round holds (o, input pair ip): packed index r*NSEQ + q = o*2 + ip
#pragma unroll
    for (int m = 0; m < BP; ++m) {
      const int f = tid + m * NT;
      const int fm = (T - f) & (T - 1);
#pragma unroll
      for (int q = 0; q < NSEQ; ++q) {
        const int pk = r * NSEQ + q;
        if (pk < 8) {
          const int o = pk >> 1, ip = pk & 1;
          const f2 A = acc[m][o][2 * ip], Bv = acc[m][o][2 * ip + 1];
          f2* zb = zbuf + q * G::LSEQ;
          if (f != 0) {
            zb[G::nat(f)] = add_pi(A, Bv);
            zb[G::nat(fm)] = conj_add_iconj(A, Bv);
          } else {
            zb[G::nat(0)] = mk2(A.x, Bv.x);
            zb[G::nat(T / 2)] = mk2(A.y, Bv.y);
          }
        }
      }
    }
    __syncthreads();
    const int pk = r * NSEQ + sq;
    if (pk < 8) {
      f2 v[P];
      nat_load<G>(v, zseq, tseq);
      seq_sync<G>();
      passA_fft_twiddle_store_lds<G, +1>(v, zseq, tseq, twl);
      seq_sync<G>();
      passB_load<G>(v, zseq, tseq);
      const int j = passB_compute<G, +1>(v, tseq, twB);
      const int o = pk >> 1, ip = pk & 1;
      const int co = ob * 4 + o, ci = ib * 4 + 2 * ip;
      if (co < a.Cog) {
        float* out0 = a.part + (size_t)slice * a.part_stride + (((size_t)(co_base + o)) * a.Cig + ci) * a.Krow + a.tap0;
        float* out1 = out0 + a.Krow;
        const int nbase = (tseq >> G::LGS) + P * P * j;
        const int kd = (a.K - 1) * a.dil + 1;
#pragma unroll
        for (int k = 0; k < P; ++k) {
          const int lag = nbase + P * k;
          const int tap = lag / a.dil;
          if (lag < kd && tap * a.dil == lag) {
            if (ci < a.Cig) out0[tap] = v[k].x * a.scale;
            if (ci + 1 < a.Cig) out1[tap] = v[k].y * a.scale;
          }
        }
      }
    }
    __syncthreads();
  }
  (void)ci_base;
}

// Depthwise variant (groups == Cin == Cout, a multiple of 8): a workgroup owns the 8 channels of block g
// and a slice of the items; per item it transforms 4 input pairs + 4 gradient pairs and accumulates ONE
// cross-spectrum per channel (8 per bin pair).  part is [slices][C][1][Krow].
template <int P, int S, int NT>
__global__ __launch_bounds__(NT, 2) void wgrad1d_diag_kernel(const WGradArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int NSEQ = 8;
  static_assert(NT == NSEQ * G::TS, "one thread slot per point group of every sequence");
  static_assert(G::TS <= 64 && (4 * G::TS) % 64 == 0, "the sequences of one role are a whole number of wavefronts");
  static_assert((T / 2) % NT == 0, "bin pairs divide evenly over the threads");
  constexpr int BP = (T / 2) / NT;
  constexpr int TWN = P * G::N2;
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  f2* twl = lds;
  f2* zbuf = lds + TWN;                     // [8][LSEQ]: X pairs 0..3, dY pairs 0..3

  const int tid = threadIdx.x;
  const int sq = tid / G::TS, tseq = tid % G::TS;
  const int role = __builtin_amdgcn_readfirstlane(sq >> 2);              // 0: input pair, 1: gradient pair
  const int pr = sq & 3;
  f2* zseq = zbuf + sq * G::LSEQ;
  const int g = blockIdx.x % a.G;           // channel block
  const int slice = blockIdx.x / a.G;
  const int item_lo = slice * a.items_per_slice;
  const int item_hi = min(item_lo + a.items_per_slice, a.n_items);
  const int c0 = g * 8 + 2 * pr;            // this sequence's two channels

  const PadMap pm = make_padmap(a.pad_mode, a.L);
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const BufRsrc xr = make_rsrc(a.x, (unsigned)((size_t)a.B * a.Cin * a.L * 4));
  const BufRsrc yr = make_rsrc(a.dy, (unsigned)((size_t)a.B * a.Cout * a.Lout * 4));
  copy_table_to_lds<TWN, NT>(twl, a.twA, tid);

  f2 acc[BP][4][2];                          // [bin pair][channel pair][even / odd channel]
#pragma unroll
  for (int m = 0; m < BP; ++m)
#pragma unroll
    for (int p = 0; p < 4; ++p) { acc[m][p][0] = mk2(0.f, 0.f); acc[m][p][1] = mk2(0.f, 0.f); }
  __syncthreads();

#pragma unroll 1
  for (int item = item_lo; item < item_hi; ++item) {
    const int b = item / a.ntiles, tile = item - b * a.ntiles;
    {
      f2 v[P];
      if (role == 0) {
        const unsigned ro0 = ((unsigned)b * (unsigned)a.Cin + (unsigned)c0) * (unsigned)a.L * 4u;
        const unsigned ro1 = ro0 + (unsigned)a.L * 4u;
        const int tile_pos = tile * a.V - a.pad + a.pos_shift;
        if (tile_pos >= 0 && tile_pos + T <= a.L) {
          const unsigned v0 = ro0 + (unsigned)(tile_pos + tseq) * 4u, v1 = ro1 + (unsigned)(tile_pos + tseq) * 4u;
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) {
            v[n1].x = buf_load_f32(xr, v0, G::N2 * n1 * 4);
            v[n1].y = buf_load_f32(xr, v1, G::N2 * n1 * 4);
          }
        } else {
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) {
            const int pos = tile_pos + G::N2 * n1 + tseq;
            v[n1].x = buf_load_f32(xr, padded_offset(ro0, pos, a.L, a.pad, pm, true), 0);
            v[n1].y = buf_load_f32(xr, padded_offset(ro1, pos, a.L, a.pad, pm, true), 0);
          }
        }
      } else {
        const unsigned ro0 = ((unsigned)b * (unsigned)a.Cout + (unsigned)c0) * (unsigned)a.Lout * 4u;
        const unsigned ro1 = ro0 + (unsigned)a.Lout * 4u;
        const int t0 = tile * a.V;
        const int limit = min(a.V, a.Lext - t0);
        const int q0 = t0 / a.stride;
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          const int n = G::N2 * n1 + tseq;
          const int nq = a.stride == 1 ? n : n / a.stride;
          const bool in = n < limit && nq * a.stride == n;
          v[n1].x = buf_load_f32(yr, in ? ro0 + (unsigned)(q0 + nq) * 4u : 0xFFFFFFFFu, 0);
          v[n1].y = buf_load_f32(yr, in ? ro1 + (unsigned)(q0 + nq) * 4u : 0xFFFFFFFFu, 0);
        }
      }
      passA_fft_twiddle_store_lds<G, -1>(v, zseq, tseq, twl);
      seq_sync<G>();
      passB_load<G>(v, zseq, tseq);
      seq_sync<G>();
      const int j = passB_compute<G, -1>(v, tseq, twB);
      const int k1 = tseq >> G::LGS;
      f2* dst = zseq + G::nat(k1 + P * P * j);
#pragma unroll
      for (int k = 0; k < P; ++k) dst[P * k] = v[k];
    }
    __syncthreads();
    static_for<0, BP>([&](auto mc) {
      constexpr int m = decltype(mc)::value;
      const int f = tid + m * NT;
      const int fm = (T - f) & (T - 1);
      const unsigned af = lds_off(zbuf + G::nat(f)), ag = lds_off(zbuf + G::nat(fm));
      f2 zf[8], zg[8];
      static_for<0, 8>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        zf[q] = lds_rd_far<q * G::LSEQ * 8>(af);
        zg[q] = lds_rd_far<q * G::LSEQ * 8>(ag);
      });
      lds_arrive(zf);
      lds_arrive(zg);
      if (f != 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          cmacc(acc[m][p][0], add_conj(zf[4 + p], zg[4 + p]), add_conj(zf[p], zg[p]));
          cmacc(acc[m][p][1], sub_conj_divi(zf[4 + p], zg[4 + p]), sub_conj_divi(zf[p], zg[p]));
        }
      } else {
        // bins 0 and T/2 are real: accumulate them as one (bin 0, bin T/2) pair per channel
        const unsigned ah = lds_off(zbuf + G::nat(T / 2));
        f2 zh[8];
        static_for<0, 8>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          zh[q] = lds_rd_far<q * G::LSEQ * 8>(ah);
        });
        lds_arrive(zh);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          acc[m][p][0] = pkfma(mk2(2.f * zf[4 + p].x, 2.f * zh[4 + p].x), mk2(2.f * zf[p].x, 2.f * zh[p].x), acc[m][p][0]);
          acc[m][p][1] = pkfma(mk2(2.f * zf[4 + p].y, 2.f * zh[4 + p].y), mk2(2.f * zf[p].y, 2.f * zh[p].y), acc[m][p][1]);
        }
      }
    });
    __syncthreads();
  }

  // -------------------------------------------------- 8 cross-spectra = 4 packed sequences -> inverse -> taps
#pragma unroll
  for (int m = 0; m < BP; ++m) {
    const int f = tid + m * NT;
    const int fm = (T - f) & (T - 1);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const f2 A = acc[m][p][0], Bv = acc[m][p][1];
      f2* zb = zbuf + p * G::LSEQ;
      if (f != 0) {
        zb[G::nat(f)] = add_pi(A, Bv);
        zb[G::nat(fm)] = conj_add_iconj(A, Bv);
      } else {
        zb[G::nat(0)] = mk2(A.x, Bv.x);
        zb[G::nat(T / 2)] = mk2(A.y, Bv.y);
      }
    }
  }
  __syncthreads();
  if (sq < 4) {
    f2 v[P];
    nat_load<G>(v, zseq, tseq);
    seq_sync<G>();
    passA_fft_twiddle_store_lds<G, +1>(v, zseq, tseq, twl);
    seq_sync<G>();
    passB_load<G>(v, zseq, tseq);
    const int j = passB_compute<G, +1>(v, tseq, twB);
    float* out0 = a.part + (size_t)slice * a.part_stride + (size_t)c0 * a.Krow + a.tap0;
    float* out1 = out0 + a.Krow;
    const int nbase = (tseq >> G::LGS) + P * P * j;
    const int kd = (a.K - 1) * a.dil + 1;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int lag = nbase + P * k;
      const int tap = lag / a.dil;
      if (lag < kd && tap * a.dil == lag) { out0[tap] = v[k].x * a.scale; out1[tap] = v[k].y * a.scale; }
    }
  }
}

}  // namespace fc
