// dense1d.hpp -- 1-D FFT convolution for MANY channels per group: spectra through HBM, contraction on the matrix pipe.
//
// The fused kernels (conv1d_pers / conv1d_wide) keep a tile's spectra in LDS and contract them on the VALU; with
// more than 8 input AND output channels per group every out-chunk workgroup of conv1d_wide repeats the forward
// transforms of all input chunks, and the per-bin contraction (functional.py:11-16, `complex_matmul`: an einsum over
// the input channels) is a dense [tiles x Cin] x [Cin x Cout] complex product per frequency bin -- the one place
// of the path where MFMA has a second large dimension to work on.  Three launches per slab of M = batch x tiles rows:
//
//   dense_fwd   tile -> packed-pair FFT -> unpacked spectra   X[g][f][m][i]   (f = 0 .. T/2, complex, 2 X really)
//   dense_gemm  per (g, f):  Y[m][o] = sum_i X[m][i] * H[i][o]  as the real product  [Xr Xi] x [[Hr Hi], [-Hi Hr]]
//               on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: same arithmetic type as the VALU kernels)
//   dense_inv   Y[g][f][m][o] -> packed pairs -> inverse FFT -> valid window + bias (functional.py:76-87)
//
// H is the kernel spectrum of spectrum1d.hpp (conjugated, scaled by 1/(2T)) re-laid bin-major by dense_spec.
// Rows of X / Y are channel-contiguous (Kc / Nc complex values), so the transposing stores and loads of the two FFT
// kernels move 256-byte runs (16 channel pairs x 16 bytes) and the GEMM reads its A panel in whole rows.
#pragma once
#include "nd_passes.hpp"

namespace fc {

struct DenseArgs {
  const float* x;        // (B, Cin, L)
  float* y;              // (B, Cout, Lout)
  const float* bias;     // (Cout) or null
  f2* X;                 // [G][NF][mcount][Kc]
  f2* Y;                 // [G][NF][mcount][Nc]
  const f2* Hd;          // [G][NF][Kc][Nc]
  const f2* twA;
  const f2* twB;
  int B, Cin, Cout, G, Cig, Cog;
  int Kc, Nc;            // channels per group padded to a multiple of 8
  int L, pad, pad_mode, V, ntiles, Lfull, Lout;
  int m0, mcount;        // this slab: rows m0 .. m0 + mcount - 1 of M = B * ntiles (row m = batch m / ntiles, tile m % ntiles)
  int cus;               // CUs of the device (grid of the persistent GEMM)
};

// ------------------------------------------------------------------------------------------ dense_fwd
template <int P, int S, int NSEQ, int NT>
__global__ __launch_bounds__(NT) void dense_fwd_kernel(const DenseArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T, NF = T / 2 + 1;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  static_assert(NT == NSEQ * G::TS, "one thread slot per point group of every sequence");
  static_assert(NT % NSEQ == 0 && (NSEQ & (NSEQ - 1)) == 0, "bins are dealt out to groups of NSEQ lanes");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  const int ncb = (a.Kc / 2 + NSEQ - 1) / NSEQ;          // channel blocks of NSEQ pairs
  int id = blockIdx.x;
  const int ml = id % a.mcount; id /= a.mcount;
  const int cb = id % ncb;
  const int g = id / ncb;
  const int m = a.m0 + ml, b = m / a.ntiles, tile = m % a.ntiles;
  const int ci0 = 2 * (cb * NSEQ + sq);                   // this sequence's channel pair inside the group
  const bool act = ci0 < a.Kc;
  const bool has0 = ci0 < a.Cig, has1 = ci0 + 1 < a.Cig;  // channels past Cig are padding: zero spectra

  f2 v[P];
  {
    const PadMap pm = make_padmap(a.pad_mode, a.L);
    const float* xbase = a.x + ((size_t)b * a.Cin + (size_t)g * a.Cig) * a.L;
    const BufRsrc xg = make_rsrc(xbase, (unsigned)((size_t)a.Cig * a.L * 4));
    const int pos0 = tile * a.V - a.pad;
    const bool interior = (pos0 >= 0) && (pos0 + T <= a.L);
    const unsigned ro0 = (unsigned)ci0 * (unsigned)a.L * 4u, ro1 = ro0 + (unsigned)a.L * 4u;
    if (interior && has1) {
      const unsigned v0 = ro0 + (unsigned)(pos0 + tseq) * 4u, v1 = ro1 + (unsigned)(pos0 + tseq) * 4u;
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        v[n1].x = buf_load_f32(xg, v0, G::N2 * n1 * 4);
        v[n1].y = buf_load_f32(xg, v1, G::N2 * n1 * 4);
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int pos = pos0 + G::N2 * n1 + tseq;
        v[n1].x = buf_load_f32(xg, padded_offset(ro0, pos, a.L, a.pad, pm, has0), 0);
        v[n1].y = buf_load_f32(xg, padded_offset(ro1, pos, a.L, a.pad, pm, has1), 0);
      }
    }
  }
  fwd_from_regs<G>(v, lds + sq * LSEQP, tseq, act, twA, twB);
  __syncthreads();
  // unpack (2 X_even, 2 X_odd of every pair) and store channel-contiguous: NSEQ neighbouring lanes write the
  // NSEQ pairs of one bin = 16 * NSEQ bytes in one run
  const int s = tid % NSEQ, fsub = tid / NSEQ;
  constexpr int NFS = NT / NSEQ;
  const bool st_ok = 2 * (cb * NSEQ + s) < a.Kc;
  const f2* z = lds + s * LSEQP;
  f4* out = reinterpret_cast<f4*>(a.X + (((size_t)g * NF) * a.mcount + ml) * a.Kc + 2 * (cb * NSEQ + s));
  const size_t fstride = (size_t)a.mcount * a.Kc / 2;      // f4 units between bins
  // (all of a lane's bins read from LDS first -- asm reads, hipcc would sink them back to their uses -- then unpacked
  // and stored: one LDS latency instead of one per bin)
  constexpr int NITER = (NF + NFS - 1) / NFS;
  f2 zf[NITER], zg[NITER];
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int f = min(fsub + NFS * it, NF - 1);            // (the last round is partly idle: clamped, not stored)
    zf[it] = lds_rd<0>(lds_off(z + G::nat(f)));
    zg[it] = lds_rd<0>(lds_off(z + G::nat((T - f) & (T - 1))));
  }
  lds_arrive(zf); lds_arrive(zg);
#pragma unroll
  for (int it = 0; it < NITER; ++it) {
    const int f = fsub + NFS * it;
    const f2 xe = add_conj(zf[it], zg[it]), xo = sub_conj_divi(zf[it], zg[it]);
    f4 o; o.x = xe.x; o.y = xe.y; o.z = xo.x; o.w = xo.y;
    if (st_ok && f < NF) out[(size_t)f * fstride] = o;
  }
}

// ------------------------------------------------------------------------------------------ dense_inv
template <int P, int S, int NSEQ, int NT>
__global__ __launch_bounds__(NT) void dense_inv_kernel(const DenseArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T, NF = T / 2 + 1;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  static_assert(NT == NSEQ * G::TS, "one thread slot per point group of every sequence");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  const int ncb = (a.Nc / 2 + NSEQ - 1) / NSEQ;
  int id = blockIdx.x;
  const int ml = id % a.mcount; id /= a.mcount;
  const int cb = id % ncb;
  const int g = id / ncb;
  const int m = a.m0 + ml, b = m / a.ntiles, tile = m % a.ntiles;
  const int co0 = 2 * (cb * NSEQ + sq);
  const bool act = co0 < a.Nc;
  const int cg0 = g * a.Cog + co0;
  const bool ok0 = co0 < a.Cog, ok1 = co0 + 1 < a.Cog;
  float bias0 = (a.bias && ok0) ? a.bias[cg0] : 0.f;
  float bias1 = (a.bias && ok1) ? a.bias[cg0 + 1] : 0.f;
  asm volatile("" : "+v"(bias0), "+v"(bias1));   // (arrived here, not inside the guarded stores: see rows_c2r)
  {
    // gather: lane (s, bin) reads the two channels of pair s at one bin (16 bytes; NSEQ lanes = one run) and
    // re-packs them as Z[f] = Ye + i Yo, Z[T-f] = conj(Ye) + i conj(Yo)
    const int s = tid % NSEQ, fsub = tid / NSEQ;
    constexpr int NFS = NT / NSEQ;
    const bool ld_ok = 2 * (cb * NSEQ + s) < a.Nc;
    f2* z = lds + s * LSEQP;
    // all of a lane's bins are requested before the first one is used (as a loop of load -> re-pack -> LDS store the
    // gather paid one memory latency per bin: 17 in a row); bins past T/2 and idle pairs read outside the resource
    constexpr int NITER = (NF + NFS - 1) / NFS;
    const BufRsrc yr = make_rsrc(a.Y + ((size_t)g * NF) * a.mcount * a.Nc, (unsigned)((size_t)NF * a.mcount * a.Nc * 8));
    const unsigned base = (unsigned)(((size_t)ml * a.Nc + 2 * (cb * NSEQ + s)) * 8);
    const unsigned fstride = (unsigned)((size_t)a.mcount * a.Nc * 8);
    f4 yv[NITER];
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int f = fsub + NFS * it;
      yv[it] = buf_load_f32x4(yr, (ld_ok && f < NF) ? base + (unsigned)f * fstride : 0xFFFFFFFFu, 0);
    }
#pragma unroll
    for (int it = 0; it < NITER; ++it) {
      const int f = fsub + NFS * it;
      if (ld_ok && f < NF) {
        const f2 ya = mk2(yv[it].x, yv[it].y), yb = mk2(yv[it].z, yv[it].w);
        z[G::nat(f)] = add_pi(ya, yb);
        if (f != 0 && f != T / 2) z[G::nat(T - f)] = conj_add_iconj(ya, yb);
      }
    }
  }
  __syncthreads();
  f2 v[P];
  const int j = inv_to_regs<G>(v, lds + sq * LSEQP, tseq, act, twA, twB);
  if (act) {
    const int o1 = tseq >> G::LGS;
    const int t0 = tile * a.V;
    const int limit = min(a.V, a.Lfull - t0);
    const int nbase = o1 + P * P * j;
    if constexpr (S == 1) {
      // as in conv1d_pers: rows below the last fully valid row ka in blocks of 8 behind one scalar branch, the block
      // that holds row ka with per-row tests and an out-of-range offset for lanes past the window; channels that do
      // not exist (padding of the pair) start from an offset outside the resource
      const int ka = limit >= P ? (limit - P) / P + 1 : 0;
      const BufRsrc yr = make_rsrc(a.y + ((size_t)b * a.Cout + (size_t)g * a.Cog) * a.Lout, (unsigned)((size_t)a.Cog * a.Lout * 4));
      const unsigned vo = (unsigned)(((size_t)co0 * a.Lout + (size_t)(t0 + nbase)) * 4);
      const unsigned vo0 = ok0 ? vo : 0x80000000u, vo1 = ok1 ? vo + (unsigned)a.Lout * 4u : 0x80000000u;
      static_for<0, P / 8>([&](auto bc) {
        constexpr int k0 = 8 * decltype(bc)::value;
        if (ka >= k0 + 8) {
          static_for<k0, k0 + 8>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            buf_store_f32(v[k].x + bias0, yr, vo0, P * k * 4);
            buf_store_f32(v[k].y + bias1, yr, vo1, P * k * 4);
          });
        } else if (ka >= k0) {
          static_for<k0, k0 + 8>([&](auto kc) {
            constexpr int k = decltype(kc)::value;
            if (k <= ka) {
              const unsigned dead = (nbase + P * k < limit) ? 0u : 0x80000000u;
              buf_store_f32(v[k].x + bias0, yr, vo0 | dead, P * k * 4);
              buf_store_f32(v[k].y + bias1, yr, vo1 | dead, P * k * 4);
            }
          });
        }
      });
    } else {
      float* y0 = a.y + ((size_t)b * a.Cout + cg0) * a.Lout + (size_t)(t0 + nbase);
      float* y1 = y0 + a.Lout;
#pragma unroll
      for (int k = 0; k < P; ++k)
        if (nbase + P * k < limit) {
          if (ok0) y0[P * k] = v[k].x + bias0;
          if (ok1) y1[P * k] = v[k].y + bias1;
        }
    }
  }
}

// ------------------------------------------------------------------------------------------ dense_gemm
// Persistent: one workgroup per CU walks units (group g, bin f, block of 8*NCT output channels, block of 128 rows m) and,
// inside a unit, K chunks of KCH input channels.  Wave w owns the 16 real output columns of column tile w % NCT (= 8
// complex output channels) and the row tiles w / NCT, w / NCT + 8 / NCT, ...  The B operand (this bin's
// [[Hr Hi], [-Hi Hr]] columns) lives in registers, the A panel (128 rows x 2*KCH floats, rows as stored by dense_fwd) in
// LDS, read by all waves.  The loads of step s+1 (panel into registers, B values) are issued BEFORE the MFMAs of step s
// and land in the other LDS panel at the top of the next step: the one-shot form of this kernel (load, multiply, store
// per workgroup; two workgroups per CU) moved its 114 MB at 1.9 TB/s and took 62 us where the MFMAs need 20.
// MFMA operand layout (checked by scripts/ubench/ubench_mfma16.hip): A[i][k] in lane 16 k + i, B[k][j] in lane
// 16 k + j, D[4 (lane / 16) + r][lane % 16] in register r.
typedef float v4f __attribute__((ext_vector_type(4)));
// MB = rows per unit: 128, or 32 for slabs of few rows (M = batch x tiles <= 64: the units are then all spectrum
// traffic and hardly any arithmetic; 35 KB of LDS and half the registers let two workgroups per CU keep twice the
// loads in flight)

// K2N = pairs of MFMA k steps per K chunk: a chunk holds KCH = 4 K2N complex input channels (16, 32 or 64; the
// launcher picks the smallest that covers Kc, or 64 and several chunks).  The k loops are unconditional -- a
// guarded MFMA makes hipcc copy the accumulator around every single instruction -- so the last chunk's missing
// channels are zero columns of the LDS panel and zero B values.
template <int NCT, int K2N, int kDenseMB>
__global__ __launch_bounds__(512, kDenseMB >= 128 ? 2 : 4) void dense_gemm_kernel(const DenseArgs a, int nf) {
  constexpr int NW = 8, MS = NW / NCT;          // waves; row-tile subsets
  constexpr int MTW = (kDenseMB / 16) / MS;     // row tiles per wave
  constexpr int KCH = 4 * K2N;                  // complex input channels per K chunk
  constexpr int RS = 2 * KCH + 4;               // LDS row pitch in floats: conflict-free 8-byte reads of (row i, k pair)
  constexpr int NCOL = 16 * NCT;                // real columns of a unit
  constexpr int OS = NCOL + 4;                  // LDS row pitch of the staged output block (floats)
  constexpr int PANEL = kDenseMB * (RS > OS ? RS : OS);   // floats per LDS panel (also holds an output block)
  constexpr int Q4 = KCH / 2;                   // float4 per panel row
  constexpr int NIT = (kDenseMB * Q4 + 511) / 512;   // float4 per thread and panel (MB = 32 with small chunks: part of the threads)
  static_assert((kDenseMB / 16) % MS == 0 && kDenseMB / 16 >= MS, "every wave owns whole row tiles");
  static_assert(NCT == 2 || NCT == 4 || NCT == 8, "column tiles per unit");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  float* pbase = reinterpret_cast<float*>(lds);
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = w % NCT, ms = w / NCT;
  const int nmb = (a.mcount + kDenseMB - 1) / kDenseMB;
  const int nnb = (a.Nc + 8 * NCT - 1) / (8 * NCT);
  const int nkc = (a.Kc + KCH - 1) / KCH;
  const long long U = (long long)a.G * nf * nnb * nmb;
  const int i16 = lane & 15, kq = lane >> 4;
  const int nr = ct * 16 + i16;                 // this lane's real output column inside the unit

  // unit u -> (bin fastest: neighbouring workgroups read neighbouring X / Y rows; row blocks of one bin share H in L2)
  auto decode = [&](long long u, int& g, int& f, int& nb, int& mb) {
    f = (int)(u % nf); u /= nf;
    mb = (int)(u % nmb); u /= nmb;
    nb = (int)(u % nnb);
    g = (int)(u / nnb);
  };
  // loads of one step: the chunk's panel rows (16 bytes per lane, rows past the slab / channels past the chunk read as
  // zero through the buffer resource) and this lane's B values: k step kk covers the real rows
  // k = 8 (kk / 2) + 2 kq + (kk % 2) of the chunk
  auto issue = [&](long long u, int kc, f4 (&val)[NIT], float (&bf)[2 * K2N]) __attribute__((always_inline)) {
    int g, f, nb, mb;
    decode(u, g, f, nb, mb);
    const int kc0 = kc * KCH, kch = min(KCH, a.Kc - kc0), q4 = kch / 2, row0 = mb * kDenseMB;
    const BufRsrc xr = make_rsrc(a.X + (((size_t)g * nf + f) * a.mcount) * a.Kc, (unsigned)((size_t)a.mcount * a.Kc * 8));
    const BufRsrc hr = make_rsrc(a.Hd + (((size_t)g * nf + f) * a.Kc) * a.Nc, (unsigned)((size_t)a.Kc * a.Nc * 8));
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 512 * it, r = idx / Q4, c4 = idx % Q4;
      const unsigned off = (c4 < q4 && r < kDenseMB) ? (unsigned)((((row0 + r) * a.Kc) + kc0 + 2 * c4) * 8) : 0xFFFFFFFFu;
      val[it] = buf_load_f32x4(xr, off, 0);
    }
    // B values: no per-step conditions at all (hipcc turns them into branches with a full wait in front of every merge):
    // input channels past Kc fall outside the resource by themselves (it covers exactly this bin's Kc x Nc matrix), a
    // lane whose output column does not exist starts from an offset that is outside it for every k
    const int o = nb * 8 * NCT + (nr >> 1), comp = nr & 1;
    const unsigned obase = (o < a.Nc) ? (unsigned)(o * 8) : 0x80000000u;
    const unsigned rowb = (unsigned)a.Nc * 8u;              // bytes per input channel row of H
#pragma unroll
    for (int kk = 0; kk < 2 * K2N; ++kk) {
      const int k = 8 * (kk >> 1) + 2 * kq + (kk & 1);
      const int part = kk & 1;                              // 0: times Xr, 1: times Xi (takes the other component of H)
      const unsigned off = obase + (unsigned)(kc0 + (k >> 1)) * rowb + (unsigned)((comp ^ part) * 4);
      bf[kk] = buf_load_f32(hr, off, 0);        // raw value: nothing may touch it here, or the wave would wait for the load
    }
  };
  const float bsign = (nr & 1) ? 1.f : -1.f;     // odd k steps (times Xi) take -Hi for a real-part column, +Hr for an imaginary-part one

  v4f acc[MTW];
#pragma unroll
  for (int t = 0; t < MTW; ++t) acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
  f4 val[NIT];
  float bnext[2 * K2N], bcur[2 * K2N];
  long long u = blockIdx.x;
  int kc = 0, cur = 0;
  if (u < U) issue(u, 0, val, bnext);
  while (u < U) {
    int g, f, nb, mb;
    decode(u, g, f, nb, mb);
    const int row0 = mb * kDenseMB;
    float* pa = pbase + cur * PANEL;
    __syncthreads();                                       // the other panel's readers (MFMAs two steps ago, staged
                                                           // output one step ago) are done
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = tid + 512 * it, r = idx / Q4, c4 = idx % Q4;
      if (kDenseMB * Q4 % 512 == 0 || r < kDenseMB) *reinterpret_cast<f4*>(pa + r * RS + 4 * c4) = val[it];
    }
#pragma unroll
    for (int kk = 0; kk < 2 * K2N; ++kk) bcur[kk] = (kk & 1) ? bsign * bnext[kk] : bnext[kk];
    __syncthreads();
    // ---- next step's loads travel while this step multiplies
    long long un = u;
    int kcn = kc + 1;
    if (kcn == nkc) { kcn = 0; un = u + gridDim.x; }
    if (un < U) issue(un, kcn, val, bnext);
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
      const int mt = ms + MS * t;
      if (row0 + mt * 16 < a.mcount) {                     // (wave-uniform: whole row tiles past the slab are skipped)
        const float* prow = pa + (mt * 16 + i16) * RS + 2 * kq;
        v4f c = acc[t];
        // the whole row tile's A values are requested first (plain ds_read_b64 through asm: left to hipcc the reads
        // sink next to their MFMAs, one full LDS latency per four of them)
        constexpr int GS = (kDenseMB >= 128 || K2N <= 8) ? K2N : K2N / 2;   // (the small unit has 128 VGPRs: half rows at a time)
        static_for<0, K2N / GS>([&](auto gc) {
          constexpr int k0 = GS * decltype(gc)::value;
          f2 av[GS];
          lds_read_strided<GS, 4>(av, reinterpret_cast<const f2*>(prow + 8 * k0));
          lds_arrive(av);
#pragma unroll
          for (int k2 = 0; k2 < GS; ++k2) {
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k2].x, bcur[2 * (k0 + k2)], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[k2].y, bcur[2 * (k0 + k2) + 1], c, 0, 0, 0);
          }
        });
        acc[t] = c;
      }
    }
    if (kc == nkc - 1) {
      // ---- unit done: register r of row tile mt is row 16 mt + 4 (lane / 16) + r, this lane's column.  The tiles go
      // through the OTHER panel (free until the top of the next step) so that whole rows leave in 16-byte pieces.
      float* po = pbase + (cur ^ 1) * PANEL;
#pragma unroll
      for (int t = 0; t < MTW; ++t) {
        const int mt = ms + MS * t;
#pragma unroll
        for (int r = 0; r < 4; ++r) po[(mt * 16 + 4 * kq + r) * OS + nr] = acc[t][r];
        acc[t] = v4f{0.f, 0.f, 0.f, 0.f};
      }
      __syncthreads();
      constexpr int O4 = NCOL / 4;                         // float4 per output row of the unit
      float* yf = reinterpret_cast<float*>(a.Y + (((size_t)g * nf + f) * a.mcount) * a.Nc) + (size_t)nb * NCOL;
      const int ncol_ok = min(NCOL, 2 * a.Nc - nb * NCOL);  // real columns that exist (a multiple of 16)
      for (int idx = tid; idx < kDenseMB * O4; idx += 512) {
        const int r = idx / O4, c4 = idx % O4;
        if (row0 + r < a.mcount && 4 * c4 < ncol_ok)
          *reinterpret_cast<f4*>(yf + (size_t)(row0 + r) * (2 * a.Nc) + 4 * c4) = *reinterpret_cast<const f4*>(po + r * OS + 4 * c4);
      }
    }
    u = un; kc = kcn; cur ^= 1;
  }
}

// ------------------------------------------------------------------------------------------ dense_spec
// Kernel spectrum of spectrum1d.hpp ([g][o][input pair][T/2] float4, bins 0 and T/2 packed into slot 0) -> bin-major
// complex matrices Hd[g][f][i][o], f = 0 .. T/2.
struct DenseSpecArgs {
  const f4* wspec;
  f2* Hd;
  int G, Kc, Nc, T;
};
template <int UNUSED>   // (a template only so that every translation unit may carry the definition)
__global__ __launch_bounds__(256) void dense_spec_kernel(const DenseSpecArgs a) {
  const int NF = a.T / 2 + 1;
  const size_t total = (size_t)a.G * NF * a.Kc * a.Nc;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int o = (int)(idx % a.Nc);
    size_t r = idx / a.Nc;
    const int i = (int)(r % a.Kc); r /= a.Kc;
    const int f = (int)(r % NF);
    const int g = (int)(r / NF);
    const f4* row = a.wspec + (((size_t)g * a.Nc + o) * (a.Kc / 2) + (i >> 1)) * (a.T / 2);
    f2 h;
    if (f == 0 || f == a.T / 2) {
      const f4 h0 = row[0];
      const float re = (f == 0) ? ((i & 1) ? h0.z : h0.x) : ((i & 1) ? h0.w : h0.y);
      h = mk2(re, 0.f);
    } else {
      const f4 hv = row[f];
      h = (i & 1) ? mk2(hv.z, hv.w) : mk2(hv.x, hv.y);
    }
    a.Hd[idx] = h;
  }
}

}  // namespace fc
