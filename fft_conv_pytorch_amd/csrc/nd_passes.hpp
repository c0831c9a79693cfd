// nd_passes.hpp -- separable passes for 2-D / 3-D FFT convolution.
//
// The N-d transform of functional.py:70-75 (rfftn / irfftn over the last n axes)
// is done one axis at a time; every pass reads sequences that are contiguous in
// memory and writes its result TRANSPOSED so that the next pass's axis is
// contiguous again (128-byte segments either way):
//
//   rows_r2c   last axis, real -> half spectrum (two rows ride one complex FFT),
//              padding of every axis folded into the load index maps (a3, a5)
//   c2c_fwd    middle axis of a 3-D problem
//   fusedc     outermost spatial axis: forward FFT, channel contraction against
//              the transformed kernel, inverse FFT (overlap-save tiles) (a7)
//   c2c_inv    middle axis back
//   rows_c2r   last axis back, valid window + stride + bias (a8, a9, a10)
//
// The same passes, fed from the dilated, zero-padded kernel taps instead of the
// signal, produce the kernel spectrum (a2, a6).
#pragma once
#include "conv1d_fused.hpp"

namespace fc {

// Index map of one padded axis: position p in [0, n_padded) -> source index or -1 (zero).
struct AxisMap {
  int size;       // unpadded extent
  int pad;        // left padding (may be negative for a transposed plan)
  int mode;       // PadMode
  int up;         // transposed plan: source spread over a grid of this step
};
__device__ __forceinline__ int axis_src(const AxisMap& m, int p) {   // p: padded coordinate
  const int pos = p - m.pad;
  if (m.up > 1) {
    const int q = pos / m.up;
    return (pos >= 0 && q * m.up == pos && q < m.size) ? q : -1;
  }
  if ((unsigned)pos < (unsigned)m.size) return pos;
  if (pos < -m.pad || pos >= m.size + m.pad || m.mode == PAD_CONSTANT) return -1;
  if (m.mode == PAD_REFLECT) return pos < 0 ? -pos : 2 * (m.size - 1) - pos;
  if (m.mode == PAD_REPLICATE) return pos < 0 ? 0 : m.size - 1;
  return pos < 0 ? pos + m.size : pos - m.size;
}
// Image index as the pass counts it -> image index in the caller's tensor.  Off (identity) for a convolution; the
// weight-gradient plans (fc_wgrad_nd) read the signal and the output gradient with batch and channels exchanged
// and write dW in the weight layout, without a transposed copy:  img = (q0*n1 + q1)*n2 + q2  ->  q0*s0 + q1*s1 + q2*s2.
struct ImgMap {
  int on, n1, n2;
  long long s0, s1, s2;
};
__device__ __forceinline__ size_t map_img(const ImgMap& m, int img) {
  if (!m.on) return (size_t)img;
  const int q2 = img % m.n2, t = img / m.n2;
  return (size_t)((long long)(t / m.n1) * m.s0 + (long long)(t % m.n1) * m.s1 + (long long)q2 * m.s2);
}
// Kernel taps: position p -> tap index p/dil if p is a multiple of dil and in range.
__device__ __forceinline__ int tap_src(int p, int dil, int k) {
  const int t = p / dil;
  return (t * dil == p && t < k) ? t : -1;
}

// ------------------------------------------------------------------------------------------
// forward passes on one sequence whose P inputs per thread are already in registers;
// leaves the natural-order spectrum in lseq.  Contains two sequence syncs (workgroup barriers only
// for the 4096-point tile): every thread of the workgroup must call it (idle threads pass act = false).
template <class G>
__device__ __forceinline__ void fwd_from_regs(f2 (&v)[G::P], f2* lseq, int tseq, bool act, BufRsrc twA, BufRsrc twB) {
  if (act) {
    passA_fft_twiddle_store<G, -1>(v, lseq, tseq, twA);
  }
  seq_sync<G>();
  if (act) passB_load<G>(v, lseq, tseq);
  seq_sync<G>();
  if (act) {
    const int j = passB_compute<G, -1>(v, tseq, twB);
    const int k1 = tseq >> G::LGS;
    f2* dst = lseq + G::nat(k1 + G::P * G::P * j);
#pragma unroll
    for (int k = 0; k < G::P; ++k) dst[G::P * k] = v[k];
  }
}

// inverse passes from the natural-order spectrum in lseq; on return element k of this lane is
// sample n = o1 + P*k + P*P*j (o1 = tseq >> log2 S, j returned).  Two barriers inside.
template <class G>
__device__ __forceinline__ int inv_to_regs(f2 (&v)[G::P], f2* lseq, int tseq, bool act, BufRsrc twA, BufRsrc twB) {
  if (act) nat_load<G>(v, lseq, tseq);
  seq_sync<G>();
  if (act) {
    passA_fft_twiddle_store<G, +1>(v, lseq, tseq, twA);
  }
  seq_sync<G>();
  int j = 0;
  if (act) {
    passB_load<G>(v, lseq, tseq);
    j = passB_compute<G, +1>(v, tseq, twB);
  }
  return j;
}

// same with the pass-A twiddles already requested by the caller (at kernel start, so that their L2 round
// trip hides behind the data loads instead of following the barrier in front of the inverse)
template <class G>
__device__ __forceinline__ int inv_to_regs_pre(f2 (&v)[G::P], const f2 (&w)[G::P], f2* lseq, int tseq, bool act, BufRsrc twB) {
  if (act) nat_load<G>(v, lseq, tseq);
  seq_sync<G>();
  if (act) {
    fft_regs<G::P, +1>(v);
    passA_twiddle_apply<G, +1>(v, w, lseq, tseq);
  }
  seq_sync<G>();
  int j = 0;
  if (act) {
    passB_load<G>(v, lseq, tseq);
    j = passB_compute<G, +1>(v, tseq, twB);
  }
  return j;
}

template <class G>
struct SeqLayout {
  // sequence stride in LDS: == 2 (mod 32) complex slots so that 16 neighbouring sequences read
  // at the same bin hit 16 different bank pairs (transposed stores / tile loads)
  static constexpr int LSEQP = G::LSEQ + ((2 - G::LSEQ % 32) + 32) % 32;
};

// ------------------------------------------------------------------------------------------ rows_r2c
struct RowsR2CArgs {
  const float* src;      // signal (B, C, [Z,] Y, X) or kernel taps (Co, Cig, [Kz,] Ky, Kx)
  f2* dst;           // [(a*NC + c)][Fx][NYa]
  const f2* twA;
  const f2* twB;
  int from_kernel;       // 0: signal with padding maps, 1: dilated kernel taps
  int transposed;        // kernel taps of a transposed plan: (Cin, Cout/g, *k), flipped, in/out swapped per group
  int Cig, Cog;          // (kernel source only)
  AxisMap mx, my, mz;    // signal: per-axis padding maps (mz: the identity {1, 0, constant, 1} for 2-D)
  int kx, ky, kz, dx, dy, dz;   // kernel: taps and dilation per axis
  int NA, NC, NY, NYa;   // images, planes per image (padded), rows per plane (padded), row stride of dst
  int SZ, SY, SX;        // source extents (signal: unpadded sizes; kernel: taps)
  int Fx;                // T/2 (odd-frequency bins)
  unsigned src_bytes;    // size of the source tensor when it fits 32-bit buffer offsets, else 0
  int nxt, Vx;           // overlap-save tiles along x (rows longer than the largest FFT): tile xt holds the padded
                         // positions [xt*Vx, xt*Vx + T); dst has nxt*Fx bin columns per plane (1, - for one tile)
  ImgMap im;             // where image `img` of this pass sits in src (identity unless a weight-gradient plan)
  FastDiv d_nyb, d_nxt, d_nc;   // unit map of the launch (filled by the dispatcher)
  int rowmajor;          // dst [(a*NC + c)][NY][nxt*Fx] instead of the transposed [..][Fx][NYa]: the 2-D pipeline whose column
                         // pass runs one THREAD per sequence with its lanes over neighbouring bin columns (planes3d.hpp colz)
};

template <int P, int S, int NSEQ, int NT>
__global__ __launch_bounds__(NT) void rows_r2c_kernel(const RowsR2CArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  constexpr int RB = 2 * NSEQ;
  static_assert(NT == NSEQ * G::TS, "one thread slot per sequence point group");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  unsigned q;
  const int yb = (int)fdivmod(blockIdx.x, a.d_nyb, &q);
  const int xt = (int)fdivmod(q, a.d_nxt, &q);
  const int c = (int)fdivmod(q, a.d_nc, &q);
  const int img = (int)q;
  const int y0 = yb * RB;
  const int x0 = xt * a.Vx;              // padded position of this tile's first sample

  f2 v[P];
  {
    // two rows per sequence
    const float* rows[2];
    bool ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int yp = y0 + 2 * sq + h;
      int ys, zs;
      if (a.from_kernel) {
        ys = yp < a.NY ? tap_src(yp, a.dy, a.ky) : -1;
        zs = tap_src(c, a.dz, a.kz);
      } else {
        ys = yp < a.NY ? axis_src(a.my, yp) : -1;
        zs = axis_src(a.mz, c);       // (2-D plans carry an identity map here)
      }
      ok[h] = ys >= 0 && zs >= 0;
      size_t simg = map_img(a.im, img);
      if (a.from_kernel && a.transposed) {
        const int o_all = img / a.Cig, i = img % a.Cig;
        simg = (size_t)((o_all / a.Cog) * a.Cig + i) * a.Cog + (o_all % a.Cog);
        ys = a.SY - 1 - ys; zs = a.SZ - 1 - zs;
      }
      rows[h] = a.src + ((simg * a.SZ + (ok[h] ? zs : 0)) * a.SY + (ok[h] ? ys : 0)) * a.SX;
    }
    if (!a.from_kernel && a.mx.up == 1 && a.mx.mode == PAD_CONSTANT && a.src_bytes != 0) {
      // fast path (zero padding): unrolled buffer loads over the whole source tensor; a sample outside
      // its row gets an out-of-range offset and reads as zero -- no masks live across the loads
      const BufRsrc sg = make_rsrc(a.src, a.src_bytes);
      const unsigned ro0 = ok[0] ? (unsigned)((rows[0] - a.src) * 4) : 0xFFFFFFFFu;
      const unsigned ro1 = ok[1] ? (unsigned)((rows[1] - a.src) * 4) : 0xFFFFFFFFu;
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int xs = x0 + G::N2 * n1 + tseq - a.mx.pad;
        const bool in = (unsigned)xs < (unsigned)a.SX;
        v[n1].x = buf_load_f32(sg, (in && ok[0]) ? ro0 + (unsigned)xs * 4u : 0xFFFFFFFFu, 0);
        v[n1].y = buf_load_f32(sg, (in && ok[1]) ? ro1 + (unsigned)xs * 4u : 0xFFFFFFFFu, 0);
      }
    } else {
      f2* col = lds + sq * LSEQP + tseq;
#pragma unroll 1
      for (int n1 = 0; n1 < P; ++n1) {
        const int xp = x0 + G::N2 * n1 + tseq;
        int xs = a.from_kernel ? tap_src(xp, a.dx, a.kx) : axis_src(a.mx, xp);
        if (a.from_kernel && a.transposed && xs >= 0) xs = a.SX - 1 - xs;
        const float v0 = (ok[0] && xs >= 0) ? rows[0][xs] : 0.f;
        const float v1 = (ok[1] && xs >= 0) ? rows[1][xs] : 0.f;
        col[n1 * G::RS] = mk2(v0, v1);
      }
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) v[n1] = col[n1 * G::RS];
    }
  }
  {
    // odd-frequency transform of the real axis: sample n is turned by e^{-i*pi*n/T} first, so that bin k sits at
    // frequency k + 1/2.  A real row then has exactly T/2 independent bins (no self-paired DC / Nyquist bins: T/2
    // columns downstream instead of T/2 + 1 -- whole rounds of workgroups in the column pass), and the product of two
    // such spectra is the negacyclic convolution, which differs from the cyclic one only in the wrapped samples that
    // overlap-save discards anyway.  n = N2*n1 + tseq: one per-thread factor times compile-time constants.
    float ws, wc;
    sincospif(-(float)tseq / (float)T, &ws, &wc);
    const f2 wt = mk2(wc, ws);
    static_for<0, P>([&](auto ic) {
      constexpr int n1 = decltype(ic)::value;
      constexpr float c = (float)cospi_q(n1, P), s = (float)sinpi_q(n1, P);
      v[n1] = cmul(v[n1], cmul_const(wt, c, -s));
    });
  }
  fwd_from_regs<G>(v, lds + sq * LSEQP, tseq, true, twA, twB);
  __syncthreads();
  // unpack the two real spectra of every pair (partner of bin k: T-1-k) and store transposed: RB rows contiguous per bin
  // (a thread keeps its row r = tid % RB over all rounds and steps NT/RB bins per round: every LDS read of the thread is
  // requested before the first value is used, and the stores are buffer stores relative to this workgroup's block of bin
  // columns -- 32-bit offsets, rows past NY get an out-of-range offset instead of a branch)
  constexpr int FXC = T / 2;                       // == a.Fx
  static_assert(NT % RB == 0 && (FXC * RB) % NT == 0, "whole rounds, fixed row per thread");
  if (a.rowmajor) {
    // rows as they are: a wavefront writes 64 neighbouring bins of one row (512 bytes), both rows of a pair from the same
    // two LDS reads; rows past NY fall outside the resource (it ends with the last row of this block)
    constexpr int NIT = FXC * NSEQ / NT;
    static_assert(NIT * NT == FXC * NSEQ, "whole rounds");
    const unsigned pitch = (unsigned)(a.nxt * a.Fx);
    f2* out = a.dst + (((size_t)img * a.NC + c) * a.NY + y0) * pitch + (size_t)xt * a.Fx;
    const BufRsrc orr = make_rsrc(out, (unsigned)(((size_t)(min(RB, a.NY - y0) - 1) * pitch + FXC) * 8));
    f2 zf[NIT], zg[NIT];
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int idx = u * NT + tid, s = idx / FXC, fx = idx % FXC;
      const f2* z = lds + s * LSEQP;
      zf[u] = z[G::nat(fx)];
      zg[u] = z[G::nat(T - 1 - fx)];
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int idx = u * NT + tid, s = idx / FXC, fx = idx % FXC;
      const f2 ev = mk2(0.5f * (zf[u].x + zg[u].x), 0.5f * (zf[u].y - zg[u].y));
      const f2 ov = mk2(0.5f * (zf[u].y + zg[u].y), 0.5f * (zg[u].x - zf[u].x));
      const unsigned off = ((unsigned)(2 * s) * pitch + (unsigned)fx) * 8u;
      buf_store_f32x2(ev, orr, off, 0);
      buf_store_f32x2(ov, orr, off + pitch * 8u, 0);
    }
    return;
  }
  constexpr int NROUND = FXC * RB / NT, FSTEP = NT / RB;
  f2* out = a.dst + (((size_t)img * a.NC + c) * a.nxt + xt) * a.Fx * a.NYa + y0;
  const BufRsrc orr = make_rsrc(out, (unsigned)(((size_t)(FXC - 1) * a.NYa + min(RB, a.NYa - y0)) * 8));
  const int r = tid % RB, fx0 = tid / RB;
  const f2* z = lds + (r >> 1) * LSEQP;
  f2 zf[NROUND], zg[NROUND];
#pragma unroll
  for (int u = 0; u < NROUND; ++u) {
    const int fx = fx0 + u * FSTEP;
    zf[u] = z[G::nat(fx)];
    zg[u] = z[G::nat(T - 1 - fx)];
  }
  const bool odd = (r & 1) != 0;
  const unsigned rofs = (unsigned)r * 8u, rbad = (y0 + r < a.NY) ? 0u : 0x80000000u;
#pragma unroll
  for (int u = 0; u < NROUND; ++u) {
    const int fx = fx0 + u * FSTEP;
    const f2 ev = mk2(0.5f * (zf[u].x + zg[u].x), 0.5f * (zf[u].y - zg[u].y));
    const f2 ov = mk2(0.5f * (zf[u].y + zg[u].y), 0.5f * (zg[u].x - zf[u].x));
    buf_store_f32x2(odd ? ov : ev, orr, (rofs + (unsigned)(fx * a.NYa) * 8u) | rbad, 0);
  }
}

// ------------------------------------------------------------------------------------------ c2c_fwd
struct C2CArgs {
  const f2* src;
  f2* dst;
  const f2* twA;
  const f2* twB;
  // sequence (a, c, bn): src element n at a*sa + c*sc + bn*sb + n        (n < NLEN, zero beyond)
  // dst: transposed  a*ta + c*tc + f*tf + bn   (store_mode 0, forward)
  //      weights     final kernel-spectrum layout (store_mode 1, forward; conj + scale + i-pair interleave)
  //      inverse     src transposed a*sa + c*sc + f*sb + bn ; dst a*ta + bn*tb + c*tc + n_out
  long long sa, sc, sb, ta, tc, tf, tb;
  int NA, NC, NB, NLEN;
  int store_mode;
  // weights mode: a = o*Cig + i (o over all Cout)
  int Cig, Cog, Cig_pad, Cog_pad;
  float scale;
  // inverse mode: valid samples and decimation; noff = position of this launch's sample 0 on the axis (overlap-save
  // tiles along the middle axis run one launch per tile)
  int NV, stride, noff;
  FastDiv d_nbb, d_nc;   // unit map of the launch (filled by the dispatcher)
};

template <int P, int S, int NSEQ, int NT>
__global__ __launch_bounds__(NT) void c2c_fwd_kernel(const C2CArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  static_assert(NT == NSEQ * G::TS, "thread count");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  unsigned q;
  const int bb = (int)fdivmod(blockIdx.x, a.d_nbb, &q);
  const int c = (int)fdivmod(q, a.d_nc, &q);
  const int img = (int)q;
  const int bn0 = bb * NSEQ;
  const bool act = bn0 + sq < a.NB;

  f2 v[P];
  {
    // (buffer loads: samples past NLEN and idle sequences read as zero without a branch per load -- with branches hipcc
    // waits for every load before the next one)
    const f2* s0 = a.src + (size_t)img * a.sa + (size_t)c * a.sc + (size_t)bn0 * a.sb;
    const long long span = (long long)(min(NSEQ, a.NB - bn0) - 1) * a.sb + a.NLEN;      // complex values reachable from s0
    if (span > 0 && span * 8 < 0x7fffffffLL && a.sb >= 0) {
      const BufRsrc sr = make_rsrc(s0, (unsigned)(span * 8));
      const unsigned rowoff = act ? (unsigned)((long long)sq * a.sb * 8) : 0x80000000u;
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int n = G::N2 * n1 + tseq;
        v[n1] = buf_load_f32x2(sr, n < a.NLEN ? rowoff + (unsigned)n * 8u : 0xFFFFFFFFu, 0);
      }
    } else {
      const f2* s = s0 + (size_t)sq * a.sb;
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int n = G::N2 * n1 + tseq;
        v[n1] = (act && n < a.NLEN) ? s[n] : mk2(0.f, 0.f);
      }
    }
  }
  fwd_from_regs<G>(v, lds + sq * LSEQP, tseq, act, twA, twB);
  __syncthreads();
  if (a.store_mode == 0) {
    f2* out = a.dst + (size_t)img * a.ta + (size_t)c * a.tc + bn0;
    for (int idx = tid; idx < T * NSEQ; idx += NT) {
      const int r = idx % NSEQ, f = idx / NSEQ;
      if (bn0 + r < a.NB) out[(size_t)f * a.tf + r] = lds[r * LSEQP + G::nat(f)];
    }
  } else {
    // kernel spectrum: H = conj(W_hat) * scale, layout [g][o][i/2][col][f][i&1]; here bn indexes the
    // columns (fx or (fx,fy)) of one (o, i) image and c is unused (NC == 1)
    const int o_all = img / a.Cig, i = img % a.Cig;
    const int g = o_all / a.Cog, o = o_all % a.Cog;
    f2* base = a.dst + (((size_t)(g * a.Cog_pad + o) * (a.Cig_pad / 2) + (i >> 1)) * a.NB) * T * 2 + (i & 1);
    for (int idx = tid; idx < T * NSEQ; idx += NT) {
      const int f = idx % T, r = idx / T;
      if (bn0 + r < a.NB) {
        const f2 wv = lds[r * LSEQP + G::nat(f)];
        base[((size_t)(bn0 + r) * T + f) * 2] = mk2(wv.x * a.scale, -wv.y * a.scale);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ c2c_inv
template <int P, int S, int NSEQ, int NT>
__global__ __launch_bounds__(NT) void c2c_inv_kernel(const C2CArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  static_assert(NT == NSEQ * G::TS, "thread count");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  unsigned q;
  const int bb = (int)fdivmod(blockIdx.x, a.d_nbb, &q);
  const int c = (int)fdivmod(q, a.d_nc, &q);
  const int img = (int)q;
  const int bn0 = bb * NSEQ;
  const bool act = bn0 + sq < a.NB;
  f2 wtw[P];
  passA_twiddle_fetch<G>(wtw, tseq, twA);      // requested first: lands while the spectra are loaded
  {
    // every bin of this thread is requested before the first is stored to LDS (as a loop of load -> store the gather
    // paid one memory latency per iteration); sequences past NB read as zero through the buffer resource
    const f2* in = a.src + (size_t)img * a.sa + (size_t)c * a.sc + bn0;
    static_assert((T * NSEQ) % NT == 0, "whole rounds of the gather");
    constexpr int NITER = T * NSEQ / NT;
    const long long span = (long long)(T - 1) * a.sb + min(NSEQ, a.NB - bn0);
    if (span > 0 && span * 8 < 0x7fffffffLL && a.sb >= 0) {
      const BufRsrc sr = make_rsrc(in, (unsigned)(span * 8));
      f2 val[NITER];
#pragma unroll
      for (int u = 0; u < NITER; ++u) {
        const int idx = tid + u * NT, r = idx % NSEQ, f = idx / NSEQ;
        val[u] = buf_load_f32x2(sr, bn0 + r < a.NB ? (unsigned)(((long long)f * a.sb + r) * 8) : 0xFFFFFFFFu, 0);
      }
#pragma unroll
      for (int u = 0; u < NITER; ++u) {
        const int idx = tid + u * NT, r = idx % NSEQ, f = idx / NSEQ;
        lds[r * LSEQP + G::nat(f)] = val[u];
      }
    } else {
      for (int idx = tid; idx < T * NSEQ; idx += NT) {
        const int r = idx % NSEQ, f = idx / NSEQ;
        lds[r * LSEQP + G::nat(f)] = (bn0 + r < a.NB) ? in[(size_t)f * a.sb + r] : mk2(0.f, 0.f);
      }
    }
  }
  __syncthreads();
  f2 v[P];
  const int j = inv_to_regs_pre<G>(v, wtw, lds + sq * LSEQP, tseq, act, twB);
  if (act) {
    f2* out = a.dst + (size_t)img * a.ta + (size_t)(bn0 + sq) * a.tb + (size_t)c * a.tc;
    const int nbase = (tseq >> G::LGS) + P * P * j;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int n = nbase + P * k;
      const int ng = a.noff + n;
      const int idx = ng / a.stride;
      if (n < a.NV && idx * a.stride == ng) out[idx] = v[k];
    }
  }
}

// ------------------------------------------------------------------------------------------ rows_c2r
struct RowsC2RArgs {
  const f2* src;     // [(a*NC + c)][Fx][NYa]  (a = b*Cout + o)
  float* dst;            // (B, Cout, [Zo,] Yo, Xo)
  const float* bias;
  const f2* twA;
  const f2* twB;
  int NA, NC, NY, NYa, Fx, Cout;
  int NV, stride, Xo;    // valid stride-1 samples along x, decimation, output row length
  int nxt, Vx;           // x tiles (see RowsR2CArgs): tile xt yields the stride-1 samples [xt*Vx, xt*Vx + Vx)
  ImgMap im;             // where image `img` of this pass goes in dst (identity unless a weight-gradient plan)
  FastDiv d_nyb, d_nxt, d_nc;   // unit map of the launch (filled by the dispatcher)
  int rowmajor;          // src [(a*NC + c)][NY][nxt*Fx] (see RowsR2CArgs)
};

template <int P, int S, int NSEQ, int NT>
__global__ __launch_bounds__(NT) void rows_c2r_kernel(const RowsC2RArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  constexpr int RB = 2 * NSEQ;
  static_assert(NT == NSEQ * G::TS, "thread count");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  unsigned q;
  const int yb = (int)fdivmod(blockIdx.x, a.d_nyb, &q);
  const int xt = (int)fdivmod(q, a.d_nxt, &q);
  const int c = (int)fdivmod(q, a.d_nc, &q);
  const int img = (int)q;
  const int y0 = yb * RB;
  const int x0 = xt * a.Vx;                    // first stride-1 sample of this tile
  const int xlim = min(a.Vx, a.NV - x0);       // valid samples the tile contributes
  f2 wtw[P];
  passA_twiddle_fetch<G>(wtw, tseq, twA);      // requested first: lands while the spectra are loaded
  {
    // rows 2s (-> real part) and 2s+1 (-> imaginary part) share one complex inverse FFT (odd-frequency bins, see rows_r2c):
    // V[k] = Ya[k] + i*Yb[k],  V[T-1-k] = conj(Ya[k]) + i*conj(Yb[k]),  k < T/2
    // ALL of a thread's bins are requested before the first is consumed, through a buffer resource over this plane
    // of bin columns: rows past NY and slots past the last bin get an offset outside it and read as zero.  (Written
    // with `if (row < NY) load`, hipcc puts every load in its own branch and a full s_waitcnt in front of each
    // merge: the gather then pays one memory latency per bin.)  A row pair is one 16-byte load when whole and aligned.
    const f2* in = a.src + (((size_t)img * a.NC + c) * a.nxt + xt) * a.Fx * a.NYa + y0;
    constexpr int FXC = T / 2;                           // == a.Fx
    constexpr int TOTAL = FXC * NSEQ;
    constexpr int NITER = (TOTAL + NT - 1) / NT;
    // (workgroup-uniform choice, made once: a branch per load would bring the waits back)
    const int rows_here = a.NY - y0;
    const bool pair16 = ((a.NYa | y0) & 1) == 0 && !(rows_here < RB && (rows_here & 1));
    const unsigned plane_bytes = (unsigned)((size_t)(FXC - 1) * a.NYa * 8 + (size_t)min(RB, a.NYa - y0) * 8);
    f2 ya[NITER], yb2[NITER];
    if (a.rowmajor) {
      // rows as they are: lanes over neighbouring bins of one row; rows past NY fall outside the resource
      const unsigned pitch = (unsigned)(a.nxt * a.Fx);
      const f2* inr = a.src + (((size_t)img * a.NC + c) * a.NY + y0) * pitch + (size_t)xt * a.Fx;
      const BufRsrc srr = make_rsrc(inr, (unsigned)(((size_t)(min(RB, a.NY - y0) - 1) * pitch + FXC) * 8));
#pragma unroll
      for (int u = 0; u < NITER; ++u) {
        const int idx = u * NT + tid;
        const int s = idx / FXC, fx = idx % FXC;
        const unsigned off = idx < TOTAL ? ((unsigned)(2 * s) * pitch + (unsigned)fx) * 8u : 0x80000000u;
        ya[u] = buf_load_f32x2(srr, off, 0);
        yb2[u] = buf_load_f32x2(srr, off == 0x80000000u ? off : off + pitch * 8u, 0);
      }
#pragma unroll
      for (int u = 0; u < NITER; ++u) {
        const int idx = u * NT + tid;
        if (idx < TOTAL) {
          const int s = idx / FXC, fx = idx % FXC;
          f2* z = lds + s * LSEQP;
          z[G::nat(fx)] = mk2(ya[u].x - yb2[u].y, ya[u].y + yb2[u].x);
          z[G::nat(T - 1 - fx)] = mk2(ya[u].x + yb2[u].y, yb2[u].x - ya[u].y);
        }
      }
    } else {
    const BufRsrc sr = make_rsrc(in, plane_bytes);
    if (pair16) {
#pragma unroll
      for (int u = 0; u < NITER; ++u) {
        const int idx = u * NT + tid;
        const int s = idx % NSEQ, fx = idx / NSEQ;
        const bool both = idx < TOTAL && y0 + 2 * s + 1 < a.NY;
        const f4 q = buf_load_f32x4(sr, both ? (unsigned)(((size_t)fx * a.NYa + 2 * s) * 8) : 0xFFFFFFFFu, 0);
        ya[u] = q.xy; yb2[u] = q.zw;
      }
    } else {
#pragma unroll
      for (int u = 0; u < NITER; ++u) {
        const int idx = u * NT + tid;
        const int s = idx % NSEQ, fx = idx / NSEQ;
        const bool both = idx < TOTAL && y0 + 2 * s + 1 < a.NY;
        const bool one = idx < TOTAL && y0 + 2 * s < a.NY;
        const unsigned off = (unsigned)(((size_t)fx * a.NYa + 2 * s) * 8);
        ya[u] = buf_load_f32x2(sr, one ? off : 0xFFFFFFFFu, 0);
        yb2[u] = buf_load_f32x2(sr, both ? off + 8u : 0xFFFFFFFFu, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < NITER; ++u) {
      const int idx = u * NT + tid;
      if (idx < TOTAL) {
        const int s = idx % NSEQ, fx = idx / NSEQ;
        f2* z = lds + s * LSEQP;
        z[G::nat(fx)] = mk2(ya[u].x - yb2[u].y, ya[u].y + yb2[u].x);
        z[G::nat(T - 1 - fx)] = mk2(ya[u].x + yb2[u].y, yb2[u].x - ya[u].y);
      }
    }
    }
  }
  __syncthreads();
  f2 v[P];
  const int j = inv_to_regs_pre<G>(v, wtw, lds + sq * LSEQP, tseq, true, twB);
  {
    // undo the half-bin shift: sample n = nbase + P*k is turned back by e^{+i*pi*n/T}
    const int nb0 = (tseq >> G::LGS) + P * P * j;
    float ws, wc;
    sincospif((float)nb0 / (float)T, &ws, &wc);
    const f2 wb = mk2(wc, ws);
    static_for<0, P>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr float c = (float)cospi_q(k, G::N2), s = (float)sinpi_q(k, G::N2);
      v[k] = cmul(v[k], cmul_const(wb, c, s));
    });
  }
  const int ya_row = y0 + 2 * sq;
  // (declared arrived before the guarded stores: met first inside them, hipcc puts a full s_waitcnt vmcnt(0) in front of
  // every store, and vmcnt counts stores too -- each store would wait for the one before it)
  float b = a.bias ? a.bias[img % a.Cout] : 0.f;
  asm volatile("" : "+v"(b));
  // buffer stores relative to this workgroup's RB output rows (32-bit offsets; samples past the valid window, decimated
  // away or in rows past NY get an out-of-range offset instead of a branch)
  float* orow = a.dst + ((map_img(a.im, img) * a.NC + c) * a.NY + y0) * a.Xo;
  const BufRsrc orr = make_rsrc(orow, (unsigned)((size_t)min(RB, a.NY - y0) * a.Xo * 4));
  const bool has0 = ya_row < a.NY, has1 = ya_row + 1 < a.NY;
  // (offsets are sums of small row and column parts; bit 31 marks a store that must not happen)
  const unsigned r0 = (unsigned)(2 * sq * a.Xo) * 4u, r1 = r0 + (unsigned)a.Xo * 4u;
  const unsigned bad0 = has0 ? 0u : 0x80000000u, bad1 = has1 ? 0u : 0x80000000u;
  const int nbase = (tseq >> G::LGS) + P * P * j;
  if (a.stride == 1) {
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int n = nbase + P * k;
      const unsigned xo = (unsigned)(x0 + n) * 4u, badx = n < xlim ? 0u : 0x80000000u;
      buf_store_f32(v[k].x + b, orr, (r0 + xo) | bad0 | badx, 0);
      buf_store_f32(v[k].y + b, orr, (r1 + xo) | bad1 | badx, 0);
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < P; ++k) {
    const int n = nbase + P * k;
    const int t = x0 + n;
    const int idx = t / a.stride;
    const unsigned xo = (unsigned)idx * 4u, badx = (n < xlim && idx * a.stride == t) ? 0u : 0x80000000u;
    buf_store_f32(v[k].x + b, orr, (r0 + xo) | bad0 | badx, 0);
    buf_store_f32(v[k].y + b, orr, (r1 + xo) | bad1 | badx, 0);
  }
}

// ------------------------------------------------------------------------------------------ fusedc
// Outermost-axis pass: CIB complex sequences (one per input channel of the group) per workgroup.
struct FusedCArgs {
  const f2* src;     // [(b*Cin + ci)][col][NLEN]   contiguous along the fused axis
  const f4* wspec;   // [g][Cog_pad][Cig_pad/2][ncol][T] f4 = {H(o,2ip), H(o,2ip+1)}
  f2* dst;           // [(b*Cout + co)][col][NVo]    valid (decimated) samples
  const f2* twA;
  const f2* twB;
  int B, Cin, Cout, G, Cig, Cog, Cig_pad, Cog_pad, cob, n_ochunks;
  int ncol, NLEN;        // columns per image, valid input length (zero beyond)
  int wfx, wty, wrep, wncol;   // kernel-spectrum column of signal column col = ((xt*wfx + fx)*wrep + yt)*wty + fy:
                         // ((col / (wty*wrep)) % wfx) * wty + col % wty -- the x tiles (xt) and middle-axis tiles (yt) of
                         // the signal share the wfx * wty columns of the kernel; wncol = wfx * wty columns per (o, ip)
  int Kd, V, ntiles, Lfull, NVo, stride;
  int accumulate;
  unsigned long long* stamps;   // profiling hook: 8 timestamps per workgroup, else null
  FastDiv d_nbb, d_ncb, d_g, d_noc;   // unit map of the launch (filled by the dispatcher)
};

// NB batch items of one (column, tile, group, out-chunk) share a workgroup and with it every
// kernel-spectrum load of the mix (the spectrum of a column is cob x Cig x T complex, several times
// the signal data of one batch item; streaming it per batch item bound this kernel).
template <int P, int S, int CIB, int NB, int NT>
__global__ __launch_bounds__(NT) void fusedc_kernel(const FusedCArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  constexpr int NSEQ = NB * CIB;
  static_assert(NT == NSEQ * G::TS, "one sequence per channel of the chunk and batch slot");
  constexpr int R = NT > T ? NT / T : 1;                // mix threads per bin (they split the output channels)
  constexpr int BPT = NT > T ? 1 : T / NT;              // bins per mix thread
  static_assert(R * T == NT || BPT * NT == T, "threads and bins divide evenly");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const int tid = threadIdx.x, sq = tid / G::TS, tseq = tid % G::TS;
  const int nbi = sq / CIB, ch = sq % CIB;              // batch slot, channel slot

  // XCD-aware unit map.  Workgroups are dealt round-robin over the 8 XCDs (id % 8 shares an L2), and
  // a column's kernel spectrum (Cog x Cig x T complex, e.g. 256 KB) is read by every batch block of
  // that column: so the column is tied to id % 8 and the batch block runs fastest behind it --
  // all workgroups of a column hit the same L2 back to back and the spectrum leaves the
  // Infinity Cache once per XCD instead of once per workgroup.
  //   id = ((((tile*n_ochunks + oc)*G + g)*ncb + colblk)*nbb + bb)*8 + xcd ,  col = colblk*8 + xcd
  const int xcd = blockIdx.x & 7;
  unsigned q;
  const int b0 = (int)fdivmod(blockIdx.x >> 3, a.d_nbb, &q) * NB;
  const int col = (int)fdivmod(q, a.d_ncb, &q) * 8 + xcd;
  const int g = (int)fdivmod(q, a.d_g, &q);
  const int oc = (int)fdivmod(q, a.d_noc, &q);
  const int tile = (int)q;
  if (col >= a.ncol) return;                 // padding of the last column block (uniform per workgroup)
  const int nbc = min(NB, a.B - b0);         // batch items of this workgroup
  auto stampc = [&](int slot) {
    if (a.stamps != nullptr && tid == 0) a.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  stampc(0);

  f2* zin = lds;
  f2* vout = a.accumulate ? lds + NSEQ * LSEQP : lds;
  const int n_ichunks = a.Cig_pad / CIB;
  const int t0 = tile * a.V;
  const size_t wcol = (size_t)a.wncol * T;                         // f4 per (o, ip)
  const int wc = ((col / (a.wty * a.wrep)) % a.wfx) * a.wty + col % a.wty;
  const f4* wgrp = a.wspec + (size_t)g * a.Cog_pad * (a.Cig_pad / 2) * wcol + (size_t)wc * T;

  for (int ic = 0; ic < n_ichunks; ++ic) {
    {
      const int ci = ic * CIB + ch;
      const bool has = ci < a.Cig && nbi < nbc;
      f2 v[P];
      // rows of this workgroup: batch items b0 .. b0+nbc-1, the CIB channels of this chunk, this column -- buffer loads
      // relative to the first of them (32-bit offsets; idle sequences and samples past NLEN get an offset outside the
      // resource and read as zero: no branch and no 64-bit address per load)
      const long long rowlen = (long long)a.ncol * a.NLEN;
      const long long ispan = ((long long)(nbc - 1) * a.Cin + CIB - 1) * rowlen + a.NLEN;
      if (ispan * 8 < 0x7fffffffLL) {
        const f2* s0 = a.src + (((size_t)b0 * a.Cin + (size_t)g * a.Cig + (size_t)ic * CIB) * a.ncol + col) * a.NLEN;
        const BufRsrc sr = make_rsrc(s0, (unsigned)(ispan * 8));
        const unsigned ro = has ? (unsigned)((((long long)nbi * a.Cin + ch) * rowlen + t0) * 8) + (unsigned)tseq * 8u : 0x80000000u;
        if (t0 + T <= a.NLEN) {
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) v[n1] = buf_load_f32x2(sr, ro, G::N2 * n1 * 8);
        } else {
          const int lim = a.NLEN - t0 - tseq;        // sample n1 exists iff N2*n1 < lim
#pragma unroll
          for (int n1 = 0; n1 < P; ++n1) v[n1] = buf_load_f32x2(sr, G::N2 * n1 < lim ? ro : 0x80000000u, G::N2 * n1 * 8);
        }
      } else {
        const f2* s = a.src + (((size_t)(b0 + (has ? nbi : 0)) * a.Cin + (size_t)g * a.Cig + (has ? ci : 0)) * a.ncol + col) * a.NLEN + t0;
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          const int n = G::N2 * n1 + tseq;
          v[n1] = (has && t0 + n < a.NLEN) ? s[n] : mk2(0.f, 0.f);
        }
      }
      if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stampc(1); }
      fwd_from_regs<G>(v, zin + sq * LSEQP, tseq, true, twA, twB);
    }
    // mix: every bin is independent in complex mode.  step = (bin of this thread, output channel o);
    // two register sets alternate so the CIB/2 spectrum loads of the next step are in flight while
    // this one is contracted for all NB batch items (same scheme as the 1-D kernel).
    {
      constexpr int NH = CIB / 2;
      const int h = R > 1 ? tid / T : 0;                  // which share of the output channels
      const int f0 = R > 1 ? tid % T : tid;
      const int no = R > 1 ? (a.cob - h + R - 1) / R : a.cob;      // outputs o = h + R*k, k < no
      const int nsteps = BPT * max(no, 0);
      const size_t orow = (size_t)(a.Cig_pad / 2) * wcol;  // float4 between output channels
      const f4* wbase = wgrp + ((size_t)(oc * a.cob + h) * (a.Cig_pad / 2) + ic * NH) * wcol;
      f4 wA[NH], wB[NH];
      f2 x[NB][CIB];
      auto issue = [&](int m, int k, f4 (&dst)[NH]) {
        const int f = f0 + m * NT;
        const f4* w = wbase + (size_t)(R * k) * orow + f;
#pragma unroll
        for (int p = 0; p < NH; ++p) dst[p] = w[(size_t)p * wcol];
      };
      auto load_x = [&](int m) {
        const unsigned base = lds_off(zin + G::nat(f0 + m * NT));
        static_for<0, NB>([&](auto bc) {
          constexpr int b = decltype(bc)::value;
          const unsigned bb = base + b * CIB * LSEQP * 8;
          static_for<0, CIB>([&](auto icn) {
            constexpr int i = decltype(icn)::value;
            x[b][i] = lds_rd_far<i * LSEQP * 8>(bb);
          });
        });
#pragma unroll
        for (int b = 0; b < NB; ++b) lds_arrive(x[b]);
      };
      auto step = [&](int m, int k, const f4 (&wc)[NH]) {
        const int o = h + R * k;
        const int f = f0 + m * NT;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f2 y = mk2(0.f, 0.f);
#pragma unroll
          for (int p = 0; p < NH; ++p) {
            cmac(y, x[b][2 * p], wc[p].xy);
            cmac(y, x[b][2 * p + 1], wc[p].zw);
          }
          if (b < nbc) {
            f2* pv = vout + (b * CIB + o) * LSEQP + G::nat(f);
            if (ic != 0) y += *pv;
            *pv = y;
          }
        }
      };
      if (nsteps > 0) issue(0, 0, wA);      // does not depend on the transforms: travels across the barrier
      stampc(2);
      __syncthreads();
      stampc(3);
      if (R > 1) {
        // several threads (in different waves) read a bin that others overwrite in place
        load_x(0);
        __syncthreads();
      }
      int m = 0, k = 0;
#pragma unroll 1
      for (int s2 = 0; s2 < nsteps; s2 += 2) {
        int k1 = k + 1, m1 = m;
        if (k1 == no) { k1 = 0; ++m1; }
        int k2 = k1 + 1, m2 = m1;
        if (k2 == no) { k2 = 0; ++m2; }
        if (s2 + 1 < nsteps) issue(m1, k1, wB);
        if (R == 1 && k == 0) load_x(m);
        step(m, k, wA);
        if (s2 + 2 < nsteps) issue(m2, k2, wA);
        if (s2 + 1 < nsteps) {
          if (R == 1 && k1 == 0) load_x(m1);
          step(m1, k1, wB);
        }
        m = m2; k = k2;
      }
    }
    stampc(4);
    __syncthreads();
    stampc(5);
  }
  // inverse + store of the valid, decimated samples
  f2 v[P];
  const bool act = ch < a.cob && nbi < nbc;
  const int j = inv_to_regs<G>(v, vout + sq * LSEQP, tseq, act, twA, twB);
  stampc(6);
  const int co = oc * a.cob + ch;
  const bool live = act && co < a.Cog;
  const int limit = min(a.V, a.Lfull - t0);
  const int nbase = (tseq >> G::LGS) + P * P * j;
  // rows of this workgroup: batch items b0 .. b0+nbc-1, all output channels of the group, this column
  f2* out0 = a.dst + (((size_t)b0 * a.Cout + (size_t)g * a.Cog) * a.ncol + col) * a.NVo;
  const long long ospan = ((long long)(nbc - 1) * a.Cout + a.Cog - 1) * a.ncol * a.NVo + a.NVo;
  if (ospan * 8 < 0x7fffffffLL) {
    // buffer stores relative to the workgroup's first row (idle sequences, samples past the valid window or decimated
    // away: bit 31 of the offset instead of a branch)
    const BufRsrc orr = make_rsrc(out0, (unsigned)(ospan * 8));
    const unsigned ro = live ? (unsigned)((((long long)nbi * a.Cout + co) * a.ncol * a.NVo) * 8) : 0u;
    const unsigned rbad = live ? 0u : 0x80000000u;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int n = nbase + P * k;
      const int t = t0 + n;
      const int idx = t / a.stride;
      buf_store_f32x2(v[k], orr, (ro + (unsigned)idx * 8u) | rbad | ((n < limit && idx * a.stride == t) ? 0u : 0x80000000u), 0);
    }
  } else if (live) {
    f2* out = a.dst + (((size_t)(b0 + nbi) * a.Cout + (size_t)g * a.Cog + co) * a.ncol + col) * a.NVo;
#pragma unroll
    for (int k = 0; k < P; ++k) {
      const int n = nbase + P * k;
      const int t = t0 + n;
      const int idx = t / a.stride;
      if (n < limit && idx * a.stride == t) out[idx] = v[k];
    }
  }
  if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stampc(7); }
}

}  // namespace fc
