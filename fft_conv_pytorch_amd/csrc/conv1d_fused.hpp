// conv1d_fused.hpp -- the fused 1-D FFT-convolution kernel (overlap-save tiles).
//
// One workgroup = one unit (batch item b, channel group g, out-channel chunk oc,
// tile t).  Everything between the input samples and the output samples stays on
// chip (replaces functional.py:60-87 of the reference, rows a3..a10 of SURVEY 8a):
//
//   load    T input samples of CIB channels, padding mode folded into the index
//           map, two real channels packed per complex sequence z = x0 + i*x1
//   forward two-pass register/LDS FFT (fft_engine.hpp)
//   mix     per bin pair (f, T-f): unpack the two real spectra of every pair,
//           contract input channels against the pre-transformed kernel
//           H[o][i][f] = conj(W_oi[f])/(2T)   (L2-resident, float4 = two i's),
//           re-pack two output channels per complex sequence, in place in LDS
//   inverse two-pass FFT, last pass writes the V = T-Kd+1 valid samples straight
//           to HBM with stride decimation and bias.
//
// Channel counts are runtime; CIB (input channels per chunk, even) is a template
// parameter so the contraction keeps its inputs in registers.  Chunks beyond the
// first accumulate into a second LDS region.
#pragma once
#include "fft_engine.hpp"

namespace fc {

enum PadMode : int { PAD_CONSTANT = 0, PAD_REFLECT = 1, PAD_REPLICATE = 2, PAD_CIRCULAR = 3 };

struct Conv1dArgs {
  const float* x;        // (B, Cin, L)
  const f4* wspec;       // [G][Cog_pad][Cig_pad/2][T/2] float4 = {H(o,2ip)[f], H(o,2ip+1)[f]}
  const float* bias;     // (Cout) or null
  float* y;              // (B, Cout, Lout)
  const f2* twA;         // [P][N2]
  const f2* twB;         // [S][P] (unused: the lane-split twiddles are compile-time constants)
  int B, Cin, Cout, G, Cig, Cog;
  int Cig_pad, Cog_pad;  // padded to CIB / COB multiples (spectrum layout)
  int cob;               // out channels per chunk (even)
  int n_ochunks;         // Cog_pad / cob
  int L, pad, pad_mode;  // source row length, left padding (may be negative for a transposed plan)
  int up;                // transposed plan: the source is spread over a grid of this step (zeros between)
  int ph;                // batch-sharing kernel: dilation run as `ph` interleaved phases (virtual batch B*ph), else 1
  int ph2;               // batch-sharing kernel: the phases run in pairs (conv1d_pers.hpp PH2; ph even, slots = batch items)
  int diag;              // batch-sharing kernel: depthwise plan (8-channel blocks, per-channel mix, [pair][T/2] spectrum)
  int slot_tiles;        // batch-sharing kernel: the slots of a work item are consecutive TILES of one (virtual) batch
                         // item instead of consecutive batch items of one tile
  int Kd, V, ntiles, Lfull, Lout, stride;
  int accumulate;        // 1 when this launch covers several input chunks (separate output region in LDS)
  int ic_begin, ic_end;  // input chunks of this launch (all of them unless the plan launches chunk by chunk)
  int segmented;         // the plan runs the kernel in segments of taps (pos_shift / add_out vary per launch)
  int pos_shift;         // kernel segment: its first tap sits this many samples into the (dilated) kernel
  int add_out;           // 1: y += result (later chunk launches of such a plan; bias went with the first)
  unsigned long long* stamps;  // optional profiling hook: 16 timestamps per workgroup (null = off)
};

// Branch-free padded load.  Outside [0, L) the index is remapped as a*pos + b with
// wave-uniform (a, b) per side, chosen from the padding mode (reflect: -pos /
// 2(L-1)-pos, replicate: 0 / L-1, circular: pos+L / pos-L); constant mode and
// positions beyond the padded extent read as zero.
struct PadMap {
  int lo_a, lo_b, hi_a, hi_b, live;   // live = 0 for constant mode
};
__device__ __forceinline__ PadMap make_padmap(int mode, int L) {
  PadMap m;
  m.live = (mode != PAD_CONSTANT);
  m.lo_a = (mode == PAD_REFLECT) ? -1 : (mode == PAD_CIRCULAR ? 1 : 0);
  m.hi_a = m.lo_a;
  m.lo_b = (mode == PAD_CIRCULAR) ? L : 0;
  m.hi_b = (mode == PAD_REFLECT) ? 2 * (L - 1) : (mode == PAD_REPLICATE ? L - 1 : (mode == PAD_CIRCULAR ? -L : 0));
  return m;
}
// Zero-spread source of a transposed convolution: grid position pos holds row[pos/up] when pos is a
// non-negative multiple of up inside the row, zero otherwise (functional.py:126-139).
__device__ __forceinline__ float load_spread(const float* __restrict__ row, int pos, int L, int up, bool chan_ok) {
  const int q = pos / up;
  const bool ok = chan_ok && pos >= 0 && q * up == pos && q < L;
  const float v = row[ok ? q : 0];
  return ok ? v : 0.0f;
}
__device__ __forceinline__ float load_padded(const float* __restrict__ row, int pos, int L, int pad, const PadMap& m,
                                             bool chan_ok) {
  const bool inside = (unsigned)pos < (unsigned)L;
  const int qm = (pos < 0) ? m.lo_a * pos + m.lo_b : m.hi_a * pos + m.hi_b;
  int q = inside ? pos : qm;
  q = min(max(q, 0), L - 1);
  const bool ok = chan_ok && (inside || (m.live && pos >= -pad && pos < L + pad));
  const float v = row[ok ? q : 0];
  return ok ? v : 0.0f;
}

// Byte offset of padded position pos inside a row starting at row_off, or an out-of-range offset
// (buffer loads then return zero) -- the mask is consumed before the load, nothing stays live.
__device__ __forceinline__ unsigned padded_offset(unsigned row_off, int pos, int L, int pad, const PadMap& m, bool chan_ok) {
  const bool inside = (unsigned)pos < (unsigned)L;
  const int qm = (pos < 0) ? m.lo_a * pos + m.lo_b : m.hi_a * pos + m.hi_b;
  int q = inside ? pos : qm;
  q = min(max(q, 0), L - 1);
  const bool ok = chan_ok && (inside || (m.live && pos >= -pad && pos < L + pad));
  return ok ? row_off + (unsigned)q * 4u : 0xFFFFFFFFu;
}

// The same for the zero-spread source of a transposed plan (see load_spread).
__device__ __forceinline__ unsigned spread_offset(unsigned row_off, int pos, int L, int up, bool chan_ok) {
  const int q = pos / up;
  const bool ok = chan_ok && pos >= 0 && q * up == pos && q < L;
  return ok ? row_off + (unsigned)q * 4u : 0xFFFFFFFFu;
}

// Profiling hook: lane 0 of each workgroup records the 100 MHz wall clock at phase boundaries.
__device__ __forceinline__ void stamp(unsigned long long* buf, int slot) {
  if (buf != nullptr && threadIdx.x == 0) buf[(size_t)blockIdx.x * 16 + slot] = __builtin_amdgcn_s_memrealtime();
}

template <int P, int S, int CIB, int NT>
__global__ __launch_bounds__(NT, 2) void conv1d_fused_kernel(const Conv1dArgs a) {
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int NPI = CIB / 2;
  constexpr int SEQ_PER_IT = NT / G::TS;  // sequences processed concurrently
  static_assert(NPI <= SEQ_PER_IT, "one sequence per thread per pass");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];

  // ---- unit decode: id = ((b*ntiles + tile)*n_ochunks + oc)*G + g  (g fastest: a
  // group's spectrum stays on one XCD's L2 when G is a multiple of 8)
  int id = blockIdx.x;
  const int g = id % a.G; id /= a.G;
  const int oc = id % a.n_ochunks; id /= a.n_ochunks;
  const int tile = id % a.ntiles;
  const int b = id / a.ntiles;

  const int tid = threadIdx.x;
  const int seq0 = tid / G::TS;
  const int tseq = tid % G::TS;
  const int npo = a.cob / 2;
  f2* zin = lds;
  f2* vout = a.accumulate ? lds + NPI * G::LSEQ : lds;

  const int tile_pos = tile * a.V - a.pad + a.pos_shift;     // signal coordinate of tile sample 0
  const bool interior = a.up == 1 && (tile_pos >= 0) && (tile_pos + T <= a.L);
  const PadMap pm = make_padmap(a.pad_mode, a.L);
  // buffer descriptors from uniform values only (no waterfall loops)
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(P * G::N2 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const BufRsrc xg = make_rsrc(a.x + ((size_t)b * a.Cin + (size_t)g * a.Cig) * a.L, (unsigned)a.Cig * (unsigned)a.L * 4u);
  const size_t wgroup = (size_t)a.Cog_pad * (a.Cig_pad / 2) * (T / 2);   // float4 per group
  const BufRsrc wg = make_rsrc(a.wspec + (size_t)g * wgroup, (unsigned)(wgroup * 16));

  stamp(a.stamps, 0);
  for (int ic = a.ic_begin; ic < a.ic_end; ++ic) {
    // ------------------------------------------------ forward pass A (global -> regs -> LDS)
    if (seq0 < NPI) {
      const int sq = seq0;
      const int ci0 = ic * CIB + 2 * sq;         // channel within the group
      f2 v[P];
      const bool has0 = ci0 < a.Cig, has1 = ci0 + 1 < a.Cig;
      // (rows of phantom channels -- ci >= Cig in the last chunk, past the tensor for the last group -- are never
      // dereferenced: every load below goes through the group's buffer resource with an out-of-range offset for them)
      if (interior && has1) {
        const unsigned v0 = ((unsigned)ci0 * (unsigned)a.L + (unsigned)(tile_pos + tseq)) * 4u;
        const unsigned v1 = v0 + (unsigned)a.L * 4u;
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) {
          v[n1].x = buf_load_f32(xg, v0, G::N2 * n1 * 4);
          v[n1].y = buf_load_f32(xg, v1, G::N2 * n1 * 4);
        }
      } else {
        // border tile / odd channel count / zero-spread source of a transposed plan: a rolled loop
        // stages this thread's own column in LDS (no long-lived masks, no register-array indexing),
        // then the column is read back -- same thread, same addresses, so no barrier.  (An unrolled
        // masked-offset variant is used by the batch-sharing kernel; here it made hipcc spill.)
        // Loads in batches of 8 positions through the group's buffer resource (a masked position or a phantom channel
        // is an offset outside it: nothing is dereferenced, nothing branches).  One sample per iteration with
        // `ok ? row[q] : 0` cost one memory latency per sample -- 32 in a row on every tile of a plan with an odd
        // channel count or a spread source.
        f2* col = zin + sq * G::LSEQ + tseq;
        const unsigned ro0 = (unsigned)ci0 * (unsigned)a.L * 4u, ro1 = ro0 + (unsigned)a.L * 4u;
        constexpr int CH = P < 8 ? P : 8;
#pragma unroll 1
        for (int c0 = 0; c0 < P; c0 += CH) {
          f2 t[CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int pos = tile_pos + G::N2 * (c0 + u) + tseq;
            const unsigned o0 = (a.up == 1) ? padded_offset(ro0, pos, a.L, a.pad, pm, has0) : spread_offset(ro0, pos, a.L, a.up, has0);
            const unsigned o1 = (a.up == 1) ? padded_offset(ro1, pos, a.L, a.pad, pm, has1) : spread_offset(ro1, pos, a.L, a.up, has1);
            t[u].x = buf_load_f32(xg, o0, 0);
            t[u].y = buf_load_f32(xg, o1, 0);
          }
#pragma unroll
          for (int u = 0; u < CH; ++u) col[(c0 + u) * G::RS] = t[u];
        }
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) v[n1] = col[n1 * G::RS];
      }
      if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(a.stamps, 1); }
      passA_fft_twiddle_store<G, -1>(v, zin + sq * G::LSEQ, tseq, twA);
    }
    stamp(a.stamps, 2);
    seq_sync<G>();
    stamp(a.stamps, 3);
    // ------------------------------------------------ forward pass B (LDS -> regs -> LDS natural)
    {
      // every row is read before anyone writes: the two layouts alias
      f2 v[P];
      if (seq0 < NPI) passB_load<G>(v, zin + seq0 * G::LSEQ, tseq);
      seq_sync<G>();
      if (seq0 < NPI) {
        const int j = passB_compute<G, -1>(v, tseq, twB);
        const int k1 = tseq >> G::LGS;
        f2* dst = zin + seq0 * G::LSEQ + G::nat(k1 + P * P * j);   // nat() pad is constant per j block
#pragma unroll
        for (int k = 0; k < P; ++k) dst[P * k] = v[k];
      }
    }
    stamp(a.stamps, 4);
    __syncthreads();
    stamp(a.stamps, 5);
    // ------------------------------------------------ mix: channel contraction per bin pair
    {
      // byte offsets inside the group's spectrum: uniform part in SGPRs, lane part = f*16
      const unsigned ostride = (unsigned)(a.Cig_pad / 2) * (T / 2) * 16u;   // bytes between output channels
      const unsigned wbase = (unsigned)(oc * a.cob) * ostride + (unsigned)(ic * NPI) * (T / 2) * 16u;
      // Self-paired bins 0 and T/2 (both spectra real there; wspec[.][0] = {Re H[0], Re H[T/2]}):
      // lane (output o, bin) of wave 0 owns one real output; its loads are issued here and consumed
      // after the main loop so their latency is hidden.
      const int sb_o = tid >> 1, sb_f = (tid & 1) ? T / 2 : 0;
      const bool sb_act = tid < 2 * a.cob;
      f4 sbw[NPI];
      f2 sbz[NPI];
      if (sb_act) {
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          sbw[p] = buf_load_f32x4(wg, (unsigned)sb_o * ostride, wbase + p * (T / 2) * 16);
          sbz[p] = zin[p * G::LSEQ + G::nat(sb_f)];
        }
      }
      // Main bins: step = (bin pair of this thread, output pair q).  Two register sets alternate so
      // the 2*NPI spectrum loads of the next step are in flight while this one is contracted.
      constexpr int BPT = (T / 2 + NT - 1) / NT;          // bin pairs per thread
      const int nsteps = BPT * npo;
      f4 wA[2 * NPI], wB[2 * NPI];
      f2 xe[NPI], xo[NPI];                                  // 2*X of the even / odd channel of every pair
      auto issue = [&](int m, int q, f4 (&dst)[2 * NPI]) {
        const unsigned vo = (unsigned)(tid + m * NT) * 16u;     // beyond T/2: out of the descriptor -> zeros
        const unsigned sa = wbase + (unsigned)(2 * q) * ostride, sb = sa + ostride;
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          dst[2 * p] = buf_load_f32x4(wg, vo, sa + p * (T / 2) * 16);
          dst[2 * p + 1] = buf_load_f32x4(wg, vo, sb + p * (T / 2) * 16);
        }
      };
      auto step = [&](int m, int q, const f4 (&wc)[2 * NPI]) {
        const int f = tid + m * NT;
        const bool live = f < T / 2 && f != 0;             // f = 0 belongs to the self-paired lanes
        const int fc = live ? f : 1;
        const int fm = T - fc;
        if (q == 0) {
#pragma unroll
          for (int p = 0; p < NPI; ++p) {
            const f2 zf = zin[p * G::LSEQ + G::nat(fc)];
            const f2 zg = zin[p * G::LSEQ + G::nat(fm)];
            xe[p] = add_conj(zf, zg);          // 2*X_even[f] = Z[f] + conj(Z[T-f])
            xo[p] = sub_conj_divi(zf, zg);     // 2*X_odd[f]  = (Z[f] - conj(Z[T-f])) / i
          }
        }
        f2 ya = mk2(0.f, 0.f), yb = mk2(0.f, 0.f);
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          const f4 ha = wc[2 * p], hb = wc[2 * p + 1];
          cmac(ya, xe[p], ha.xy); cmac(ya, xo[p], ha.zw);
          cmac(yb, xe[p], hb.xy); cmac(yb, xo[p], hb.zw);
        }
        if (live) {
          // V[f] = Ya + i*Yb ; V[T-f] = conj(Ya) + i*conj(Yb)
          f2 vf = add_pi(ya, yb);
          f2 vg = conj_add_iconj(ya, yb);
          f2* pf = vout + q * G::LSEQ + G::nat(fc);
          f2* pg = vout + q * G::LSEQ + G::nat(fm);
          if (ic != a.ic_begin) { vf += *pf; vg += *pg; }
          *pf = vf; *pg = vg;
        }
      };
      {
        int m = 0, q = 0;                                   // position of the step being contracted
        issue(0, 0, wA);
#pragma unroll 1
        for (int s2 = 0; s2 < nsteps; s2 += 2) {
          int q1 = q + 1, m1 = m;
          if (q1 == npo) { q1 = 0; ++m1; }
          int q2 = q1 + 1, m2 = m1;
          if (q2 == npo) { q2 = 0; ++m2; }
          if (s2 + 1 < nsteps) issue(m1, q1, wB);
          step(m, q, wA);
          if (s2 + 2 < nsteps) issue(m2, q2, wA);
          if (s2 + 1 < nsteps) step(m1, q1, wB);
          m = m2; q = q2;
        }
      }
      if (sb_act) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          acc = fmaf(2.f * sbz[p].x, (tid & 1) ? sbw[p].y : sbw[p].x, acc);
          acc = fmaf(2.f * sbz[p].y, (tid & 1) ? sbw[p].w : sbw[p].z, acc);
        }
        float* dstf = reinterpret_cast<float*>(vout + (sb_o >> 1) * G::LSEQ + G::nat(sb_f)) + (sb_o & 1);
        if (ic != a.ic_begin) acc += *dstf;
        *dstf = acc;
      }
    }
    stamp(a.stamps, 6);
    __syncthreads();
    stamp(a.stamps, 7);
  }

  // -------------------------------------------------- inverse pass A' (LDS natural -> regs -> LDS rows)
  {
    f2 v[P];
    const bool act = seq0 < npo;
    if (act) {
      const f2* src = vout + seq0 * G::LSEQ;
#pragma unroll
      for (int i1 = 0; i1 < P; ++i1) v[i1] = src[G::nat(G::N2 * i1 + tseq)];
    }
    seq_sync<G>();
    if (act) {
      passA_fft_twiddle_store<G, +1>(v, vout + seq0 * G::LSEQ, tseq, twA);
    }
  }
  stamp(a.stamps, 8);
  seq_sync<G>();
  stamp(a.stamps, 9);
  // -------------------------------------------------- inverse pass B' (LDS -> regs -> HBM)
  if (seq0 < npo) {
    const int sq = seq0;
    f2 v[P];
    passB_load<G>(v, vout + sq * G::LSEQ, tseq);
    const int j = passB_compute<G, +1>(v, tseq, twB);
    const int o1 = tseq >> G::LGS;
    const int co0 = oc * a.cob + 2 * sq;           // out channel within the group
    const bool has0 = co0 < a.Cog, has1 = co0 + 1 < a.Cog;
    const int cg0 = g * a.Cog + co0;
    // (declared arrived before the guarded stores: met first inside them, hipcc puts a full s_waitcnt vmcnt(0) in front
    // of every store, and vmcnt counts stores too -- each store then waits for the one before it)
    float bias0 = (a.bias && has0) ? a.bias[cg0] : 0.f;
    float bias1 = (a.bias && has1) ? a.bias[cg0 + 1] : 0.f;
    asm volatile("" : "+v"(bias0), "+v"(bias1));
    const int t0 = tile * a.V;
    const int limit = min(a.V, a.Lfull - t0);      // valid samples of this tile
    const int nbase = o1 + P * P * j;
    if (a.add_out) {
      // chunk-by-chunk plan, chunks after the first: accumulate into the output already in HBM
      float* y0 = a.y + ((size_t)b * a.Cout + cg0) * a.Lout;
      float* y1 = y0 + a.Lout;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const int n = nbase + P * k;
        const int t = t0 + n;
        const int idx = t / a.stride;
        if (n < limit && idx * a.stride == t) {
          if (has0) y0[idx] += v[k].x;
          if (has1) y1[idx] += v[k].y;
        }
      }
    } else if (a.stride == 1) {
      float* y0 = a.y + ((size_t)b * a.Cout + cg0) * a.Lout + t0 + nbase;
      float* y1 = y0 + a.Lout;
      if (has1) {
#pragma unroll
        for (int k = 0; k < P; ++k)
          if (nbase + P * k < limit) { y0[P * k] = v[k].x + bias0; y1[P * k] = v[k].y + bias1; }
      } else if (has0) {
#pragma unroll
        for (int k = 0; k < P; ++k)
          if (nbase + P * k < limit) y0[P * k] = v[k].x + bias0;
      }
    } else {
      float* y0 = a.y + ((size_t)b * a.Cout + cg0) * a.Lout;
      float* y1 = y0 + a.Lout;
#pragma unroll
      for (int k = 0; k < P; ++k) {
        const int n = nbase + P * k;
        const int t = t0 + n;
        const int idx = t / a.stride;
        if (n < limit && idx * a.stride == t) {
          if (has0) y0[idx] = v[k].x + bias0;
          if (has1) y1[idx] = v[k].y + bias1;
        }
      }
    }
    stamp(a.stamps, 10);
    if (a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(a.stamps, 11); }
  }
}

}  // namespace fc
