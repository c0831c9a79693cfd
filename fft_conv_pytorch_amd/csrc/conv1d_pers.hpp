// conv1d_pers.hpp -- persistent fused 1-D FFT convolution, NB batch items per workgroup.
//
// Same arithmetic as conv1d_fused.hpp (which stays the general fallback), organised for throughput:
//   * work items come from a host-built list; an item is (tile, group, out-chunk, first batch item,
//     number of batch items <= NB).  A workgroup runs up to TWO items back to back (written out twice,
//     not as a loop: hipcc spills the FFT registers around every barrier when the body sits in a loop)
//     and requests the second item's input samples while the first is in its inverse passes, so the
//     HBM reads of one item hide behind the arithmetic of the other;
//   * the NB batch items of an item share every kernel-spectrum load of the mix step, which is what
//     bounds the one-item-per-workgroup kernel (L2 -> L1 traffic of the spectrum);
//   * the pass-A twiddle table lives in LDS for the lifetime of the workgroup;
//   * the mix streams the spectrum through two register sets that are both in flight (set A holds step
//     s, set B step s+1; each is refilled with step s+2 right after use); the first two sets of an item
//     are requested before the barrier in front of the mix;
//   * with PHASES a dilation d runs as d interleaved phases on the batch-sharing axis (virtual batch
//     B*d against the undilated kernel spectrum).
// Fast-path restrictions (everything else runs conv1d_wide_kernel or conv1d_fused_kernel): a single
// input-channel chunk (Cin/groups <= CIB), full output chunks (Cout/groups a multiple of CIB), stride 1.
#pragma once
#include "conv1d_fused.hpp"

namespace fc {

struct WorkItem {
  int b0, nbc, tile, goc;   // goc = g * n_ochunks + oc
};

struct Conv1dPersArgs {
  Conv1dArgs c;
  const WorkItem* items;
  int n_items;
};

// profiling hook: lane 0 of EVERY wave records the 100 MHz wall clock, [item][wave (16 slots)][stamp (16 slots)]
__device__ __forceinline__ void stamp_item(unsigned long long* buf, int item, int slot) {
  if (buf != nullptr && (threadIdx.x & 63) == 0)
    buf[((size_t)item * 16 + (threadIdx.x >> 6)) * 16 + slot] = __builtin_amdgcn_s_memrealtime();
}
// same with the shader clock counter (slots 12 / 13 bracket an item: cycles per 100 MHz tick = the clock the CU held)
__device__ __forceinline__ void stamp_clock(unsigned long long* buf, int item, int slot) {
  if (buf != nullptr && (threadIdx.x & 63) == 0)
    buf[((size_t)item * 16 + (threadIdx.x >> 6)) * 16 + slot] = __builtin_amdgcn_s_memtime();
}

// PH2 (with PHASES, an even number of phases): the two sequences of a wave are the phases 2j and 2j+1 of ONE channel
// pair instead of two channel pairs of one phase, so that a lane can load / store both phases' samples of a position
// as 8 bytes and trade halves with its partner lane (v_permlane32_swap): half the memory instructions, each
// touching half the cache lines of the 4-byte accesses at a 4*d-byte pitch.
// STAMPS: the profiling hook (fc_forward_stamped) is a build of its own -- compiled into the product kernel, even
// switched off, its guarded stores made hipcc put a full s_waitcnt vmcnt(0) between the two items of a workgroup
// (a pending store's registers are reused), i.e. a wait for every output store of the first item.
// PH4 (with PHASES, a multiple of four phases, 8 -> 8 channel blocks, four slots): the packing is turned round -- a complex
// sequence carries two PHASES of ONE channel, a wave owns one channel and its two halves the phase pairs (0,1) and (2,3) --
// so a lane loads / stores all four phases of a position as 16 contiguous bytes (full cache lines per wave-instruction, where
// PH2's 8 bytes at a 16-byte pitch cost partial-line write-backs and re-fetches: 1.6x the output bytes written at cfgD).
// The mix contracts the same 8 x 8 matrix per bin; its slots are phases and its steps single output channels.
template <int P, int S, int CIB, int NB, int NT, bool PHASES = false, int RING = 2, bool DIAG = false, bool SEG = false, bool PH2 = false,
          bool STAMPS = false, bool PH4 = false>
__global__ __launch_bounds__(NT, 2) void conv1d_pers_kernel(const Conv1dPersArgs pa) {
  static_assert(!PH2 || (PHASES && !DIAG && !SEG && S == 1 && CIB == 8), "paired phases: plain phase build on a P*P tile");
  static_assert(!PH4 || (PHASES && !DIAG && !SEG && !PH2 && S == 1 && CIB == 8 && NB == 4), "phase quads: 8 channels x 2 phase pairs");
  using G = Geo<P, S>;
  constexpr int T = G::T;
  constexpr int NPI = CIB / 2;
  constexpr int NSEQ = NB * NPI;
  static_assert(NT == NSEQ * G::TS, "one thread slot per point group of every sequence");
  static_assert((T / 2) % NT == 0, "bin pairs divide evenly over the threads");
  static_assert(NPI % 2 == 0, "the spectrum pipeline alternates two register sets");
  static_assert(NB * CIB * 2 <= 64, "the self-paired bins are handled by one wave");
  constexpr int BP = (T / 2) / NT;          // bin pairs per thread
  constexpr int TWN = P * G::N2;            // pass-A twiddle table entries
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  const Conv1dArgs& a = pa.c;
  auto stampi = [&](int item, int slot) { if constexpr (STAMPS) stamp_item(a.stamps, item, slot); };
  auto stampc = [&](int item, int slot) { if constexpr (STAMPS) stamp_clock(a.stamps, item, slot); };
  const int nph = PHASES ? a.ph : 1;          // dilation phases (compile-time 1 in the plain build)

  const int tid = threadIdx.x;
  const int sq = tid / G::TS;               // sequence slot: batch slot nb, channel pair p
  const int tseq = tid % G::TS;
  static_assert(G::TS <= 64 && (NPI * G::TS) % 64 == 0, "a batch slot is a whole number of wavefronts");
  // batch slot and channel pair of this thread's sequence: slot wave-uniform, except with PH2 where the two halves of
  // a wave are slots 2j and 2j+1 (even / odd phase) of pair wave % 4
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (PH4: nb = phase pair of this half-wave, pr = the wave's channel, sequence (pair, channel) at pair * 8 + channel)
  const int nb = PH4 ? ((tid >> 5) & 1) : (PH2 ? 2 * (wv / NPI) + ((tid >> 5) & 1) : __builtin_amdgcn_readfirstlane(tid / (NPI * G::TS)));
  const int pr = PH4 ? wv : (PH2 ? wv % NPI : sq % NPI);
  f2* twl = lds;
  f2* zbuf = lds + TWN;                     // [NSEQ][LSEQ], sequence (slot, pair) at slot * NPI + pair
  f2* zseq = PH4 ? zbuf + (nb * 8 + pr) * G::LSEQ : zbuf + (nb * NPI + pr) * G::LSEQ;

  const PadMap pm = make_padmap(a.pad_mode, a.L);
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(S * P * 8));
  const size_t wgroup = (size_t)a.Cog_pad * (a.Cig_pad / 2) * (T / 2);   // float4 per group

  // ---- input samples of one item -> registers (unrolled buffer loads; border tiles get per-sample
  // padded / out-of-range offsets).  Only requests: nothing waits here.
  // Dilation d (stride 1) runs as d interleaved phases: y[t*d + p] = sum_k w[k] x[(t + k)*d + p] is an
  // undilated convolution of the phase-p samples, and all phases use the same spectrum -- so they ride
  // the batch-sharing axis as a virtual batch of B*d items (slot vb -> batch vb / d, phase vb % d).
  auto fetch = [&](const WorkItem& wi, f2 (&v)[P]) {
    const int g = wi.goc / a.n_ochunks;
    const int vb = a.slot_tiles ? wi.b0 : wi.b0 + nb;
    const int tile = a.slot_tiles ? wi.tile + nb : wi.tile;
    const int bfirst = wi.b0 / nph, blast = a.slot_tiles ? bfirst : (wi.b0 + wi.nbc - 1) / nph;
    const int b = vb / nph, phase = vb - b * nph;
    const int pos0 = tile * a.V * nph + phase - a.pad + (SEG ? a.pos_shift : 0);   // source position of the tile's first sample
    const bool interior = (pos0 >= 0) && (pos0 + (T - 1) * nph < a.L);
    const bool act_in = nb < wi.nbc;
    const float* xbase = a.x + ((size_t)bfirst * a.Cin + (size_t)g * a.Cig) * a.L;
    const BufRsrc xg = make_rsrc(xbase, (unsigned)(((size_t)(blast - bfirst) * a.Cin + a.Cig) * a.L * 4));
    const int ci0 = 2 * pr;
    // (a depthwise plan's last block may reach past the last channel)
    const bool has0 = act_in && ci0 < a.Cig && (!DIAG || g * CIB + ci0 < a.Cin);
    const bool has1 = act_in && ci0 + 1 < a.Cig && (!DIAG || g * CIB + ci0 + 1 < a.Cin);
    const unsigned ro0 = ((unsigned)(b - bfirst) * (unsigned)a.Cin + (unsigned)ci0) * (unsigned)a.L * 4u;
    const unsigned ro1 = ro0 + (unsigned)a.L * 4u;
    if constexpr (PH4) {
      // wave = channel pr of batch item bA, phases phA .. phA + 3 (phA a multiple of 4: items hold four slots)
      const int bA = wi.b0 / nph, phA = wi.b0 - bA * nph;
      const int posA = wi.tile * a.V * nph + phA - a.pad;
      const bool all_in = (posA >= 0) && (posA + 3 + (T - 1) * nph < a.L);
      const bool chan = pr < a.Cig;
      const unsigned rA = ((unsigned)(bA - bfirst) * (unsigned)a.Cin + (unsigned)pr) * (unsigned)a.L * 4u;
      // 16-byte accesses need the row base and the tile position on 16-byte boundaries (uniform test; else 4-byte loads)
      const bool al16 = (((size_t)xbase | ((size_t)a.L * 4) | ((size_t)(unsigned)posA * 4)) & 15) == 0;
      if (all_in && al16) {
        const unsigned q0 = chan ? rA + (unsigned)(posA + (tid & 63) * nph) * 4u : 0x80000000u;
        const unsigned step4 = 256u * (unsigned)nph;
#pragma unroll
        for (int m = 0; m < P / 2; ++m) {
          const f4 q = buf_load_f32x4(xg, q0, step4 * m);
          v[2 * m] = q.xy; v[2 * m + 1] = q.zw;
        }
        return;        // (the halves are traded in fetch_finish)
      }
      // border tiles / unaligned rows: per-sample padded loads of this half-wave's two phases
      const unsigned ro = ((unsigned)(bA - bfirst) * (unsigned)a.Cin + (unsigned)pr) * (unsigned)a.L * 4u;
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int pos = posA + 2 * nb + (G::N2 * n1 + tseq) * nph;
        v[n1].x = buf_load_f32(xg, padded_offset(ro, pos, a.L, a.pad, pm, chan), 0);
        v[n1].y = buf_load_f32(xg, padded_offset(ro, pos + 1, a.L, a.pad, pm, chan), 0);
      }
      return;
    }
    if constexpr (PH2) {
      // even slot of this wave (wave-uniform), its batch item and (even) phase; the odd slot is the next phase
      const int vbA = wi.b0 + 2 * (wv / NPI), bA = vbA / nph, phA = vbA - bA * nph;
      const int posA = wi.tile * a.V * nph + phA - a.pad;
      const bool both_in = (posA >= 0) && (posA + 1 + (T - 1) * nph < a.L);
      if (both_in && __all(has1)) {
        // lane L of the wave asks for (phase, phase + 1) at the positions 64 m + L, m = 0 .. P/2 - 1; the swap hands
        // lanes 0-31 (even phase) the even-phase halves of both lane groups and lanes 32-63 the odd-phase halves:
        // rows n1 = 2m and 2m + 1 of this thread's sequence
        const unsigned rA = ((unsigned)(bA - bfirst) * (unsigned)a.Cin + (unsigned)ci0) * (unsigned)a.L * 4u;
        const unsigned q0 = rA + (unsigned)(posA + (tid & 63) * nph) * 4u, q1 = q0 + (unsigned)a.L * 4u;
        const unsigned step2 = 256u * (unsigned)nph;
#pragma unroll
        for (int m = 0; m < P / 2; ++m) {
          f2 cx = buf_load_f32x2(xg, q0, step2 * m), cy = buf_load_f32x2(xg, q1, step2 * m);
          v[2 * m].x = cx.x; v[2 * m + 1].x = cx.y; v[2 * m].y = cy.x; v[2 * m + 1].y = cy.y;
        }
        return;        // (the halves are traded in fetch_finish, when the item starts: here nothing may wait for the loads)
      }
    }
    if (interior && has1 && !PHASES) {
      const unsigned v0 = ro0 + (unsigned)(pos0 + tseq) * 4u, v1 = ro1 + (unsigned)(pos0 + tseq) * 4u;
#if FC_DIAG == 7
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) { v[n1] = mk2(1e-3f * n1, (float)tseq); asm volatile("" : "+v"(v[n1])); }
      (void)v0; (void)v1;
      return;
#endif
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        v[n1].x = buf_load_f32(xg, v0, G::N2 * n1 * 4);
        v[n1].y = buf_load_f32(xg, v1, G::N2 * n1 * 4);
      }
    } else if (interior && has1) {
      const unsigned v0 = ro0 + (unsigned)(pos0 + tseq * nph) * 4u, v1 = ro1 + (unsigned)(pos0 + tseq * nph) * 4u;
      const unsigned step = (unsigned)(G::N2 * 4) * (unsigned)nph;
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        v[n1].x = buf_load_f32(xg, v0, step * n1);
        v[n1].y = buf_load_f32(xg, v1, step * n1);
      }
    } else {
#pragma unroll
      for (int n1 = 0; n1 < P; ++n1) {
        const int pos = pos0 + (G::N2 * n1 + tseq) * nph;
        v[n1].x = buf_load_f32(xg, padded_offset(ro0, pos, a.L, a.pad, pm, has0), 0);
        v[n1].y = buf_load_f32(xg, padded_offset(ro1, pos, a.L, a.pad, pm, has1), 0);
      }
    }
  };

  // second half of a paired-phase fetch, run when the item starts (its loads have long landed): the partner lanes trade
  // the halves they loaded for each other.  Recomputes which path fetch() took for this item.
  auto fetch_finish = [&](const WorkItem& wi, f2 (&v)[P]) {
    if constexpr (PH4) {
      // lanes 0-31 keep (phase 0, 1) of the positions both lane groups loaded, lanes 32-63 get (phase 2, 3)
      const int bA = wi.b0 / nph, phA = wi.b0 - bA * nph;
      const int posA = wi.tile * a.V * nph + phA - a.pad;
      const int bfirst = wi.b0 / nph;
      const float* xbase = a.x + ((size_t)bfirst * a.Cin + (size_t)(wi.goc / a.n_ochunks) * a.Cig) * a.L;
      const bool all_in = (posA >= 0) && (posA + 3 + (T - 1) * nph < a.L);
      const bool al16 = (((size_t)xbase | ((size_t)a.L * 4) | ((size_t)(unsigned)posA * 4)) & 15) == 0;
      if (all_in && al16) {
#pragma unroll
        for (int m = 0; m < P / 2; ++m) { swap_halves_c<0>(v[2 * m], v[2 * m + 1]); swap_halves_c<1>(v[2 * m], v[2 * m + 1]); }
      }
    }
    if constexpr (PH2) {
      const bool has1 = nb < wi.nbc && 2 * pr + 1 < a.Cig;
      const int vbA = wi.b0 + 2 * (wv / NPI), bA = vbA / nph, phA = vbA - bA * nph;
      const int posA = wi.tile * a.V * nph + phA - a.pad;
      const bool both_in = (posA >= 0) && (posA + 1 + (T - 1) * nph < a.L);
      if (both_in && __all(has1)) {
#pragma unroll
        for (int m = 0; m < P / 2; ++m) { swap_halves_c<0>(v[2 * m], v[2 * m + 1]); swap_halves_c<1>(v[2 * m], v[2 * m + 1]); }
      }
    }
  };

  // ---- one item: forward FFT, mix, [request the next item's samples], inverse FFT, store
  auto run = [&](const WorkItem& wi, int it, f2 (&v)[P], bool more, const WorkItem& wnext, f2 (&vnext)[P]) {
    const int g = wi.goc / a.n_ochunks, oc = wi.goc % a.n_ochunks;
    const bool act_in = PH4 ? true : nb < wi.nbc;          // (phase quads: every item holds its four slots)
    // DIAG (depthwise): the spectrum is [channel pair][T/2] float4 = {H(2p)[f], H(2p+1)[f]}, 8 channels per block g
    const BufRsrc wg = DIAG ? make_rsrc(a.wspec + (size_t)g * NPI * (T / 2), (unsigned)(NPI * (T / 2) * 16))
                            : make_rsrc(a.wspec + (size_t)g * wgroup, (unsigned)(wgroup * 16));
    stampi(it, 0);
    stampc(it, 12);
    // bias of this lane's two output channels, requested now and used in the last pass
    const int cg0 = g * a.Cog + oc * a.cob + (PH4 ? pr : 2 * pr);     // (PH4: one output channel per wave)
    const bool ok0 = !DIAG || cg0 < a.Cout, ok1 = !DIAG || cg0 + 1 < a.Cout;
    // (buffer loads: a missing bias is an empty resource, a missing channel an offset outside it -- no branch, so
    // nothing waits for these two values here; they are pinned as arrived after the mix, see below)
    const BufRsrc brs = make_rsrc(a.bias, a.bias ? (unsigned)a.Cout * 4u : 0u);
    float bias0 = buf_load_f32(brs, ok0 ? (unsigned)cg0 * 4u : 0xFFFFFFFFu, 0);
    float bias1 = buf_load_f32(brs, ok1 ? (unsigned)(cg0 + 1) * 4u : 0xFFFFFFFFu, 0);
    // ------------------------------------------------ forward pass A
    if (STAMPS && a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stampi(it, 1); }
    // (act_in is wave-uniform and the two passes of a sequence only need wave-level ordering)
    if (act_in) {
      fetch_finish(wi, v);
      passA_fft_twiddle_store_lds<G, -1>(v, zseq, tseq, twl);
      stampi(it, 2);
      seq_sync<G>();
      stampi(it, 3);
      // ---------------------------------------------- forward pass B
      passB_load<G>(v, zseq, tseq);
      seq_sync<G>();
      const int j = passB_compute<G, -1>(v, tseq, twB);
      const int k1 = tseq >> G::LGS;
      f2* dst = zseq + G::nat(k1 + P * P * j);
#if FC_DIAG == 2
      { f2 sink = v[0];
#pragma unroll
        for (int k = 1; k < P; ++k) sink += v[k];
        if (sink.x == 123.456f) dst[0] = sink; }
#else
#pragma unroll
      for (int k = 0; k < P; ++k) dst[P * k] = v[k];
#endif
    }
    // ------------------------------------------------ mix, depthwise: every channel meets only its own kernel
    if constexpr (DIAG) {
      f4 wd[BP][NPI];
#pragma unroll
      for (int m = 0; m < BP; ++m)
#pragma unroll
        for (int p = 0; p < NPI; ++p) wd[m][p] = buf_load_f32x4(wg, (unsigned)(tid + m * NT) * 16u, p * (T / 2) * 16);
      // self-paired bins 0 and T/2: lane (batch b, pair p, bin) of wave 0; wspec[p][0] = {Re H_e[0], Re H_e[T/2], Re H_o[0], Re H_o[T/2]}
      const int sd_b = tid / (2 * NPI), sd_p = (tid >> 1) % NPI, sd_f = (tid & 1) ? T / 2 : 0;
      const bool sd_act = tid < NB * NPI * 2 && sd_b < wi.nbc;
      f4 sdw = {0.f, 0.f, 0.f, 0.f};
      if (sd_act) sdw = buf_load_f32x4(wg, 0u, sd_p * (T / 2) * 16);
      stampi(it, 4);
      __syncthreads();
      stampi(it, 5);
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
            xe[b][p] = lds_rd<p * G::LSEQ * 8>(bf);
            xo[b][p] = lds_rd<p * G::LSEQ * 8>(bg);
          });
        });
#pragma unroll
        for (int b = 0; b < NB; ++b) { lds_arrive(xe[b]); lds_arrive(xo[b]); }
        if (f != 0) {
#pragma unroll
          for (int b = 0; b < NB; ++b)
            if (b < wi.nbc) {
#pragma unroll
              for (int p = 0; p < NPI; ++p) {
                const f2 zf = xe[b][p], zg = xo[b][p];
                const f2 ye = cmul(add_conj(zf, zg), wd[m][p].xy);        // 2 X_even[f] * H_even[f]
                const f2 yo = cmul(sub_conj_divi(zf, zg), wd[m][p].zw);   // 2 X_odd[f]  * H_odd[f]
                f2* zb = zbuf + (b * NPI + p) * G::LSEQ;
                zb[G::nat(f)] = add_pi(ye, yo);
                zb[G::nat(fm)] = conj_add_iconj(ye, yo);
              }
            }
        }
      });
      if (sd_act) {
        f2* zp = zbuf + (sd_b * NPI + sd_p) * G::LSEQ + G::nat(sd_f);
        const f2 z = *zp;
        *zp = mk2(2.f * z.x * ((tid & 1) ? sdw.y : sdw.x), 2.f * z.y * ((tid & 1) ? sdw.w : sdw.z));
      }
    } else if constexpr (PH4) {
    // ------------------------------------------------ mix, phase quads: slots are the four phases, sequence (pair hp, channel c)
    // holds phases 2hp (re) and 2hp+1 (im) of channel c; a step is one output channel (its 8 inputs = 4 float4)
    const unsigned ostride = (unsigned)(a.Cig_pad / 2) * (T / 2) * 16u;
    const unsigned wbase = (unsigned)(oc * a.cob) * ostride;
    constexpr int R4 = 4;
    static_assert(BP == 1, "one bin pair per thread");
    const int sb_s = tid / (2 * CIB), sb_o = (tid >> 1) % CIB, sb_f = (tid & 1) ? T / 2 : 0;     // (phase, output, bin) of wave 0
    const bool sb_act = tid < 4 * CIB * 2;
    f4 sbw[NPI];
    if (sb_act) {
#pragma unroll
      for (int p = 0; p < NPI; ++p) sbw[p] = buf_load_f32x4(wg, (unsigned)sb_o * ostride, wbase + p * (T / 2) * 16);
    }
    f4 w4[R4][NPI];
    auto issue4 = [&](auto oc_) {
      constexpr int o = decltype(oc_)::value;
#pragma unroll
      for (int p = 0; p < NPI; ++p) w4[o % R4][p] = buf_load_f32x4(wg, (unsigned)tid * 16u, wbase + (unsigned)o * ostride + p * (T / 2) * 16);
    };
    static_for<0, R4>([&](auto oc_) { issue4(oc_); });
    stampi(it, 4);
    __syncthreads();
    stampi(it, 5);
    {
      float sbv[CIB];
      if (sb_act) {
#pragma unroll
        for (int c = 0; c < CIB; ++c) {
          const f2 z = zbuf[((sb_s >> 1) * 8 + c) * G::LSEQ + G::nat(sb_f)];
          sbv[c] = (sb_s & 1) ? z.y : z.x;
        }
      }
      const int f = tid, fm = (T - f) & (T - 1);
      f2 xe[2][CIB], xo[2][CIB];          // 2 * X of phase 2hp / 2hp+1 of every channel at bin f
      {
        const unsigned af = lds_off(zbuf + G::nat(f)), ag = lds_off(zbuf + G::nat(fm));
        static_for<0, 2>([&](auto hc) {
          constexpr int hp = decltype(hc)::value;
          static_for<0, CIB>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            xe[hp][c] = lds_rd_far<(hp * 8 + c) * G::LSEQ * 8>(af);
            xo[hp][c] = lds_rd_far<(hp * 8 + c) * G::LSEQ * 8>(ag);
          });
        });
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) { lds_arrive(xe[hp]); lds_arrive(xo[hp]); }
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
          for (int c = 0; c < CIB; ++c) {
            const f2 zf = xe[hp][c], zg = xo[hp][c];
            xe[hp][c] = add_conj(zf, zg);
            xo[hp][c] = sub_conj_divi(zf, zg);
          }
      }
      static_for<0, CIB>([&](auto oc_) {
        constexpr int o = decltype(oc_)::value;
        const f4 (&wc)[NPI] = w4[o % R4];
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
          f2 ya = mk2(0.f, 0.f), yb = mk2(0.f, 0.f);
#pragma unroll
          for (int p = 0; p < NPI; ++p) {
            // (ya += xe_a h_a + xe_b h_b ; yb += xo_a h_a + xo_b h_b: two chains, interleaved)
            asm("v_pk_fma_f32 %0, %2, %6, %0 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %1, %4, %6, %1 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %0, %2, %6, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                "v_pk_fma_f32 %1, %4, %6, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                "v_pk_fma_f32 %0, %3, %7, %0 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %1, %5, %7, %1 op_sel_hi:[0,1,1]\n\t"
                "v_pk_fma_f32 %0, %3, %7, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
                "v_pk_fma_f32 %1, %5, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
                : "+v"(ya), "+v"(yb)
                : "v"(xe[hp][2 * p]), "v"(xe[hp][2 * p + 1]), "v"(xo[hp][2 * p]), "v"(xo[hp][2 * p + 1]), "v"(wc[p].xy), "v"(wc[p].zw));
          }
          if (f != 0) {
            f2* zb = zbuf + (hp * 8 + o) * G::LSEQ;
            zb[G::nat(f)] = add_pi(ya, yb);
            zb[G::nat(fm)] = conj_add_iconj(ya, yb);
          }
        }
        if constexpr (o + R4 < CIB) issue4(std::integral_constant<int, o + R4>{});
      });
      if (sb_act) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          acc = fmaf(2.f * sbv[2 * p], (tid & 1) ? sbw[p].y : sbw[p].x, acc);
          acc = fmaf(2.f * sbv[2 * p + 1], (tid & 1) ? sbw[p].w : sbw[p].z, acc);
        }
        float* dstf = reinterpret_cast<float*>(zbuf + ((sb_s >> 1) * 8 + sb_o) * G::LSEQ + G::nat(sb_f)) + (sb_s & 1);
        *dstf = acc;
      }
    }
    } else {
    // ------------------------------------------------ mix
    // The first two spectrum sets (and the self-paired bins' weights) do not depend on this item's
    // transforms: they are requested BEFORE the barrier and travel while the slower waves finish.
    const unsigned ostride = (unsigned)(a.Cig_pad / 2) * (T / 2) * 16u;
    const unsigned wbase = (unsigned)(oc * a.cob) * ostride;
    // Self-paired bins 0 and T/2 (both spectra real there; wspec[.][0] = {Re H[0], Re H[T/2]}):
    // lane (batch b, output o, bin) of wave 0 owns one real output.
    const int sb_b = tid / (2 * CIB), sb_o = (tid >> 1) % CIB, sb_f = (tid & 1) ? T / 2 : 0;
    const bool sb_act = tid < NB * CIB * 2 && sb_b < wi.nbc;
    f4 sbw[NPI];
    if (sb_act) {
#pragma unroll
      for (int p = 0; p < NPI; ++p) sbw[p] = buf_load_f32x4(wg, (unsigned)sb_o * ostride, wbase + p * (T / 2) * 16);
    }
    // spectrum pipeline: step = (bin pair m, output pair q); two named register sets (static
    // indices), both in flight: set A holds step s, set B step s+1, each is refilled with step s+2
    // right after it has been contracted
    auto issue = [&](int m, int q, f4 (&dst)[2 * NPI]) {
      const unsigned vo = (unsigned)(tid + m * NT) * 16u;
      const unsigned sa = wbase + (unsigned)(2 * q) * ostride, sb = sa + ostride;
#if FC_DIAG == 1
#pragma unroll
      for (int p = 0; p < 2 * NPI; ++p) { f4 c; c.x = 1e-3f * (p + q); c.y = c.x; c.z = c.x; c.w = c.x; asm volatile("" : "+v"(c)); dst[p] = c; }
      return;
#endif
#pragma unroll
      for (int p = 0; p < NPI; ++p) {
        dst[2 * p] = buf_load_f32x4(wg, vo, sa + p * (T / 2) * 16);
        dst[2 * p + 1] = buf_load_f32x4(wg, vo, sb + p * (T / 2) * 16);
      }
    };
    // RING register sets: step s lives in set s % RING; the first RING steps are requested here
    // (RING = 3 fits in 244 VGPRs at NB = 4 but measured no faster than 2: 37.4 vs 36.9 us at cfgA)
    f4 wr[RING][2 * NPI];
    static_for<0, RING>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      if constexpr (s < BP * NPI) issue(s / NPI, s % NPI, wr[s]);
    });
    stampi(it, 4);
#if FC_DIAG != 5
    __syncthreads();
#endif
    stampi(it, 5);
    {
      f2 sbz[NPI];
      if (sb_act) {
#pragma unroll
        for (int p = 0; p < NPI; ++p) sbz[p] = zbuf[(sb_b * NPI + p) * G::LSEQ + G::nat(sb_f)];
      }
      f2 xe[NB][NPI], xo[NB][NPI];       // 2*X of the even / odd channel of every pair
      auto contract = [&](int f, int fm, int q, const f4 (&wc)[2 * NPI]) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          f2 ya = mk2(0.f, 0.f), yb = mk2(0.f, 0.f);
#pragma unroll
          for (int p = 0; p < NPI; ++p) {
            const f4 ha = wc[2 * p], hb = wc[2 * p + 1];
            cmac2x2(ya, yb, xe[b][p], xo[b][p], ha.xy, ha.zw, hb.xy, hb.zw);
          }
          if (f != 0 && b < wi.nbc) {
            f2* zb = zbuf + (b * NPI + q) * G::LSEQ;
            zb[G::nat(f)] = add_pi(ya, yb);
            zb[G::nat(fm)] = conj_add_iconj(ya, yb);
          }
        }
      };
      static_for<0, BP>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int f = tid + m * NT;
        const int fm = (T - f) & (T - 1);
        {
          // all 2*NB*NPI bin reads are requested first (plain ds_read_b64), then combined
          const unsigned af = lds_off(zbuf + G::nat(f)), ag = lds_off(zbuf + G::nat(fm));
          static_for<0, NB>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const unsigned bf = af + b * NPI * G::LSEQ * 8, bg = ag + b * NPI * G::LSEQ * 8;
            static_for<0, NPI>([&](auto pc) {
              constexpr int p = decltype(pc)::value;
              xe[b][p] = lds_rd<p * G::LSEQ * 8>(bf);
              xo[b][p] = lds_rd<p * G::LSEQ * 8>(bg);
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
        }
        static_for<0, NPI>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          constexpr int s = m * NPI + q;               // this step; its set is refilled with step s + RING
          contract(f, fm, q, wr[s % RING]);
          if constexpr (s + RING < BP * NPI) issue((s + RING) / NPI, (s + RING) % NPI, wr[s % RING]);
        });
      });
      if (sb_act) {
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < NPI; ++p) {
          acc = fmaf(2.f * sbz[p].x, (tid & 1) ? sbw[p].y : sbw[p].x, acc);
          acc = fmaf(2.f * sbz[p].y, (tid & 1) ? sbw[p].w : sbw[p].z, acc);
        }
        float* dstf = reinterpret_cast<float*>(zbuf + (sb_b * NPI + (sb_o >> 1)) * G::LSEQ + G::nat(sb_f)) + (sb_o & 1);
        *dstf = acc;
      }
    }
    }   // dense / depthwise mix
    stampi(it, 6);
#if FC_DIAG != 5
    __syncthreads();
#endif
    stampi(it, 7);
    // gfx9 counts loads and stores in ONE vmcnt: a value first used after later memory instructions went out through
    // branches gets a full s_waitcnt vmcnt(0).  The bias is therefore declared arrived here (only this item's spectrum
    // loads are older, all consumed), and the next item's samples right before this item's output stores.
    asm volatile("" : "+v"(bias0), "+v"(bias1));
    // the next item's samples travel from HBM while this item's inverse passes run
    if (more) fetch(wnext, vnext);
    // ------------------------------------------------ inverse pass A'
    if (act_in) {
      nat_load<G>(v, zseq, tseq);
      seq_sync<G>();
      passA_fft_twiddle_store_lds<G, +1>(v, zseq, tseq, twl);
      stampi(it, 8);
      seq_sync<G>();
      stampi(it, 9);
      // ---------------------------------------------- inverse pass B' + store
      passB_load<G>(v, zseq, tseq);
      const int j = passB_compute<G, +1>(v, tseq, twB);
      if (more) {
        // (requested a whole inverse transform ago: no stall; but behind the stores the next item would have to wait
        // for every one of them to be acknowledged before it may touch its samples)
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) asm volatile("" : "+v"(vnext[n1]));
      }
      const int o1 = tseq >> G::LGS;
      const int vb = a.slot_tiles ? wi.b0 : wi.b0 + nb;
      const int tile = a.slot_tiles ? wi.tile + nb : wi.tile;
      const int b = vb / nph, phase = vb - b * nph;
      const int t0 = tile * a.V;                                      // in samples of this phase
      const int limit = min(a.V, (a.Lfull - phase + nph - 1) / nph - t0);
      const int nbase = o1 + P * P * j;
      float* y0 = a.y + ((size_t)b * a.Cout + cg0) * a.Lout + (size_t)(t0 + nbase) * nph + phase;
      float* y1 = y0 + a.Lout;
      if constexpr (DIAG) {
        // depthwise: the last block may hold fewer than 8 channels, so every row is guarded
        const int ystep = P * nph;
        const bool addo = SEG && a.add_out;
#pragma unroll
        for (int k = 0; k < P; ++k)
          if (nbase + P * k < limit) {
            if (ok0) y0[ystep * k] = v[k].x + (addo ? y0[ystep * k] : bias0);
            if (ok1) y1[ystep * k] = v[k].y + (addo ? y1[ystep * k] : bias1);
          }
      } else if (SEG && a.add_out) {
        // later segments of a long kernel accumulate into the output of the first
        const int ystep = P * nph;
#pragma unroll
        for (int k = 0; k < P; ++k)
          if (nbase + P * k < limit) { y0[ystep * k] += v[k].x; y1[ystep * k] += v[k].y; }
      } else if (!PHASES && S == 1) {
        // Lane o1 holds samples o1 + P k.  For k < ka every lane is inside the valid window, k = ka is the one partly
        // valid row of stores, later k are outside for every lane (ka is wave-uniform).  Buffer stores: one 32-bit
        // offset register per output row, the step P k as the instruction's immediate.
        const int ka = limit >= P ? (limit - P) / P + 1 : 0;
        const BufRsrc yr = make_rsrc(a.y + ((size_t)b * a.Cout + (size_t)(g * a.Cog + oc * a.cob)) * a.Lout, (unsigned)((size_t)a.cob * a.Lout * 4));
        const unsigned vo0 = (unsigned)((size_t)(2 * pr) * a.Lout + (size_t)(t0 + nbase)) * 4u, vo1 = vo0 + (unsigned)a.Lout * 4u;
        // Blocks of 8 rows: a block below ka is straight-line code behind ONE scalar branch (per-row tests, even
        // wave-uniform ones, cost more instructions than the stores they guard); only the block that holds row ka
        // tests its rows, and there the lanes past the window get an offset outside the resource (store dropped).
        static_for<0, P / 8>([&](auto bc) {
          constexpr int k0 = 8 * decltype(bc)::value;
          if (ka >= k0 + 8) {
#if FC_DIAG == 6
            { f2 sink = v[k0];
              static_for<k0 + 1, k0 + 8>([&](auto kc) { sink += v[decltype(kc)::value]; });
              if (sink.x == 123.456f) buf_store_f32(sink.x + bias0 + bias1, yr, vo0, 0); }
#else
            static_for<k0, k0 + 8>([&](auto kc) {
              constexpr int k = decltype(kc)::value;
              buf_store_f32(v[k].x + bias0, yr, vo0, P * k * 4);
              buf_store_f32(v[k].y + bias1, yr, vo1, P * k * 4);
            });
#endif
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
      } else if (!PHASES) {
#pragma unroll
        for (int k = 0; k < P; ++k)
          if (nbase + P * k < limit) { y0[P * k] = v[k].x + bias0; y1[P * k] = v[k].y + bias1; }
      } else if constexpr (PH4) {
        // all four phases of a position leave as 16 bytes: after the trade lane L of the wave holds (phases 0, 1) in
        // row 2m and (phases 2, 3) in row 2m + 1 of sample 64 m + L.  Phases past the end of the row are shorter by one.
        const int bA = wi.b0 / nph, phA = wi.b0 - bA * nph;
        int lim[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lim[j] = min(a.V, (a.Lfull - (phA + j) + nph - 1) / nph - t0);
        const BufRsrc yr = make_rsrc(a.y + ((size_t)bA * a.Cout + (size_t)(g * a.Cog + oc * a.cob)) * a.Lout, (unsigned)((size_t)a.cob * a.Lout * 4));
        const int L64 = tid & 63;
        const unsigned w0 = (unsigned)(((size_t)pr * a.Lout + (size_t)(t0 + L64) * nph + phA) * 4);
        const unsigned step4 = 256u * (unsigned)nph;
        const bool al16 = (((size_t)a.y | ((size_t)a.Lout * 4)) & 15) == 0;
        const int mfull = (al16 && lim[3] >= 64) ? (lim[3] - 64) / 64 + 1 : 0;       // rows m below it: every lane, every phase valid
#pragma unroll
        for (int m = 0; m < P / 2; ++m) { swap_halves_c<0>(v[2 * m], v[2 * m + 1]); swap_halves_c<1>(v[2 * m], v[2 * m + 1]); }
        static_for<0, P / 8>([&](auto bc) {
          constexpr int m0 = 4 * decltype(bc)::value;
          if (mfull >= m0 + 4) {
            static_for<m0, m0 + 4>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              u32x4 d;
              d.x = __float_as_uint(v[2 * m].x + bias0); d.y = __float_as_uint(v[2 * m].y + bias0);
              d.z = __float_as_uint(v[2 * m + 1].x + bias0); d.w = __float_as_uint(v[2 * m + 1].y + bias0);
              __builtin_amdgcn_raw_buffer_store_b128(d, yr, w0, step4 * m, 0);
            });
          } else if (64 * m0 < lim[0]) {
            static_for<m0, m0 + 4>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              if (64 * m < lim[0]) {
                const int n = 64 * m + L64;
                buf_store_f32(v[2 * m].x + bias0, yr, w0 | (n < lim[0] ? 0u : 0x80000000u), step4 * m);
                buf_store_f32(v[2 * m].y + bias0, yr, (w0 + 4u) | (n < lim[1] ? 0u : 0x80000000u), step4 * m);
                buf_store_f32(v[2 * m + 1].x + bias0, yr, (w0 + 8u) | (n < lim[2] ? 0u : 0x80000000u), step4 * m);
                buf_store_f32(v[2 * m + 1].y + bias0, yr, (w0 + 12u) | (n < lim[3] ? 0u : 0x80000000u), step4 * m);
              }
            });
          }
        });
      } else if constexpr (PH2) {
        // both phases of a position leave as 8 bytes: rows 2m and 2m + 1 trade halves so that lane L holds (even
        // phase, odd phase) of sample 64 m + L.  The odd phase may be one sample shorter at the very end of a row.
        const int vbA = wi.b0 + 2 * (wv / NPI), bA = vbA / nph, phA = vbA - bA * nph;
        const int limA = min(a.V, (a.Lfull - phA + nph - 1) / nph - t0), limB = min(a.V, (a.Lfull - phA - 1 + nph - 1) / nph - t0);
        const BufRsrc yr = make_rsrc(a.y + ((size_t)bA * a.Cout + (size_t)(g * a.Cog + oc * a.cob)) * a.Lout, (unsigned)((size_t)a.cob * a.Lout * 4));
        const int L64 = tid & 63;
        const unsigned w0 = (unsigned)(((size_t)(2 * pr) * a.Lout + (size_t)(t0 + L64) * nph + phA) * 4), w1 = w0 + (unsigned)a.Lout * 4u;
        const unsigned step2 = 256u * (unsigned)nph;
        const int mfull = limB >= 64 ? (limB - 64) / 64 + 1 : 0;        // rows m below it: every lane valid in both phases
#pragma unroll
        for (int m = 0; m < P / 2; ++m) { swap_halves_c<0>(v[2 * m], v[2 * m + 1]); swap_halves_c<1>(v[2 * m], v[2 * m + 1]); }
        static_for<0, P / 8>([&](auto bc) {
          constexpr int m0 = 4 * decltype(bc)::value;
          if (mfull >= m0 + 4) {
            static_for<m0, m0 + 4>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              buf_store_f32x2(mk2(v[2 * m].x + bias0, v[2 * m + 1].x + bias0), yr, w0, step2 * m);
              buf_store_f32x2(mk2(v[2 * m].y + bias1, v[2 * m + 1].y + bias1), yr, w1, step2 * m);
            });
          } else if (64 * m0 < limA) {
            static_for<m0, m0 + 4>([&](auto mc) {
              constexpr int m = decltype(mc)::value;
              if (64 * m < limA) {
                const int n = 64 * m + L64;
                const unsigned dA = n < limA ? 0u : 0x80000000u, dB = n < limB ? 0u : 0x80000000u;
                buf_store_f32(v[2 * m].x + bias0, yr, w0 | dA, step2 * m);
                buf_store_f32(v[2 * m + 1].x + bias0, yr, (w0 + 4u) | dB, step2 * m);
                buf_store_f32(v[2 * m].y + bias1, yr, w1 | dA, step2 * m);
                buf_store_f32(v[2 * m + 1].y + bias1, yr, (w1 + 4u) | dB, step2 * m);
              }
            });
          }
        });
      } else {
        const int ystep = P * nph;
#pragma unroll
        for (int k = 0; k < P; ++k)
          if (nbase + P * k < limit) { y0[ystep * k] = v[k].x + bias0; y1[ystep * k] = v[k].y + bias1; }
      }
    }
    stampi(it, 10);
    if (STAMPS && a.stamps) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stampi(it, 11); stampc(it, 13); }
    seq_sync<G>();     // this wave's sequences are free again (the mix barriers order the other waves)
  };

  const int it0 = blockIdx.x, it1 = blockIdx.x + gridDim.x;
  const bool two = it1 < pa.n_items;
  const WorkItem w0 = pa.items[it0];
  const WorkItem w1 = pa.items[two ? it1 : it0];
  f2 va[P], vb[P];
  fetch(w0, va);                             // first item's samples and the twiddle table travel together
  copy_table_to_lds<TWN, NT>(twl, a.twA, tid);
  __syncthreads();
  run(w0, it0, va, two, w1, vb);
  if (two) run(w1, it1, vb, false, w1, va);
}

}  // namespace fc
