// planes3d.hpp -- plane-major 3-D pipeline: three launches instead of five.
//
// The reference transforms all three axes in one rfftn / irfftn (functional.py:66-75).  The separable scheme of
// nd_passes.hpp runs one launch per axis and side (rows_r2c, c2c_fwd, fusedc, c2c_inv, rows_c2r): 5.1x the algorithmic
// HBM bytes at cfgC.  When the padded (y, x) plane fits a 64 x 64 transform and the z tile is 64 points, a whole
// plane's half-spectrum (32 x 64 complex = 16 KB) fits one workgroup's LDS, so both plane axes are fused on each side
// of the z pass and every intermediate is kept PLANE-MAJOR -- [(b, c)][z][col], col = fx*64 + fy, 2048 bin columns:
//
//   planes_fwd  one (b, ci, z) plane per workgroup: 16 KB in (contiguous), x rows (two real rows per complex FFT,
//               odd-frequency bins, see rows_r2c) -> unpack through DPP -> y columns -> 16 KB out (contiguous)
//   colz        one block of 16 neighbouring columns x NB batch items per workgroup.  ONE THREAD OWNS ONE SEQUENCE:
//               lane = column, so the 64 z samples of a thread are 64 loads of which every wave-instruction reads
//               4 x 128 contiguous bytes of a plane -- no transposition anywhere -- and the 64-point transforms run
//               entirely in registers (no LDS exchange).  Only the channel contraction crosses threads: the bins
//               travel through LDS in chunks of FC frequencies (threads swap roles: bin owner <-> sequence owner),
//               NB batch items share every kernel-spectrum load.  Stores mirror the loads.
//   planes_inv  one (b, co, z_out) plane per workgroup: y columns back, x rows back (c2r, valid window, stride, bias),
//               output plane staged in LDS and written as one contiguous run.
//
// HBM bytes at cfgC: x 67 + S 2x67 + H 67 + O 2x59 + y 45 = 430 MB (separable: 577 MB), three launches.
// The kernel transform keeps the separable passes (it runs once per weight version and produces the same
// [g][o][i/2][col][fz] layout).
#pragma once
#include "nd_passes.hpp"

namespace fc {

constexpr int kPlFx = 32;                 // odd-frequency bins along x
constexpr int kPlCols = kPlFx * 64;       // bin columns per plane
constexpr int kPlNT = 256;
constexpr int kPlPitch = 72;              // complex slots per LDS sequence: the engine's 8 rows of 9, and conflict-free for
                                          // 8 lanes x 4 sequences reading the same element index

__device__ __forceinline__ float dpp_half_mirror(float v) {   // lane i <- lane 7 - i inside every group of 8 lanes
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));
}

// ------------------------------------------------------------------------------------------ planes_fwd
struct PlaneFwdArgs {
  const float* src;      // signal (B, C, Z, Y, X)
  f2* dst;               // S [(b*C + ci)][NZ][2048]
  const f2* twA;         // tables of the 64-point tile
  const f2* twB;
  AxisMap mx, my, mz;    // padding maps (functional.py:60-62 folded into the loads)
  int SZ, SY, SX;        // source extents
  int NZ;                // padded planes per image
  int nxt, nyt, Vx, Vy;  // overlap-save tiles of the plane (padded plane larger than 64 x 64): tile (yt, xt) is the window of padded
                         // positions [yt*Vy, yt*Vy + 64) x [xt*Vx, xt*Vx + 64); dst then holds nyt*nxt blocks of 2048 columns per plane
  FastDiv d_nz, d_nt, d_nx;   // unit map of the launch (filled by the dispatcher): blockIdx = (img*NZ + zp)*ntile + tile, tile = yt*nxt + xt
};

template <int NT_>
__global__ __launch_bounds__(kPlNT) void planes_fwd_kernel(const PlaneFwdArgs a) {
  using G = Geo<8, 1>;
  // ONE 18 KB region serves, one after the other, as row buffer, x-pass exchange, [fx][y] transpose, y-pass exchange and
  // output staging (a workgroup barrier at every change of hands): eight workgroups per CU instead of four
  __shared__ __attribute__((aligned(16))) f2 reg[32 * kPlPitch];
  float* rowbuf = reinterpret_cast<float*>(reg);          // [64][68] floats while the plane is loaded
  constexpr int RP = 68;
  static_assert(64 * RP * 4 <= 32 * kPlPitch * 8, "row buffer fits the region");
  const BufRsrc twA = make_rsrc(a.twA, 8 * 8 * 8), twB = make_rsrc(a.twB, 8 * 8);
  const int tid = threadIdx.x, sq = tid >> 3, tseq = tid & 7;
  unsigned qq;
  const int tile = (int)fdivmod(blockIdx.x, a.d_nt, &qq);          // qq = img*NZ + zp
  const int img = (int)fdiv(qq, a.d_nz), zp = (int)qq - img * a.NZ;
  const int yt = (int)fdiv((unsigned)tile, a.d_nx), xt = tile - yt * a.nxt;
  const int x0 = xt * a.Vx, y0 = yt * a.Vy;                         // padded position of this tile's corner
  f4* outp = reinterpret_cast<f4*>(a.dst + (size_t)blockIdx.x * kPlCols);
  const int zs = axis_src(a.mz, zp);
  if (zs < 0) {                                            // a plane of padding zeros (workgroup-uniform)
    const f4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) outp[tid + kPlNT * u] = z;
    return;
  }
  // pass-A twiddles of lane tseq: the same for the x rows and the y columns, requested once, ahead of the plane
  f2 w[8];
  passA_twiddle_fetch<G>(w, tseq, twA);
  // ---- 1. the padded plane -> LDS (dword per lane, 256 contiguous bytes per wave-instruction; padding of both axes
  // through the index maps, positions past the padded extent read as zero)
  {
    const float* plane = a.src + ((size_t)img * a.SZ + zs) * a.SY * a.SX;
    const BufRsrc pr = make_rsrc(plane, (unsigned)(a.SY * a.SX * 4));
    const int xp = tid & 63;
    float val[16];
    if (a.mx.up == 1 && a.my.up == 1 && a.mx.mode == PAD_CONSTANT) {
      // zero padding, no spread (the usual case): a shift and a bounds test per element -- the general index map below is
      // a division and three mode branches per element, 17 times per thread: most of this kernel's instructions
      const int xs = x0 + xp - a.mx.pad;
      const bool xok = (unsigned)xs < (unsigned)a.SX;
      const unsigned base = (unsigned)((y0 + (tid >> 6) - a.my.pad) * a.SX + xs) * 4u, step = (unsigned)(4 * a.SX) * 4u;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int ys = y0 + (tid >> 6) + 4 * u - a.my.pad;
        val[u] = buf_load_f32(pr, (xok && (unsigned)ys < (unsigned)a.SY) ? base + step * u : 0xFFFFFFFFu, 0);
      }
    } else {
      const int xs = axis_src(a.mx, x0 + xp);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int yp = y0 + (tid >> 6) + 4 * u;
        const int ys = axis_src(a.my, yp);
        val[u] = buf_load_f32(pr, (ys >= 0 && xs >= 0) ? (unsigned)(ys * a.SX + xs) * 4u : 0xFFFFFFFFu, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) rowbuf[((tid >> 6) + 4 * u) * RP + xp] = val[u];
  }
  __syncthreads();
  // ---- 2. x rows: rows 2sq and 2sq+1 ride one complex transform at odd frequencies (nd_passes.hpp, rows_r2c)
  f2 v[8];
#pragma unroll
  for (int n1 = 0; n1 < 8; ++n1) {
    const int x = 8 * n1 + tseq;
    v[n1] = mk2(rowbuf[(2 * sq) * RP + x], rowbuf[(2 * sq + 1) * RP + x]);
  }
  {
    float ws, wc;
    sincospif(-(float)tseq / 64.0f, &ws, &wc);
    const f2 wt = mk2(wc, ws);
    static_for<0, 8>([&](auto ic) {
      constexpr int n1 = decltype(ic)::value;
      constexpr float c = (float)cospi_q(n1, 8), s = (float)sinpi_q(n1, 8);
      v[n1] = cmul(v[n1], cmul_const(wt, c, -s));
    });
  }
  fft_regs<8, -1>(v);
  __syncthreads();                                         // every wave has read its rows: the region changes hands
  f2* seq = reg + sq * kPlPitch;
  passA_twiddle_apply<G, -1>(v, w, seq, tseq);
  seq_sync<G>();
  passB_load<G>(v, seq, tseq);
  passB_compute<G, -1>(v, tseq, twB);                      // v[k] = bin tseq + 8k of the packed pair
  // unpack the two real rows' spectra at bins fx = tseq + 8k < 32; the partner bin 63 - fx is element 7 - k of lane
  // 7 - tseq of the same sequence: one DPP half-row mirror, no LDS
  f2 ev[4], ov[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f2 zf = v[k];
    const f2 zg = mk2(dpp_half_mirror(v[7 - k].x), dpp_half_mirror(v[7 - k].y));
    ev[k] = mk2(0.5f * (zf.x + zg.x), 0.5f * (zf.y - zg.y));
    ov[k] = mk2(0.5f * (zf.y + zg.y), 0.5f * (zg.x - zf.x));
  }
  __syncthreads();                                         // every x row is in registers
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    f4 q; q.x = ev[k].x; q.y = ev[k].y; q.z = ov[k].x; q.w = ov[k].y;
    *reinterpret_cast<f4*>(reg + (tseq + 8 * k) * kPlPitch + 2 * sq) = q;     // [fx][y]: rows 2sq, 2sq+1
  }
  __syncthreads();
  // ---- 3. y columns: sequence fx = sq, in place in its slot
#pragma unroll
  for (int n1 = 0; n1 < 8; ++n1) v[n1] = seq[8 * n1 + tseq];
  fft_regs<8, -1>(v);
  passA_twiddle_apply<G, -1>(v, w, seq, tseq);
  seq_sync<G>();
  passB_load<G>(v, seq, tseq);
  passB_compute<G, -1>(v, tseq, twB);                      // v[k] = bin fy = tseq + 8k of column fx = sq
  seq_sync<G>();                                           // (the slot is rewritten by its own eight lanes only)
#pragma unroll
  for (int k = 0; k < 8; ++k) seq[tseq + 8 * k] = v[k];
  __syncthreads();
  // ---- 4. the plane's spectrum leaves as one contiguous 16 KB run, 16 bytes per lane
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + kPlNT * u, fx = idx >> 5, wq = idx & 31;
    outp[idx] = *reinterpret_cast<const f4*>(reg + fx * kPlPitch + 2 * wq);
  }
}

// ------------------------------------------------------------------------------------------ planes_inv
struct PlaneInvArgs {
  const f2* src;         // O [(b*Cout + co)][NZo][2048]
  float* dst;            // (B, Cout, Zo, Yo, Xo)
  const float* bias;
  const f2* twA;
  const f2* twB;
  int NZo, Cout;
  int NVy, sy, Yo;       // valid stride-1 rows, decimation, output rows
  int NVx, sx, Xo;
  int nxt, nyt, Vx, Vy;  // overlap-save tiles of the plane (see PlaneFwdArgs): tile (yt, xt) yields the stride-1 samples
                         // [yt*Vy, yt*Vy + Vy) x [xt*Vx, xt*Vx + Vx) of the plane; src holds nyt*nxt blocks of 2048 columns per plane
  FastDiv d_nz, d_nt, d_nx;   // unit map of the launch (filled by the dispatcher): blockIdx = (img*NZo + zi)*ntile + tile, tile = yt*nxt + xt
};

template <int NT_>
__global__ __launch_bounds__(kPlNT) void planes_inv_kernel(const PlaneInvArgs a) {
  using G = Geo<8, 1>;
  __shared__ __attribute__((aligned(16))) f2 reg[32 * kPlPitch];     // one region, six changes of hands (see planes_fwd)
  constexpr int YP = 36;                                   // Yt[yo][fx] pitch (complex): row pairs land 64 bytes apart mod 256
  static_assert(64 * YP <= 32 * kPlPitch, "transposed rows fit the region");
  const BufRsrc twA = make_rsrc(a.twA, 8 * 8 * 8), twB = make_rsrc(a.twB, 8 * 8);
  const int tid = threadIdx.x, sq = tid >> 3, tseq = tid & 7;
  unsigned qq;
  const int tile = (int)fdivmod(blockIdx.x, a.d_nt, &qq);          // qq = img*NZo + zi
  const int img = (int)fdiv(qq, a.d_nz);
  const int yt = (int)fdiv((unsigned)tile, a.d_nx), xt = tile - yt * a.nxt;
  // this tile's window of stride-1 samples and of output samples: local sample n is sample g0 + n of the plane's axis and,
  // where that is a multiple of the stride, output (g0 + n)/s -- row / column (g0 + n)/s - o0 of this tile's output block
  const int gy0 = yt * a.Vy, gx0 = xt * a.Vx;
  const int NVy = min(a.Vy, a.NVy - gy0), NVx = min(a.Vx, a.NVx - gx0);
  // (stride 1 -- the usual case -- without the emulated divisions: they are uniform but run on the vector pipeline)
  int oy0 = gy0, ox0 = gx0, Yo = max(NVy, 0), Xo = max(NVx, 0);
  if (a.sy != 1) { oy0 = (gy0 + a.sy - 1) / a.sy; Yo = NVy > 0 ? (gy0 + NVy - 1) / a.sy - oy0 + 1 : 0; }
  if (a.sx != 1) { ox0 = (gx0 + a.sx - 1) / a.sx; Xo = NVx > 0 ? (gx0 + NVx - 1) / a.sx - ox0 + 1 : 0; }
  f2 w[8];
  passA_twiddle_fetch<G>(w, tseq, twA);                    // one batch for both inverse passes, ahead of the plane
  {
    const f4* inp = reinterpret_cast<const f4*>(a.src + (size_t)blockIdx.x * kPlCols);
    f4 q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) q[u] = inp[tid + kPlNT * u];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = tid + kPlNT * u, fx = idx >> 5, wq = idx & 31;
      *reinterpret_cast<f4*>(reg + fx * kPlPitch + 2 * wq) = q[u];
    }
  }
  float b = a.bias ? a.bias[img % a.Cout] : 0.f;
  __syncthreads();
  // ---- y columns back: sequence fx = sq; sample n = tseq + 8k
  f2 v[8];
  f2* seq = reg + sq * kPlPitch;
  nat_load<G>(v, seq, tseq);
  seq_sync<G>();
  fft_regs<8, +1>(v);
  passA_twiddle_apply<G, +1>(v, w, seq, tseq);
  seq_sync<G>();
  passB_load<G>(v, seq, tseq);
  passB_compute<G, +1>(v, tseq, twB);
  __syncthreads();                                         // every column is in registers
  if (a.sy == 1) {                                          // (no emulated division per sample)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int n = tseq + 8 * k;
      if (n < NVy) reg[n * YP + sq] = v[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int n = tseq + 8 * k, yo = (gy0 + n) / a.sy;
      if (n < NVy && yo * a.sy == gy0 + n) reg[(yo - oy0) * YP + sq] = v[k];
    }
  }
  __syncthreads();
  // ---- x rows back: output rows 2sq (-> real part) and 2sq+1 (-> imaginary part) share one inverse transform
  // (odd-frequency bins: V[f] = Ya[f] + i Yb[f], V[63-f] = conj(Ya[f]) + i conj(Yb[f]), f < 32; rows_c2r)
  const int ra = 2 * sq, rb = ra + 1;
  const bool has_a = ra < Yo, has_b = rb < Yo;
  static_for<0, 8>([&](auto ic) {
    constexpr int i1 = decltype(ic)::value;
    const int f = 8 * i1 + tseq, fs = i1 < 4 ? f : 63 - f;
    f2 ya = reg[ra * YP + fs], yb = reg[rb * YP + fs];
    ya = has_a ? ya : mk2(0.f, 0.f);
    yb = has_b ? yb : mk2(0.f, 0.f);
    v[i1] = i1 < 4 ? mk2(ya.x - yb.y, ya.y + yb.x) : mk2(ya.x + yb.y, yb.x - ya.y);
  });
  fft_regs<8, +1>(v);
  __syncthreads();                                         // every row pair has been read
  passA_twiddle_apply<G, +1>(v, w, seq, tseq);
  seq_sync<G>();
  passB_load<G>(v, seq, tseq);
  passB_compute<G, +1>(v, tseq, twB);                       // sample n = tseq + 8k
  {
    float ws, wc;
    sincospif((float)tseq / 64.0f, &ws, &wc);
    const f2 wb = mk2(wc, ws);
    static_for<0, 8>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
      constexpr float c = (float)cospi_q(k, 8), s = (float)sinpi_q(k, 8);
      v[k] = cmul(v[k], cmul_const(wb, c, s));
    });
  }
  __syncthreads();                                          // every exchange has been read: the region becomes the output plane
  float* ob = reinterpret_cast<float*>(reg);                // [Yo][Xo] of this tile
  if (a.sx == 1) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int n = tseq + 8 * k;
      if (n < NVx) {
        if (has_a) ob[ra * Xo + n] = v[k].x + b;
        if (has_b) ob[rb * Xo + n] = v[k].y + b;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int n = tseq + 8 * k, xo = (gx0 + n) / a.sx;
      if (n < NVx && xo * a.sx == gx0 + n) {
        if (has_a) ob[ra * Xo + xo - ox0] = v[k].x + b;
        if (has_b) ob[rb * Xo + xo - ox0] = v[k].y + b;
      }
    }
  }
  __syncthreads();
  float* op = a.dst + (size_t)qq * a.Yo * a.Xo;             // the (b, co, z_out) plane
  const int total = Yo * Xo;
  if (a.nxt * a.nyt == 1) {                                 // (the whole plane: one contiguous run)
    for (int idx = tid; idx < total; idx += kPlNT) op[idx] = ob[idx];
  } else {
    for (int idx = tid; idx < total; idx += kPlNT) {
      const int r = idx / Xo, c = idx - r * Xo;
      op[(size_t)(oy0 + r) * a.Xo + ox0 + c] = ob[idx];
    }
  }
}

// ------------------------------------------------------------------------------------------ colz
struct ColZArgs {
  const f2* src;         // S [(b*Cin + ci)][NZ][2048]
  const f4* wspec;       // [g][Cog_pad][Cig_pad/2 = 4][2048][64] f4 = {H(o,2ip), H(o,2ip+1)}
  f2* dst;               // O [(b*Cout + co)][NZo][2048]
  int B, Cin, Cout, G, Cig, Cog, Cog_pad, cob, n_ochunks;
  int NZ, NZo;           // planes per image on the signal / output side
  int V, ntiles, Lfull, stride;     // overlap-save tiles along z (tile t: padded planes [t*V, t*V + 64))
  unsigned long long* stamps;       // profiling build only: 8 timestamps (100 MHz) per workgroup
  int ncol;                         // bin columns per plane (a multiple of 16): kPlCols in 3-D, nxt * Tx/2 in 2-D
  int hcol;                         // columns of the kernel spectrum: ncol in 3-D, Tx/2 in 2-D (the x tiles share them)
  FastDiv d_nbp, d_ntiles, d_per, d_g, d_hcol;   // unit map of the launch (filled by the dispatcher; d_per.d = column blocks per XCD)
};

constexpr int colz_fc(int nb) { return nb >= 2 ? 16 : 8; }      // frequencies per exchange chunk
constexpr size_t colz_lds_bytes(int nb) { return (size_t)2 * nb * 8 * 16 * (colz_fc(nb) + 1) * sizeof(f2); }

// NB batch items per workgroup (1, 2 or 4) share every kernel-spectrum load.  A chunk of FC frequencies x 16 columns
// is mixed by NB*128 threads: R = NB*8/FC threads per bin, each owning 8/R output channels (with NB = 4 the two
// halves of the workgroup).  RING spectrum sets (one output channel x 8 inputs = 4 float4) travel per thread.
// DIAG (timestamp builds only, FFTCONV_COLZ_DIAG): 1 = the kernel-spectrum loads are replaced by register constants,
// 2 = no LDS exchange barriers' partner work (the mix arithmetic is skipped) -- what-bounds-the-mix experiments.
// NCOLC: bin columns per plane as a compile-time constant (the 3-D pipeline: kPlCols), 0 = a.ncol (the 2-D pipeline,
// where the "planes" are the rows of the half-spectrum and the columns its Tx/2 bins).
template <int NB, int RING, bool STAMPS = false, int DIAG = 0, int NCOLC = kPlCols>
__global__ __launch_bounds__(NB * 128, 2) void colz_kernel(const ColZArgs a) {
  const int ncol = NCOLC > 0 ? NCOLC : a.ncol;
  constexpr int NT = NB * 128;
  constexpr int FC = colz_fc(NB);
  constexpr int R = NT / (16 * FC);         // threads per bin
  constexpr int SPC = 8 / R;                // mix steps (output channels) per thread and chunk
  constexpr int NCH = 64 / FC;
  constexpr int PF = FC + 1;                // padded chunk rows: conflict-free for lanes over columns AND over frequencies
  constexpr int ROW = 16 * PF;              // complex slots per (batch slot, channel)
  static_assert(R * FC * 16 == NT && NCH * FC == 64 && SPC * R == 8, "bins, threads and output channels divide evenly");
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  f2* xs = lds;                             // [NB*8][16][PF] forward spectra of the chunk
  f2* ys = lds + NB * 8 * ROW;              // [NB*8][16][PF] mixed spectra of the chunk
  const int tid = threadIdx.x;
  auto stampc = [&](int slot, bool drain) {
    if constexpr (STAMPS) {
      if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) a.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
  };
  stampc(0, false);
  // XCD-aware unit map (see fusedc): workgroups that stream one block of the kernel spectrum sit on one XCD (equal
  // blockIdx % 8) and run back to back (the batch block is the fastest index behind it)
  //   id = (((((oc*G + g)*per + colblk_local)*ntiles + tile)*nbp + bb)*8 + xcd : the tiles of a column block follow
  //   each other on one XCD too (they share the spectrum block and the overlap rows of their inputs)
  const int xcd = blockIdx.x & 7;
  unsigned q;
  const int b0 = (int)fdivmod(blockIdx.x >> 3, a.d_nbp, &q) * NB;
  const int tile = (int)fdivmod(q, a.d_ntiles, &q);
  const int colblk = xcd * (int)a.d_per.d + (int)fdivmod(q, a.d_per, &q);
  const int g = (int)fdivmod(q, a.d_g, &q);
  const int oc = (int)q;
  const int col0 = colblk * 16;
  if (col0 >= ncol) return;                  // padding of the last XCD's range of column blocks (uniform per workgroup)
  const int nbc = min(NB, a.B - b0);
  const int t0 = tile * a.V;
  // (plane step of the 64 loads / stores as an opaque scalar: as 64 literal offsets hipcc keeps every one of them in
  // its own SGPR across the kernel and spills)
  unsigned zstep = ncol * 8;
  asm volatile("" : "+s"(zstep));

  // ---- sequence owner: (batch slot nb, channel ch, column c)
  const int c = tid & 15, ch = (tid >> 4) & 7, nb = tid >> 7;
  f2 v[64];
  // (the output resource is built here, while scalar registers are plentiful: built behind the transforms, whose
  // twiddle constants fill the SGPR file, hipcc assembled it in VECTOR registers and wrapped every store in a
  // readfirstlane loop)
  const int cend = min(a.cob, a.Cog - oc * a.cob);
  f2* obase = a.dst + ((size_t)b0 * a.Cout + (size_t)g * a.Cog + (size_t)oc * a.cob) * a.NZo * ncol;
  const BufRsrc orr = make_rsrc(obase, (unsigned)((((size_t)(nbc - 1) * a.Cout + cend) * a.NZo) * ncol * 8));
  {
    const f2* sbase = a.src + ((size_t)b0 * a.Cin + (size_t)g * a.Cig) * a.NZ * ncol;
    const BufRsrc sr = make_rsrc(sbase, (unsigned)((((size_t)(nbc - 1) * a.Cin + a.Cig) * a.NZ) * ncol * 8));
    const bool has_in = ch < a.Cig && nb < nbc;
    // (dead loads / stores: bit 31 of the offset -- every resource here is below 2 GiB, fc_api.cpp plan_nd -- which,
    // unlike an all-ones offset, cannot wrap back into range when the instruction's constant offset is added)
    const unsigned voff = has_in ? (unsigned)((((size_t)nb * a.Cin + ch) * a.NZ + t0) * ncol + col0 + c) * 8u : 0x80000000u;
    if (t0 + 64 <= a.NZ) {                                  // (the usual case without a select per load)
      static_for<0, 64>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        v[n] = buf_load_f32x2(sr, voff, n * zstep);
      });
    } else {
      static_for<0, 64>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        v[n] = buf_load_f32x2(sr, (t0 + n < a.NZ) ? voff : 0x80000000u, n * zstep);
      });
    }
  }
  // ---- bin owner: (share h of the output channels, frequency fzl of the chunk, column cm); its kernel-spectrum
  // stream does not depend on data
  const int h = tid / (16 * FC), fzl = tid % FC, cm = (tid / FC) % 16;
  // (the kernel spectrum has hcol columns: all of them in 3-D; in 2-D the Tx/2 columns of ONE x tile, which the x tiles of
  // the signal share -- a block of 16 columns lies inside one tile, its spectrum columns start at col0 % hcol)
  const int hcol = NCOLC > 0 ? NCOLC : a.hcol;
  const int hc0 = NCOLC > 0 ? col0 : (int)(col0 - (int)fdiv((unsigned)col0, a.d_hcol) * hcol);
  const size_t wrow = (size_t)hcol * 64;                 // f4 per (o, ip)
  const f4* wbase = a.wspec + ((size_t)g * a.Cog_pad + (size_t)oc * a.cob) * 4 * wrow;
  const BufRsrc wr_ = make_rsrc(wbase, (unsigned)((size_t)a.cob * 4 * wrow * 16));
  // (lane offset: column, frequency and this thread's first output channel h*SPC -- at most 64 MB)
  const unsigned wvo = (unsigned)((hc0 + cm) * 64 + fzl) * 16u + (unsigned)(h * SPC * 4) * (unsigned)(hcol * 64 * 16);
  f4 ring[RING][4];
  auto issue = [&](auto stc) {
    constexpr int st = decltype(stc)::value;
    constexpr int chunk = st / SPC, k = st % SPC;
    // (rows past the chunk's last output channel lie outside the spectrum buffer: the scalar offset is not part of
    // the range check, so those loads are switched off through the lane offset)
    const unsigned vo = h * SPC + k < a.cob ? wvo + (unsigned)(chunk * FC * 16) : 0x80000000u;
    if constexpr (DIAG == 1) {
#pragma unroll
      for (int ip = 0; ip < 4; ++ip) {
        f4 q; q.x = 1e-3f * (float)(st + ip); q.y = q.x; q.z = q.x; q.w = q.x;
        asm volatile("" : "+v"(q));
        ring[st % RING][ip] = q;
      }
      return;
    }
#pragma unroll
    for (int ip = 0; ip < 4; ++ip)
      ring[st % RING][ip] = buf_load_f32x4(wr_, vo, (unsigned)((k * 4 + ip) * (hcol * 64 * 16)));
  };
  // (the first sets travel during the forward transform, whose temporaries leave room for two of them)
  constexpr int EARLY = RING < 2 ? RING : 2;
  static_for<0, EARLY>([&](auto sc) { issue(sc); });
  stampc(1, true);
  fft_regs<64, -1>(v);                                      // 64-point forward transform along z, in registers
  static_for<EARLY, RING>([&](auto sc) { issue(sc); });
  stampc(2, false);

  static_for<0, NCH>([&](auto cc) {
    constexpr int chunk = decltype(cc)::value;
    {
      f2* xrow = xs + ((nb * 8 + ch) * 16 + c) * PF;
#pragma unroll
      for (int j = 0; j < FC; ++j) xrow[j] = v[chunk * FC + j];
    }
    __syncthreads();
    {
      const unsigned xa = lds_off(xs + cm * PF + fzl);
      f2* yrow = ys + cm * PF + fzl;
      // two batch items at a time: their inputs (16 reads in flight together) and two interleaved accumulation chains
      // (a chain of dependent v_pk_fma_f32 issues every 8 cycles, two of them every 4)
      constexpr int NP = NB >= 2 ? NB / 2 : 1, PB = NB >= 2 ? 2 : 1;
      f2 x[PB][8];
      auto load_x = [&](auto pc) {
        constexpr int pr = decltype(pc)::value;
        static_for<0, PB>([&](auto bc) {
          constexpr int bb = pr * PB + decltype(bc)::value;
          static_for<0, 8>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            x[decltype(bc)::value][i] = lds_rd_far<(bb * 8 + i) * ROW * 8>(xa);
          });
        });
#pragma unroll
        for (int q = 0; q < PB; ++q) lds_arrive(x[q]);
      };
      if constexpr (NP == 1) load_x(std::integral_constant<int, 0>{});      // held across the chunk's output channels
      static_for<0, SPC>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int st = chunk * SPC + k;
        f4 (&w)[4] = ring[st % RING];
        const int o = h * SPC + k;
        if (o < a.cob && DIAG != 2) {
          static_for<0, NP>([&](auto pc) {
            constexpr int pr = decltype(pc)::value;
            if constexpr (NP > 1) load_x(pc);
            // (2 * PB independent accumulation chains -- even / odd input channels of every batch item: a chain of
            // dependent v_pk_fma_f32 issues every 8 cycles, and the mix's arithmetic alone was 6.4 us of its 20)
            f2 y[PB], y2[PB];
#pragma unroll
            for (int q = 0; q < PB; ++q) { y[q] = mk2(0.f, 0.f); y2[q] = mk2(0.f, 0.f); }
#pragma unroll
            for (int ip = 0; ip < 4; ++ip) {
#pragma unroll
              for (int q = 0; q < PB; ++q) cmac(y[q], x[q][2 * ip], w[ip].xy);
#pragma unroll
              for (int q = 0; q < PB; ++q) cmac(y2[q], x[q][2 * ip + 1], w[ip].zw);
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) yrow[((pr * PB + q) * 8 + o) * ROW] = y[q] + y2[q];
          });
        }
        if constexpr (st + RING < NCH * SPC) issue(std::integral_constant<int, st + RING>{});
      });
    }
    __syncthreads();
    {
      const f2* yrow = ys + ((nb * 8 + ch) * 16 + c) * PF;
#pragma unroll
      for (int j = 0; j < FC; ++j) v[chunk * FC + j] = yrow[j];
    }
  });
  stampc(3, false);
  fft_regs<64, +1>(v);                                      // back along z (the 1/N factors live in the kernel spectrum)
  stampc(4, false);
  {
    const bool live = nb < nbc && ch < a.cob && oc * a.cob + ch < a.Cog;
    const int limit = min(a.V, a.Lfull - t0);
    const unsigned vo = live ? (unsigned)((((size_t)nb * a.Cout + ch) * a.NZo) * ncol + col0 + c) * 8u : 0x80000000u;
    if (a.stride == 1) {
      unsigned zstep_st = ncol * 8;               // (a copy of its own: shared with the loads, the 64 products stay live in SGPRs)
      asm volatile("" : "+s"(zstep_st));
      const unsigned vo1 = live ? vo + (unsigned)t0 * (ncol * 8) : 0x80000000u;
      // blocks of 8 planes: a block inside the valid window is straight-line code behind one scalar branch
      static_for<0, 8>([&](auto bc) {
        constexpr int n0 = 8 * decltype(bc)::value;
        if (n0 + 8 <= limit) {
          static_for<n0, n0 + 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            buf_store_f32x2(v[n], orr, vo1, n * zstep_st);
          });
        } else if (n0 < limit) {
          static_for<n0, n0 + 8>([&](auto nc) {
            constexpr int n = decltype(nc)::value;
            buf_store_f32x2(v[n], orr, n < limit ? vo1 : 0x80000000u, n * zstep_st);
          });
        }
      });
    } else {
      static_for<0, 64>([&](auto nc) {
        constexpr int n = decltype(nc)::value;
        const int t = t0 + n, idx = t / a.stride;
        const bool ok = live && n < limit && idx * a.stride == t;
        buf_store_f32x2(v[n], orr, ok ? vo + (unsigned)idx * (ncol * 8) : 0x80000000u, 0);
      });
    }
  }
  stampc(5, false);
  stampc(6, true);
  stampc(7, false);
}

// ------------------------------------------------------------------------------------------ colz, two threads per sequence
// The mix of colz_kernel is bound by the LATENCY of its kernel-spectrum stream: the 64-point register transforms hold 128
// VGPRs per thread, which leaves room for two spectrum sets in flight (64 KB per CU against 1 MB to stream at ~1.3 us per
// load: 21 us, profiles/r03_experiments.md block 1).  Here a sequence is split over the two lanes of a pair -- lane h owns
// the samples z = 2m + h, runs a 32-point register transform and meets its partner in ONE radix-2 stage through DPP (as the
// engine's S = 2 split) -- so a thread carries 32 points (64 VGPRs) and the ring grows to RING sets.  512 threads = 2 batch
// slots x 8 channels x 16 columns x 2 halves, one workgroup per CU; a chunk of 16 frequencies x 16 columns is mixed by two
// threads per bin (each 4 of the 8 output channels).  After the forward stage lane h holds the bins f + 32 h, which is also
// what the inverse stage wants; after the inverse lane h holds the samples 2m + h again: loads and stores mirror each other.
constexpr size_t colz2_lds_bytes() { return (size_t)2 * 16 * 16 * 17 * sizeof(f2); }

template <int RING, bool STAMPS = false>
__global__ __launch_bounds__(512, 2) void colz2_kernel(const ColZArgs a) {
  constexpr int NB = 2, NT = 512, FC = 16, NCH = 4, PF = FC + 1, ROW = 16 * PF, SPC = 4;
  extern __shared__ __attribute__((aligned(16))) f2 lds[];
  f2* xs = lds;                             // [NB*8][16][PF] forward spectra of the chunk
  f2* ys = lds + NB * 8 * ROW;              // [NB*8][16][PF] mixed spectra of the chunk
  const int tid = threadIdx.x;
  auto stampc = [&](int slot, bool drain) {
    if constexpr (STAMPS) {
      if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) a.stamps[(size_t)blockIdx.x * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
  };
  stampc(0, false);
  int id = blockIdx.x;
  const int xcd = id & 7; id >>= 3;
  const int nbp = (a.B + NB - 1) / NB;
  const int b0 = (id % nbp) * NB; id /= nbp;
  const int colblk = xcd * (kPlCols / 128) + id % (kPlCols / 128); id /= (kPlCols / 128);
  const int g = id % a.G; id /= a.G;
  const int oc = id % a.n_ochunks;
  const int tile = id / a.n_ochunks;
  const int col0 = colblk * 16;
  const int nbc = min(NB, a.B - b0);
  const int t0 = tile * a.V;
  unsigned zstep2 = 2 * kPlCols * 8;        // two planes (opaque scalar, see colz_kernel)
  asm volatile("" : "+s"(zstep2));

  // ---- sequence half owner: (batch slot nb, channel ch, column c, half h: samples z = 2m + h)
  const int h = tid & 1, c = (tid >> 1) & 15, ch = (tid >> 5) & 7, nb = tid >> 8;
  const float sgf = h ? -1.0f : 1.0f;
  const f2 sg = mk2(sgf, sgf);
  f2 v[32];
  const int cend = min(a.cob, a.Cog - oc * a.cob);
  f2* obase = a.dst + ((size_t)b0 * a.Cout + (size_t)g * a.Cog + (size_t)oc * a.cob) * a.NZo * kPlCols;
  const BufRsrc orr = make_rsrc(obase, (unsigned)((((size_t)(nbc - 1) * a.Cout + cend) * a.NZo) * kPlCols * 8));
  {
    const f2* sbase = a.src + ((size_t)b0 * a.Cin + (size_t)g * a.Cig) * a.NZ * kPlCols;
    const BufRsrc sr = make_rsrc(sbase, (unsigned)((((size_t)(nbc - 1) * a.Cin + a.Cig) * a.NZ) * kPlCols * 8));
    const bool has_in = ch < a.Cig && nb < nbc;
    const unsigned voff = has_in ? (unsigned)((((size_t)nb * a.Cin + ch) * a.NZ + t0 + h) * kPlCols + col0 + c) * 8u : 0x80000000u;
    if (t0 + 64 <= a.NZ) {
      static_for<0, 32>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        v[m] = buf_load_f32x2(sr, voff, m * zstep2);
      });
    } else {
      static_for<0, 32>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        v[m] = buf_load_f32x2(sr, (t0 + 2 * m + h < a.NZ) ? voff : 0x80000000u, m * zstep2);
      });
    }
  }
  // ---- bin owner: (share hh of the output channels, frequency fzl of the chunk, column cm)
  const int hh = tid >> 8, fzl = tid & 15, cm = (tid >> 4) & 15;
  const size_t wrow = (size_t)kPlCols * 64;
  const f4* wbase = a.wspec + ((size_t)g * a.Cog_pad + (size_t)oc * a.cob) * 4 * wrow;
  const BufRsrc wr_ = make_rsrc(wbase, (unsigned)((size_t)a.cob * 4 * wrow * 16));
  const unsigned wvo = (unsigned)((col0 + cm) * 64 + fzl) * 16u + (unsigned)(hh * SPC * 4) * (unsigned)(kPlCols * 64 * 16);
  f4 ring[RING][4];
  auto issue = [&](auto stc) {
    constexpr int st = decltype(stc)::value;
    constexpr int chunk = st / SPC, k = st % SPC;
    const unsigned vo = hh * SPC + k < a.cob ? wvo + (unsigned)(chunk * FC * 16) : 0x80000000u;
#pragma unroll
    for (int ip = 0; ip < 4; ++ip)
      ring[st % RING][ip] = buf_load_f32x4(wr_, vo, (unsigned)((k * 4 + ip) * (kPlCols * 64 * 16)));
  };
  constexpr int EARLY = RING < 3 ? RING : 3;                 // (the rest of the ring goes out behind the forward transform)
  static_for<0, EARLY>([&](auto sc) { issue(sc); });
  stampc(1, true);
  // ---- forward: 32-point transform of this lane's samples, then the radix-2 stage across the lane pair
  fft_regs<32, -1>(v);
  if (h) {
    static_for<1, 32>([&](auto fc_) {
      constexpr int f = decltype(fc_)::value;
      v[f] = cmul_const(v[f], cos64(f), -sin64(f));           // w_64^f
    });
  }
#pragma unroll
  for (int f = 0; f < 32; ++f) v[f] = pkfma(sg, v[f], dpp_xor1(v[f]));     // h = 0: E + t -> X[f]; h = 1: E - t -> X[f + 32]
  static_for<EARLY, RING>([&](auto sc) { issue(sc); });
  stampc(2, false);

  static_for<0, NCH>([&](auto cc) {
    constexpr int chunk = decltype(cc)::value;
    if (h == chunk / 2) {                 // the chunk's 16 bins live in the lanes of one half
      f2* xrow = xs + ((nb * 8 + ch) * 16 + c) * PF;
#pragma unroll
      for (int j = 0; j < FC; ++j) xrow[j] = v[(chunk % 2) * FC + j];
    }
    __syncthreads();
    {
      const unsigned xa = lds_off(xs + cm * PF + fzl);
      f2* yrow = ys + cm * PF + fzl;
      f2 x[NB][8];
      static_for<0, NB>([&](auto bc) {
        constexpr int bb = decltype(bc)::value;
        static_for<0, 8>([&](auto ic) {
          constexpr int i = decltype(ic)::value;
          x[bb][i] = lds_rd<(bb * 8 + i) * ROW * 8>(xa);
        });
      });
#pragma unroll
      for (int q = 0; q < NB; ++q) lds_arrive(x[q]);
      static_for<0, SPC>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr int st = chunk * SPC + k;
        f4 (&w)[4] = ring[st % RING];
        const int o = hh * SPC + k;
        if (o < a.cob) {
          f2 y[NB];
#pragma unroll
          for (int q = 0; q < NB; ++q) y[q] = mk2(0.f, 0.f);
#pragma unroll
          for (int ip = 0; ip < 4; ++ip) {
#pragma unroll
            for (int q = 0; q < NB; ++q) cmac(y[q], x[q][2 * ip], w[ip].xy);
#pragma unroll
            for (int q = 0; q < NB; ++q) cmac(y[q], x[q][2 * ip + 1], w[ip].zw);
          }
#pragma unroll
          for (int q = 0; q < NB; ++q) yrow[(q * 8 + o) * ROW] = y[q];
        }
        if constexpr (st + RING < NCH * SPC) issue(std::integral_constant<int, st + RING>{});
      });
    }
    __syncthreads();
    if (h == chunk / 2) {
      const f2* yrow = ys + ((nb * 8 + ch) * 16 + c) * PF;
#pragma unroll
      for (int j = 0; j < FC; ++j) v[(chunk % 2) * FC + j] = yrow[j];
    }
  });
  stampc(3, false);
  // ---- inverse: radix-2 stage across the pair, twiddle, 32-point transform -> samples z = 2m + h
#pragma unroll
  for (int f = 0; f < 32; ++f) v[f] = pkfma(sg, v[f], dpp_xor1(v[f]));     // h = 0: Y[f] + Y[f+32]; h = 1: Y[f] - Y[f+32]
  if (h) {
    static_for<1, 32>([&](auto fc_) {
      constexpr int f = decltype(fc_)::value;
      v[f] = cmul_const(v[f], cos64(f), sin64(f));            // w_64^(-f)
    });
  }
  fft_regs<32, +1>(v);
  stampc(4, false);
  {
    const bool live = nb < nbc && ch < a.cob && oc * a.cob + ch < a.Cog;
    const int limit = min(a.V, a.Lfull - t0);
    const unsigned vo = live ? (unsigned)((((size_t)nb * a.Cout + ch) * a.NZo) * kPlCols + col0 + c) * 8u : 0x80000000u;
    if (a.stride == 1) {
      unsigned zstep_st = 2 * kPlCols * 8;
      asm volatile("" : "+s"(zstep_st));
      const unsigned vo1 = live ? vo + (unsigned)(t0 + h) * (kPlCols * 8) : 0x80000000u;
      // blocks of 8 sample pairs: a block inside the valid window is straight-line code behind one scalar branch
      static_for<0, 4>([&](auto bc) {
        constexpr int m0 = 8 * decltype(bc)::value;
        if (2 * (m0 + 8) <= limit) {
          static_for<m0, m0 + 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            buf_store_f32x2(v[m], orr, vo1, m * zstep_st);
          });
        } else if (2 * m0 < limit) {
          static_for<m0, m0 + 8>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            buf_store_f32x2(v[m], orr, 2 * m + h < limit ? vo1 : 0x80000000u, m * zstep_st);
          });
        }
      });
    } else {
      static_for<0, 32>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
        const int n = 2 * m + h, t = t0 + n, idx = t / a.stride;
        const bool ok = live && n < limit && idx * a.stride == t;
        buf_store_f32x2(v[m], orr, ok ? vo + (unsigned)idx * (kPlCols * 8) : 0x80000000u, 0);
      });
    }
  }
  stampc(5, false);
  stampc(6, true);
  stampc(7, false);
}

}  // namespace fc
