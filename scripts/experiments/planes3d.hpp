// planes3d.hpp -- fused (y, x) plane passes of the 3-D path for planes that fit one 64 x 64 FFT tile.
//
// The separable 3-D scheme (nd_passes.hpp) runs five kernels: rows_r2c (x), c2c_fwd (y), fusedc (z + channel mix),
// c2c_inv (y), rows_c2r (x), and every arrow between them is a round trip through HBM.  When the padded (y, x) plane
// of the problem fits a 64 x 64 transform (cfgC of BASELINE.json: 64^3, k = 9) a whole plane's half-spectrum is
// 33 x 64 complex = 16.9 KB, so both plane axes are transformed inside one workgroup:
//
//   planes_fwd   x rows (two real rows per complex FFT) -> unpack + transpose in LDS -> y columns; the column pass
//                leaves element k of lane tseq = bin fy = tseq + 8k in REGISTERS, so a thread that walks NZ
//                consecutive planes ends up holding NZ consecutive z samples of its 8 bins and stores them as
//                8*NZ-byte runs of the z-contiguous layout the z pass (fusedc) reads:  S2[(b,ci)][fx*64 + fy][zp]
//   planes_inv   the mirror image: loads NZ z samples of its 8 bins (exactly the inputs of its inverse column
//                pass A'), inverse y columns -> LDS -> inverse x rows (c2r) with the valid window, stride and bias.
//
// Five launches become three and the S1 / O1 intermediates (2 x 2 x 70 MB at cfgC) disappear.  The kernel
// transform keeps the separable passes (it runs once per weight version).
#pragma once
#include "nd_passes.hpp"

namespace fc {

struct PlanesArgs {
  // forward
  const float* src;      // signal (B, C, Z, Y, X)
  f2* dst;               // S2 [(b,ci)][fx*64 + fy][NZP]
  AxisMap mx, my, mz;    // per-axis padding maps
  int SZ, SY, SX;        // source extents
  unsigned src_bytes;    // size of the source tensor (< 4 GiB: the planner only takes this path then)
  int NZP;               // z extent of the spectrum rows (padded planes)
  // inverse
  const f2* isrc;        // O2 [(b,co)][fx*64 + fy][Lzo]
  float* out;            // (B, Cout, Zo, Yo, Xo)
  const float* bias;
  int Lzo, Cout;         // planes of the output (already valid + decimated along z), channels
  int NVy, sy, Yo;       // valid stride-1 rows, decimation, output rows
  int NVx, sx, Xo;       // the same along x
  // both
  const f2* twA;         // 64-point tile: [8][8]
  const f2* twB;
  int NA;                // images
  int nz;                // planes per workgroup (8 or 4)
  unsigned long long* stamps;   // optional profiling hook (ubench_planes): 64 timestamps per workgroup, null = off
};

// 64-point forward / inverse transform of one sequence with the pass-A twiddles already in registers (they are the
// same for the x rows, the y columns and every plane of a workgroup: fetched once); the result stays in registers:
// element k of lane tseq is bin (or sample) tseq + 8k.
template <int DIR>
__device__ __forceinline__ void plane_fft64(f2 (&v)[8], const f2 (&w)[8], f2* ls, int tseq, BufRsrc twB) {
  using G = Geo<8, 1>;
  fft_regs<8, DIR>(v);
  passA_twiddle_apply<G, DIR>(v, w, ls, tseq);
  seq_sync<G>();
  passB_load<G>(v, ls, tseq);
  passB_compute<G, DIR>(v, tseq, twB);
}

constexpr int kPlaneNT = 320;          // 5 waves: 33 column sequences x 8 threads = 264 active in the column pass
constexpr int kPlaneCols = 33;         // Tx/2 + 1
constexpr int kPlanePitch = 66;        // plane buffer: [fx][y], rows padded to 66 complex

template <int NZ>
__global__ __launch_bounds__(kPlaneNT) void planes_fwd_kernel(const PlanesArgs a) {
  using G = Geo<8, 1>;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  __shared__ __attribute__((aligned(16))) f2 lseq[kPlaneCols * LSEQP];
  __shared__ __attribute__((aligned(16))) f2 plane[kPlaneCols * kPlanePitch];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(8 * 8 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(8 * 8));
  const int tid = threadIdx.x, sq = tid >> 3, tseq = tid & 7;
  const int nzb = (a.NZP + NZ - 1) / NZ;
  const int img = blockIdx.x / nzb, z0 = (blockIdx.x % nzb) * NZ;
  f2 keep[NZ][8];                       // indexed with compile-time zl only (the plane loop is written out)
  // Row samples of one plane -> registers, as unconditional buffer loads: a sample outside the source (zero padding,
  // rows / planes beyond the padded extent) gets an out-of-range offset and reads as zero.  (A per-element
  // "load or zero" select makes hipcc branch around every load and wait for each one: 16 dependent round trips per
  // plane, measured 97 us for this kernel at cfgC.)  The next plane's samples are requested before this plane's
  // transforms start.
  const BufRsrc sg = make_rsrc(a.src, a.src_bytes);
  auto load_plane = [&](int zp, f2 (&v)[8]) {
    const int zs = zp < a.NZP ? axis_src(a.mz, zp) : -1;
    unsigned ro[2];
    bool ok[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ys = axis_src(a.my, 2 * sq + h);
      ok[h] = zs >= 0 && ys >= 0;
      ro[h] = (unsigned)((((size_t)img * a.SZ + (ok[h] ? zs : 0)) * a.SY + (ok[h] ? ys : 0)) * a.SX * 4);
    }
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {
      const int xs = axis_src(a.mx, 8 * n1 + tseq);
      v[n1].x = buf_load_f32(sg, (ok[0] && xs >= 0) ? ro[0] + (unsigned)xs * 4u : 0xFFFFFFFFu, 0);
      v[n1].y = buf_load_f32(sg, (ok[1] && xs >= 0) ? ro[1] + (unsigned)xs * 4u : 0xFFFFFFFFu, 0);
    }
  };
  auto stamp = [&](int slot) {
    if (a.stamps != nullptr && tid == 0) a.stamps[(size_t)blockIdx.x * 64 + slot] = __builtin_amdgcn_s_memrealtime();
  };
  stamp(0);
  f2 vr[2][8];
  if (tid < 256) load_plane(z0, vr[0]);
  f2 wtw[8];
  passA_twiddle_fetch<G>(wtw, tseq, twA);
  static_for<0, NZ>([&](auto zc) {
    constexpr int zl = decltype(zc)::value;
    // ---- x rows: sequence sq = padded rows 2*sq, 2*sq+1
    if (tid < 256) {
      if constexpr (zl + 1 < NZ) load_plane(z0 + zl + 1, vr[(zl + 1) & 1]);
      f2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = vr[zl & 1][k];
      f2* ls = lseq + sq * LSEQP;
      plane_fft64<-1>(v, wtw, ls, tseq, twB);
      seq_sync<G>();                      // (the row loads above are done before the natural-order stores land)
#pragma unroll
      for (int k = 0; k < 8; ++k) ls[G::nat(tseq + 8 * k)] = v[k];
    }
    stamp(1 + 4 * zl);
    __syncthreads();
    stamp(2 + 4 * zl);
    // ---- unpack the two real rows of every sequence, transposed into plane[fx][y]
    for (int idx = tid; idx < kPlaneCols * 64; idx += kPlaneNT) {
      const int y = idx & 63, fx = idx >> 6;
      const f2* z = lseq + (y >> 1) * LSEQP;
      const f2 zf = z[G::nat(fx)], zg = z[G::nat((64 - fx) & 63)];
      plane[fx * kPlanePitch + y] = (y & 1) ? mk2(0.5f * (zf.y + zg.y), 0.5f * (zg.x - zf.x))
                                            : mk2(0.5f * (zf.x + zg.x), 0.5f * (zf.y - zg.y));
    }
    __syncthreads();
    stamp(3 + 4 * zl);
    // ---- y columns: sequence sq = bin column fx; the result stays in registers (bin fy = tseq + 8k)
    if (tid < kPlaneCols * 8) {
      f2 v[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) v[n1] = plane[sq * kPlanePitch + 8 * n1 + tseq];
      plane_fft64<-1>(v, wtw, lseq + sq * LSEQP, tseq, twB);
#pragma unroll
      for (int k = 0; k < 8; ++k) keep[zl][k] = v[k];
    }
    stamp(4 + 4 * zl);
    __syncthreads();
  });
  stamp(40);
  if (tid < kPlaneCols * 8) {
    f2* base = a.dst + ((size_t)img * (kPlaneCols * 64) + (size_t)sq * 64) * a.NZP + z0;
    const int nz = min(NZ, a.NZP - z0);
    if (nz == NZ) {                      // whole block: NZ consecutive complex values per bin, mergeable into 16-byte stores
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f2* p = base + (size_t)(tseq + 8 * k) * a.NZP;
#pragma unroll
        for (int zl = 0; zl < NZ; ++zl) p[zl] = keep[zl][k];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        f2* p = base + (size_t)(tseq + 8 * k) * a.NZP;
#pragma unroll
        for (int zl = 0; zl < NZ; ++zl)
          if (zl < nz) p[zl] = keep[zl][k];
      }
    }
  }
  if (a.stamps != nullptr) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(41); }
}

template <int NZ>
__global__ __launch_bounds__(kPlaneNT) void planes_inv_kernel(const PlanesArgs a) {
  using G = Geo<8, 1>;
  constexpr int LSEQP = SeqLayout<G>::LSEQP;
  __shared__ __attribute__((aligned(16))) f2 lseq[kPlaneCols * LSEQP];
  __shared__ __attribute__((aligned(16))) f2 plane[kPlaneCols * kPlanePitch];
  const BufRsrc twA = make_rsrc(a.twA, (unsigned)(8 * 8 * 8));
  const BufRsrc twB = make_rsrc(a.twB, (unsigned)(8 * 8));
  const int tid = threadIdx.x, sq = tid >> 3, tseq = tid & 7;
  const int nzb = (a.Lzo + NZ - 1) / NZ;
  const int img = blockIdx.x / nzb, z0 = (blockIdx.x % nzb) * NZ;
  const int nz = min(NZ, a.Lzo - z0);
  f2 keep[NZ][8];
  if (tid < kPlaneCols * 8) {
    const f2* base = a.isrc + ((size_t)img * (kPlaneCols * 64) + (size_t)sq * 64) * a.Lzo + z0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const f2* p = base + (size_t)(tseq + 8 * k) * a.Lzo;
#pragma unroll
      for (int zl = 0; zl < NZ; ++zl) keep[zl][k] = p[min(zl, nz - 1)];     // unconditional loads (planes >= nz are never used)
    }
  }
  const float b = a.bias ? a.bias[img % a.Cout] : 0.f;
  f2 wtw[8];
  passA_twiddle_fetch<G>(wtw, tseq, twA);
  static_for<0, NZ>([&](auto zc) {
    constexpr int zl = decltype(zc)::value;
    if (zl >= nz) return;               // (uniform over the workgroup: the barriers below stay matched)
    // ---- inverse y columns: the 8 bins fy = tseq + 8*i1 of column sq are exactly the inputs of pass A'
    if (tid < kPlaneCols * 8) {
      f2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = keep[zl][k];
      plane_fft64<+1>(v, wtw, lseq + sq * LSEQP, tseq, twB);
#pragma unroll
      for (int k = 0; k < 8; ++k) plane[sq * kPlanePitch + tseq + 8 * k] = v[k];     // sample y = tseq + 8k of bin column sq
    }
    __syncthreads();
    // ---- inverse x rows: stride-1 rows 2*sq (-> real part) and 2*sq+1 (-> imaginary part) share one complex FFT:
    //      V[f] = Ya[f] + i*Yb[f],  V[64-f] = conj(Ya[f]) + i*conj(Yb[f])
    if (tid < 256) {
      const int ya = 2 * sq;
      f2 v[8];
#pragma unroll
      for (int i1 = 0; i1 < 8; ++i1) {
        const int f = 8 * i1 + tseq;
        const int fx = f <= 32 ? f : 64 - f;
        const f2 pa = plane[fx * kPlanePitch + ya], pb = plane[fx * kPlanePitch + ya + 1];
        if (f == 0 || f == 32) v[i1] = mk2(pa.x, pb.x);
        else if (f < 32) v[i1] = mk2(pa.x - pb.y, pa.y + pb.x);
        else v[i1] = mk2(pa.x + pb.y, pb.x - pa.y);
      }
      plane_fft64<+1>(v, wtw, lseq + sq * LSEQP, tseq, twB);
      // element k = sample x = tseq + 8k; rows ya / ya+1 are kept when valid and on the stride grid
      const int ia = ya / a.sy, ib = (ya + 1) / a.sy;
      const bool ha = ya < a.NVy && ia * a.sy == ya, hb = ya + 1 < a.NVy && ib * a.sy == ya + 1;
      float* oa = a.out + (((size_t)img * a.Lzo + z0 + zl) * a.Yo + ia) * a.Xo;
      float* ob = a.out + (((size_t)img * a.Lzo + z0 + zl) * a.Yo + ib) * a.Xo;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int n = tseq + 8 * k;
        const int ix = n / a.sx;
        if (n < a.NVx && ix * a.sx == n) {
          if (ha) oa[ix] = v[k].x + b;
          if (hb) ob[ix] = v[k].y + b;
        }
      }
    }
    __syncthreads();
  });
}

}  // namespace fc
