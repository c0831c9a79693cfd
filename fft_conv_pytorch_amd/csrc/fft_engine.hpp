// fft_engine.hpp -- workgroup-level complex FFT engine for gfx950 (MI355X, wave64).
//
// A sequence of T = P*P*S complex points lives in LDS and is transformed by
// TS = P*S threads that hold exactly P points each, in two register passes
// ("four-step" FFT, T = N1*N2 with N1 = P, N2 = P*S):
//
//   pass A  thread n2 owns the N1-point sub-FFT over n1 of x[N2*n1 + n2]
//           (all in VGPRs), multiplies by w_T^(n2*k1) and writes A[k1][n2]
//           into LDS rows padded to RS = N2+S complex (conflict-free b64).
//   pass B  the N2-point sub-FFT of row k1 is spread over S neighbouring
//           lanes: lane r takes the decimated inputs a[S*m + r], runs a
//           P-point register FFT, applies w_N2^(r*k) and finishes with a
//           radix-S butterfly across the lane quad through DPP (no LDS).
//           Lane (k1, j) then owns X[k1 + P*(k + P*j)], k = 0..P-1.
//
// Register FFTs are radix-2 DIT with compile-time twiddles; non-trivial
// butterflies use the 6-FMA form  t = a + w*b ; u = 2a - t.
// No vendor FFT library is used anywhere.
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

// cos(2*pi*q/64), q = 0..16 (correctly rounded from float64)
__device__ constexpr float kCos64[17] = {
    1.0f,                 0.99518472667219693f, 0.98078528040323043f, 0.95694033573220882f,
    0.92387953251128674f, 0.88192126434835505f, 0.83146961230254524f, 0.77301045336273699f,
    0.70710678118654757f, 0.63439328416364549f, 0.55557023301960229f, 0.47139673682599770f,
    0.38268343236508984f, 0.29028467725446239f, 0.19509032201612833f, 0.09801714032956077f,
    0.0f};

__host__ __device__ constexpr float cos64(int q) {
  q &= 63;
  if (q > 32) q = 64 - q;
  return (q > 16) ? -kCos64[32 - q] : kCos64[q];
}
__host__ __device__ constexpr float sin64(int q) { return cos64(q - 16); }

// cos(2*pi*q/128) for odd q = 1,3,..,31 (the 128th roots that are not 64th roots)
__device__ constexpr float kCos128odd[16] = {
    0.99879545620517241f, 0.98917650996478101f, 0.97003125319454397f, 0.94154406518302081f,
    0.90398929312344334f, 0.85772861000027212f, 0.80320753148064494f, 0.74095112535495911f,
    0.67155895484701833f, 0.59569930449243336f, 0.51410274419322166f, 0.42755509343028208f,
    0.33688985339222005f, 0.24298017990326387f, 0.14673047445536175f, 0.04906767432741801f};
__host__ __device__ constexpr float cos128(int q) {
  q &= 127;
  if (q > 64) q = 128 - q;
  if ((q & 1) == 0) return cos64(q >> 1);
  return (q > 32) ? -kCos128odd[(64 - q) >> 1] : kCos128odd[q >> 1];
}
__host__ __device__ constexpr float sin128(int q) { return cos128(q - 32); }

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
__host__ __device__ constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

// one radix-2 DIT butterfly with twiddle w = exp(DIR*2*pi*i*q/64); q is a
// compile-time constant after unrolling, so the branches fold away.
template <int DIR>
__device__ __forceinline__ void bfly(float& ar, float& ai, float& br, float& bi, int q) {
  if (q == 0) {
    const float tr = ar - br, ti = ai - bi;
    ar += br; ai += bi; br = tr; bi = ti;
  } else if (q == 16) {            // w = DIR*i :  w*b = (-DIR*bi, DIR*br)
    const float wr = (DIR > 0) ? -bi : bi;
    const float wi = (DIR > 0) ? br : -br;
    const float tr = ar - wr, ti = ai - wi;
    ar += wr; ai += wi; br = tr; bi = ti;
  } else {
    const float c = cos64(q), s = (DIR > 0) ? sin64(q) : -sin64(q);
    const float tr = fmaf(c, br, fmaf(-s, bi, ar));
    const float ti = fmaf(c, bi, fmaf(s, br, ai));
    br = fmaf(2.0f, ar, -tr);
    bi = fmaf(2.0f, ai, -ti);
    ar = tr; ai = ti;
  }
}

// P-point FFT on registers, natural order in, natural order out.
template <int P, int DIR>
__device__ __forceinline__ void fft_regs(float (&re)[P], float (&im)[P]) {
  constexpr int LG = ilog2(P);
  float tr[P], ti[P];
#pragma unroll
  for (int i = 0; i < P; ++i) { tr[bitrev(i, LG)] = re[i]; ti[bitrev(i, LG)] = im[i]; }
#pragma unroll
  for (int len = 2; len <= P; len <<= 1) {
#pragma unroll
    for (int blk = 0; blk < P; blk += len) {
#pragma unroll
      for (int j = 0; j < len / 2; ++j)
        bfly<DIR>(tr[blk + j], ti[blk + j], tr[blk + j + len / 2], ti[blk + j + len / 2], j * (64 / len));
    }
  }
#pragma unroll
  for (int i = 0; i < P; ++i) { re[i] = tr[i]; im[i] = ti[i]; }
}

// ---- buffer (SRSRC) loads: one VGPR byte offset + SGPR/constant offset per load ----
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;
// build from wave-uniform values only (kernel arguments / blockIdx arithmetic)
__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(BufRsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ float2 buf_load_f32x2(BufRsrc r, unsigned voff, unsigned soff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ float4 buf_load_f32x4(BufRsrc r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// ---- cross-lane helpers (DPP quad permutes: VALU rate, no LDS) --------------
__device__ __forceinline__ float dpp_xor1(float v) {   // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {   // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

template <int P_, int S_>
struct Geo {
  static constexpr int P = P_, S = S_;
  static constexpr int N1 = P, N2 = P * S, T = P * P * S;
  static constexpr int TS = N2;                       // threads per sequence
  static constexpr int RS = N2 + S;                   // padded row stride (complex)
  static constexpr int NATPAD = (S > 1) ? 32 / S : 0; // natural layout: f + NATPAD*(f/(P*P))
  static constexpr int LSEQ = N1 * RS;                // complex slots per sequence (>= natural size)
  static constexpr int LGS = ilog2(S);
  __device__ static constexpr int nat(int f) { return f + NATPAD * (f / (P * P)); }
};

// Multiply by the pass-A twiddles and write A[k1][n2] (padded rows).
//   twA[k1*N2 + n2] = exp(-2*pi*i*n2*k1/T)  (forward sign; conjugated for DIR=+1)
template <class G, int DIR>
__device__ __forceinline__ void passA_twiddle_store(float (&re)[G::P], float (&im)[G::P], float2* __restrict__ lseq,
                                                    int n2, BufRsrc twA) {
  lseq[n2] = make_float2(re[0], im[0]);
  const unsigned off = (unsigned)n2 * 8u;
#pragma unroll
  for (int k1 = 1; k1 < G::P; ++k1) {
    const float2 w = buf_load_f32x2(twA, off, k1 * G::N2 * 8);
    const float c = w.x, s = (DIR > 0) ? -w.y : w.y;
    const float xr = re[k1], xi = im[k1];
    lseq[k1 * G::RS + n2] = make_float2(fmaf(c, xr, -s * xi), fmaf(c, xi, s * xr));
  }
}

// Pass-B load: lane (k1, r) of the sequence reads a[S*m + r] of row k1.
template <class G>
__device__ __forceinline__ void passB_load(float (&re)[G::P], float (&im)[G::P], const float2* __restrict__ lseq, int tseq) {
  const int k1 = tseq >> G::LGS, r = tseq & (G::S - 1);
  const float2* row = lseq + k1 * G::RS + r;
#pragma unroll
  for (int m = 0; m < G::P; ++m) {
    const float2 v = row[G::S * m];
    re[m] = v.x; im[m] = v.y;
  }
}

// Pass-B compute: register FFT + lane-split finish.  Returns j such that element
// k of this lane is X[k1 + P*(k + P*j)]   (k1 = tseq >> log2(S)).
//   twB[r*P + k] = exp(-2*pi*i*r*k/N2)  (forward sign), only used when S > 1.
template <class G, int DIR>
__device__ __forceinline__ int passB_compute(float (&re)[G::P], float (&im)[G::P], int tseq, BufRsrc twB) {
  fft_regs<G::P, DIR>(re, im);
  if constexpr (G::S == 1) {
    return 0;
  } else {
    const int r = tseq & (G::S - 1);
    // lane twiddle w_N2^(r*k): compile-time roots of unity (N2 <= 128), chosen per lane
    static_assert(G::N2 <= 128, "lane-split twiddles come from the 128th-root table");
    (void)twB;
    if (r != 0) {
#pragma unroll
      for (int k = 1; k < G::P; ++k) {
        constexpr int STEP = 128 / G::N2;
        float c = cos128(STEP * k), s = -sin128(STEP * k);
        if constexpr (G::S == 4) {
          const float c2 = cos128(2 * STEP * k), s2 = -sin128(2 * STEP * k);
          const float c3 = cos128(3 * STEP * k), s3 = -sin128(3 * STEP * k);
          c = (r == 1) ? c : ((r == 2) ? c2 : c3);
          s = (r == 1) ? s : ((r == 2) ? s2 : s3);
        }
        if (DIR > 0) s = -s;
        const float xr = re[k], xi = im[k];
        re[k] = fmaf(c, xr, -s * xi);
        im[k] = fmaf(c, xi, s * xr);
      }
    }
    if constexpr (G::S == 2) {
      const float sg = r ? -1.0f : 1.0f;
#pragma unroll
      for (int k = 0; k < G::P; ++k) {
        re[k] = fmaf(sg, re[k], dpp_xor1(re[k]));
        im[k] = fmaf(sg, im[k], dpp_xor1(im[k]));
      }
      return r;
    } else {  // S == 4 : radix-4 across the quad, output order bit-reversed
      const float s2 = (r & 2) ? -1.0f : 1.0f;
      const float s1 = (r & 1) ? -1.0f : 1.0f;
      const bool rot = (r == 3);
#pragma unroll
      for (int k = 0; k < G::P; ++k) {
        float ar = fmaf(s2, re[k], dpp_xor2(re[k]));
        float ai = fmaf(s2, im[k], dpp_xor2(im[k]));
        // lane 3 holds (E1 - E3): multiply by w4 = DIR*i
        const float rr = (DIR > 0) ? -ai : ai;
        const float ri = (DIR > 0) ? ar : -ar;
        ar = rot ? rr : ar;
        ai = rot ? ri : ai;
        re[k] = fmaf(s1, ar, dpp_xor1(ar));
        im[k] = fmaf(s1, ai, dpp_xor1(ai));
      }
      return ((r & 1) << 1) | (r >> 1);
    }
  }
}

}  // namespace fc
