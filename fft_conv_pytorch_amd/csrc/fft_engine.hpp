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
#include <type_traits>

#ifndef FC_DIAG
#define FC_DIAG 0     // diagnostic builds (scripts/experiments/exp_cfgA_diag.sh): 2 = no LDS stores in the transform passes of the
#endif                // batch-sharing kernel, 3 = no twiddle-table reads; timing only, the results are wrong

namespace fc {

// One complex value = one 64-bit VGPR pair.  gfx950 issues a wave64 VALU instruction in 4 cycles
// whether it is v_fma_f32 or v_pk_fma_f32, so all complex arithmetic is written on packed
// (re, im) pairs: a twiddle butterfly is 3 v_pk_fma_f32, a complex MAC is 2.  Where hipcc does not
// fold the half-swaps / sign flips into op_sel / neg modifiers by itself, small asm helpers do.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
// a + (-i)*b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ f2 add_mi(f2 a, f2 b) {
  f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// a + (+i)*b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ f2 add_pi(f2 a, f2 b) {
  f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// a + conj(b) = (a.x + b.x, a.y - b.y)
__device__ __forceinline__ f2 add_conj(f2 a, f2 b) {
  f2 d; asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// (a - conj(b)) / i = (a.y + b.y, b.x - a.x)
__device__ __forceinline__ f2 sub_conj_divi(f2 a, f2 b) {
  f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// conj(a) + i*conj(b) = (a.x + b.y, b.x - a.y)
__device__ __forceinline__ f2 conj_add_iconj(f2 a, f2 b) {
  f2 d; asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b)); return d;
}
// x * w and x * conj(w), w = (cos, sin) in registers
__device__ __forceinline__ f2 cmul(f2 x, f2 w) {
  f2 t;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=&v"(t) : "v"(w), "v"(x));
  return t;
}
__device__ __forceinline__ f2 cmulc(f2 x, f2 w) {
  f2 t;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_hi:[1,0,0]" : "=&v"(t) : "v"(w), "v"(x));
  return t;
}
// y += x * h  (complex multiply-accumulate, 2 instructions in ONE asm statement: between two asm statements that
// depend on each other hipcc puts an s_nop -- 250 of them in a cfgA mix of 576 instructions)
__device__ __forceinline__ void cmac(f2& y, f2 x, f2 h) {
  asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "+v"(y) : "v"(x), "v"(h));
}
// ya += xa * ha + xb * hb ; yb += xa * ga + xb * gb, the two accumulation chains interleaved: a dependent v_pk_fma_f32
// issues every 8 cycles, two independent ones every 4 -- written as consecutive cmac() calls the compiler kept every
// chain back to back (the mix of one work item: ~4k instead of ~2k cycles of a wave's own time)
__device__ __forceinline__ void cmac2x2(f2& ya, f2& yb, f2 xa, f2 xb, f2 ha, f2 hb, f2 ga, f2 gb) {
  asm("v_pk_fma_f32 %0, %2, %4, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %2, %6, %1 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %2, %4, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %2, %6, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %0, %3, %5, %0 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %1, %3, %7, %1 op_sel_hi:[0,1,1]\n\t"
      "v_pk_fma_f32 %0, %3, %5, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
      "v_pk_fma_f32 %1, %3, %7, %1 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "+v"(ya), "+v"(yb) : "v"(xa), "v"(xb), "v"(ha), "v"(hb), "v"(ga), "v"(gb));
}
// x * (c + i*s) with compile-time c, s (hipcc folds the modifiers itself)
__device__ __forceinline__ f2 cmul_const(f2 x, float c, float s) {
  f2 t = mk2(c, c) * x;
  return pkfma(mk2(-s, s), x.yx, t);
}

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

// cos(pi*num/den), sin(pi*num/den) at compile time (argument folded into [0, pi/2], 12 Taylor terms in double)
__host__ __device__ constexpr double cospi_q(long long num, long long den) {
  num %= 2 * den;
  if (num < 0) num += 2 * den;
  if (num > den) num = 2 * den - num;
  bool neg = false;
  if (2 * num > den) { num = den - num; neg = true; }
  const double a = 3.14159265358979323846 * (double)num / (double)den;
  double term = 1.0, sum = 1.0;
  for (int k = 1; k <= 12; ++k) { term *= -a * a / ((2.0 * k - 1.0) * (2.0 * k)); sum += term; }
  return neg ? -sum : sum;
}
__host__ __device__ constexpr double sinpi_q(long long num, long long den) { return cospi_q(2 * num - den, 2 * den); }

__host__ __device__ constexpr int ilog2(int v) { return v <= 1 ? 0 : 1 + ilog2(v >> 1); }
__host__ __device__ constexpr int bitrev(int v, int bits) {
  int r = 0;
  for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1) << (bits - 1 - i);
  return r;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// one radix-2 DIT butterfly with twiddle w = exp(DIR*2*pi*i*q/64); q is a
// compile-time constant after unrolling, so the branches fold away.
template <int DIR>
__device__ __forceinline__ void bfly(f2& a, f2& b, int q) {
  if (q == 0) {
    const f2 t = a - b;
    a = a + b; b = t;
  } else if (q == 16) {            // w = DIR*i
    const f2 t = (DIR > 0) ? add_pi(a, b) : add_mi(a, b);
    b = (DIR > 0) ? add_mi(a, b) : add_pi(a, b);
    a = t;
  } else {
    const float c = cos64(q), s = (DIR > 0) ? sin64(q) : -sin64(q);
    f2 t = pkfma(mk2(-s, s), b.yx, a);    // (a.re - s*b.im, a.im + s*b.re)
    t = pkfma(mk2(c, c), b, t);
    b = pkfma(mk2(2.0f, 2.0f), a, -t);
    a = t;
  }
}

// P-point FFT on registers, natural order in, natural order out.  `hook(stage)` runs after every radix-2 stage
// (stage = 0 .. log2(P)-1 as an integral_constant): the batch-sharing kernel issues a few of the next work item's
// global loads there, so their issue cost spreads over the butterflies instead of blocking the wave in one burst.
struct NoHook {
  template <class T> __device__ __forceinline__ void operator()(T) const {}
};
template <int P, int DIR, class H = NoHook>
__device__ __forceinline__ void fft_regs(f2 (&v)[P], H&& hook = H{}) {
#if FC_DIAG == 8
  if constexpr (P == 32) {          // timing experiment: the 32-point register transforms do nothing
#pragma unroll
    for (int i = 0; i < P; ++i) asm volatile("" : "+v"(v[i]));
    return;
  }
#endif
  constexpr int LG = ilog2(P);
  f2 t[P];
#pragma unroll
  for (int i = 0; i < P; ++i) t[bitrev(i, LG)] = v[i];
  static_for<0, LG>([&](auto sc) {
    constexpr int len = 2 << decltype(sc)::value;
#pragma unroll
    for (int blk = 0; blk < P; blk += len) {
#pragma unroll
      for (int j = 0; j < len / 2; ++j) bfly<DIR>(t[blk + j], t[blk + j + len / 2], j * (64 / len));
    }
    hook(sc);
  });
#pragma unroll
  for (int i = 0; i < P; ++i) v[i] = t[i];
}

// ---- buffer (SRSRC) loads: one VGPR byte offset + SGPR/constant offset per load ----
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
using BufRsrc = __amdgpu_buffer_rsrc_t;
// build from wave-uniform values only (kernel arguments / blockIdx arithmetic)
// Division of a workgroup index by a launch constant.  A 32-bit integer division costs ~15 VALU instructions per
// wave (v_rcp_iflag sequence) even though operand and result are uniform; with the multiplier prepared on the host
// (Granlund-Montgomery: m = floor(2^(31+l) / d) + 1, l = ceil(log2 d), exact for n < 2^31) it is two scalar multiplies
// and a shift, and the unit maps of the pass kernels (3-7 divisions each) leave the vector pipeline entirely.
struct FastDiv {
  unsigned m, sh, d;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  if (d == 0) d = 1;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.m = (unsigned)(((1ull << (31 + l)) / d) + 1ull);
  f.sh = 31 + l;
  f.d = d;
  return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return (unsigned)(((unsigned long long)n * f.m) >> f.sh); }
// n -> (n / d, n % d):  q returned, n replaced by the remainder's complement use:  r = n - q*d
__device__ __forceinline__ unsigned fdivmod(unsigned n, const FastDiv& f, unsigned* q) {
  *q = fdiv(n, f);
  return n - *q * f.d;
}

__device__ __forceinline__ BufRsrc make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(BufRsrc r, unsigned voff, unsigned soff) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f2 buf_load_f32x2(BufRsrc r, unsigned voff, unsigned soff) {
  const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  return mk2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ f4 buf_load_f32x4(BufRsrc r, unsigned voff, unsigned soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  f4 o; o.x = __uint_as_float(v.x); o.y = __uint_as_float(v.y); o.z = __uint_as_float(v.z); o.w = __uint_as_float(v.w);
  return o;
}

__device__ __forceinline__ void buf_store_f32x2(f2 v, BufRsrc r, unsigned voff, unsigned soff) {
  u32x2 d; d.x = __float_as_uint(v.x); d.y = __float_as_uint(v.y);
  __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, 0);
}
// v_permlane32_swap (gfx950): lanes 32-63 of `a` trade places with lanes 0-31 of `b`
__device__ __forceinline__ void swap_halves(float& a, float& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]); b = __uint_as_float(r[1]);
}
// the same on one component (0 = x, 1 = y) of two packed complex values
template <int C>
__device__ __forceinline__ void swap_halves_c(f2& a, f2& b) {
  float fa = C ? a.y : a.x, fb = C ? b.y : b.x;
  swap_halves(fa, fb);
  if (C) { a.y = fa; b.y = fb; } else { a.x = fa; b.x = fb; }
}
__device__ __forceinline__ void buf_store_f32(float v, BufRsrc r, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}

// ---- cross-lane helpers (DPP quad permutes: VALU rate, no LDS) --------------
__device__ __forceinline__ float dpp_xor1(float v) {   // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {   // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ f2 dpp_xor1(f2 v) { return mk2(dpp_xor1(v.x), dpp_xor1(v.y)); }
__device__ __forceinline__ f2 dpp_xor2(f2 v) { return mk2(dpp_xor2(v.x), dpp_xor2(v.y)); }

// ---------------------------------------------------------------- LDS reads
// hipcc fuses neighbouring 8-byte LDS reads into ds_read2_b64 / ds_read2st64_b64, which gfx950 serves
// at HALF the rate of two ds_read_b64 (8 vs 2 x 2 LDS cycles per wave-instruction).  The hot loops
// therefore issue their reads through asm (one base VGPR + immediate offsets) and wait explicitly:
//   lds_read_strided(v, base)   requests v[i] = base[i * STRIDE]           (nothing waits)
//   lds_arrive(v)               s_waitcnt lgkmcnt(0), then hands the values to the compiler
// LDS returns in order, so compiler-issued LDS traffic mixed in between stays correct.
__device__ __forceinline__ unsigned lds_off(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}
template <int BYTES>
__device__ __forceinline__ f2 lds_rd(unsigned addr) {
  static_assert(BYTES >= 0 && BYTES < 65536, "ds offset field is 16 bits");
  f2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(BYTES));
  return v;
}
// same for offsets beyond the 16-bit field: the excess goes into the address register
template <int BYTES>
__device__ __forceinline__ f2 lds_rd_far(unsigned addr) {
  if constexpr (BYTES < 65536) return lds_rd<BYTES>(addr);
  else return lds_rd<BYTES % 32768>(addr + (unsigned)(BYTES - BYTES % 32768));
}
template <int N, int STRIDE, int FIRST = 0>
__device__ __forceinline__ void lds_read_strided(f2 (&v)[N], const f2* base) {
  const unsigned addr = lds_off(base);
  static_for<FIRST, N>([&](auto ic) {
    constexpr int i = decltype(ic)::value;
    v[i] = lds_rd<i * STRIDE * 8>(addr);
  });
}
template <int N, int FIRST = 0>
__device__ __forceinline__ void lds_arrive(f2 (&v)[N]) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = FIRST; i < N; ++i) asm volatile("" : "+v"(v[i]));
}

// Twiddle table -> LDS at workgroup start: all of a thread's entries are requested before the first is stored (a
// copy loop waits for every load: one memory latency per iteration).
template <int TWN, int NT>
__device__ __forceinline__ void copy_table_to_lds(f2* __restrict__ twl, const f2* __restrict__ src, int tid) {
  constexpr int N = (TWN + NT - 1) / NT;
  f2 tw[N];
#pragma unroll
  for (int i = 0; i < N; ++i) tw[i] = (TWN % NT == 0 || tid + i * NT < TWN) ? src[tid + i * NT] : mk2(0.f, 0.f);
#pragma unroll
  for (int i = 0; i < N; ++i) if (TWN % NT == 0 || tid + i * NT < TWN) twl[tid + i * NT] = tw[i];
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

// Synchronisation between the two passes of ONE sequence.  A sequence is owned by TS = P*S
// consecutive threads; for TS <= 64 they all sit in one wavefront, whose LDS instructions execute
// in program order, so a compiler-level fence is enough and the workgroup barrier (and the skew
// it exposes between waves) disappears.  Only the 4096-point tile (TS = 128) needs s_barrier.
template <class G>
__device__ __forceinline__ void seq_sync() {
  if constexpr (G::TS > 64) {
    __syncthreads();
  } else {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  }
}

// Multiply by the pass-A twiddles and write A[k1][n2] (padded rows).
//   twA[k1*N2 + n2] = exp(-2*pi*i*n2*k1/T)  (forward sign; conjugated for DIR=+1)
template <class G>
__device__ __forceinline__ void passA_twiddle_fetch(f2 (&w)[G::P], int n2, BufRsrc twA) {
  const unsigned off = (unsigned)n2 * 8u;
#pragma unroll
  for (int k1 = 1; k1 < G::P; ++k1) w[k1] = buf_load_f32x2(twA, off, k1 * G::N2 * 8);
}
template <class G, int DIR>
__device__ __forceinline__ void passA_twiddle_apply(f2 (&v)[G::P], const f2 (&w)[G::P], f2* __restrict__ lseq, int n2) {
  lseq[n2] = v[0];
#pragma unroll
  for (int k1 = 1; k1 < G::P; ++k1) lseq[k1 * G::RS + n2] = (DIR > 0) ? cmulc(v[k1], w[k1]) : cmul(v[k1], w[k1]);
}
// P-point register FFT + pass-A twiddles + row store.  The whole twiddle batch is requested before
// the FFT (and pinned there) so its L2 latency hides behind the butterflies.
template <class G, int DIR>
__device__ __forceinline__ void passA_fft_twiddle_store(f2 (&v)[G::P], f2* __restrict__ lseq, int n2, BufRsrc twA) {
  f2 w[G::P];
  passA_twiddle_fetch<G>(w, n2, twA);
  __builtin_amdgcn_sched_barrier(0);
  fft_regs<G::P, DIR>(v);
  passA_twiddle_apply<G, DIR>(v, w, lseq, n2);
}
// same with the twiddle table in LDS
template <class G, int DIR, class H = NoHook>
__device__ __forceinline__ void passA_fft_twiddle_store_lds(f2 (&v)[G::P], f2* __restrict__ lseq, int n2,
                                                            const f2* __restrict__ twl, H&& hook = H{}) {
  fft_regs<G::P, DIR>(v, hook);
  f2 w[G::P];
#if FC_DIAG == 3 || FC_DIAG == 4
#pragma unroll
  for (int k1 = 1; k1 < G::P; ++k1) { w[k1] = mk2(1.f, 1e-3f * k1); asm volatile("" : "+v"(w[k1])); }
#else
  lds_read_strided<G::P, G::N2, 1>(w, twl + n2);
  lds_arrive<G::P, 1>(w);
#endif
#if FC_DIAG == 2
  f2 sink = v[0];
#pragma unroll
  for (int k1 = 1; k1 < G::P; ++k1) sink += (DIR > 0) ? cmulc(v[k1], w[k1]) : cmul(v[k1], w[k1]);
  if (sink.x == 123.456f) lseq[n2] = sink;          // (keeps the arithmetic alive, never stores)
#else
  lseq[n2] = v[0];
#pragma unroll
  for (int k1 = 1; k1 < G::P; ++k1) lseq[k1 * G::RS + n2] = (DIR > 0) ? cmulc(v[k1], w[k1]) : cmul(v[k1], w[k1]);
#endif
}

// same, with the table read in two halves (half the twiddle registers live at a time): for kernels that
// carry many accumulators across the transform
template <class G, int DIR>
__device__ __forceinline__ void passA_fft_twiddle_store_lds_lowreg(f2 (&v)[G::P], f2* __restrict__ lseq, int n2,
                                                                   const f2* __restrict__ twl) {
  fft_regs<G::P, DIR>(v);
  lseq[n2] = v[0];
  constexpr int H = G::P / 2;
  const unsigned addr = lds_off(twl + n2);
  static_for<0, 2>([&](auto hc) {
    constexpr int h = decltype(hc)::value;
    f2 w[H];
    static_for<0, H>([&](auto ic) {
      constexpr int k1 = h * H + decltype(ic)::value;
      w[decltype(ic)::value] = lds_rd<k1 * G::N2 * 8>(addr);
    });
    lds_arrive(w);
#pragma unroll
    for (int i = (h == 0 ? 1 : 0); i < H; ++i) {
      const int k1 = h * H + i;
      lseq[k1 * G::RS + n2] = (DIR > 0) ? cmulc(v[k1], w[i]) : cmul(v[k1], w[i]);
    }
  });
}

// Pass-B load: lane (k1, r) of the sequence reads a[S*m + r] of row k1.
template <class G>
__device__ __forceinline__ void passB_load(f2 (&v)[G::P], const f2* __restrict__ lseq, int tseq) {
  const int k1 = tseq >> G::LGS, r = tseq & (G::S - 1);
#if FC_DIAG == 4
#pragma unroll
  for (int i = 0; i < G::P; ++i) asm volatile("" : "+v"(v[i]));
  (void)k1; (void)r; (void)lseq;
#else
  lds_read_strided<G::P, G::S>(v, lseq + k1 * G::RS + r);
  lds_arrive(v);
#endif
}
// Inverse pass-A load from the natural layout: v[i1] = Z[N2*i1 + tseq]
template <class G>
__device__ __forceinline__ void nat_load(f2 (&v)[G::P], const f2* __restrict__ lseq, int tseq) {
  const unsigned addr = lds_off(lseq + tseq);
#if FC_DIAG == 4
#pragma unroll
  for (int i = 0; i < G::P; ++i) { v[i] = mk2(1e-3f * i, 1.f); asm volatile("" : "+v"(v[i])); }
  (void)addr;
#else
  static_for<0, G::P>([&](auto ic) {
    constexpr int i1 = decltype(ic)::value;
    v[i1] = lds_rd<(G::N2 * i1 + G::NATPAD * (i1 / (G::P / G::S))) * 8>(addr);
  });
  lds_arrive(v);
#endif
}

// Pass-B compute: register FFT + lane-split finish.  Returns j such that element
// k of this lane is X[k1 + P*(k + P*j)]   (k1 = tseq >> log2(S)).
//   twB[r*P + k] = exp(-2*pi*i*r*k/N2) (forward sign): used by the S = 4 split, whose per-lane
//   twiddles differ between the three non-trivial lanes of a quad (S = 2 uses compile-time roots)
template <class G, int DIR, class H = NoHook>
__device__ __forceinline__ int passB_compute(f2 (&v)[G::P], int tseq, BufRsrc twB, H&& hook = H{}) {
  if constexpr (G::S == 4) {
    const int r4 = tseq & 3;
    f2 w[G::P];
#pragma unroll
    for (int k = 1; k < G::P; ++k) w[k] = buf_load_f32x2(twB, (unsigned)(r4 * G::P * 8), k * 8);
    __builtin_amdgcn_sched_barrier(0);
    fft_regs<G::P, DIR>(v, hook);
#pragma unroll
    for (int k = 1; k < G::P; ++k) v[k] = (DIR > 0) ? cmulc(v[k], w[k]) : cmul(v[k], w[k]);
  } else {
    fft_regs<G::P, DIR>(v, hook);
  }
  if constexpr (G::S == 1) {
    return 0;
  } else {
    const int r = tseq & (G::S - 1);
    // lane twiddle w_N2^(r*k): compile-time roots of unity (N2 <= 128), chosen per lane
    static_assert(G::N2 <= 128, "lane-split twiddles come from the 128th-root table");
    constexpr int STEP = 128 / G::N2;
    if constexpr (G::S == 2) {
      if (r != 0) {
#pragma unroll
        for (int k = 1; k < G::P; ++k)
          v[k] = cmul_const(v[k], cos128(STEP * k), (DIR > 0) ? sin128(STEP * k) : -sin128(STEP * k));
      }
      const float sg = r ? -1.0f : 1.0f;
      const f2 sg2 = mk2(sg, sg);
#pragma unroll
      for (int k = 0; k < G::P; ++k) v[k] = pkfma(sg2, v[k], dpp_xor1(v[k]));
      return r;
    } else {  // S == 4 : radix-4 across the quad, output order bit-reversed (lane twiddles applied above)
      const float f2s = (r & 2) ? -1.0f : 1.0f;
      const float f1s = (r & 1) ? -1.0f : 1.0f;
      const f2 sg2 = mk2(f2s, f2s), sg1 = mk2(f1s, f1s);
      const bool rot = (r == 3);
#pragma unroll
      for (int k = 0; k < G::P; ++k) {
        f2 a = pkfma(sg2, v[k], dpp_xor2(v[k]));
        // lane 3 holds (E1 - E3): multiply by w4 = DIR*i
        const f2 ar = (DIR > 0) ? mk2(-a.y, a.x) : mk2(a.y, -a.x);
        a = rot ? ar : a;
        v[k] = pkfma(sg1, a, dpp_xor1(a));
      }
      return ((r & 1) << 1) | (r >> 1);
    }
  }
}

}  // namespace fc
