// fft_f64.h -- float64 1-D FFT convolution (fft_f64.hip): arguments and launchers.
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

struct FftF64Args {
  const double* x;       // (B, Cin, L)
  const double* w;       // (Cout, Cin/G, K)            (kernel transform only)
  double2* wspec;        // [G][Cog][Cig][T]  H = conj(FFT_T(dilated taps)) / T
  const double* bias;    // (Cout) or null
  double* y;             // (B, Cout, Lout)
  int B, Cin, Cout, G, Cig, Cog;
  int L, pad, pad_mode, K, dil, stride;
  int T, V, ntiles, Lfull, Lout;
  int cob, n_ochunks;    // output channels per workgroup (<= 8)
};

// which = 0: kernel transform (w -> wspec), 1: forward convolution
hipError_t launch_fft_f64(int which, const FftF64Args& a, hipStream_t st);
inline size_t fft_f64_lds_bytes(int T) { return (size_t)(2 * T + T / 2) * 16; }   // two sequence buffers + twiddle table

}  // namespace fc
