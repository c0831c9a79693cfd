// direct_f64.h -- arguments of the float64 direct-convolution kernel (direct_f64.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace fc {

struct DirectF64Args {
  const double* x;      // (B, Cin, S0, S1, S2)   (leading axes of extent 1 for 1-D / 2-D)
  const double* w;      // (Cout, Cin/G, K0, K1, K2) or, transposed, (Cin, Cout/G, K0, K1, K2)
  const double* bias;   // (Cout) or null
  double* y;            // (B, Cout, O0, O1, O2)
  int B, Cin, Cout, G;
  int S[3], K[3], O[3], stride[3], pad[3], dil[3];
  int pad_mode, transposed;
};

hipError_t launch_direct_f64(const DirectF64Args& a, hipStream_t st);

}  // namespace fc
