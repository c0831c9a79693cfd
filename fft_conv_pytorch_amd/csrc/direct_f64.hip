// direct_f64.hip -- float64 tensors: time-domain convolution, one thread per output element.
//
// The reference is dtype-agnostic (torch.fft works in float64); this library's FFT engine is written on
// packed fp32 pairs.  So that float64 callers are served on the device instead of refused, fc_desc.dtype == FC_F64
// plans run this kernel: the same function (padding modes, stride, dilation, groups, 1-3 axes, transposed form with
// output_padding), summed directly in float64 -- O(outputs x Cin/g x taps), meant for validation-grade use, not for
// speed.  Results agree with the reference's float64 FFT path to ~1e-13 relative.
#include <hip/hip_runtime.h>
#include "direct_f64.h"

namespace fc {
namespace {

__device__ __forceinline__ int src_index(int pos, int size, int pad, int mode) {   // unpadded coordinate or -1 (zero)
  if ((unsigned)pos < (unsigned)size) return pos;
  if (pos < -pad || pos >= size + pad || mode == 0) return -1;
  if (mode == 1) return pos < 0 ? -pos : 2 * (size - 1) - pos;      // reflect
  if (mode == 2) return pos < 0 ? 0 : size - 1;                     // replicate
  return pos < 0 ? pos + size : pos - size;                          // circular
}

__global__ __launch_bounds__(256) void direct_f64_kernel(const DirectF64Args a) {
  const long long total = (long long)a.B * a.Cout * a.O[0] * a.O[1] * a.O[2];
  for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < total; id += (long long)gridDim.x * blockDim.x) {
    long long r = id;
    const int o2 = (int)(r % a.O[2]); r /= a.O[2];
    const int o1 = (int)(r % a.O[1]); r /= a.O[1];
    const int o0 = (int)(r % a.O[0]); r /= a.O[0];
    const int co = (int)(r % a.Cout);
    const int b = (int)(r / a.Cout);
    const int cog = a.Cout / a.G, cig = a.Cin / a.G;
    const int g = co / cog;
    const int o[3] = {o0, o1, o2};
    double acc = a.bias ? a.bias[co] : 0.0;
    for (int ci = 0; ci < cig; ++ci) {
      const double* xrow = a.x + ((size_t)b * a.Cin + (size_t)g * cig + ci) * a.S[0] * a.S[1] * a.S[2];
      // weight (Cout, Cin/g, *k) for the forward op, (Cin, Cout/g, *k) for the transposed one
      const double* wrow = a.transposed
          ? a.w + ((size_t)(g * cig + ci) * cog + (co - g * cog)) * a.K[0] * a.K[1] * a.K[2]
          : a.w + ((size_t)co * cig + ci) * a.K[0] * a.K[1] * a.K[2];
      for (int k0 = 0; k0 < a.K[0]; ++k0)
        for (int k1 = 0; k1 < a.K[1]; ++k1)
          for (int k2 = 0; k2 < a.K[2]; ++k2) {
            const int k[3] = {k0, k1, k2};
            int q[3];
            bool ok = true;
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
              if (a.transposed) {
                // y[t] += x[q] w[k] with t = q*stride - padding + k*dilation   (functional.py:126-154 of the reference)
                const int num = o[ax] + a.pad[ax] - k[ax] * a.dil[ax];
                const int qq = num / a.stride[ax];
                ok = ok && num >= 0 && qq * a.stride[ax] == num && qq < a.S[ax];
                q[ax] = qq;
              } else {
                const int pos = o[ax] * a.stride[ax] + k[ax] * a.dil[ax] - a.pad[ax];
                q[ax] = src_index(pos, a.S[ax], a.pad[ax], a.pad_mode);
                ok = ok && q[ax] >= 0;
              }
            }
            if (ok) acc = fma(xrow[((size_t)q[0] * a.S[1] + q[1]) * a.S[2] + q[2]], wrow[((size_t)k0 * a.K[1] + k1) * a.K[2] + k2], acc);
          }
    }
    a.y[id] = acc;
  }
}

}  // namespace

hipError_t launch_direct_f64(const DirectF64Args& a, hipStream_t st) {
  const long long total = (long long)a.B * a.Cout * a.O[0] * a.O[1] * a.O[2];
  if (total <= 0) return hipErrorInvalidValue;
  const long long blocks = (total + 255) / 256;
  const unsigned grid = (unsigned)(blocks < (1LL << 20) ? blocks : (1LL << 20));
  hipLaunchKernelGGL(direct_f64_kernel, dim3(grid), dim3(256), 0, st, a);
  return hipGetLastError();
}

}  // namespace fc
