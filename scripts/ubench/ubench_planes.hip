// ubench_planes.hip -- timeline of planes_fwd_kernel (fused x/y plane pass of the 3-D path) at the cfgC shape:
// 64 images of 64^3, per-plane timestamps of the first workgroups.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../experiments/planes3d.hpp"

using namespace fc;

int main() {
  const int NA = 64, S = 64;
  float* src; f2 *dst, *twA, *twB; unsigned long long* st;
  hipMalloc(&src, (size_t)NA * S * S * S * 4);
  hipMalloc(&dst, (size_t)NA * 33 * 64 * S * 8);
  hipMalloc(&twA, 64 * 8); hipMalloc(&twB, 8 * 8);
  const int grid = NA * (S / 8);
  hipMalloc(&st, (size_t)grid * 64 * 8);
  hipMemset(st, 0, (size_t)grid * 64 * 8);
  hipMemset(src, 0, (size_t)NA * S * S * S * 4);
  std::vector<f2> ta(64);
  for (int k1 = 0; k1 < 8; ++k1) for (int n2 = 0; n2 < 8; ++n2) { const double ang = -6.283185307179586 * (k1 * n2) / 64.0; ta[k1 * 8 + n2] = f2{(float)cos(ang), (float)sin(ang)}; }
  hipMemcpy(twA, ta.data(), 64 * 8, hipMemcpyHostToDevice);
  hipMemcpy(twB, ta.data(), 8 * 8, hipMemcpyHostToDevice);
  PlanesArgs a{};
  a.src = src; a.dst = dst; a.SZ = a.SY = a.SX = S; a.src_bytes = (unsigned)((size_t)NA * S * S * S * 4); a.NZP = S;
  AxisMap m; m.size = S; m.pad = 0; m.mode = 0; m.up = 1;
  a.mx = a.my = a.mz = m; a.twA = twA; a.twB = twB; a.NA = NA; a.nz = 8; a.stamps = nullptr;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(planes_fwd_kernel<8>, dim3(grid), dim3(kPlaneNT), 0, 0, a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("planes_fwd<8> without stamps: %.1f us\n", ms * 1e3);
  }
  a.stamps = st;
  hipLaunchKernelGGL(planes_fwd_kernel<8>, dim3(grid), dim3(kPlaneNT), 0, 0, a);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h((size_t)grid * 64);
  hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
  unsigned long long t0 = ~0ull, tend = 0;
  for (int w = 0; w < grid; ++w) { t0 = std::min(t0, h[(size_t)w * 64]); tend = std::max(tend, h[(size_t)w * 64 + 41]); }
  printf("kernel span by stamps %.2f us (100 MHz ticks)\n", (tend - t0) * 0.01);
  const char* names[4] = {"rows done", "barrier", "unpack done+barrier", "columns done"};
  for (int w : {0, 1, 255, 511}) {
    printf("workgroup %d: start +%.2f us;", w, (h[(size_t)w * 64] - t0) * 0.01);
    unsigned long long prev = h[(size_t)w * 64];
    for (int zl = 0; zl < 8; ++zl) {
      printf("\n  plane %d:", zl);
      for (int j = 1; j <= 4; ++j) { const unsigned long long t = h[(size_t)w * 64 + j + 4 * zl]; printf(" %s %.2f |", names[j - 1], (t - prev) * 0.01); prev = t; }
    }
    printf("\n  loop end %.2f, stores landed %.2f; lifetime %.2f us\n", (h[(size_t)w * 64 + 40] - prev) * 0.01,
           (h[(size_t)w * 64 + 41] - h[(size_t)w * 64 + 40]) * 0.01, (h[(size_t)w * 64 + 41] - h[(size_t)w * 64]) * 0.01);
  }
  return 0;
}
