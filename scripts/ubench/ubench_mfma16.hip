// ubench_mfma16.hip -- operand layout check of v_mfma_f32_16x16x4_f32 on gfx950 (diagnostic, not product):
//   A[i][k]: lane (k*16 + i), B[k][j]: lane (k*16 + j), D[4*(lane/16) + r][lane%16] in register r.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* A, const float* B, float* D) {
  const int l = threadIdx.x;
  const float a = A[(l % 16) * 4 + l / 16];      // A is 16 x 4 row-major
  const float b = B[(l / 16) * 16 + l % 16];     // B is 4 x 16 row-major
  f4 d = {0.f, 0.f, 0.f, 0.f};
  d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(4 * (l / 16) + r) * 16 + l % 16] = d[r];
}
int main() {
  float hA[64], hB[64], hD[256], ref[256];
  for (int i = 0; i < 64; ++i) { hA[i] = (float)(rand() % 17) - 8.f; hB[i] = (float)(rand() % 13) - 6.f; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
  float *A, *B, *D;
  hipMalloc(&A, 256); hipMalloc(&B, 256); hipMalloc(&D, 1024);
  hipMemcpy(A, hA, 256, hipMemcpyHostToDevice); hipMemcpy(B, hB, 256, hipMemcpyHostToDevice);
  k<<<1, 64>>>(A, B, D);
  hipMemcpy(hD, D, 1024, hipMemcpyDeviceToHost);
  double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(hD[i] - ref[i]));
  printf("v_mfma_f32_16x16x4_f32 layout check: max abs diff %.3g (%s)\n", e, e == 0 ? "layout confirmed" : "LAYOUT WRONG");
  return e == 0 ? 0 : 1;
}
