// Is x*x + y*y evaluated fused or unfused on the device?  (debugging aid, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__device__ __forceinline__ float cnrm_unfused(float2 x) {
#pragma clang fp contract(off)
  const float a = x.x * x.x;
  const float b = x.y * x.y;
  return a + b;
}
__global__ void k(const float2* in, float* out_unf, float* out_plain, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { out_unf[i] = cnrm_unfused(in[i]); out_plain[i] = in[i].x * in[i].x + in[i].y * in[i].y; }
}
int main() {
  const int n = 1 << 16;
  float2* h = (float2*)malloc(sizeof(float2) * n);
  srand(1);
  for (int i = 0; i < n; i++) { h[i].x = (float)rand() / RAND_MAX * 100 - 50; h[i].y = (float)rand() / RAND_MAX * 100 - 50; }
  float2* d; float *a, *b;
  hipMalloc(&d, sizeof(float2) * n); hipMalloc(&a, 4 * n); hipMalloc(&b, 4 * n);
  hipMemcpy(d, h, sizeof(float2) * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, n);
  float* ha = (float*)malloc(4 * n); float* hb = (float*)malloc(4 * n);
  hipMemcpy(ha, a, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(hb, b, 4 * n, hipMemcpyDeviceToHost);
  int unf_eq_unf = 0, unf_eq_fx = 0, plain_eq_unf = 0, plain_eq_fx = 0, plain_eq_fy = 0;
  for (int i = 0; i < n; i++) {
    volatile float xx = h[i].x * h[i].x, yy = h[i].y * h[i].y;
    volatile float unf = xx + yy;
    float fx = fmaf(h[i].x, h[i].x, yy), fy = fmaf(h[i].y, h[i].y, xx);
    unf_eq_unf += ha[i] == unf; unf_eq_fx += ha[i] == fx;
    plain_eq_unf += hb[i] == unf; plain_eq_fx += hb[i] == fx; plain_eq_fy += hb[i] == fy;
  }
  printf("n=%d  helper: ==unfused %d ==fma(x,x,yy) %d | plain: ==unfused %d ==fma(x,x,yy) %d ==fma(y,y,xx) %d\n", n, unf_eq_unf, unf_eq_fx, plain_eq_unf, plain_eq_fx, plain_eq_fy);
  return 0;
}
