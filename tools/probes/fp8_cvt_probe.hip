// What does v_cvt_pk_fp8_f32 (OCP e4m3 on gfx950) return for inputs beyond the format's maximum (448)?  NaN (0x7f) or 448 (0x7e)?
// (Decides how ce_attn_fp8.hip detects an overflowing P.)   build: hipcc --offload-arch=gfx950 -O3 tools/probes/fp8_cvt_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const float* x, unsigned* o, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x[i], 1.0f, w, false);
    o[i] = (unsigned)w;
  }
}
int main() {
  const float h[] = {100.f, 440.f, 448.f, 449.f, 463.9f, 464.f, 465.f, 480.f, 500.f, 512.f, 1e4f, 1e30f, INFINITY, NAN, -500.f, -INFINITY, 1e-3f, 9.8e-4f, 0.0019f};
  const int n = sizeof h / sizeof h[0];
  float* dx; unsigned* dout;
  (void)hipMalloc(&dx, sizeof h); (void)hipMalloc(&dout, n * 4);
  (void)hipMemcpy(dx, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dout, n);
  unsigned ho[32];
  (void)hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt_fp8(%g) = 0x%02x (second byte 0x%02x)\n", h[i], ho[i] & 0xff, (ho[i] >> 8) & 0xff);
  return 0;
}
