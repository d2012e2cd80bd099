// v_cvt_scalef32_pk_fp8_f32 (gfx950): does the scale operand multiply or divide, and is its mantissa ignored (E8M0 use of an f32)?
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/fp8_cvt_scale_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((ext_vector_type(2))) short s16x2;
__global__ void k(const float* x, const float* sc, unsigned* o, int nx, int ns) {
  const int i = threadIdx.x;
  if (i < nx * ns) {
    s16x2 r = {0, 0};
    r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(r, x[i % nx], 1.0f, sc[i / nx], false);
    o[i] = __builtin_bit_cast(unsigned, r);
  }
}
static float fp8(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  if ((v & 0x7f) == 0x7f) return NAN;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}
int main() {
  const float hx[] = {1.f, 3.f, 100.f, 1792.f, 0.01f, 7000.f}, hs[] = {1.f, 4.f, 0.25f, 6.f, 7.99f};
  const int nx = 6, ns = 5;
  float *dx, *ds; unsigned* dout;
  (void)hipMalloc(&dx, sizeof hx); (void)hipMalloc(&ds, sizeof hs); (void)hipMalloc(&dout, 64 * 4);
  (void)hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice); (void)hipMemcpy(ds, hs, sizeof hs, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, ds, dout, nx, ns);
  unsigned ho[64];
  (void)hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  for (int j = 0; j < ns; ++j) {
    printf("scale %-5g:", hs[j]);
    for (int i = 0; i < nx; ++i) printf("  x=%-6g -> 0x%02x = %-8g (1.0 -> %g)", hx[i], ho[j * nx + i] & 0xff, fp8(ho[j * nx + i] & 0xff), fp8((ho[j * nx + i] >> 8) & 0xff));
    printf("\n");
  }
  return 0;
}
