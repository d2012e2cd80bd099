// Operand packing of v_mfma_scale_f32_16x16x128_f8f6f4 (fp8 e4m3 x fp8 e4m3, unit block scales) on gfx950:
// lane l holds 32 bytes of A row (l & 15) and 32 bytes of B column (l & 15); which k's?  Tested hypothesis: k = 32 (l >> 4) + j
// for byte j (and, the contraction being a sum over k, ANY packing that is the same for A and B gives A.B).  C/D: col = lane & 15,
// row = 4 (lane >> 4) + reg.   build: hipcc --offload-arch=gfx950 -O3 tools/probes/mx_probe.hip -o tools/probes/mx_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const unsigned char* A, const unsigned char* B, float* C, int scale_word) {
  const int l = threadIdx.x, r = l & 15, g = l >> 4;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = *reinterpret_cast<const int*>(A + r * 128 + 32 * g + 4 * w);       // A [16][128] row-major
    b[w] = *reinterpret_cast<const int*>(B + r * 128 + 32 * g + 4 * w);       // B^T [16][128]: column r of B, k contiguous
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, scale_word, 0, scale_word);
  for (int i = 0; i < 4; ++i) C[(4 * g + i) * 16 + r] = acc[i];
}

static float fp8_e4m3_to_float(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 0) x = ldexpf((float)m, -9);            // subnormal: m * 2^-3 * 2^-6
  else if (e == 15 && m == 7) x = NAN;
  else x = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

int main() {
  unsigned char hA[16 * 128], hB[16 * 128];
  srand(1);
  for (int i = 0; i < 16 * 128; ++i) {
    hA[i] = (unsigned char)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 24));  // |x| in [0.25, 2): exact products / sums in fp32
    hB[i] = (unsigned char)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 24));
  }
  unsigned char *dA, *dB;
  float* dC;
  hipMalloc(&dA, sizeof hA);
  hipMalloc(&dB, sizeof hB);
  hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int sw : {(int)0x7f7f7f7f, (int)0x80808080, (int)0x7f7f7f7e}) {
    k<<<1, 64>>>(dA, dB, dC, sw);
    float hC[256];
    hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0, ratio = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double ref = 0;
        for (int kk = 0; kk < 128; ++kk) ref += (double)fp8_e4m3_to_float(hA[i * 128 + kk]) * fp8_e4m3_to_float(hB[j * 128 + kk]);
        maxerr = fmax(maxerr, fabs(hC[i * 16 + j] - ref));
        maxref = fmax(maxref, fabs(ref));
        if (i == 3 && j == 5) ratio = hC[i * 16 + j] / ref;
      }
    printf("scale word 0x%08x: max |C - A.B| = %.3e (max |ref| %.3f), C/ref at (3,5) = %.4f\n", sw, maxerr, maxref, ratio);
  }
  return 0;
}
