// Operand packing, C/D layout and per-lane block-scale semantics of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3) on gfx950.
// Hypotheses tested (what ce_attn_fp8.hip assumes):
//   H1  lane l = (r = l & 31, g = l >> 5) supplies bytes j = 0..31 of A row r / B column r for k = 32 g + j;
//   H2  C/D: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)   (the 32x32 layout of the bf16 forms);
//   H3  the scale operand of lane (r, g), byte `opsel`, is the E8M0 scale of the 32-element block (row r, k-block g).
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mx32_probe.hip -o tools/probes/mx32_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int OPSEL>
__global__ void k(const unsigned char* A, const unsigned char* B, const unsigned char* SA, const unsigned char* SB, float* C) {
  const int l = threadIdx.x, r = l & 31, g = l >> 5;
  i32x8 a, b;
  for (int w = 0; w < 8; ++w) {
    a[w] = *reinterpret_cast<const int*>(A + r * 64 + 32 * g + 4 * w);  // A [32][64] row-major
    b[w] = *reinterpret_cast<const int*>(B + r * 64 + 32 * g + 4 * w);  // B^T [32][64]: column r of B, k contiguous
  }
  const int sa = (int)SA[r * 2 + g] << (8 * OPSEL), sb = (int)SB[r * 2 + g] << (8 * OPSEL);  // the scale byte in position OPSEL
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, OPSEL, sa, OPSEL, sb);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = acc[i];
}

static float fp8(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -x : x;
}

int main() {
  unsigned char hA[32 * 64], hB[32 * 64], hSA[64], hSB[64];
  srand(2);
  for (int i = 0; i < 32 * 64; ++i) {
    hA[i] = (unsigned char)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 24));
    hB[i] = (unsigned char)((rand() % 2 ? 0x80 : 0) | (0x28 + rand() % 24));
  }
  unsigned char *dA, *dB, *dSA, *dSB;
  float* dC;
  (void)hipMalloc(&dA, sizeof hA); (void)hipMalloc(&dB, sizeof hB); (void)hipMalloc(&dSA, 64); (void)hipMalloc(&dSB, 64); (void)hipMalloc(&dC, 1024 * 4);
  (void)hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice);
  (void)hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
  for (int mode = 0; mode < 3; ++mode) {  // 0: unit scales, 1: random scales on A only, 2: random scales on both
    for (int i = 0; i < 64; ++i) {
      hSA[i] = (unsigned char)(mode >= 1 ? 124 + rand() % 7 : 127);
      hSB[i] = (unsigned char)(mode >= 2 ? 124 + rand() % 7 : 127);
    }
    (void)hipMemcpy(dSA, hSA, 64, hipMemcpyHostToDevice);
    (void)hipMemcpy(dSB, hSB, 64, hipMemcpyHostToDevice);
    for (int opsel = 0; opsel < 2; ++opsel) {
      if (opsel == 0) k<0><<<1, 64>>>(dA, dB, dSA, dSB, dC); else k<2><<<1, 64>>>(dA, dB, dSA, dSB, dC);
      float hC[1024];
      (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
      double maxerr = 0, maxref = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double ref = 0;
          for (int kk = 0; kk < 64; ++kk)
            ref += (double)fp8(hA[i * 64 + kk]) * ldexp(1.0, hSA[i * 2 + kk / 32] - 127) * fp8(hB[j * 64 + kk]) * ldexp(1.0, hSB[j * 2 + kk / 32] - 127);
          maxerr = fmax(maxerr, fabs(hC[i * 32 + j] - ref));
          maxref = fmax(maxref, fabs(ref));
        }
      printf("mode %d (0 unit scales, 1 scaled A, 2 scaled A and B), opsel %d: max |C - ref| = %.3e (max |ref| %.2f) %s\n", mode, opsel ? 2 : 0,
             maxerr, maxref, maxerr <= 1e-3 * maxref ? "OK" : "MISMATCH");
    }
  }
  return 0;
}
