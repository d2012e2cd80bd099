// Which lane's scale byte does v_mfma_scale_f32_32x32x64_f8f6f4 apply to WHICH operand bytes?  (tools/probes/mx32_probe.hip showed:
// data packing and C/D layout as assumed, but "lane (r, g)'s scale covers lane (r, g)'s 32 bytes" is wrong.)
// A = 0 except ONE byte (lane L, byte b) = 1.0; B = all 1.0; scale_a of lane S = 2^(S - 40) (all distinct), scale_b = 1.
// Then C[row L % 32][*] = the scale that was applied to that byte -> S*(L, b).  Same for the B side.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mx32_probe2.hip -o tools/probes/mx32_probe2
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool SIDE_B>
__global__ void k(int L, int b, float* C) {
  const int l = threadIdx.x;
  i32x8 one, sel;
  for (int w = 0; w < 8; ++w) { one[w] = 0x38383838; sel[w] = 0; }
  if (l == L) sel[b >> 2] = 0x38 << (8 * (b & 3));
  const int code = 127 - 40 + l, unit = 127;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  if (!SIDE_B) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(sel, one, acc, 0, 0, 0, code, 0, unit);
  else         acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(one, sel, acc, 0, 0, 0, unit, 0, code);
  for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[i];
}

int main() {
  float* dC;
  (void)hipMalloc(&dC, 1024 * 4);
  float hC[1024];
  const int Ls[] = {0, 1, 5, 31, 32, 33, 37, 63}, bs[] = {0, 3, 4, 7, 8, 15, 16, 17, 24, 31};
  for (int side = 0; side < 2; ++side) {
    printf("%s side: rows = data lane L, columns = byte b; entry = lane whose scale was applied\n      ", side ? "B" : "A");
    for (int b : bs) printf("b=%-3d ", b);
    printf("\n");
    for (int L : Ls) {
      printf("L=%-3d ", L);
      for (int b : bs) {
        if (side == 0) k<false><<<1, 64>>>(L, b, dC); else k<true><<<1, 64>>>(L, b, dC);
        (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // A side: the nonzero lives in row L % 32 (all columns equal); B side: in column L % 32 (all rows equal)
        const float v = side == 0 ? hC[(L & 31) * 32 + 3] : hC[3 * 32 + (L & 31)];
        const int S = v > 0 ? (int)lrintf(log2f(v)) + 40 : -1;
        printf("%-5d ", S);
      }
      printf("\n");
    }
  }
  return 0;
}
