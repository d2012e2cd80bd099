// Which lane's scale byte does v_mfma_scale_f32_16x16x128_f8f6f4 apply to WHICH operand bytes, and which byte of the scale VGPR does
// op_sel pick?  (The twin of mx32_probe2.hip for the 16x16x128 shape the fp8 GEMMs run: lane (r = l & 15, g = l >> 4) supplies 32 bytes
// of row r - tools/probes/mx_probe.hip.)  A = 0 except ONE byte (lane L, byte b) = 1.0; B = all 1.0; scale_a of lane S = 2^(S - 40)
// (all distinct) in byte `sel` of its scale word (the other three bytes hold 2^(S - 40 + 64)), scale_b = 1.  Then C[row L % 16][*] = the
// scale that was applied to that byte -> S*(L, b) (and + 64 if the wrong byte of the word was read).  Same for the B side.
// C/D layout assumed (mx_probe.hip): col = lane & 15, row = 4 (lane >> 4) + reg.
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/mx16_probe.hip -o tools/probes/mx16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <bool SIDE_B, int SEL>
__global__ void k(int L, int b, float* C) {
  const int l = threadIdx.x;
  i32x8 one, sel;
  for (int w = 0; w < 8; ++w) { one[w] = 0x38383838; sel[w] = 0; }
  if (l == L) sel[b >> 2] = 0x38 << (8 * (b & 3));
  const int good = 127 - 40 + l, bad = good + 64;
  int code = 0;
  for (int j = 0; j < 4; ++j) code |= (j == SEL ? good : bad) << (8 * j);
  const int unit = 0x7f7f7f7f;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (!SIDE_B) acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(sel, one, acc, 0, 0, SEL, code, 0, unit);
  else         acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(one, sel, acc, 0, 0, 0, unit, SEL, code);
  for (int i = 0; i < 4; ++i) C[(4 * (l >> 4) + i) * 16 + (l & 15)] = acc[i];
}

template <int SEL>
void table(float* dC) {
  float hC[256];
  const int Ls[] = {0, 1, 5, 15, 16, 17, 31, 32, 37, 47, 48, 63}, bs[] = {0, 3, 4, 7, 8, 15, 16, 17, 24, 31};
  for (int side = 0; side < 2; ++side) {
    printf("op_sel %d, %s side: rows = data lane L, columns = byte b; entry = lane whose scale was applied (+64: another byte of the word)\n      ", SEL, side ? "B" : "A");
    for (int b : bs) printf("b=%-3d ", b);
    printf("\n");
    for (int L : Ls) {
      printf("L=%-3d ", L);
      for (int b : bs) {
        if (side == 0) k<false, SEL><<<1, 64>>>(L, b, dC); else k<true, SEL><<<1, 64>>>(L, b, dC);
        (void)hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost);
        // C = A.B^T: A side: the nonzero lives in row L % 16 (all columns equal); B side: in column L % 16 (all rows equal)
        const float v = side == 0 ? hC[(L & 15) * 16 + 3] : hC[3 * 16 + (L & 15)];
        const int S = v > 0 ? (int)lrintf(log2f(v)) + 40 : -1;
        printf("%-5d ", S);
      }
      printf("\n");
    }
  }
}

int main() {
  float* dC;
  (void)hipMalloc(&dC, 256 * 4);
  table<0>(dC);
  table<2>(dC);
  return 0;
}
