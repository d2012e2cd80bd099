// Does a SIMD overlap one wave's MFMAs with ANOTHER wave's VALU work?  And with its OWN VALU work?
// 512-thread workgroups, one per CU: waves 0-3 land one per SIMD, waves 4-7 are their partners.
// role: 0 idle, 1 = 16 MFMA 32x32x16 per iteration, 2 = 32 v_exp_f32 per iteration, 3 = 64 v_fma_f32 per iteration,
//       4 = 16 MFMA + 32 v_exp interleaved in ONE instruction stream (sched_group_barrier), 5 = same, clustered (MFMAs then exps)
// build: hipcc --offload-arch=gfx950 -O3 tools/probes/pipe_probe.hip -o tools/probes/pipe_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int ROLE_>
__device__ __forceinline__ void body(int iters, float* out, int tid) {
  constexpr int ROLE = ROLE_ == 6 ? 2 : ROLE_ == 7 ? 1 : ROLE_ == 8 ? 3 : ROLE_;  // 6/7/8: exp32 / mfma16 / fma64 at s_setprio 3
  if (ROLE_ >= 6) __builtin_amdgcn_s_setprio(3);
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (tid + i)); b[i] = (__bf16)(0.002f * (tid - i)); }
  float e[32];
  for (int i = 0; i < 32; ++i) e[i] = 0.001f * (tid + i);
  for (int it = 0; it < iters; ++it) {
    if (ROLE == 1 || ROLE == 4 || ROLE == 5) {
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
    }
    if (ROLE == 2 || ROLE == 4 || ROLE == 5) {
#pragma unroll
      for (int i = 0; i < 32; ++i) e[i] = __builtin_amdgcn_exp2f(e[i]);
    }
    if (ROLE == 3) {
#pragma unroll
      for (int i = 0; i < 32; ++i) e[i] = __builtin_fmaf(e[i], 1.0001f, 0.5f);
#pragma unroll
      for (int i = 0; i < 32; ++i) e[i] = __builtin_fmaf(e[i], 0.9999f, -0.5f);
    }
    if (ROLE == 4) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);  // 2 transcendental
      }
    }
    if (ROLE == 5) {
      __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 32, 0);
    }
    if (ROLE != 4 && ROLE != 5) __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  for (int i = 0; i < 32; ++i) s += e[i];
  if (s == 123.456f) out[tid] = s;
}

// roles 9..12: per iteration 16 units of {1 MFMA + NV plain VALU + NE v_exp}, each unit its own scheduling region;
// PRIO: s_setprio 1 around the VALU part of every unit, 0 around the MFMA
template <int NE, int NV, bool PRIO>
__device__ __forceinline__ void body_mix(int iters, float* out, int tid) {
  f32x16 acc[4];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (tid + i)); b[i] = (__bf16)(0.002f * (tid - i)); }
  float e[64], f[64];
  for (int i = 0; i < 64; ++i) { e[i] = 0.001f * (tid + i); f[i] = 0.002f * (tid + i); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < NE; ++i) e[(u * NE + i) & 63] = __builtin_amdgcn_exp2f(e[(u * NE + i) & 63]);
#pragma unroll
      for (int i = 0; i < NV; ++i) f[(u * NV + i) & 63] = __builtin_fmaf(f[(u * NV + i) & 63], 1.0001f, 0.5f);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  for (int i = 0; i < 64; ++i) s += e[i] + f[i];
  if (s == 123.456f) out[tid] = s;
}

template <int NE, int NV, bool PRIO>
__global__ __launch_bounds__(512) void probe_mix(int iters, float* out) { body_mix<NE, NV, PRIO>(iters, out, threadIdx.x); }

template <int NE, int NV, bool PRIO>
float run_mix(int iters, float* out, int threads = 512) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe_mix<NE, NV, PRIO><<<256, threads>>>(iters, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe_mix<NE, NV, PRIO><<<256, threads>>>(iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int R0, int R1>
__global__ __launch_bounds__(512) void probe(int iters, float* out) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (R0) body<R0>(iters, out, threadIdx.x);
  } else {
    if (R1) body<R1>(iters, out, threadIdx.x);
  }
}

template <int R0, int R1>
float run(int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  probe<R0, R1><<<256, 512>>>(iters, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<R0, R1><<<256, 512>>>(iters, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 512 * 4);
  const int it = 20000;
  const char* names[] = {"idle", "mfma16", "exp32", "fma64", "mfma16+exp32 interleaved (one wave)", "mfma16+exp32 clustered (one wave)",
                         "exp32 @prio3", "mfma16 @prio3", "fma64 @prio3"};
#define RUN(A, B) printf("waves0-3: %-38s waves4-7: %-38s %8.3f ms\n", names[A], names[B], run<A, B>(it, out));
  for (int rep = 0; rep < 2; ++rep) {
    RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(1, 1) RUN(2, 2) RUN(3, 3) RUN(1, 2) RUN(1, 3) RUN(4, 0) RUN(5, 0) RUN(4, 4) RUN(5, 5) RUN(4, 2)
    RUN(1, 6) RUN(6, 1) RUN(7, 2) RUN(2, 7) RUN(1, 8) RUN(8, 1) RUN(4, 6) RUN(5, 6)
#define MIX(NE, NV) printf("8 waves, per MFMA %d v_exp + %d v_fma: %8.3f ms | with s_setprio 1 on the VALU part: %8.3f ms\n", NE, NV, \
                          run_mix<NE, NV, false>(it, out), run_mix<NE, NV, true>(it, out));
    MIX(0, 0) MIX(2, 0) MIX(2, 4) MIX(4, 0) MIX(2, 8) MIX(4, 8) MIX(0, 8) MIX(0, 16)
#define MIX1(NE, NV) printf("4 waves (one per SIMD), per MFMA %d v_exp + %d v_fma: %8.3f ms\n", NE, NV, run_mix<NE, NV, false>(it, out, 256));
    MIX1(0, 0) MIX1(2, 0) MIX1(2, 4) MIX1(4, 0) MIX1(2, 8) MIX1(4, 8) MIX1(0, 8) MIX1(0, 16)
    printf("--\n");
  }
  return 0;
}
