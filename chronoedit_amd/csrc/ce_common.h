// Shared device helpers for the ChronoEdit gfx950 (CDNA4) kernels.
// Wave = 64 lanes everywhere; MFMA fragment typedefs follow the gfx950 bf16 forms.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define CE_WAVE 64

// The shared library is built with -fvisibility=hidden: CE_API marks the entry points include/chronoedit_hip.h declares (and, in the
// -DCE_DIAGNOSTICS build, the selectors of include/chronoedit_hip_diag.h); cross-TU helpers stay `extern "C"` and hidden.
#define CE_API extern "C" __attribute__((visibility("default")))
// A kernel-body selector: a constant at its default in the product build, a process-wide variable with an exported setter in the diagnostic build
#ifdef CE_DIAGNOSTICS
#define CE_KNOB static int
// (diagnostic build) how often a wave took the EXACT route of an attention kernel's speculative softmax for one key tile: counted per wave and
// tile, read and reset through ce_diag_attention_exact_route_hits (include/chronoedit_hip_diag.h; tools/attn_peaked.py)
#define CE_DIAG_COUNT_EXACT(counter)                                        \
  do {                                                                      \
    if ((threadIdx.x & 63) == 0) atomicAdd(&(counter), 1ull);               \
  } while (0)
#else
#define CE_KNOB static constexpr int
#define CE_DIAG_COUNT_EXACT(counter) do { } while (0)
#endif

// error codes returned by every extern "C" launcher (include/chronoedit_hip.h)
#define CE_OK 0
#define CE_ERR_ARG (-1)
#define CE_ERR_SHAPE (-2)
#define CE_ERR_ALIGN (-3)

// hipFuncSetAttribute applies to the CURRENT device only: launchers remember "attribute set" per device, not per process
#define CE_MAX_DEVICES 16
inline int ce_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= CE_MAX_DEVICES) d = 0;
  return d;
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) { return __uint_as_float(bits16 << 16); }
__device__ __forceinline__ float bf16lo(uint32_t packed) { return __uint_as_float(packed << 16); }
__device__ __forceinline__ float bf16hi(uint32_t packed) { return __uint_as_float(packed & 0xffff0000u); }

// round-to-nearest-even fp32 -> bf16 (v_cvt_pk_bf16_f32 on gfx950), packed pair
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
  f32x2 v = {lo, hi};
  bf16x2 r = __builtin_convertvector(v, bf16x2);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ float round_bf16(float x) {
  bf16 b = (bf16)x;
  return (float)b;
}

// r + y * g with BOTH fp32 roundings (x.float() + y * gate of transformer_chronoedit.py:281,293 is a multiply and an add in eager
// torch).  hipcc compiles HIP with -ffp-contract=fast and __fmul_rn / __fadd_rn are plain `*` / `+` in its headers: without the
// pragma the pair becomes one v_pk_fma_f32 (a single rounding).
__device__ __forceinline__ float mul_then_add(float y, float g, float r) {
#pragma clang fp contract(off)
  const float t = y * g;
  return r + t;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// tanh-approximated GELU, F.gelu(x, approximate="tanh") in fp32:  0.5 x (1 + tanh(u)) == x / (1 + exp(-2u)),
// u = sqrt(2/pi) (x + 0.044715 x^3).  One v_exp_f32 + one v_rcp_f32 (1 ulp each) instead of libm tanhf: the result is
// rounded to bf16 by every caller, and this runs in GEMM epilogues where VALU time is not hidden behind MFMA.
__device__ __forceinline__ float gelu_tanh(float x) {
  constexpr float kC1 = -2.0f * 1.4426950408889634f * 0.7978845608028654f;  // -2 log2(e) sqrt(2/pi)
  constexpr float kC2 = kC1 * 0.044715f;
  const float x2 = x * x;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(kC2, x2, kC1));  // exp(-2u); +inf / 0 at the tails
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// the same on the hardware transcendentals (v_exp_f32, v_rcp_f32: 1 ulp each) for results that are rounded to bf16 at once: libm expf
// plus an IEEE division are ~25 VALU instructions per element, which made the HBM-bound RMS-norm + SiLU pass of the VAE VALU-bound
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// Bijective XCD-aware remap of a 1-D block id (cdna guide T1): block b runs on XCD b % 8,
// so give every XCD one contiguous chunk of the logical tile sequence.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, local = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + local;
}

// ---- OCP MXFP8 helpers (e4m3 elements, one E8M0 scale per 32 elements): shared by the attention producers (ce_attn_fp8.hip) and the
// GEMM operand quantisers (ce_gemm_fp8.hip, ce_rowops.hip).  E8M0 scale byte and its inverse (as a float) for a block with the given
// amax: scale = 2^(floor(log2 amax) - 8) (byte 1 = 2^-126 for an all-zero block); elements are clamped to +-448 before the conversion
// (v_cvt_pk_fp8_f32 does not saturate: above 464 it returns NaN, tools/probes/fp8_cvt_probe.hip).
__device__ __forceinline__ int mx_scale_byte(float amax) {
  const int ef = (int)(__float_as_uint(amax) >> 23);  // biased exponent (amax >= 0)
  return max(ef - 8, 1);
}
// The GEMM operands use the NON-SATURATING choice: the smallest power of two with amax / scale <= 448 (one step up when the mantissa of amax
// exceeds 1.75).  With the floor rule above a block whose amax / scale lands in (448, 512) has its largest elements clipped to 448 - up to
// 12.5 % off on an eighth of the blocks, which made the MX form NOISIER than one scale per row on the same operands (measured: 4.2e-2 vs
// 3.6e-2 on random blocks; e4m3 is a floating-point format, so nothing is gained at the small end by the lower scale).
__device__ __forceinline__ int mx_scale_byte_nosat(float amax) {
  const uint32_t b = __float_as_uint(amax);
  return max((int)(b >> 23) - 8 + ((b & 0x7fffffu) > 0x600000u ? 1 : 0), 1);
}
__device__ __forceinline__ float mx_inv_scale(int byte) { return __uint_as_float((uint32_t)(254 - byte) << 23); }
__device__ __forceinline__ float clamp448(float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); }
// Byte offset of the scale of elements [32 blk, 32 blk + 32) of row `row` in the TILED scale layout of the MX GEMM operands
// ([ceil(rows / 128)][K / 128][4][16][8]; ce_gemm_fp8w4.hip reads 8 consecutive bytes - the 8 row fragments of a wave tile - per lane)
__device__ __forceinline__ size_t mx_gemm_scale_offset(int row, int blk, int ktiles) {
  return ((size_t)(row >> 7) * ktiles + (blk >> 2)) * 512 + (blk & 3) * 128 + (row & 15) * 8 + ((row >> 4) & 7);
}
// The same for the W operand (weights, quantised once): [ceil(rows / 128)][K / 128][4][128 rows in order].  The register-direct epilogue of
// ce_gemm_fp8w4.hip (round 6) feeds fragment G of a wave tile with W rows G + 8 i (i = fragment row), so the eight scale bytes ONE lane needs -
// rows 8 i .. 8 i + 7 - are eight consecutive bytes of this order (the A order keeps rows i, i + 16, ... together: the A fragments are unpermuted).
__device__ __forceinline__ size_t mx_gemm_wscale_offset(int row, int blk, int ktiles) {
  return ((size_t)(row >> 7) * ktiles + (blk >> 2)) * 512 + (blk & 3) * 128 + (row & 127);
}
