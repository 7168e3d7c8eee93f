// kurosiwo_amd — gfx950 (MI355X / CDNA4) kernels: shared device helpers.
// Wave = 64 lanes everywhere (hard-coded; cdna_hip_programming.md §1).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KSMI_WAVE 64

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

typedef uint16_t bf16_t;   // storage type for bf16 activations (raw bits)

// ---- scalar conversions -----------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even (v_cvt_pk_bf16_f32: one VALU instruction per PAIR instead of five per value)
typedef float ksmi_f2 __attribute__((ext_vector_type(2)));
typedef __bf16 ksmi_b2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t f32x2_to_bf16x2(float lo, float hi) {
  const ksmi_f2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, ksmi_b2));
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(f32x2_to_bf16x2(f, 0.f) & 0xffffu); }

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
  static constexpr int kVec = 4;              // elements per 16-byte vector
  __device__ static __forceinline__ float ld(const float* p) { return *p; }
  __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
  __device__ static __forceinline__ float cvt(float v) { return v; }
};
template <> struct ElemTraits<bf16_t> {
  static constexpr int kVec = 8;
  __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  __device__ static __forceinline__ float cvt(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// 16-byte vector <-> 4/8 floats
template <typename T> __device__ __forceinline__ void vec_unpack(const u32x4& v, float* f);
template <> __device__ __forceinline__ void vec_unpack<float>(const u32x4& v, float* f) {
  f[0] = __uint_as_float(v[0]); f[1] = __uint_as_float(v[1]); f[2] = __uint_as_float(v[2]); f[3] = __uint_as_float(v[3]);
}
template <> __device__ __forceinline__ void vec_unpack<bf16_t>(const u32x4& v, float* f) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(v[i] << 16); f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ u32x4 vec_pack(const float* f);
template <> __device__ __forceinline__ u32x4 vec_pack<float>(const float* f) {
  u32x4 v; v[0] = __float_as_uint(f[0]); v[1] = __float_as_uint(f[1]); v[2] = __float_as_uint(f[2]); v[3] = __float_as_uint(f[3]); return v;
}
template <> __device__ __forceinline__ u32x4 vec_pack<bf16_t>(const float* f) {
  u32x4 v;
#pragma unroll
  for (int i = 0; i < 4; ++i) v[i] = f32x2_to_bf16x2(f[2 * i], f[2 * i + 1]);
  return v;
}

// ---- MFMA: one "k-chunk" = 64 bytes of K per row (32 bf16 or 16 fp32) ---------
// A/B operand of a lane = 16 bytes = its 8 (bf16) / 4 (fp32) consecutive k values at
// k-group g = lane>>4; row/col = lane&15.  C/D: col = lane&15, row = (lane>>4)*4 + reg.
// For fp32 the 4 values are fed to 4 v_mfma_f32_16x16x4_f32 steps; step s covers
// k = {g*4+s : g=0..3}; any k permutation is fine as A and B use the same one.
template <typename T> __device__ __forceinline__ void mma16(f32x4& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma16<bf16_t>(f32x4& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma16<float>(f32x4& acc, const u32x4& a, const u32x4& b) {
#pragma unroll
  for (int s = 0; s < 4; ++s)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[s]), __uint_as_float(b[s]), acc, 0, 0, 0);
}

// ---- reductions -------------------------------------------------------------
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
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// XCD-aware bijective block remap (cdna_hip_programming.md T1): consecutive logical
// tiles land on the same XCD (block b is observed to run on XCD b % 8) so neighbouring
// tiles share that XCD's L2.  Speed only; never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// ---- counter-based random stream of the stochastic layers ------------------------------------------------------------
// nn.Dropout / DropPath of the reference draw from torch's global generator; here every Bernoulli draw is a pure function
// of (seed, step, site, element index) so that the backward pass regenerates the forward's mask instead of storing it and
// the CPU oracle (oracle/rng_ref.py) reproduces the same masks.  state[0] = seed, state[1] = step (device memory).
__device__ __forceinline__ uint32_t ksmi_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x21f0aaadu; x ^= x >> 15; x *= 0x735a2d97u; x ^= x >> 15;
  return x;
}
__device__ __forceinline__ uint32_t ksmi_rng_key(const uint32_t* state, uint32_t site) {
  return ksmi_mix32(state[0] ^ ksmi_mix32(state[1] ^ ksmi_mix32(site + 0x9e3779b9u)));
}
__device__ __forceinline__ uint32_t ksmi_rng_u32(uint32_t key, uint32_t idx) { return ksmi_mix32(ksmi_mix32(idx) ^ key); }
// element kept with probability 1 - thr / 2^32
__device__ __forceinline__ bool ksmi_rng_keep(uint32_t key, uint32_t idx, uint32_t thr) { return ksmi_rng_u32(key, idx) >= thr; }

// ---- host: the launchers tell which kernel instantiation they start (the host-side function pointer).  ksmi_last_kernels() (api.hip)
// resolves the pointers to names (hipKernelNameRefByPtr + demangling) so that a timed launch can be tied to the row of a rocprofv3
// table it appears in (bench.py `roofline`, profiles/summarize.py).
void ksmi_note_kernel(const void* host_function);
#define KSMI_NOTE(...) ksmi_note_kernel((const void*)(__VA_ARGS__))
