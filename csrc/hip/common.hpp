// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.
//
// Everything here assumes a 64-lane wavefront. Kernels are exported with a C ABI
// (raw pointers + hipStream_t) and called from Python through ctypes, so no
// framework headers are needed to build them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16_t;
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define WAVE 64
#define DEVINL __device__ __forceinline__

#define HIP_RET(expr)                         \
  do {                                        \
    hipError_t _e = (expr);                   \
    if (_e != hipSuccess) return (int)_e;     \
  } while (0)

DEVINL float bf2f(bf16_t x) { return (float)x; }
DEVINL bf16_t f2bf(float x) { return (bf16_t)x; }

// Exchanges across 16- and 32-lane rows in the VALU (v_permlane16_swap / v_permlane32_swap of gfx950) instead of
// through the LDS crossbar (`__shfl_xor(v, 16 | 32)` compiles to ds_bpermute_b32: an LDS-pipe round trip of ~100
// cycles on the dependent path, and LDS traffic in kernels that otherwise use none). Fed the same register twice
// the swap leaves, in every lane, the value of the lane's even row in [0] and of its odd row in [1]: own value and
// partner's value, in an order that is the same for both partners - op([0], [1]) is bit-identical on both lanes.
DEVINL f32x2 rows16_pair(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  return (f32x2){__uint_as_float(r[0]), __uint_as_float(r[1])};
}
DEVINL f32x2 rows32_pair(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  return (f32x2){__uint_as_float(r[0]), __uint_as_float(r[1])};
}
DEVINL float sum_xor16(float v) { const f32x2 r = rows16_pair(v); return r[0] + r[1]; }
DEVINL float sum_xor32(float v) { const f32x2 r = rows32_pair(v); return r[0] + r[1]; }
DEVINL float max_xor16(float v) { const f32x2 r = rows16_pair(v); return fmaxf(r[0], r[1]); }
DEVINL float max_xor32(float v) { const f32x2 r = rows32_pair(v); return fmaxf(r[0], r[1]); }
DEVINL float min_xor16(float v) { const f32x2 r = rows16_pair(v); return fminf(r[0], r[1]); }
DEVINL float min_xor32(float v) { const f32x2 r = rows32_pair(v); return fminf(r[0], r[1]); }

DEVINL float wave_sum(float v) {
  v = sum_xor16(sum_xor32(v));
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);  // within a 16-lane row: DPP
  return v;
}
DEVINL float wave_max(float v) {
  v = max_xor16(max_xor32(v));
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
DEVINL float wave_min(float v) {
  v = min_xor16(min_xor32(v));
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// 16-byte vector load/store of 8 bf16.
// v_exp_f32 without the denormal-safe wrapper of exp2f(): softmax terms do not need denormal results
DEVINL float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

DEVINL bf16x8 ld8(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
DEVINL void st8(bf16_t* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }

// ---- fp8 (e4m3, the format of the gfx950 conversion instructions = torch.float8_e4m3fn) for the KV cache.
// Values are clamped to the finite range (+-448) before the conversion; every fp8 value is exact in bf16.
typedef unsigned char fp8_t;
DEVINL float fp8_clamp(float x) { return fminf(fmaxf(x, -448.f), 448.f); }
DEVINL unsigned pack_fp8x4(float a, float b, float c, float d) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(a), fp8_clamp(b), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(c), fp8_clamp(d), w, true);
  return (unsigned)w;
}
DEVINL fp8_t f2fp8(float x) { return (fp8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(fp8_clamp(x), 0.f, 0, false) & 0xff); }
DEVINL u32x2 bf16x8_to_fp8(bf16x8 v) {
  return (u32x2){pack_fp8x4(bf2f(v[0]), bf2f(v[1]), bf2f(v[2]), bf2f(v[3])),
                 pack_fp8x4(bf2f(v[4]), bf2f(v[5]), bf2f(v[6]), bf2f(v[7]))};
}
DEVINL bf16x8 fp8x8_to_bf16(u32x2 w) {
  const f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[0], true);
  const f32x2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[1], true);
  return (bf16x8){f2bf(a[0]), f2bf(a[1]), f2bf(b[0]), f2bf(b[1]), f2bf(c[0]), f2bf(c[1]), f2bf(d[0]), f2bf(d[1])};
}
DEVINL float fp82f(fp8_t x) { return __builtin_amdgcn_cvt_pk_f32_fp8((int)x, false)[0]; }
// 8 cache elements at element offset `i` of a cache of bf16 (FP8 = false) or fp8 (true) values, as bf16
template <bool FP8>
DEVINL bf16x8 ld8_cache(const void* base, size_t i) {
  if (FP8) return fp8x8_to_bf16(*reinterpret_cast<const u32x2*>(reinterpret_cast<const fp8_t*>(base) + i));
  return ld8(reinterpret_cast<const bf16_t*>(base) + i);
}
DEVINL void st8_cache(void* base, size_t i, bf16x8 v, int fp8) {
  if (fp8) *reinterpret_cast<u32x2*>(reinterpret_cast<fp8_t*>(base) + i) = bf16x8_to_fp8(v);
  else st8(reinterpret_cast<bf16_t*>(base) + i, v);
}
DEVINL void st1_cache(void* base, size_t i, float v, int fp8) {
  if (fp8) reinterpret_cast<fp8_t*>(base)[i] = f2fp8(v);
  else reinterpret_cast<bf16_t*>(base)[i] = f2bf(v);
}

// Block-wide sum for blocks of up to 1024 threads; `red` needs 16 floats of LDS.
DEVINL float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : 0.f;
  return wave_sum(t);
}
DEVINL float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float t = (lane < nw) ? red[lane] : -INFINITY;
  return wave_max(t);
}

// Counter-based RNG (Philox-4x32-10): stateless, reproducible across launches and
// graph replays when the (seed, offset) pair lives in device memory.
struct Philox {
  uint32_t k0, k1;
  DEVINL static void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
  }
  DEVINL static u32x4 round(u32x4 c, uint32_t k0, uint32_t k1) {
    uint32_t h0, l0, h1, l1;
    mulhilo(0xD2511F53u, c[0], h0, l0);
    mulhilo(0xCD9E8D57u, c[2], h1, l1);
    u32x4 r;
    r[0] = h1 ^ c[1] ^ k0;
    r[1] = l1;
    r[2] = h0 ^ c[3] ^ k1;
    r[3] = l0;
    return r;
  }
  DEVINL static u32x4 gen(uint64_t seed, uint64_t subseq, uint64_t offset) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    u32x4 c;
    c[0] = (uint32_t)offset;
    c[1] = (uint32_t)(offset >> 32);
    c[2] = (uint32_t)subseq;
    c[3] = (uint32_t)(subseq >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      c = round(c, k0, k1);
      k0 += 0x9E3779B9u;
      k1 += 0xBB67AE85u;
    }
    return c;
  }
};
DEVINL float u32_to_unit(uint32_t x) {  // (0, 1]
  return ((x >> 8) + 1) * (1.0f / 16777216.0f);
}
