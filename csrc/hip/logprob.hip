// Policy log-prob / entropy kernels over the vocabulary axis (gfx950).
//
// Replaces the torch.compile'd log_softmax+gather(+entropy) of the reference
// (areal/utils/functional.py:10-80) and the 3-all-reduce vocab-parallel cross entropy
// (realhf/impl/model/parallelism/tensor_parallel/modules.py:1060-1177): a single pass
// over each logits row keeps an online (max, sum exp, sum exp*z, target) tuple, so
// [T, V] fp32 is never materialised and a vocab shard only has to exchange four floats
// per row with its tensor-parallel peers.
#include "common.hpp"

template <typename T>
struct Vec8;
template <>
struct Vec8<bf16_t> {
  static DEVINL void load(const bf16_t* p, float (&o)[8]) {
    bf16x8 v = ld8(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bf2f(v[j]);
  }
  static DEVINL void store(bf16_t* p, const float (&o)[8]) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = f2bf(o[j]);
    st8(p, v);
  }
};
template <>
struct Vec8<float> {
  static DEVINL void load(const float* p, float (&o)[8]) {
    f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[j + 4] = b[j]; }
  }
  static DEVINL void store(float* p, const float (&o)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { a[j] = o[j]; b[j] = o[j + 4]; }
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
  }
};

// Online-softmax state merge: (m, s, w) with s = sum exp(z-m), w = sum exp(z-m) * z.
DEVINL void merge(float& m, float& s, float& w, float m2, float s2, float w2) {
  const float mn = fmaxf(m, m2);
  const float a = (m == -INFINITY) ? 0.f : __expf(m - mn);
  const float b = (m2 == -INFINITY) ? 0.f : __expf(m2 - mn);
  s = s * a + s2 * b;
  w = w * a + w2 * b;
  m = mn;
}

// One 256-thread block per row. Outputs per row: m, s, w, tgt (target logit * inv_temp, or
// 0 when the label is outside [vocab_start, vocab_start+V)), plus min/max raw logits.
// If `finalize`, also writes logp / entropy / lse / ez directly (single-shard case).
template <typename T>
__global__ __launch_bounds__(256) void logprob_stats_kernel(
    const T* __restrict__ logits, int64_t ld, const int64_t* __restrict__ labels, int V,
    int vocab_start, float inv_temp, float* __restrict__ out_m, float* __restrict__ out_s,
    float* __restrict__ out_w, float* __restrict__ out_tgt, float* __restrict__ out_logp,
    float* __restrict__ out_ent, float* __restrict__ out_lse, float* __restrict__ out_ez,
    int finalize) {
  __shared__ float red[3][4];
  const int row = blockIdx.x;
  const T* p = logits + (size_t)row * ld;
  float m = -INFINITY, s = 0.f, w = 0.f;
  const int nvec = V >> 3;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    float z[8];
    Vec8<T>::load(p + c * 8, z);
    float lm = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) { z[j] *= inv_temp; lm = fmaxf(lm, z[j]); }
    if (lm > m) {
      const float a = (m == -INFINITY) ? 0.f : __expf(m - lm);
      s *= a; w *= a; m = lm;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float e = __expf(z[j] - m);
      s += e; w += e * z[j];
    }
  }
  // scalar tail when V is not a multiple of 8
  for (int c = (nvec << 3) + threadIdx.x; c < V; c += blockDim.x) {
    const float z = (float)p[c] * inv_temp;
    merge(m, s, w, z, 1.f, z);
  }
  // wave reduce
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64), w2 = __shfl_xor(w, o, 64);
    merge(m, s, w, m2, s2, w2);
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { red[0][wid] = m; red[1][wid] = s; red[2][wid] = w; }
  __syncthreads();
  if (threadIdx.x == 0) {
    m = red[0][0]; s = red[1][0]; w = red[2][0];
    for (int k = 1; k < 4; ++k) merge(m, s, w, red[0][k], red[1][k], red[2][k]);
    const int64_t lab = labels[row] - vocab_start;
    float tgt = 0.f;
    const bool has = lab >= 0 && lab < V;
    if (has) tgt = (float)p[lab] * inv_temp;
    if (out_m) { out_m[row] = m; out_s[row] = s; out_w[row] = w; out_tgt[row] = tgt; }
    if (finalize) {
      const float lse = m + __logf(s);
      const float ez = w / s;
      if (out_logp) out_logp[row] = has ? tgt - lse : 0.f;
      if (out_ent) out_ent[row] = lse - ez;
      if (out_lse) out_lse[row] = lse;
      if (out_ez) out_ez[row] = ez;
    }
  }
}

// dlogits[t, v] = inv_temp * ( dlogp[t] * (1[v == label] - p) - dent[t] * p * (z - ez) )
// p = exp(z - lse). Written to `dlogits` (may alias `logits`) in the logits dtype.
template <typename T>
__global__ __launch_bounds__(256) void logprob_bwd_kernel(
    const T* __restrict__ logits, T* __restrict__ dlogits, int64_t ld,
    const int64_t* __restrict__ labels, int V, int vocab_start, float inv_temp,
    const float* __restrict__ lse, const float* __restrict__ ez, const float* __restrict__ dlogp,
    const float* __restrict__ dent) {
  const int row = blockIdx.x;
  const T* p = logits + (size_t)row * ld;
  T* q = dlogits + (size_t)row * ld;
  const float l = lse[row];
  const float e = dent ? ez[row] : 0.f;
  const float gl = dlogp ? dlogp[row] : 0.f;
  const float ge = dent ? dent[row] : 0.f;
  const int64_t lab = labels[row] - vocab_start;
  const int nvec = V >> 3;
  for (int c = threadIdx.x; c < nvec; c += blockDim.x) {
    float z[8], d[8];
    Vec8<T>::load(p + c * 8, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float zz = z[j] * inv_temp;
      const float pr = __expf(zz - l);
      float g = -gl * pr - ge * pr * (zz - e);
      if ((int64_t)(c * 8 + j) == lab) g += gl;
      d[j] = g * inv_temp;
    }
    Vec8<T>::store(q + c * 8, d);
  }
  for (int c = (nvec << 3) + threadIdx.x; c < V; c += blockDim.x) {
    const float zz = (float)p[c] * inv_temp;
    const float pr = __expf(zz - l);
    float g = -gl * pr - ge * pr * (zz - e);
    if ((int64_t)c == lab) g += gl;
    q[c] = (T)(g * inv_temp);
  }
}

// dtype: 0 = bf16, 1 = fp32. Row stride `ld` in elements must keep rows 16-byte aligned.
extern "C" int areal_logprob_stats(const void* logits, int dtype, int64_t T, int V, int64_t ld,
                                   const int64_t* labels, int vocab_start, float inv_temp,
                                   float* out_m, float* out_s, float* out_w, float* out_tgt,
                                   float* out_logp, float* out_ent, float* out_lse, float* out_ez,
                                   int finalize, hipStream_t st) {
  if (T == 0) return 0;
  const int elt = dtype == 0 ? 2 : 4;
  if ((ld * elt) % 16 != 0 || ((uintptr_t)logits) % 16 != 0) return -1;
  if (dtype == 0)
    hipLaunchKernelGGL(logprob_stats_kernel<bf16_t>, dim3((unsigned)T), dim3(256), 0, st,
                       (const bf16_t*)logits, ld, labels, V, vocab_start, inv_temp, out_m, out_s,
                       out_w, out_tgt, out_logp, out_ent, out_lse, out_ez, finalize);
  else
    hipLaunchKernelGGL(logprob_stats_kernel<float>, dim3((unsigned)T), dim3(256), 0, st,
                       (const float*)logits, ld, labels, V, vocab_start, inv_temp, out_m, out_s,
                       out_w, out_tgt, out_logp, out_ent, out_lse, out_ez, finalize);
  return (int)hipGetLastError();
}

extern "C" int areal_logprob_bwd(const void* logits, void* dlogits, int dtype, int64_t T, int V,
                                 int64_t ld, const int64_t* labels, int vocab_start, float inv_temp,
                                 const float* lse, const float* ez, const float* dlogp,
                                 const float* dent, hipStream_t st) {
  if (T == 0) return 0;
  const int elt = dtype == 0 ? 2 : 4;
  if ((ld * elt) % 16 != 0 || ((uintptr_t)logits) % 16 != 0 || ((uintptr_t)dlogits) % 16 != 0)
    return -1;
  if (dtype == 0)
    hipLaunchKernelGGL(logprob_bwd_kernel<bf16_t>, dim3((unsigned)T), dim3(256), 0, st,
                       (const bf16_t*)logits, (bf16_t*)dlogits, ld, labels, V, vocab_start, inv_temp,
                       lse, ez, dlogp, dent);
  else
    hipLaunchKernelGGL(logprob_bwd_kernel<float>, dim3((unsigned)T), dim3(256), 0, st,
                       (const float*)logits, (float*)dlogits, ld, labels, V, vocab_start, inv_temp,
                       lse, ez, dlogp, dent);
  return (int)hipGetLastError();
}
