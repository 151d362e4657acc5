// Flat-buffer optimizer kernels for gfx950: sum of squares (grad norm), scale, fused
// AdamW with fp32 master weights and bf16 shadow parameters, and dtype casts.
//
// The trainer keeps parameters, gradients and optimizer state in flat buffers, so the
// "multi-tensor" ops the reference borrows from TransformerEngine/apex
// (areal/utils/fsdp.py:27-60, 116-179: multi_tensor_l2norm / multi_tensor_scale) and
// torch.optim.AdamW (areal/engine/base_hf_engine.py:197-246) become single launches over
// one buffer. The clip coefficient, the non-finite check and the bias-correction step
// counter live in a small device-side state vector so an optimizer step needs no host
// synchronisation.
#include "common.hpp"

static inline int ew_grid(int64_t total, int block) {
  int64_t g = (total + block - 1) / block;
  if (g > 256 * 8) g = 256 * 8;
  if (g < 1) g = 1;
  return (int)g;
}

// sum of squares -> atomicAdd into out[0] (caller zeroes it).
template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ x, int64_t n,
                                                    float* __restrict__ out) {
  __shared__ float red[16];
  float acc = 0.f;
  const int64_t nvec = n >> 3;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (sizeof(T) == 2) {
      bf16x8 v = ld8(reinterpret_cast<const bf16_t*>(x) + i * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float f = bf2f(v[j]); acc += f * f; }
    } else {
      const float* p = reinterpret_cast<const float*>(x) + i * 8;
      f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc += a[j] * a[j] + b[j] * b[j];
    }
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (nvec << 3) + threadIdx.x; i < n; i += blockDim.x) {
      float f;
      if constexpr (sizeof(T) == 2) f = bf2f(reinterpret_cast<const bf16_t*>(x)[i]);
      else f = reinterpret_cast<const float*>(x)[i];
      acc += f * f;
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) atomicAdd(out, acc);
}

template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(T* __restrict__ x, int64_t n,
                                                    const float* __restrict__ scale_ptr,
                                                    float scale) {
  const float s = scale_ptr ? scale_ptr[0] * scale : scale;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if constexpr (sizeof(T) == 2) {
      bf16_t* p = reinterpret_cast<bf16_t*>(x);
      p[i] = f2bf(bf2f(p[i]) * s);
    } else {
      float* p = reinterpret_cast<float*>(x);
      p[i] *= s;
    }
  }
}

extern "C" int areal_sumsq(const void* x, int dtype, int64_t n, float* out, hipStream_t st) {
  if (n == 0) return 0;
  if (((uintptr_t)x) % 16 != 0) return -1;
  const int grid = ew_grid((n + 7) / 8, 256);
  if (dtype == 0)
    hipLaunchKernelGGL(sumsq_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, n, out);
  else
    hipLaunchKernelGGL(sumsq_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, n, out);
  return (int)hipGetLastError();
}

extern "C" int areal_scale(void* x, int dtype, int64_t n, const float* scale_ptr, float scale,
                           hipStream_t st) {
  if (n == 0) return 0;
  const int grid = ew_grid(n, 256);
  if (dtype == 0)
    hipLaunchKernelGGL(scale_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (bf16_t*)x, n, scale_ptr,
                       scale);
  else
    hipLaunchKernelGGL(scale_kernel<float>, dim3(grid), dim3(256), 0, st, (float*)x, n, scale_ptr,
                       scale);
  return (int)hipGetLastError();
}

// Optimizer state vector (floats):
//  [0] sumsq (input, already reduced across ranks)   [1] grad norm (output)
//  [2] clip coefficient * grad_prescale (output)     [3] 1.0 if the update is applied else 0.0
//  [4] step count (float, incremented when applied)  [5] bias correction 1   [6] bias correction 2
__global__ void adam_prepare_kernel(float* __restrict__ state, float max_norm, float grad_prescale,
                                    float beta1, float beta2) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float norm = sqrtf(state[0]) * fabsf(grad_prescale);
  state[1] = norm;
  const bool ok = isfinite(norm);
  float coef = grad_prescale;
  if (max_norm > 0.f && ok) coef *= fminf(1.f, max_norm / (norm + 1e-6f));
  state[2] = coef;
  state[3] = ok ? 1.f : 0.f;
  if (ok) state[4] += 1.f;
  const float step = fmaxf(state[4], 1.f);
  state[5] = 1.f - powf(beta1, step);
  state[6] = 1.f - powf(beta2, step);
}

// AdamW on a flat shard. grad dtype: bf16 (0) or fp32 (1). Writes the updated bf16 shadow.
template <typename G>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ master, float* __restrict__ m,
                                                    float* __restrict__ v, const G* __restrict__ grad,
                                                    bf16_t* __restrict__ shadow, int64_t n,
                                                    const float* __restrict__ state, float lr,
                                                    float beta1, float beta2, float eps, float wd) {
  if (state[3] == 0.f) return;  // non-finite gradient: skip the update, keep everything
  const float coef = state[2];
  const float bc1 = state[5], bc2 = state[6];
  const float rsqrt_bc2 = rsqrtf(bc2);
  const int64_t nvec = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec;
       i += (int64_t)gridDim.x * blockDim.x) {
    f32x4 p = *reinterpret_cast<f32x4*>(master + i * 4);
    f32x4 mm = *reinterpret_cast<f32x4*>(m + i * 4);
    f32x4 vv = *reinterpret_cast<f32x4*>(v + i * 4);
    float g[4];
    if constexpr (sizeof(G) == 2) {
      bf16x4 gb = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(grad) + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = bf2f(gb[j]) * coef;
    } else {
      f32x4 gf = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(grad) + i * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = gf[j] * coef;
    }
    bf16x4 sh;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mm[j] = beta1 * mm[j] + (1.f - beta1) * g[j];
      vv[j] = beta2 * vv[j] + (1.f - beta2) * g[j] * g[j];
      const float denom = sqrtf(vv[j]) * rsqrt_bc2 + eps;
      p[j] = p[j] * (1.f - lr * wd) - (lr / bc1) * (mm[j] / denom);
      sh[j] = f2bf(p[j]);
    }
    *reinterpret_cast<f32x4*>(master + i * 4) = p;
    *reinterpret_cast<f32x4*>(m + i * 4) = mm;
    *reinterpret_cast<f32x4*>(v + i * 4) = vv;
    if (shadow) *reinterpret_cast<bf16x4*>(shadow + i * 4) = sh;
  }
  if (blockIdx.x == 0) {
    for (int64_t i = (nvec << 2) + threadIdx.x; i < n; i += blockDim.x) {
      float gg;
      if constexpr (sizeof(G) == 2) gg = bf2f(reinterpret_cast<const bf16_t*>(grad)[i]) * coef;
      else gg = reinterpret_cast<const float*>(grad)[i] * coef;
      float mj = beta1 * m[i] + (1.f - beta1) * gg;
      float vj = beta2 * v[i] + (1.f - beta2) * gg * gg;
      const float denom = sqrtf(vj) * rsqrt_bc2 + eps;
      float pj = master[i] * (1.f - lr * wd) - (lr / bc1) * (mj / denom);
      master[i] = pj; m[i] = mj; v[i] = vj;
      if (shadow) shadow[i] = f2bf(pj);
    }
  }
}

extern "C" int areal_adam_prepare(float* state, float max_norm, float grad_prescale, float beta1,
                                  float beta2, hipStream_t st) {
  hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(64), 0, st, state, max_norm, grad_prescale,
                     beta1, beta2);
  return (int)hipGetLastError();
}

extern "C" int areal_adamw(float* master, float* m, float* v, const void* grad, int grad_dtype,
                           void* shadow, int64_t n, const float* state, float lr, float beta1,
                           float beta2, float eps, float wd, hipStream_t st) {
  if (n == 0) return 0;
  if (((uintptr_t)master | (uintptr_t)m | (uintptr_t)v) % 16 != 0) return -1;
  if (((uintptr_t)grad) % (grad_dtype == 0 ? 8 : 16) != 0) return -1;
  if (shadow && ((uintptr_t)shadow) % 8 != 0) return -1;
  const int grid = ew_grid((n + 3) / 4, 256);
  if (grad_dtype == 0)
    hipLaunchKernelGGL(adamw_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, master, m, v,
                       (const bf16_t*)grad, (bf16_t*)shadow, n, state, lr, beta1, beta2, eps, wd);
  else
    hipLaunchKernelGGL(adamw_kernel<float>, dim3(grid), dim3(256), 0, st, master, m, v,
                       (const float*)grad, (bf16_t*)shadow, n, state, lr, beta1, beta2, eps, wd);
  return (int)hipGetLastError();
}
