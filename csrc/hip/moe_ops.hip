// Token routing glue of mixture-of-experts layers on gfx950: the permutation of token rows into expert order and
// the weighted recombination of the expert outputs, both as GATHERS (every output row is written once by one
// workgroup, no atomics, deterministic) - the scatter-add formulations (index_add / indexing backward) serialise on
// the k copies of every token. The reference does this inside Megatron-style token dispatchers
// (realhf/impl/model/modules/moe/token_dispatcher.py).
//
//   gather_rows : out[p, :] = scale[p] * src[idx[p], :]                  (permute forward; combine backward)
//   combine_rows: out[t, :] = sum_j w[t, j] * src[inv[t, j], :]         (combine forward; permute backward, w = 1)
//   combine_dots: dw[t, j]  = < g[t, :], src[inv[t, j], :] >             (gradient of the combine weights)
// Rows are H bf16 values, H a multiple of 8; one workgroup of 256 threads walks a row in 16-byte pieces.
#include "common.hpp"

__global__ __launch_bounds__(256) void moe_gather_rows_kernel(const bf16_t* __restrict__ src,
                                                              const int64_t* __restrict__ idx,
                                                              const bf16_t* __restrict__ scale,
                                                              bf16_t* __restrict__ out, int64_t n_rows, int H) {
  const int64_t p = blockIdx.x;
  if (p >= n_rows) return;
  const bf16_t* s = src + idx[p] * (int64_t)H;
  bf16_t* o = out + p * (int64_t)H;
  const float sc = scale ? bf2f(scale[p]) : 1.f;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    bf16x8 v = ld8(s + c);
    if (scale) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = f2bf(bf2f(v[j]) * sc);
    }
    st8(o + c, v);
  }
}

template <int KMAX>
__global__ __launch_bounds__(256) void moe_combine_rows_kernel(const bf16_t* __restrict__ src,
                                                               const int64_t* __restrict__ inv,
                                                               const bf16_t* __restrict__ w,
                                                               bf16_t* __restrict__ out, int64_t T, int k, int H) {
  const int64_t t = blockIdx.x;
  if (t >= T) return;
  int64_t rows[KMAX];
  float wt[KMAX];
#pragma unroll
  for (int j = 0; j < KMAX; ++j) {
    rows[j] = j < k ? inv[t * k + j] : 0;
    wt[j] = j < k ? (w ? bf2f(w[t * k + j]) : 1.f) : 0.f;
  }
  bf16_t* o = out + t * (int64_t)H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
#pragma unroll
    for (int j = 0; j < KMAX; ++j) {
      if (j < k) {
        const bf16x8 v = ld8(src + rows[j] * (int64_t)H + c);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += wt[j] * bf2f(v[i]);
      }
    }
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r[i] = f2bf(acc[i]);
    st8(o + c, r);
  }
}

__global__ __launch_bounds__(256) void moe_combine_dots_kernel(const bf16_t* __restrict__ g,
                                                               const bf16_t* __restrict__ src,
                                                               const int64_t* __restrict__ inv,
                                                               bf16_t* __restrict__ dw, int64_t T, int k, int H) {
  __shared__ float red[16];
  const int64_t t = blockIdx.x;
  if (t >= T) return;
  const bf16_t* gr = g + t * (int64_t)H;
  for (int j = 0; j < k; ++j) {
    const bf16_t* s = src + inv[t * k + j] * (int64_t)H;
    float acc = 0.f;
    for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
      const bf16x8 a = ld8(gr + c), b = ld8(s + c);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += bf2f(a[i]) * bf2f(b[i]);
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) dw[t * k + j] = f2bf(acc);
  }
}

static int check_rows(const void* a, const void* b, int H) {
  if (H % 8 != 0) return -1;
  if (((uintptr_t)a | (uintptr_t)b) % 16 != 0) return -2;
  return 0;
}

extern "C" int areal_moe_gather_rows(const void* src, const int64_t* idx, const void* scale, void* out,
                                     int64_t n_rows, int H, hipStream_t st) {
  if (n_rows == 0) return 0;
  if (int e = check_rows(src, out, H)) return e;
  hipLaunchKernelGGL(moe_gather_rows_kernel, dim3((unsigned)n_rows), dim3(256), 0, st, (const bf16_t*)src, idx,
                     (const bf16_t*)scale, (bf16_t*)out, n_rows, H);
  return (int)hipGetLastError();
}

extern "C" int areal_moe_combine_rows(const void* src, const int64_t* inv, const void* w, void* out, int64_t T,
                                      int k, int H, hipStream_t st) {
  if (T == 0) return 0;
  if (int e = check_rows(src, out, H)) return e;
  if (k < 1 || k > 16) return -3;
  dim3 grid((unsigned)T), block(256);
  if (k <= 2)
    hipLaunchKernelGGL(moe_combine_rows_kernel<2>, grid, block, 0, st, (const bf16_t*)src, inv, (const bf16_t*)w,
                       (bf16_t*)out, T, k, H);
  else if (k <= 8)
    hipLaunchKernelGGL(moe_combine_rows_kernel<8>, grid, block, 0, st, (const bf16_t*)src, inv, (const bf16_t*)w,
                       (bf16_t*)out, T, k, H);
  else
    hipLaunchKernelGGL(moe_combine_rows_kernel<16>, grid, block, 0, st, (const bf16_t*)src, inv, (const bf16_t*)w,
                       (bf16_t*)out, T, k, H);
  return (int)hipGetLastError();
}

extern "C" int areal_moe_combine_dots(const void* g, const void* src, const int64_t* inv, void* dw, int64_t T,
                                      int k, int H, hipStream_t st) {
  if (T == 0) return 0;
  if (int e = check_rows(g, src, H)) return e;
  hipLaunchKernelGGL(moe_combine_dots_kernel, dim3((unsigned)T), dim3(256), 0, st, (const bf16_t*)g,
                     (const bf16_t*)src, inv, (bf16_t*)dw, T, k, H);
  return (int)hipGetLastError();
}
