// GAE and interval gather/scatter kernels for gfx950.
//
// GAE: the reference kernels (csrc/cugae/gae.cu:10-28, 62-89, 136-161) give each
// sequence (or episode segment) to ONE thread that walks it serially. Here a segment
// belongs to one 64-lane wave: A_t = delta_t + c * A_{t+1} is a linear recurrence, so
// each lane scans a contiguous chunk, the 64 chunk summaries (decay, value) are combined
// with a log-step suffix scan over the wave, and a second pass applies the carry.
//
// Interval ops: csrc/interval_op/interval_op.cu:12-104 semantics (gather many
// [start, end) ranges of a flat parameter buffer into a packed buffer and back).
#include "common.hpp"

// ---------------------------------------------------------------------------- GAE core
struct GaeSeg {
  const float* r;   // rewards of the segment (length L)
  const float* v;   // values of the segment (length L)
  float* adv;
  float* ret;
  int L;
  float last_next_value;  // next value of the final step (0 when not bootstrapping)
};

DEVINL void gae_wave_scan(const GaeSeg& s, float gamma, float lam, int lane) {
  const int L = s.L;
  if (L <= 0) return;
  const float c = gamma * lam;
  const int chunk = (L + 63) / 64;
  const int beg = lane * chunk;
  const int end = beg + chunk < L ? beg + chunk : L;
  // pass 1: local scan with zero carry; f_lane(x) = a + decay * x
  float a = 0.f, decay = 1.f;
  for (int t = end - 1; t >= beg; --t) {
    const float nv = (t == L - 1) ? s.last_next_value : s.v[t + 1];
    const float delta = s.r[t] + gamma * nv - s.v[t];
    a = delta + c * a;
    decay *= c;
  }
  // suffix scan over lanes: F_i = f_i o f_{i+1} o ... o f_63
  float fa = a, fm = decay;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float ga = __shfl_down(fa, o, 64), gm = __shfl_down(fm, o, 64);
    if (lane + o < 64) { fa = fa + fm * ga; fm = fm * gm; }
  }
  // carry into this lane = A at index `end` = F_{lane+1}(0)
  float carry = __shfl_down(fa, 1, 64);
  if (lane == 63) carry = 0.f;
  // pass 2: rescan with the carry
  a = carry;
  for (int t = end - 1; t >= beg; --t) {
    const float nv = (t == L - 1) ? s.last_next_value : s.v[t + 1];
    const float delta = s.r[t] + gamma * nv - s.v[t];
    a = delta + c * a;
    s.adv[t] = a;
    s.ret[t] = a + s.v[t];
  }
}

// Packed 1-D: rewards[total], values[total + bs] (one bootstrap slot per sequence).
__global__ __launch_bounds__(256) void gae_1d_packed_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const int* __restrict__ cu_seqlens, const uint8_t* __restrict__ bootstrap,
    float* __restrict__ adv, float* __restrict__ ret, int bs, float gamma, float lam) {
  const int seq = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (seq >= bs) return;
  const int lane = threadIdx.x & 63;
  const int rs = cu_seqlens[seq], re = cu_seqlens[seq + 1];
  GaeSeg s;
  s.L = re - rs;
  s.r = rewards + rs;
  s.v = values + rs + seq;
  s.adv = adv + rs;
  s.ret = ret + rs;
  s.last_next_value = (s.L > 0 && bootstrap[seq]) ? s.v[s.L] : 0.f;
  gae_wave_scan(s, gamma, lam, lane);
}

// Padded 2-D without episode boundaries (AReaL-lite actor semantics,
// areal/engine/ppo/actor.py:131-147): rewards/values [B, S]; advantages for t in
// [0, S-2], column S-1 is zero; the next value of t = S-2 is values[:, S-1] * no_eos[b].
__global__ __launch_bounds__(256) void gae_padded_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const uint8_t* __restrict__ no_eos, float* __restrict__ adv, float* __restrict__ ret, int B,
    int S, float gamma, float lam) {
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  GaeSeg s;
  s.L = S - 1;
  s.r = rewards + (size_t)b * S;
  s.v = values + (size_t)b * S;
  s.adv = adv + (size_t)b * S;
  s.ret = ret + (size_t)b * S;
  s.last_next_value = (S >= 2 && no_eos[b]) ? s.v[S - 1] : 0.f;
  gae_wave_scan(s, gamma, lam, lane);
  if (lane == 0 && S >= 1) { s.adv[S - 1] = 0.f; s.ret[S - 1] = s.v[S - 1]; }
}

// Padded 2-D with episode segmentation. rewards [B, H], values/done [B, H+1].
// Segment j of row b spans [t_start, t_end) where boundaries come from
// done_idx[cu_dones[b] .. cu_dones[b+1]); the last segment ends at H.
//  mode 0 ("olp", overlapped terminations): every step in the segment is used; the next
//      value of the final step is 0 when done[b, t_end] is set; if flag[k] (truncated) is
//      set for a non-final segment, the final step's advantage is forced to 0.
//  mode 1 ("nolp"): the step t_end-1 of a non-final segment is the terminal step itself and
//      is excluded; the next value of step t_end-2 is values[b, t_end-1] when flag[k]
//      (bootstrap) is set, else 0. The final segment runs to H-1 and bootstraps from
//      values[b, H].
__global__ __launch_bounds__(256) void gae_2d_seg_kernel(
    const float* __restrict__ rewards, const float* __restrict__ values,
    const uint8_t* __restrict__ done, const int* __restrict__ done_idx,
    const int* __restrict__ cu_dones, const uint8_t* __restrict__ flag, float* __restrict__ adv,
    float* __restrict__ ret, int B, int H, int max_segs, int mode, float gamma, float lam) {
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int b = w / max_segs, j = w % max_segs;
  if (b >= B) return;
  const int lane = threadIdx.x & 63;
  const int d0 = cu_dones[b], nd = cu_dones[b + 1] - d0;
  if (j > nd) return;
  const bool is_last = (j == nd);
  const int t_start = j == 0 ? 0 : done_idx[d0 + j - 1];
  int t_end = is_last ? H : done_idx[d0 + j];
  const float* vrow = values + (size_t)b * (H + 1);
  GaeSeg s;
  int zero_final = 0;
  if (mode == 0) {
    s.last_next_value = (t_end > t_start && done[(size_t)b * (H + 1) + t_end]) ? 0.f : vrow[t_end];
    zero_final = (!is_last && t_end > t_start && flag[d0 + j]) ? 1 : 0;
    if (zero_final) {
      // the final step's advantage is forced to 0 and the chain restarts below it
      t_end -= 1;
      s.last_next_value = vrow[t_end];
    }
  } else {
    if (!is_last) {
      t_end -= 1;  // drop the terminal step
      s.last_next_value = (t_end > t_start && flag[d0 + j]) ? vrow[t_end] : 0.f;
    } else {
      s.last_next_value = vrow[H];
    }
  }
  s.L = t_end - t_start;
  s.r = rewards + (size_t)b * H + t_start;
  s.v = vrow + t_start;
  s.adv = adv + (size_t)b * H + t_start;
  s.ret = ret + (size_t)b * H + t_start;
  if (zero_final && lane == 0) {
    adv[(size_t)b * H + t_end] = 0.f;
    ret[(size_t)b * H + t_end] = vrow[t_end];
  }
  gae_wave_scan(s, gamma, lam, lane);
}

extern "C" int areal_gae_1d_packed(const float* rewards, const float* values, const int* cu_seqlens,
                                   const uint8_t* bootstrap, float* adv, float* ret, int bs,
                                   float gamma, float lam, hipStream_t st) {
  if (bs == 0) return 0;
  hipLaunchKernelGGL(gae_1d_packed_kernel, dim3((bs + 3) / 4), dim3(256), 0, st, rewards, values,
                     cu_seqlens, bootstrap, adv, ret, bs, gamma, lam);
  return (int)hipGetLastError();
}
extern "C" int areal_gae_padded(const float* rewards, const float* values, const uint8_t* no_eos,
                                float* adv, float* ret, int B, int S, float gamma, float lam,
                                hipStream_t st) {
  if (B == 0 || S == 0) return 0;
  hipLaunchKernelGGL(gae_padded_kernel, dim3((B + 3) / 4), dim3(256), 0, st, rewards, values, no_eos,
                     adv, ret, B, S, gamma, lam);
  return (int)hipGetLastError();
}
extern "C" int areal_gae_2d(const float* rewards, const float* values, const uint8_t* done,
                            const int* done_idx, const int* cu_dones, const uint8_t* flag, float* adv,
                            float* ret, int B, int H, int max_dones, int mode, float gamma, float lam,
                            hipStream_t st) {
  if (B == 0 || H == 0) return 0;
  const int max_segs = max_dones + 1;
  const int64_t waves = (int64_t)B * max_segs;
  hipLaunchKernelGGL(gae_2d_seg_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, rewards,
                     values, done, done_idx, cu_dones, flag, adv, ret, B, H, max_segs, mode, gamma,
                     lam);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------- interval copy
// intervals: int64 [n, 2] (start, end) into `flat`; packed offsets: int64 [n] exclusive
// prefix sum of lengths. One block column per interval, grid.y tiles long intervals.
// dir = 0: packed <- flat (slice), dir = 1: flat <- packed (set). elt = element bytes.
template <typename T>
__global__ __launch_bounds__(256) void interval_copy_kernel(T* __restrict__ flat,
                                                            T* __restrict__ packed,
                                                            const int64_t* __restrict__ intervals,
                                                            const int64_t* __restrict__ offsets,
                                                            int n, int dir) {
  for (int i = blockIdx.x; i < n; i += gridDim.x) {
    const int64_t s = intervals[2 * i], e = intervals[2 * i + 1];
    const int64_t o = offsets[i];
    const int64_t len = e - s;
    for (int64_t k = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; k < len;
         k += (int64_t)gridDim.y * blockDim.x) {
      if (dir == 0) packed[o + k] = flat[s + k];
      else flat[s + k] = packed[o + k];
    }
  }
}

extern "C" int areal_interval_copy(void* flat, void* packed, const int64_t* intervals,
                                   const int64_t* offsets, int n, int64_t max_len, int elt_bytes,
                                   int dir, hipStream_t st) {
  if (n == 0) return 0;
  int gy = (int)((max_len + 256 * 8 - 1) / (256 * 8));
  gy = gy < 1 ? 1 : (gy > 64 ? 64 : gy);
  int gx = n > 4096 ? 4096 : n;
  dim3 grid(gx, gy), block(256);
  if (elt_bytes == 2)
    hipLaunchKernelGGL(interval_copy_kernel<uint16_t>, grid, block, 0, st, (uint16_t*)flat,
                       (uint16_t*)packed, intervals, offsets, n, dir);
  else if (elt_bytes == 4)
    hipLaunchKernelGGL(interval_copy_kernel<uint32_t>, grid, block, 0, st, (uint32_t*)flat,
                       (uint32_t*)packed, intervals, offsets, n, dir);
  else if (elt_bytes == 1)
    hipLaunchKernelGGL(interval_copy_kernel<uint8_t>, grid, block, 0, st, (uint8_t*)flat,
                       (uint8_t*)packed, intervals, offsets, n, dir);
  else
    return -1;
  return (int)hipGetLastError();
}

