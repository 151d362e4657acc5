// Grouped GEMM for routed experts on gfx950: tokens are sorted by expert, every expert multiplies its
// contiguous slab of rows with its own weight matrix:
//     out[t, :] = x[t, :] . W[e]^T      for offsets[e] <= t < offsets[e+1]
// x [T, K] bf16 row-major, W [E, N, K] bf16 (K contiguous, the layout of nn.Linear weights), out [T, N].
// Replaces the CUTLASS grouped_gemm the reference's MoE experts call
// (realhf/impl/model/modules/moe/experts.py:195-210).
//
// Structure: one workgroup (4 waves) per 64 x 64 output tile; a tile never straddles two experts (the
// m-tiles of an expert cover ceil(rows/64) * 64 rows, the tail rows are masked). K is walked in 64-wide
// steps through LDS with register prefetch of the next step (loads issue before the MFMA block, the LDS
// write happens after the barrier). Both operands are K-contiguous, so MFMA fragments are plain 16-byte
// LDS reads; rows are 128 bytes with the chunk swizzle  chunk ^ ((row >> 1) & 7), which makes the 16 rows
// of a ds_read_b128 lane group hit 16 different 16-byte bank slots. The product is computed transposed
// (W rows as the A operand) so that each lane ends up with 4 consecutive output columns of one token and
// stores 8 bytes at a time.
#include "common.hpp"

DEVINL int gg_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

__global__ __launch_bounds__(256) void grouped_gemm_nt_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
    const int* __restrict__ offsets, int E, int N, int K, int64_t w_expert_stride) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * 128];
  char* as = smem;             // x tile  [64 rows][64 k]
  char* bs = smem + 64 * 128;  // W tile  [64 n][64 k]

  // ---- which expert / m-tile is this block? (tiles are numbered expert by expert)
  int tile = blockIdx.y, e = 0, row_lo = 0, row_hi = 0;
  for (; e < E; ++e) {
    row_lo = offsets[e];
    row_hi = offsets[e + 1];
    const int nt = (row_hi - row_lo + 63) >> 6;
    if (tile < nt) break;
    tile -= nt;
  }
  if (e >= E) return;  // beyond the last tile (the grid is an upper bound)
  const int m0 = row_lo + tile * 64;
  const int n0 = blockIdx.x * 64;
  const bf16_t* we = w + (size_t)e * w_expert_stride;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;  // wave -> 32 x 32 sub-tile (tokens 32wm.., columns 32wn..)

  // cooperative loads: thread -> (row = tid >> 3 (+32), chunk = tid & 7)
  const int lrow = tid >> 3, lch = tid & 7;
  bf16x8 areg[2], breg[2];
  auto load = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int m = m0 + lrow + 32 * i;
      m = m < row_hi ? m : row_hi - 1;
      areg[i] = ld8(x + (size_t)m * K + k0 + lch * 8);
      breg[i] = ld8(we + (size_t)(n0 + lrow + 32 * i) * K + k0 + lch * 8);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<bf16x8*>(as + gg_off(lrow + 32 * i, lch)) = areg[i];
      *reinterpret_cast<bf16x8*>(bs + gg_off(lrow + 32 * i, lch)) = breg[i];
    }
  };

  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 16; ++j) acc[j] = 0.f;

  load(0);
  for (int k0 = 0; k0 < K; k0 += 64) {
    __syncthreads();  // previous tile fully consumed
    stage();
    if (k0 + 64 < K) load(k0 + 64);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // A operand = W rows (output columns), B operand = token rows: acc[n][m]
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(bs + gg_off(32 * wn + r, 2 * s + hf));
      const bf16x8 xf = *reinterpret_cast<const bf16x8*>(as + gg_off(32 * wm + r, 2 * s + hf));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc, 0, 0, 0);
    }
  }
  // ---- epilogue: lane (token = r, hf) holds columns (reg&3) + 8(reg>>2) + 4hf of the wave's block
  const int m = m0 + 32 * wm + r;
  if (m < row_hi) {
    bf16_t* op = out + (size_t)m * N + n0 + 32 * wn;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      bf16x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = f2bf(acc[4 * g + j]);
      *reinterpret_cast<bf16x4*>(op + 8 * g + 4 * hf) = v;
    }
  }
}

// offsets: int32 [E+1] on the device; max_tiles: host-side upper bound of sum_e ceil(rows_e / 64)
// (T / 64 + E is always enough).
extern "C" int areal_grouped_gemm_nt(const void* x, const void* w, void* out, const int* offsets, int T,
                                     int E, int N, int K, int64_t w_expert_stride, int max_tiles,
                                     hipStream_t st) {
  if (T == 0 || E == 0) return 0;
  if (N % 64 != 0 || K % 64 != 0) return -1;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) % 16 != 0) return -2;
  if (max_tiles < 1) return -3;
  dim3 grid(N / 64, max_tiles), block(256);
  hipLaunchKernelGGL(grouped_gemm_nt_kernel, grid, block, 0, st, (const bf16_t*)x, (const bf16_t*)w,
                     (bf16_t*)out, offsets, E, N, K, w_expert_stride);
  return (int)hipGetLastError();
}
