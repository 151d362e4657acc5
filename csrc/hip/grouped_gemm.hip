// Grouped GEMM for routed experts on gfx950: tokens are sorted by expert, every expert multiplies its
// contiguous slab of rows with its own weight matrix:
//     out[t, :] = x[t, :] . W[e]^T      for offsets[e] <= t < offsets[e+1]
// x [T, K] bf16 row-major, W [E, N, K] bf16 (K contiguous, the layout of nn.Linear weights), out [T, N].
// Replaces the CUTLASS grouped_gemm the reference's MoE experts call
// (realhf/impl/model/modules/moe/experts.py:195-210).
//
// Structure: one workgroup (4 waves) per output tile of 64 or 128 rows x 64 or 128 columns; a tile never straddles
// two experts (the m-tiles of an expert cover ceil(rows/TM) * TM rows, the tail rows are masked). K is walked in 64-wide
// steps through LDS with register prefetch of the next step (loads issue before the MFMA block, the LDS
// write happens after the barrier). Both operands are K-contiguous, so MFMA fragments are plain 16-byte
// LDS reads; rows are 128 bytes with the chunk swizzle  chunk ^ ((row >> 1) & 7), which makes the 16 rows
// of a ds_read_b128 lane group hit 16 different 16-byte bank slots. The product is computed transposed
// (W rows as the A operand) so that each lane ends up with 4 consecutive output columns of one token and
// stores 8 bytes at a time.
#include <stdlib.h>

#include "flash_tiles.hpp"  // swizzled 256-byte-row LDS tiles + transposed reads (weight-gradient kernel)

DEVINL int gg_off(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

// TM x TN output tile per workgroup (64 or 128 each): the four waves form a 2 x 2 grid, a wave owns a
// (TM/2) x (TN/2) block = MI x NI MFMA tiles of 32 x 32. With 128 x 128 every 16-byte LDS fragment read feeds two
// MFMAs instead of one (64 x 64: the K loop is bound by LDS reads and barriers, not by the matrix cores).
// KH = number of 64-deep sub-tiles staged per step: with 2 a step holds twice the MFMA work between the barrier
// pairs and behind every register prefetch of the next step's global loads.
template <int TM, int TN, int KH>
__global__ __launch_bounds__(256) void grouped_gemm_nt_kernel(
    const bf16_t* __restrict__ x, const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
    const int* __restrict__ offsets, int E, int N, int K, int64_t w_expert_stride) {
  constexpr int MI = TM / 64, NI = TN / 64;
  __shared__ __attribute__((aligned(16))) char smem[KH * (TM + TN) * 128];
  char* as = smem;                  // x tile  [KH][TM rows][64 k]
  char* bs = smem + KH * TM * 128;  // W tile  [KH][TN n][64 k]

  // ---- which expert / m-tile is this block? (tiles are numbered expert by expert)
  int tile = blockIdx.y, e = 0, row_lo = 0, row_hi = 0;
  for (; e < E; ++e) {
    row_lo = offsets[e];
    row_hi = offsets[e + 1];
    const int nt = (row_hi - row_lo + TM - 1) / TM;
    if (tile < nt) break;
    tile -= nt;
  }
  if (e >= E) return;  // beyond the last tile (the grid is an upper bound)
  const int m0 = row_lo + tile * TM;
  const int n0 = blockIdx.x * TN;
  const bf16_t* we = w + (size_t)e * w_expert_stride;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;  // wave -> (TM/2) x (TN/2) block (tokens (TM/2)wm.., columns (TN/2)wn..)

  // cooperative loads: thread -> (row = tid >> 3 (+32 i), chunk = tid & 7)
  const int lrow = tid >> 3, lch = tid & 7;
  bf16x8 areg[KH * TM / 32], breg[KH * TN / 32];
  auto load = [&](int k0) {
#pragma unroll
    for (int h = 0; h < KH; ++h) {
#pragma unroll
      for (int i = 0; i < TM / 32; ++i) {
        int m = m0 + lrow + 32 * i;
        m = m < row_hi ? m : row_hi - 1;
        areg[h * (TM / 32) + i] = ld8(x + (size_t)m * K + k0 + 64 * h + lch * 8);
      }
#pragma unroll
      for (int i = 0; i < TN / 32; ++i)
        breg[h * (TN / 32) + i] = ld8(we + (size_t)(n0 + lrow + 32 * i) * K + k0 + 64 * h + lch * 8);
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int h = 0; h < KH; ++h) {
#pragma unroll
      for (int i = 0; i < TM / 32; ++i)
        *reinterpret_cast<bf16x8*>(as + h * TM * 128 + gg_off(lrow + 32 * i, lch)) = areg[h * (TM / 32) + i];
#pragma unroll
      for (int i = 0; i < TN / 32; ++i)
        *reinterpret_cast<bf16x8*>(bs + h * TN * 128 + gg_off(lrow + 32 * i, lch)) = breg[h * (TN / 32) + i];
    }
  };

  f32x16 acc[MI * NI];
#pragma unroll
  for (int t = 0; t < MI * NI; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  load(0);
  for (int k0 = 0; k0 < K; k0 += 64 * KH) {
    __syncthreads();  // previous tile fully consumed
    stage();
    if (k0 + 64 * KH < K) load(k0 + 64 * KH);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4 * KH; ++s) {
      // A operand = W rows (output columns), B operand = token rows: acc[n][m]
      bf16x8 wf[NI], xf[MI];
      const char* bt = bs + (s >> 2) * TN * 128;
      const char* at = as + (s >> 2) * TM * 128;
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        wf[ni] = *reinterpret_cast<const bf16x8*>(bt + gg_off((TN / 2) * wn + 32 * ni + r, 2 * (s & 3) + hf));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        xf[mi] = *reinterpret_cast<const bf16x8*>(at + gg_off((TM / 2) * wm + 32 * mi + r, 2 * (s & 3) + hf));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ni], xf[mi], acc[mi * NI + ni], 0, 0, 0);
    }
  }
  // ---- epilogue: lane (token = r, hf) holds columns (reg&3) + 8(reg>>2) + 4hf of each 32 x 32 tile
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + (TM / 2) * wm + 32 * mi + r;
    if (m < row_hi) {
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        bf16_t* op = out + (size_t)m * N + n0 + (TN / 2) * wn + 32 * ni;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = f2bf(acc[mi * NI + ni][4 * g + j]);
          *reinterpret_cast<bf16x4*>(op + 8 * g + 4 * hf) = v;
        }
      }
    }
  }
}

// offsets: int32 [E+1] on the device. tile_m: rows per workgroup tile, 64 or 128 (0: 128 when the experts average
// at least 256 rows, else 64 - a tile never straddles two experts, so small experts waste less with 64).
// max_tiles: host-side upper bound of sum_e ceil(rows_e / tile_m) (T / tile_m + E is always enough).
extern "C" int areal_grouped_gemm_tile_m(int T, int E, int N, int tile_m) {
  if (tile_m == 64 || tile_m == 128) return tile_m;
  return (E > 0 && T / E >= 256) ? 128 : 64;
}

extern "C" int areal_grouped_gemm_nt(const void* x, const void* w, void* out, const int* offsets, int T,
                                     int E, int N, int K, int64_t w_expert_stride, int max_tiles, int tile_m,
                                     hipStream_t st) {
  if (T == 0 || E == 0) return 0;
  if (N % 64 != 0 || K % 64 != 0) return -1;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)out) % 16 != 0) return -2;
  if (max_tiles < 1) return -3;
  const int tm = areal_grouped_gemm_tile_m(T, E, N, tile_m);
  if (max_tiles < T / tm + E) return -4;  // the caller sized its bound for another tile height
  const int tn = (N % 128 == 0) ? 128 : 64;
  dim3 grid(N / tn, max_tiles), block(256);
  // 128-deep steps (AREAL_GROUPED_GEMM_KSTEP=128) halve the barrier pairs of the large tile but cost a wave of
  // occupancy per SIMD (3 -> 2): measured 3-15 % slower on the Qwen3-MoE expert shapes, so 64 is the default
  static const int deep = [] {
    const char* e = getenv("AREAL_GROUPED_GEMM_KSTEP");
    return e && atoi(e) == 128;
  }();
#define AREAL_GG_LAUNCH(TM_, TN_, KH_)                                                                    \
  hipLaunchKernelGGL((grouped_gemm_nt_kernel<TM_, TN_, KH_>), grid, block, 0, st, (const bf16_t*)x,      \
                     (const bf16_t*)w, (bf16_t*)out, offsets, E, N, K, w_expert_stride)
  if (tm == 128 && tn == 128) {
    if (deep && K % 128 == 0) AREAL_GG_LAUNCH(128, 128, 2); else AREAL_GG_LAUNCH(128, 128, 1);
  } else if (tm == 128) AREAL_GG_LAUNCH(128, 64, 1);
  else if (tn == 128) AREAL_GG_LAUNCH(64, 128, 1);
  else AREAL_GG_LAUNCH(64, 64, 1);
#undef AREAL_GG_LAUNCH
  return (int)hipGetLastError();
}

// =====================================================================================
// Input gradient of the grouped GEMM without a transposed copy of the weights:
//     dx[t, :] = dy[t, :] . W[e]          (W[e] is [R, C]: R = rows of W = columns of dy, C = columns of dx)
// Same tiling as the forward kernel (TM tokens x 128 output columns per workgroup, reduction in 64-deep steps),
// but the reduction dimension is the SLOW dimension of W: the W tile is staged as [64 r][128 c] in the swizzled
// 256-byte-row layout of flash_tiles.hpp and becomes the A operand through transposed LDS reads of the rows
// 8hf .. 8hf+7 of a 16-deep slab - the order in which the plain 16-byte row read hands the dy fragment (B operand)
// its values.
// =====================================================================================
template <int TM>
__global__ __launch_bounds__(256) void grouped_gemm_nn_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
    const int* __restrict__ offsets, int E, int R, int C, int64_t w_expert_stride) {
  constexpr int MI = TM / 64;
  __shared__ __attribute__((aligned(16))) char smem[TM * 128 + 64 * 256];
  char* as = smem;             // dy tile [TM tokens][64 r], 128-byte rows (gg_off)
  char* ws = smem + TM * 128;  // W tile  [64 r][128 c], 256-byte rows (swz_off)

  int tile = blockIdx.y, e = 0, row_lo = 0, row_hi = 0;
  for (; e < E; ++e) {
    row_lo = offsets[e];
    row_hi = offsets[e + 1];
    const int nt = (row_hi - row_lo + TM - 1) / TM;
    if (tile < nt) break;
    tile -= nt;
  }
  if (e >= E) return;
  const int m0 = row_lo + tile * TM;
  const int c0 = blockIdx.x * 128;
  const bf16_t* we = w + (size_t)e * w_expert_stride;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5, gl = lane & 15, grp = lane >> 4;
  const int wm = wid >> 1, wn = wid & 1;  // wave -> tokens (TM/2) wm .., output columns 64 wn ..

  const int arow = tid >> 3, ach = tid & 7;    // dy tile: (token arow + 32 i, 16-byte chunk ach)
  const int wrow = tid >> 4, wch = tid & 15;   // W tile: (r row wrow + 16 i, 16-byte chunk wch)
  bf16x8 areg[TM / 32], wreg[4];
  auto load = [&](int r0) {
#pragma unroll
    for (int i = 0; i < TM / 32; ++i) {
      int m = m0 + arow + 32 * i;
      m = m < row_hi ? m : row_hi - 1;
      areg[i] = ld8(dy + (size_t)m * R + r0 + ach * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) wreg[i] = ld8(we + (size_t)(r0 + wrow + 16 * i) * C + c0 + wch * 8);
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < TM / 32; ++i) *reinterpret_cast<bf16x8*>(as + gg_off(arow + 32 * i, ach)) = areg[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) lds_row_write(ws, wrow + 16 * i, wch, wreg[i]);
  };
  f32x16 acc[MI * 2];
#pragma unroll
  for (int t = 0; t < MI * 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  // LDS addresses of the fragment reads are the same every step: computed once, outside the loop
  const char* wa[4][2][2];
  const char* xa[4][MI];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int col = 64 * wn + 32 * ni + 16 * (grp & 1);
      wa[s][ni][0] = tr_addr(ws, 16 * s + 8 * hf, col, gl);
      wa[s][ni][1] = tr_addr(ws, 16 * s + 8 * hf + 4, col, gl);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xa[s][mi] = as + gg_off((TM / 2) * wm + 32 * mi + r, 2 * s + hf);
  }

  load(0);
  for (int r0 = 0; r0 < R; r0 += 64) {
    __syncthreads();
    stage();
    if (r0 + 64 < R) load(r0 + 64);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 wf[2], xf[MI];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) wf[ni] = cat4(tr_load(wa[s][ni][0]), tr_load(wa[s][ni][1]));
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[mi] = *reinterpret_cast<const bf16x8*>(xa[s][mi]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi * 2 + ni] = mfma32(wf[ni], xf[mi], acc[mi * 2 + ni]);
    }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + (TM / 2) * wm + 32 * mi + r;
    if (m < row_hi) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        bf16_t* op = out + (size_t)m * C + c0 + 64 * wn + 32 * ni;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          bf16x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = f2bf(acc[mi * 2 + ni][4 * g + j]);
          *reinterpret_cast<bf16x4*>(op + 8 * g + 4 * hf) = v;
        }
      }
    }
  }
}

// dy [T, R], w [E, R, C] (C contiguous), out [T, C]; R a multiple of 64, C a multiple of 128.
extern "C" int areal_grouped_gemm_nn(const void* dy, const void* w, void* out, const int* offsets, int T, int E,
                                     int R, int C, int64_t w_expert_stride, int max_tiles, int tile_m,
                                     hipStream_t st) {
  if (T == 0 || E == 0) return 0;
  if (R % 64 != 0 || C % 128 != 0) return -1;
  if (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)out) % 16 != 0) return -2;
  const int tm = areal_grouped_gemm_tile_m(T, E, C, tile_m);
  if (max_tiles < T / tm + E) return -4;
  dim3 grid(C / 128, max_tiles), block(256);
  if (tm == 128)
    hipLaunchKernelGGL(grouped_gemm_nn_kernel<128>, grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)w,
                       (bf16_t*)out, offsets, E, R, C, w_expert_stride);
  else
    hipLaunchKernelGGL(grouped_gemm_nn_kernel<64>, grid, block, 0, st, (const bf16_t*)dy, (const bf16_t*)w,
                       (bf16_t*)out, offsets, E, R, C, w_expert_stride);
  return (int)hipGetLastError();
}

// =====================================================================================
// Weight gradient of the grouped GEMM:  dW[e][n][k] = sum over the rows t of expert e of dy[t][n] * x[t][k].
// One launch for all experts (a loop of library GEMMs is launch-bound for many small experts). Grid
// (K/128, N/128 or N/256, E); workgroup = 128 x 128 or 256 x 128 tile of dW[e], four waves 2 x 2.
// The reduction runs over TOKENS, the slow dimension of both operands, so both MFMA operands are transposed LDS
// reads (ds_read_b64_tr_b16) of token-major tiles [64 tokens][128 columns] in the swizzled layout of
// flash_tiles.hpp; rows past the expert's last token are staged as zeros. fp32 accumulation, one bf16 store.
// =====================================================================================
// OUT = bf16_t: dW is stored; OUT = float: dW is ADDED to an fp32 accumulator of the same shape (gradient
// accumulation over micro-batches straight into the optimizer's buffer, every element touched by one workgroup).
template <int NS, typename OUT>  // the tile holds NS x 128 rows of dW (columns of dy): 128 x 128 or 256 x 128
__global__ __launch_bounds__(256) void grouped_gemm_wgrad_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, OUT* __restrict__ dw,
    const int* __restrict__ offsets, int N, int K, int64_t dw_expert_stride) {
  constexpr int YT = 64 * 256;  // bytes of one [64 tokens][128 columns] sub-tile
  __shared__ __attribute__((aligned(16))) char smem[(NS + 1) * YT];
  char* ys = smem;            // dy tile [NS][64 tokens][128 n]
  char* xs = smem + NS * YT;  // x tile  [64 tokens][128 k]
  const int e = blockIdx.z;
  const int row_lo = offsets[e], row_hi = offsets[e + 1];
  const int k0 = blockIdx.x * 128, n0 = blockIdx.y * 128 * NS;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, hf = lane >> 5, gl = lane & 15, grp = lane >> 4;
  const int wn = wid >> 1, wk = wid & 1;  // wave -> rows 64 NS wn .. of the tile, k columns 64 wk ..

  const int lrow = tid >> 4, lch = tid & 15;  // thread -> (token row lrow + 16 i, 16-byte chunk lch)
  bf16x8 yreg[4 * NS], xreg[4];
  auto load = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int t = t0 + lrow + 16 * i;
      const bool ok = t < row_hi;
      const int tc = ok ? t : row_hi - 1;
      bf16x8 b = ld8(x + (size_t)tc * K + k0 + lch * 8);
      bf16x8 a[NS];
#pragma unroll
      for (int h = 0; h < NS; ++h) a[h] = ld8(dy + (size_t)tc * N + n0 + 128 * h + lch * 8);
      if (!ok) {  // rows past the expert's last token contribute nothing
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          b[j] = (bf16_t)0.f;
#pragma unroll
          for (int h = 0; h < NS; ++h) a[h][j] = (bf16_t)0.f;
        }
      }
      xreg[i] = b;
#pragma unroll
      for (int h = 0; h < NS; ++h) yreg[NS * i + h] = a[h];
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      lds_row_write(xs, lrow + 16 * i, lch, xreg[i]);
#pragma unroll
      for (int h = 0; h < NS; ++h) lds_row_write(ys + h * YT, lrow + 16 * i, lch, yreg[NS * i + h]);
    }
  };
  constexpr int NI = 2 * NS;  // 32-row MFMA tiles of the wave along n
  f32x16 acc[NI * 2];
#pragma unroll
  for (int t = 0; t < NI * 2; ++t)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;

  // LDS addresses of the transposed fragment reads: the same every step, computed once (the swizzle costs ~10
  // vector ALU operations per address, more than the MFMAs they feed if redone inside the loop)
  const char* ya[4][NI][2];
  const char* xa[4][2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int col = 64 * NS * wn + 32 * i;  // row of the tile = column of dy, 0 .. 128 NS
      const int c0 = (col & 127) + 16 * (grp & 1);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) ya[s][i][h2] = tr_addr(ys + (col >> 7) * YT, 16 * s + 8 * h2 + 4 * hf, c0, gl);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c0 = 64 * wk + 32 * i + 16 * (grp & 1);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) xa[s][i][h2] = tr_addr(xs, 16 * s + 8 * h2 + 4 * hf, c0, gl);
    }
  }

  if (row_lo < row_hi) load(row_lo);
  for (int t0 = row_lo; t0 < row_hi; t0 += 64) {
    __syncthreads();  // previous tile fully consumed
    stage();
    if (t0 + 64 < row_hi) load(t0 + 64);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 4; ++s) {  // 16-token slabs
      bf16x8 af[NI], bf[2];
#pragma unroll
      for (int i = 0; i < NI; ++i) af[i] = cat4(tr_load(ya[s][i][0]), tr_load(ya[s][i][1]));
#pragma unroll
      for (int i = 0; i < 2; ++i) bf[i] = cat4(tr_load(xa[s][i][0]), tr_load(xa[s][i][1]));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int ki = 0; ki < 2; ++ki) acc[2 * ni + ki] = mfma32(af[ni], bf[ki], acc[2 * ni + ki]);
    }
  }
  // ---- epilogue: lane (k column = r, hf), reg -> n row (reg&3) + 8(reg>>2) + 4hf of each 32 x 32 tile
  OUT* de = dw + (size_t)e * dw_expert_stride;
  if (sizeof(OUT) == 4 && row_lo >= row_hi) return;  // accumulating: an expert without rows adds nothing
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int ki = 0; ki < 2; ++ki) {
      OUT* op = de + (size_t)(n0 + 64 * NS * wn + 32 * ni + 4 * hf) * K + k0 + 64 * wk + 32 * ki + r;
      if constexpr (sizeof(OUT) == 4) {
        float old[16];  // all sixteen loads in flight before the first store (the compiler cannot reorder them itself)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) old[reg] = op[(size_t)((reg & 3) + 8 * (reg >> 2)) * K];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          op[(size_t)((reg & 3) + 8 * (reg >> 2)) * K] = old[reg] + acc[2 * ni + ki][reg];
      } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg)
          op[(size_t)((reg & 3) + 8 * (reg >> 2)) * K] = f2bf(acc[2 * ni + ki][reg]);
      }
    }
}

// dy [T, N], x [T, K] bf16 row-major (rows sorted by expert); N and K multiples of 128. accumulate = 0: dw [E, N, K]
// bf16, every element is written (experts without rows get zeros); accumulate = 1: dw [E, N, K] fp32, dW is added.
extern "C" int areal_grouped_gemm_wgrad(const void* dy, const void* x, void* dw, const int* offsets, int T, int E,
                                        int N, int K, int64_t dw_expert_stride, int accumulate, hipStream_t st) {
  if (E == 0) return 0;
  if (N % 128 != 0 || K % 128 != 0) return -1;
  if (((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dw) % 16 != 0) return -2;
  if (E > 65535 || N / 128 > 65535) return -3;
  dim3 block(256);
  // 256 x 128 tiles feed four MFMAs per transposed x fragment but hold one wave per SIMD (282 registers); the
  // 128 x 128 tile runs two and measured 3 % faster on the Qwen3-MoE shapes. AREAL_GROUPED_WGRAD_TALL=1 selects it.
  static const bool want_tall = [] { const char* e = getenv("AREAL_GROUPED_WGRAD_TALL"); return e && e[0] == '1'; }();
  const bool tall = want_tall && N % 256 == 0;
  dim3 grid(K / 128, tall ? N / 256 : N / 128, E);
#define AREAL_GG_WGRAD(NS_, OUT_)                                                                            \
  hipLaunchKernelGGL((grouped_gemm_wgrad_kernel<NS_, OUT_>), grid, block, 0, st, (const bf16_t*)dy,         \
                     (const bf16_t*)x, (OUT_*)dw, offsets, N, K, dw_expert_stride)
  if (accumulate) {
    if (tall) AREAL_GG_WGRAD(2, float); else AREAL_GG_WGRAD(1, float);
  } else {
    if (tall) AREAL_GG_WGRAD(2, bf16_t); else AREAL_GG_WGRAD(1, bf16_t);
  }
#undef AREAL_GG_WGRAD
  return (int)hipGetLastError();
}
