// Dense bf16 GEMM of the trainer and of prefill on gfx950:
//
//     C[M, N] = A[M, K] . B[N, K]^T (+ bias[N])          bf16 operands, fp32 accumulation, bf16 result
//
// Both operands are K-contiguous: A = activations, B = an nn.Linear weight (forward) or its transposed copy
// (input gradient). Role in the reference: the cuBLAS GEMMs behind torch.nn.functional.linear in
// realhf/impl/model/parallelism/tensor_parallel/modules.py:242-402 and the HF model code.
//
// Structure (one workgroup = one 256 x 256 output tile, 8 waves as 2 (M) x 4 (N), a wave owns 128 x 64):
//   * K is walked in 64-deep tiles through two LDS buffers of 64 KiB. A tile is staged as four 16-KiB PIECES
//     (128 rows x 128 B each), chosen by WHEN they are read, not by where they sit in the matrix:
//       A0 = rows 0-63 of both wave rows, A1 = rows 64-127 of both wave rows,
//       B0 = columns 0-31 of all four wave columns, B1 = columns 32-63 of all four wave columns.
//   * a K tile is four phases, each one quadrant (64 x 32) of the wave's output = 16 MFMA 16x16x32:
//       phase 1 reads B0 + A0 (12 ds_read_b128), phase 2 reads B1 (4), phase 3 reads A1 (8), phase 4 reads nothing.
//     Every phase also issues the LDS-DMA (global_load_lds_dwordx4, two per thread) of ONE piece, as soon as the piece
//     it overwrites has been read by everyone:
//       phase 1 -> A1 of tile t+1, phase 2 -> B0 of t+2, phase 3 -> A0 of t+2, phase 4 -> B1 of t+2.
//     One counted wait per tile (phase 4: vmcnt(6) = the three newest pieces may still be in flight) makes tile t+1
//     complete; it is read from the next phase on. Loads are never drained inside the loop.
//   * raw s_barrier pairs around each MFMA cluster (s_setprio keeps hipcc from moving MFMAs across them); with
//     STAGGER the waves of the second wave row run one barrier behind the first, so that one group's LDS reads and DMA
//     issue overlap the other group's MFMA cluster (the two groups share every SIMD).
//   * LDS rows are 128 B; the XOR swizzle chunk ^ ((row >> 1) & 7) makes every 16-lane group of a ds_read_b128 hit 16
//     different 16-byte slots. The DMA writes LDS linearly (wave base + lane x 16), so the swizzle sits on the per-lane
//     SOURCE address and on the read address.
//   * the product is computed transposed (B rows as the MFMA A operand): a lane ends up with 4 consecutive output
//     columns of one row and stores 8 bytes.
//   * block id -> tile: contiguous id ranges per XCD (ids are dealt round-robin to the 8 XCDs), walked in bands of 4
//     tile rows so that the blocks running together on an XCD share A and B panels in its L2.
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "flash_tiles.hpp"  // swizzled 256-byte-row LDS tiles + transposed reads (weight-gradient kernel)

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

struct TGArgs {
  const bf16_t* A;
  long lda;
  const bf16_t* B;
  long ldb;
  bf16_t* C;
  long ldc;
  const bf16_t* bias;
  int M, N, K;
  int tiles_m, tiles_n;
};

constexpr int TG_PIECE = 128 * 128;   // bytes of one staged piece
constexpr int TG_BUF = 4 * TG_PIECE;  // A0 A1 B0 B1 of one K tile
constexpr int TG_LDS = 2 * TG_BUF;
enum { PA0 = 0, PA1 = 1, PB0 = 2, PB1 = 3 };
constexpr int TG_GROUP_M = 4;

template <int N>
DEVINL void tg_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
DEVINL void tg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

typedef bf16x8 tg_bfrag[2][2];  // [n tile][k step]
typedef f32x4 tg_acc[4][2];     // [m tile][n tile] of one quadrant

// EPI 0: C = A.B^T (+ bias), output tiles of 256 x 256. EPI 1 (gated MLP, no-grad forward): B = [gate rows; up rows]
// of width N each, C[M, N] = silu(A.gate^T) * (A.up^T); an output tile is 256 rows x 128 columns, its B pieces hold
// the 128 gate rows (read in the phases of the first column half) and the 128 matching up rows (second half), so a
// lane finds gate and up of the same output element in the same registers of the two accumulator quadrants.
template <bool STAGGER, int EPI>
__global__ __launch_bounds__(512) void tgemm_tn_kernel(const TGArgs a) {
  constexpr int TN = EPI == 1 ? 128 : 256;  // output columns per tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- tile sequence number -> output tile
  const int ntiles = a.tiles_m * a.tiles_n;
  auto tile_origin = [&](int id, int& m0, int& n0) __attribute__((always_inline)) {
    const int q = ntiles >> 3, r = ntiles & 7, x = id & 7;
    const int wg = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    const int per_band = TG_GROUP_M * a.tiles_n;
    const int band = wg / per_band, first = band * TG_GROUP_M;
    const int rows = min(a.tiles_m - first, TG_GROUP_M);
    const int in_band = wg - band * per_band;
    m0 = (first + in_band % rows) * 256;
    n0 = (in_band / rows) * TN;
  };
  const int KT = a.K >> 6;

  // ---- staging: thread -> row (wave * 8 + lane / 8) of a 64-row issue, physical chunk lane & 7
  const int srow = wave * 8 + (lane >> 3);
  const int schunk = (lane & 7) ^ ((srow >> 1) & 7);  // logical chunk whose data lands at physical chunk lane & 7
  const bf16_t* srcA[2][2];  // [piece][issue]
  const bf16_t* srcB[2][2];
  auto set_sources = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int row = m0 + i * 128 + h * 64 + srow;
        row = row < a.M ? row : a.M - 1;
        srcA[h][i] = a.A + (size_t)row * a.lda + schunk * 8;
        int col;
        if (EPI == 1) {  // piece h: gate (0) or up (1) rows of the tile's 128 columns, 32 per wave column
          col = n0 + (i * 2 + (wave >> 2)) * 32 + (wave & 3) * 8 + (lane >> 3);
          col = (col < a.N ? col : a.N - 1) + h * a.N;
        } else {
          col = n0 + (i * 2 + (wave >> 2)) * 64 + h * 32 + (wave & 3) * 8 + (lane >> 3);
          col = col < a.N ? col : a.N - 1;
        }
        srcB[h][i] = a.B + (size_t)col * a.ldb + schunk * 8;
      }
  };
  auto stage = [&](int buf, int piece, int kt) __attribute__((always_inline)) {
    char* dst = smem + buf * TG_BUF + piece * TG_PIECE + wave * 1024;
    const bf16_t* s0 = piece < 2 ? srcA[piece][0] : srcB[piece - 2][0];
    const bf16_t* s1 = piece < 2 ? srcA[piece][1] : srcB[piece - 2][1];
    __builtin_amdgcn_global_load_lds((gptr_t)(s0 + (size_t)kt * 64), (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(s1 + (size_t)kt * 64), (lptr_t)(dst + 8192), 16, 0, 0);
  };

  // ---- fragment reads
  const int fr = lane & 15, fq = lane >> 4;
  const int sw = (fr >> 1) & 7;
  const int cb0 = ((fq) ^ sw) << 4, cb1 = ((4 + fq) ^ sw) << 4;
  const int a_off = (wr * 64 + fr) * 128, b_off = (wc * 32 + fr) * 128;
  bf16x8 af[4][2];
  tg_bfrag bf0, bf1;
  auto read_a = [&](int buf, int h) __attribute__((always_inline)) {
    const char* p = smem + buf * TG_BUF + h * TG_PIECE + a_off;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      af[mt][0] = *reinterpret_cast<const bf16x8*>(p + mt * 2048 + cb0);
      af[mt][1] = *reinterpret_cast<const bf16x8*>(p + mt * 2048 + cb1);
    }
  };
  auto read_b = [&](int buf, int h, tg_bfrag& bf) __attribute__((always_inline)) {
    const char* p = smem + buf * TG_BUF + (2 + h) * TG_PIECE + b_off;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      bf[nt][0] = *reinterpret_cast<const bf16x8*>(p + nt * 2048 + cb0);
      bf[nt][1] = *reinterpret_cast<const bf16x8*>(p + nt * 2048 + cb1);
    }
  };
  tg_acc acc00, acc01, acc10, acc11;  // [m half][n half]
  auto mma = [&](tg_acc& acc, tg_bfrag& bf) __attribute__((always_inline)) {
    tg_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[nt][ks], af[mt][ks], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    tg_barrier();
  };

  // ---- result stores. Lane (row fr, column quad fq) of tile (mt, nt) holds 4 consecutive columns; the two 16-column
  // tiles of a quadrant exchange halves between neighbouring 16-lane rows (v_permlane16_swap), after which a lane holds
  // 8 consecutive columns of one of them: 16-byte stores, the four lanes of a row write 64 contiguous bytes.
  auto put = [&](tg_acc& acc, int m0, int n0, int mh, int nh) __attribute__((always_inline)) {
    const int n = n0 + wc * 64 + nh * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wr * 128 + mh * 64 + mt * 16 + fr;
      unsigned pk[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 v = acc[mt][nt];
        if (a.bias) {
          const int nb = n0 + wc * 64 + nh * 32 + nt * 16 + fq * 4;
          if (nb < a.N) {
            const bf16x4 bb = *reinterpret_cast<const bf16x4*>(a.bias + nb);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += bf2f(bb[j]);
          }
        }
        bf16x2 lo = {f2bf(v[0]), f2bf(v[1])}, hi = {f2bf(v[2]), f2bf(v[3])};
        pk[nt][0] = __builtin_bit_cast(unsigned, lo);
        pk[nt][1] = __builtin_bit_cast(unsigned, hi);
      }
      const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
      if (m < a.M && n < a.N) {
        if (n + 8 <= a.N) {
          *reinterpret_cast<u32x4*>(a.C + (size_t)m * a.ldc + n) = o;
        } else {  // the last 4 columns of a width that is not a multiple of 8
          *reinterpret_cast<u32x2*>(a.C + (size_t)m * a.ldc + n) = (u32x2){o[0], o[1]};
        }
      }
    }
  };
  // gated form: `g` and `u` are the two column quadrants of one row half
  auto put_gated = [&](tg_acc& g, tg_acc& u, int m0, int n0, int mh) __attribute__((always_inline)) {
    const int n = n0 + wc * 32 + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const int m = m0 + wr * 128 + mh * 64 + mt * 16 + fr;
      unsigned pk[2][2];
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          // the unfused path carries gate|up as a bf16 tensor
          const float gg = bf2f(f2bf(g[mt][nt][j])), uu = bf2f(f2bf(u[mt][nt][j]));
          v[j] = gg / (1.f + __expf(-gg)) * uu;
        }
        bf16x2 lo = {f2bf(v[0]), f2bf(v[1])}, hi = {f2bf(v[2]), f2bf(v[3])};
        pk[nt][0] = __builtin_bit_cast(unsigned, lo);
        pk[nt][1] = __builtin_bit_cast(unsigned, hi);
      }
      const auto r0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
      const auto r1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
      const u32x4 o = {r0[0], r1[0], r0[1], r1[1]};
      if (m < a.M && n < a.N) {
        if (n + 8 <= a.N) {
          *reinterpret_cast<u32x4*>(a.C + (size_t)m * a.ldc + n) = o;
        } else {
          *reinterpret_cast<u32x2*>(a.C + (size_t)m * a.ldc + n) = (u32x2){o[0], o[1]};
        }
      }
    }
  };
  int cm0 = 0, cn0 = 0;  // origin of the output tile being computed

  // One K tile = four phases. The K tiles of the block's successive output tiles form ONE stream: the pieces staged
  // while the last two K tiles of an output tile are computed are K tiles 0 and 1 of the NEXT output tile, so the
  // prologue's load latency is paid once per block and the result stores run under the next tile's loads.
  // MODE 0: inside an output tile (K tiles t+1, t+2 belong to it). MODE 1: second to last K tile (t+1 is the last one;
  // "t+2" is K tile 0 of the next output tile, whose source addresses replace the current ones after phase 1).
  // MODE 2: last K tile ("t+1", "t+2" are K tiles 0, 1 of the next output tile). Without a next tile nothing is staged
  // and the loads still in flight are drained.
  int nm0 = 0, nn0 = 0;
  bool has_next = false;
  auto tile = [&](auto buf_c, int t, auto mode_c) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_c)::value, MODE = decltype(mode_c)::value;
    const bool go = MODE == 0 || has_next;
    const int t1 = MODE == 2 ? 0 : t + 1, t2 = MODE == 0 ? t + 2 : MODE - 1;
    // phase 1: quadrant (0, 0)
    read_b(buf, 0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf, 0);
    if (MODE < 2 || go) stage(buf ^ 1, PA1, t1);
    // the four B0 reads (issued first) have returned before this phase's first barrier: B0 is restaged one phase
    // from now, when the staggered wave group has only just passed that barrier
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    mma(acc00, bf0);
    // last K tile: a quadrant is final once its phase is over; its stores are issued while the other wave group
    // computes
    if (MODE == 2 && EPI == 0) put(acc00, cm0, cn0, 0, 0);
    if (MODE == 1 && go) set_sources(nm0, nn0);
    // phase 2: quadrant (0, 1)
    read_b(buf, 1, bf1);
    if (go) stage(buf, PB0, t2);
    mma(acc01, bf1);
    if (MODE == 2 && EPI == 0) put(acc01, cm0, cn0, 0, 1);
    if (MODE == 2 && EPI == 1) put_gated(acc00, acc01, cm0, cn0, 0);
    // phase 3: quadrant (1, 1)
    read_a(buf, 1);
    if (go) stage(buf, PA0, t2);
    mma(acc11, bf1);
    if (MODE == 2 && EPI == 0) put(acc11, cm0, cn0, 1, 1);
    // phase 4: quadrant (1, 0); the next K tile of the stream becomes readable
    if (go) {
      stage(buf, PB1, t2);
      tg_vmcnt<6>();
    } else if (MODE == 1) {
      tg_vmcnt<0>();
    }
    mma(acc10, bf0);
    if (MODE == 2 && EPI == 0) put(acc10, cm0, cn0, 1, 0);
    if (MODE == 2 && EPI == 1) put_gated(acc10, acc11, cm0, cn0, 1);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>;

  // ---- prologue: K tile 0 whole, B0 A0 B1 of K tile 1 (its A1 leaves in phase 1 of K tile 0)
  int id = blockIdx.x, m0, n0;
  tile_origin(id, m0, n0);
  set_sources(m0, n0);
  stage(0, PB0, 0);
  stage(0, PA0, 0);
  stage(0, PB1, 0);
  stage(0, PA1, 0);
  stage(1, PB0, 1);
  stage(1, PA0, 1);
  stage(1, PB1, 1);
  tg_vmcnt<6>();
  tg_barrier();
  if (STAGGER && wr == 1) tg_barrier();

  while (true) {
    const int nid = id + gridDim.x;
    has_next = nid < ntiles;
    if (has_next) tile_origin(nid, nm0, nn0);
    cm0 = m0;
    cn0 = n0;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        acc00[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc01[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc10[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc11[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    // K tiles come in pairs (K is a multiple of 128)
    int t = 0;
    for (; t + 2 < KT; t += 2) {
      tile(B0{}, t, M0{});
      tile(B1{}, t + 1, M0{});
    }
    tile(B0{}, t, M1{});
    tile(B1{}, t + 1, M2{});
    if (!has_next) break;
    id = nid;
    m0 = nm0;
    n0 = nn0;
  }
  if (STAGGER && wr == 0) tg_barrier();
}

// =====================================================================================
// Weight gradient:  dW[N, K] (fp32) += dY[M, N]^T . X[M, K]     (bf16 operands, both token-major)
//
// The same phase structure as above with the roles  rows -> n (columns of dY), columns -> k (columns of X), reduction
// -> tokens. Both operands are stored with the REDUCTION index as the slow dimension, so the staged pieces are
// token-major [64 tokens][128 columns] (256-byte rows, the swizzle of flash_tiles.hpp) and every MFMA operand is a pair
// of transposed LDS reads (ds_read_b64_tr_b16: a lane receives 4 consecutive tokens of one column). 32x32x16 MFMA:
// its transposed-read pattern (4 token rows x 64 contiguous bytes per 32 lanes) is conflict-free in that layout, and
// a lane ends up with ONE k column per register, so a register is stored (or added) as 128 contiguous bytes per row.
//   A0 / A1 = dY columns [0, 64) / [64, 128) of both wave rows; B0 / B1 = X columns [0, 32) / [32, 64) of all four
//   wave columns - as for the forward kernel, pieces are what one phase reads.
// grid = tiles x splits: a workgroup reduces `chunk` 128-token pairs of one output tile. A tile whose whole reduction
// falls to one workgroup is added with a plain read-add-write; partial reductions are added with fp32 atomics (they
// execute at the memory side, ~1.3 TB/s chip-wide), which is how a gradient of 48 tiles is reduced by four or five
// workgroups per tile.
// =====================================================================================
struct TWArgs {
  const bf16_t* dY;
  long ldy;
  const bf16_t* X;
  long ldx;
  float* dW;
  long ldw;
  int M, N, K;
  int tiles_n, tiles_k;
  int pairs;          // 128-token pairs of K tiles in M
  int whole;          // the first `whole` tiles are reduced by one workgroup each
  int splits, chunk;  // the others: workgroups per tile, pairs per workgroup (the last one of a tile may get fewer)
};

typedef f32x16 tw_acc[2];  // [32-row n tile] of one quadrant (64 n x 32 k)

template <bool STAGGER>
__global__ __launch_bounds__(512) void tgemm_wgrad_kernel(const TWArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // ---- this workgroup's range of work units (tile, pair): `chunk` pairs of one tile. Ids are dealt round-robin to
  // the 8 XCDs: the workgroups of one XCD take neighbouring tiles, which share dY / X panels in its L2.
  // (whole tiles and split tiles are renumbered separately: the hardware starts workgroups in id order, and the long
  // whole-tile workgroups have to start first; `whole` is 0, a multiple of 256, or the whole grid)
  const bool in_whole = (int)blockIdx.x < a.whole;
  const int base = in_whole ? 0 : a.whole, nb = in_whole ? a.whole : (int)gridDim.x - a.whole;
  const int id = blockIdx.x - base;
  const int xq = nb >> 3, xr = nb & 7, xx = id & 7;
  const int b = base + (xx < xr ? xx * (xq + 1) : xr * (xq + 1) + (xx - xr) * xq) + (id >> 3);
  int b_tile = b, b_split = 0, b_chunk = a.pairs;
  if (b >= a.whole) {
    b_tile = a.whole + (b - a.whole) / a.splits;
    b_split = (b - a.whole) - (b_tile - a.whole) * a.splits;
    b_chunk = a.chunk;
  }
  int w = b_tile * a.pairs + b_split * b_chunk;
  const int w_end = min(w + b_chunk, (b_tile + 1) * a.pairs);
  auto tile_origin = [&](int wg, int& n0, int& k0) __attribute__((always_inline)) {
    const int per_band = TG_GROUP_M * a.tiles_k;
    const int band = wg / per_band, first = band * TG_GROUP_M;
    const int rows = min(a.tiles_n - first, TG_GROUP_M);
    const int in_band = wg - band * per_band;
    n0 = (first + in_band % rows) * 256;
    k0 = (in_band / rows) * 256;
  };

  // ---- staging: thread -> token row (wave * 4 + lane / 16) of a 32-row issue, physical 16-byte chunk lane & 15
  const int srow = wave * 4 + (lane >> 4);
  const int pc = ((lane & 15) ^ (((lane >> 4) << 2) | (wave & 3))) * 8;  // piece column whose chunk lands here
  const bf16_t *pA = nullptr, *pB = nullptr;  // set per segment
  auto stage = [&](int buf, int piece, int kt) __attribute__((always_inline)) {
    char* dst = smem + buf * TG_BUF + piece * TG_PIECE + wave * 1024;
    const bool isA = piece < 2;
    const long ld = isA ? a.ldy : a.ldx;
    const bf16_t* s0 = (isA ? pA + (piece & 1) * 64 : pB + (piece & 1) * 32) + (size_t)kt * 64 * ld;
    __builtin_amdgcn_global_load_lds((gptr_t)s0, (lptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(s0 + 32 * ld), (lptr_t)(dst + 8192), 16, 0, 0);
  };

  // ---- transposed fragment reads: addresses inside a piece (the same for both buffers and every 16-token slab)
  const int gl = lane & 15, grp = lane >> 4, hf = lane >> 5, r32 = lane & 31;
  int a_adr[2][2], b_adr[2];  // [32-row tile][token half]
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
      a_adr[mi][h2] = (int)(tr_addr((const char*)0, 8 * h2 + 4 * hf, wr * 64 + mi * 32 + 16 * (grp & 1), gl) - (const char*)0);
    b_adr[h2] = (int)(tr_addr((const char*)0, 8 * h2 + 4 * hf, wc * 32 + 16 * (grp & 1), gl) - (const char*)0);
  }
  // The reads are issued from inline assembly: behind the builtin, hipcc cannot tell the transposed read from the
  // pending LDS-DMA writes and puts s_waitcnt vmcnt(0) in front of every group of reads, draining the pipeline. The
  // results are ordered by the explicit lgkmcnt waits of the phase.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
  auto trl = [&](uint32_t adr, int imm) __attribute__((always_inline)) {
    bf16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(adr), "i"(imm));
    return v;
  };
  bf16x8 af[2][4];    // [32-row tile][16-token slab]
  bf16x8 bf0[4], bf1[4];
  auto read_a = [&](int buf, int h) __attribute__((always_inline)) {
    const uint32_t p = lds0 + buf * TG_BUF;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        af[mi][s] = cat4(trl(p + a_adr[mi][0], h * TG_PIECE + s * 4096), trl(p + a_adr[mi][1], h * TG_PIECE + s * 4096));
  };
  auto read_b = [&](int buf, int h, bf16x8 (&bf)[4]) {
    const uint32_t p = lds0 + buf * TG_BUF;
#pragma unroll
    for (int s = 0; s < 4; ++s)
      bf[s] = cat4(trl(p + b_adr[0], (2 + h) * TG_PIECE + s * 4096), trl(p + b_adr[1], (2 + h) * TG_PIECE + s * 4096));
  };
  tw_acc acc00, acc01, acc10, acc11;  // [n half][k half]
  auto mma = [&](tw_acc& acc, bf16x8 (&bf)[4]) {
    tg_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) acc[mi] = mfma32(af[mi][s], bf[s], acc[mi]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    tg_barrier();
  };
  // TAIL 0: tiles t+1 and t+2 exist; 1: only t+1 exists; 2: last tile
  auto tile = [&](auto buf_c, int t, auto tail_c) __attribute__((always_inline)) {
    constexpr int buf = decltype(buf_c)::value, TAIL = decltype(tail_c)::value;
    read_b(buf, 0, bf0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(buf, 0);
    if (TAIL < 2) stage(buf ^ 1, PA1, t + 1);
    asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");  // the eight B0 reads, issued first of 24, have returned (see the forward kernel; the counter holds at most 15)
    mma(acc00, bf0);
    read_b(buf, 1, bf1);
    if (TAIL < 1) stage(buf, PB0, t + 2);
    mma(acc01, bf1);
    read_a(buf, 1);
    if (TAIL < 1) stage(buf, PA0, t + 2);
    mma(acc11, bf1);
    if (TAIL < 1) stage(buf, PB1, t + 2);
    if (TAIL == 0) tg_vmcnt<6>();
    if (TAIL == 1) tg_vmcnt<0>();
    mma(acc10, bf0);
  };
  using B0 = std::integral_constant<int, 0>;
  using B1 = std::integral_constant<int, 1>;
  using T0 = std::integral_constant<int, 0>;
  using T1 = std::integral_constant<int, 1>;
  using T2 = std::integral_constant<int, 2>;

  // ---- epilogue: lane (k column r32, hf) holds rows n = (reg & 3) + 8 (reg >> 2) + 4 hf of each 32 x 32 tile
  int n0 = 0, k0 = 0;
  bool whole = false;
  auto flush = [&](tw_acc& acc, int mh, int nh) __attribute__((always_inline)) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      float* op = a.dW + (size_t)(n0 + wr * 128 + mh * 64 + mi * 32 + 4 * hf) * a.ldw + k0 + wc * 64 + nh * 32 + r32;
      if (whole) {
        float old[16];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) old[reg] = op[(size_t)((reg & 3) + 8 * (reg >> 2)) * a.ldw];
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) op[(size_t)((reg & 3) + 8 * (reg >> 2)) * a.ldw] = old[reg] + acc[mi][reg];
      } else {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) unsafeAtomicAdd(op + (size_t)((reg & 3) + 8 * (reg >> 2)) * a.ldw, acc[mi][reg]);
      }
    }
  };

  while (w < w_end) {
    // ---- one segment: pairs [p0, p1) of one output tile
    const int tile_id = w / a.pairs, p0 = w - tile_id * a.pairs;
    const int p1 = min(a.pairs, p0 + (w_end - w));
    w += p1 - p0;
    tile_origin(tile_id, n0, k0);
    whole = p0 == 0 && p1 == a.pairs;
    const int tok0 = p0 * 128, KT = (p1 - p0) * 2;
    // N may end half way through the last tile row (N % 256 == 128): the second wave row then re-reads the first
    // one's columns and does not write
    const bool half_n = n0 + 128 >= a.N;
    pA = a.dY + (size_t)(tok0 + srow) * a.ldy + n0 + (half_n ? 0 : (pc >> 6) * 128) + (pc & 63);
    pB = a.X + (size_t)(tok0 + srow) * a.ldx + k0 + (pc >> 5) * 64 + (pc & 31);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc00[mi][j] = 0.f;
        acc01[mi][j] = 0.f;
        acc10[mi][j] = 0.f;
        acc11[mi][j] = 0.f;
      }
    stage(0, PB0, 0);
    stage(0, PA0, 0);
    stage(0, PB1, 0);
    stage(0, PA1, 0);
    stage(1, PB0, 1);
    stage(1, PA0, 1);
    stage(1, PB1, 1);
    tg_vmcnt<6>();
    tg_barrier();
    if (STAGGER && wr == 1) tg_barrier();
    int t = 0;
    for (; t + 2 < KT; t += 2) {
      tile(B0{}, t, T0{});
      tile(B1{}, t + 1, T0{});
    }
    tile(B0{}, t, T1{});
    tile(B1{}, t + 1, T2{});
    if (STAGGER && wr == 0) tg_barrier();
    if (!(half_n && wr == 1)) {
      flush(acc00, 0, 0);
      flush(acc01, 0, 1);
      flush(acc10, 1, 0);
      flush(acc11, 1, 1);
    }
  }
}

}  // namespace

// C[M, N] = A[M, K] . B[N, K]^T (+ bias). Leading dimensions in elements (multiples of 8: 16-byte rows), K a multiple
// of 128, N a multiple of 4. Returns -1 for shapes outside that (the caller keeps its library path).
// One launch = min(tiles, max_blocks) persistent workgroups (256 = one per CU).
// gated = 1: B holds 2 N rows (gate rows, then up rows), C[M, N] = silu(A.gate^T) * (A.up^T), no bias.
static int tgemm_tn_launch(const void* A, long lda, const void* B, long ldb, void* C, long ldc, const void* bias, int M,
                           int N, int K, int stagger, int max_blocks, int gated, hipStream_t st) {
  if (M <= 0 || N <= 0) return 0;
  if (K % 128 != 0 || N % 4 != 0 || lda % 8 != 0 || ldb % 8 != 0 || ldc % 4 != 0) return -1;
  if (((uintptr_t)A | (uintptr_t)B) % 16 != 0 || (uintptr_t)C % 8 != 0 || (bias && (uintptr_t)bias % 8 != 0)) return -1;
  if (gated && bias) return -1;
  TGArgs a{(const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C, ldc, (const bf16_t*)bias, M, N, K,
           (M + 255) / 256, gated ? (N + 127) / 128 : (N + 255) / 256};
  static bool attr_set = false;  // 128 KiB of dynamic LDS: above the default limit
  if (!attr_set) {
    HIP_RET(hipFuncSetAttribute((const void*)tgemm_tn_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS));
    HIP_RET(hipFuncSetAttribute((const void*)tgemm_tn_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS));
    HIP_RET(hipFuncSetAttribute((const void*)tgemm_tn_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS));
    attr_set = true;
  }
  // max_blocks workgroups (a multiple of 8: block id mod 8 is the XCD for every tile of a block's sequence) walk the
  // tiles; 0 = one workgroup per tile
  const int ntiles = a.tiles_m * a.tiles_n;
  if (max_blocks > 0 && max_blocks % 8 != 0) return -1;
  const dim3 grid(max_blocks > 0 ? std::min(ntiles, max_blocks) : ntiles), block(512);
  if (gated)
    hipLaunchKernelGGL((tgemm_tn_kernel<true, 1>), grid, block, TG_LDS, st, a);
  else if (stagger)
    hipLaunchKernelGGL((tgemm_tn_kernel<true, 0>), grid, block, TG_LDS, st, a);
  else
    hipLaunchKernelGGL((tgemm_tn_kernel<false, 0>), grid, block, TG_LDS, st, a);
  return (int)hipGetLastError();
}

extern "C" int areal_tgemm_tn(const void* A, long lda, const void* B, long ldb, void* C, long ldc, const void* bias,
                              int M, int N, int K, int stagger, int max_blocks, hipStream_t st) {
  return tgemm_tn_launch(A, lda, B, ldb, C, ldc, bias, M, N, K, stagger, max_blocks, 0, st);
}

// C[M, F] = silu(A . Wg^T) * (A . Wu^T) with W = [Wg; Wu] of shape [2 F, K] (the gate|up projection of a gated MLP and
// its activation in one kernel: the [M, 2 F] intermediate is never written). For forward passes that keep no
// activations (scoring, prefill).
extern "C" int areal_tgemm_swiglu(const void* A, long lda, const void* W, long ldw, void* C, long ldc, int M, int F,
                                  int K, int max_blocks, hipStream_t st) {
  return tgemm_tn_launch(A, lda, W, ldw, C, ldc, nullptr, M, F, K, 1, max_blocks, 1, st);
}

// dW[N, K] (fp32, leading dimension ldw) += dY[M, N]^T . X[M, K]. N a multiple of 128, K of 256, M a multiple of 128 (the
// caller adds the last M % 128 tokens with a library GEMM), rows 16-byte aligned. splits: 0 = stream-K.
extern "C" int areal_tgemm_wgrad(const void* dY, long ldy, const void* X, long ldx, float* dW, long ldw, int M, int N,
                                 int K, int splits, int stagger, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  if (M % 128 != 0 || N % 128 != 0 || K % 256 != 0 || ldy % 8 != 0 || ldx % 8 != 0) return -1;
  if (((uintptr_t)dY | (uintptr_t)X) % 16 != 0 || (uintptr_t)dW % 4 != 0) return -1;
  const int tiles = ((N + 255) / 256) * (K / 256), pairs = M / 128;
  // splits >= 1: that many workgroups per tile (1 = every tile added without atomics). splits <= 0: choose. Blocks run
  // in rounds of 256 (one per CU); a block costs its share of the pairs (~3.1 us each) plus ~8 us of prologue and
  // flush, ~25 us more when the tile is flushed with atomic adds (a CU retires one 256-byte atomic instruction per
  // ~50 ns) - so splitting pays for gradients of few tiles (qkv: 48 tiles x 4) and not for 210 or 420 tiles. An even
  // deal of ALL work over 256 workgroups (stream-K, nearly every tile partial) was measured 20-25 % slower than whole
  // tiles for those.
  // With more than 256 tiles the full rounds run whole tiles and only the tiles of the last, partly filled round are
  // split (420 tiles: 256 whole + 164 x 3).
  int whole = 0;
  if (splits <= 0) {
    auto block_us = [&](int s) { return ((pairs + s - 1) / s) * 3.1 + 8.0 + (s > 1 ? 25.0 : 0.0); };
    const int full = tiles / 256 * 256, rest = tiles - full;
    double best = 1e30;
    for (int s = 1; s <= 16 && pairs / s >= 4; ++s) {
      const int chunk = (pairs + s - 1) / s;
      if ((pairs + chunk - 1) / chunk != s) continue;  // this count leaves a workgroup without work
      // every tile split s ways
      double t = ((tiles * s + 255) / 256) * block_us(s);
      if (t < best - 1e-9) {
        best = t;
        splits = s;
        whole = s == 1 ? tiles : 0;
      }
      // full rounds whole, the rest split s ways
      if (full > 0 && rest > 0 && s > 1) {
        t = (full / 256) * block_us(1) + ((rest * s + 255) / 256) * block_us(s);
        if (t < best - 1e-9) {
          best = t;
          splits = s;
          whole = full;
        }
      }
    }
  } else if (splits == 1) {
    whole = tiles;
  }
  splits = std::max(1, std::min(splits, pairs));
  const int chunk = (pairs + splits - 1) / splits;
  splits = (pairs + chunk - 1) / chunk;
  const int nblocks = whole + (tiles - whole) * splits;
  TWArgs a{(const bf16_t*)dY, ldy, (const bf16_t*)X, ldx, dW, ldw, M, N, K, (N + 255) / 256, K / 256, pairs, whole, splits, chunk};
  static bool attr_set = false;
  if (!attr_set) {
    HIP_RET(hipFuncSetAttribute((const void*)tgemm_wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS));
    HIP_RET(hipFuncSetAttribute((const void*)tgemm_wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, TG_LDS));
    attr_set = true;
  }
  const dim3 grid(nblocks), block(512);
  if (stagger)
    hipLaunchKernelGGL(tgemm_wgrad_kernel<true>, grid, block, TG_LDS, st, a);
  else
    hipLaunchKernelGGL(tgemm_wgrad_kernel<false>, grid, block, TG_LDS, st, a);
  return (int)hipGetLastError();
}
