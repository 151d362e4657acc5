// LDS tile helpers shared by the flash-attention kernels (flash_attn.hip: head size 128, flash_attn_d256.hip: 256).
// Tiles use 256-byte rows (128 bf16) with the 16-byte-chunk XOR swizzle
//   off(row, ch) = 256*row + 16*(ch ^ (((row&3)<<2) | ((row>>2)&3)))
// which is conflict-free for both ds_read_b128 row reads and transposed reads.
#pragma once
#include "common.hpp"

typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

DEVINL f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// (exponentials go through fast_exp2 of common.hpp: v_exp_f32 directly; exp2f() expands to a denormal-safe
// sequence (scale, v_exp, ldexp, compare) that the compiler then guards with per-element branches)
DEVINL int swz_off(int row, int ch) {  // byte offset of 16-byte chunk `ch` of `row`
  return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3)));
}
// Transposed read of a 4-row x 16-col block: every lane supplies the address of
// (row0 + q, col0 + 4p) with its group-lane index 4q+p and receives column `i` of the four
// rows. `base` is the LDS tile, rows are 256 bytes, col0 must be a multiple of 16.
DEVINL bf16x4 tr_read(const char* tile, int row0, int col0, int gl) {
  const int q = gl >> 2, p = gl & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  const char* a = tile + swz_off(row0 + q, ch) + 8 * (p & 1);
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a));
}
// The same read split in two: the address (loop-invariant when a tile is restaged in place) and the load.
DEVINL const char* tr_addr(const char* tile, int row0, int col0, int gl) {
  const int q = gl >> 2, p = gl & 3;
  const int ch = (col0 >> 3) + (p >> 1);
  return tile + swz_off(row0 + q, ch) + 8 * (p & 1);
}
DEVINL bf16x4 tr_load(const char* a) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(a)); }
DEVINL bf16x8 cat4(bf16x4 a, bf16x4 b) {
  bf16x8 r;
  r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
  r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
  return r;
}
DEVINL bf16x8 lds_row_read(const char* tile, int row, int ch) {
  return *reinterpret_cast<const bf16x8*>(tile + swz_off(row, ch));
}
DEVINL void lds_row_write(char* tile, int row, int ch, bf16x8 v) {
  *reinterpret_cast<bf16x8*>(tile + swz_off(row, ch)) = v;
}
