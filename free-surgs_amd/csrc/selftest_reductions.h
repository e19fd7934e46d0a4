// Transposing wave reductions that NO product kernel calls: kept for the reduction selftest (csrc/common.hip:
// fsgs_selftest_transpose_reduce*, tests/test_raster_gpu.py) and the micro-benchmarks (scripts/ubench/mfma_reduce.hip),
// out of the product header fsgs_device.h.  Built from the same primitives (swap32_add, swap16_add,
// fold_bit3 / fold_bit2, fold_dpp, dpp_add) the shipping reductions use.
#pragma once
#include "fsgs_device.h"

namespace fsgs {

// 64 values per lane: lane l ends with the total of v[l] (the full 64 x 64 transpose-and-sum).
__device__ __forceinline__ float wave_transpose_reduce64(const float (&v)[64], int lane) {
  float w[32], x[16], y[8], z[4], u[2];
#pragma unroll
  for (int i = 0; i < 32; i++) w[i] = swap32_add(v[i], v[i + 32]);
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = swap16_add(w[i], w[i + 16]);
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
#pragma unroll
  for (int i = 0; i < 8; i++) y[i] = fold_bit3(x[i], x[i + 8]);  // row_ror:8
#pragma unroll
  for (int i = 0; i < 4; i++) z[i] = fold_bit2(y[i], y[i + 4]);  // row_half_mirror
#pragma unroll
  for (int i = 0; i < 2; i++) u[i] = fold_dpp<0x4E>(z[i], z[i + 2], b1);   // quad_perm [2,3,0,1]
  return fold_dpp<0xB1>(u[0], u[1], b0);                                  // quad_perm [1,0,3,2]
}


// The same when v[12..15] and v[28..31] are known to be ZERO (two Gaussians x 16 slots of which 12 are used): the
// first halving step pairs v[i] with v[i + 16], so four of its sixteen swap + add pairs only move zeros.  (The swap-first
// order blend_bwd used until the cheap-first one below replaced it; scripts/ubench/mfma_reduce.hip still times it.)
__device__ __forceinline__ float wave_transpose_reduce32_12of16(const float (&v)[32], int lane) {
  float w[16], x[8], y[4], z[2];
#pragma unroll
  for (int i = 0; i < 12; i++) w[i] = swap32_add(v[i], v[i + 16]);
#pragma unroll
  for (int i = 12; i < 16; i++) w[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = swap16_add(w[i], w[i + 8]);
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int i = 0; i < 4; i++) y[i] = fold_bit3(x[i], x[i + 4]);
#pragma unroll
  for (int i = 0; i < 2; i++) z[i] = fold_bit2(y[i], y[i + 2]);
  float u = fold_dpp<0x4E>(z[0], z[1], b1);
  return dpp_add<0xB1>(u);
}


// 16-value variant: lane l ends with the total of v[l >> 2] (four lanes hold the same value); ~40 VALU.
__device__ __forceinline__ float wave_transpose_reduce16(const float (&v)[16], int lane) {
  float w[8], x[4], y[2];
#pragma unroll
  for (int i = 0; i < 8; i++) w[i] = swap32_add(v[i], v[i + 8]);     // lane bit 5 <-> index bit 3
#pragma unroll
  for (int i = 0; i < 4; i++) x[i] = swap16_add(w[i], w[i + 4]);     // lane bit 4 <-> index bit 2
  const bool b3 = lane & 8, b2 = lane & 4;
#pragma unroll
  for (int i = 0; i < 2; i++) y[i] = fold_bit3(x[i], x[i + 2]);  // lane bit 3 <-> index bit 1
  float z = fold_bit2(y[0], y[1]);                                // lane bit 2 <-> index bit 0
  z = dpp_add<0x4E>(z);                                                     // lanes l, l^2, l^1, l^3: plain sum
  return dpp_add<0xB1>(z);
}


}  // namespace fsgs
