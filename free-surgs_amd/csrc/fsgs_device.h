// fsgs_device.h -- device-side helpers shared by the gfx950 kernels.
// Written for CDNA4 only: 64-lane wavefronts, DPP row operations, v_readlane /
// v_writelane scalar broadcast.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FSGS_TILE 16
#define FSGS_WAVE 64
#define FSGS_PX_PER_LANE 4  // one wave owns a 16x16 tile = four 8x8 quadrants; lane l owns pixel (l&7, l>>3) of each
#define FSGS_QUAD 8

namespace fsgs {

typedef float float2v __attribute__((ext_vector_type(2)));  // a register pair: elementwise fma on it is one v_pk_fma_f32

// ---- scalar broadcast of one lane's value (v_readlane_b32: the index must be wave-uniform) ----
__device__ __forceinline__ float readlane(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ uint32_t readlane(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ int readlane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---- DPP wave64 reductions -------------------------------------------------------------------
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_add(float v) {
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, true);
  return v + __int_as_float(t);
}
// Sum over the 64 lanes; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_lane63(float v) {
  v = dpp_add<0xB1>(v);        // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);        // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);       // row_half_mirror
  v = dpp_add<0x140>(v);       // row_mirror      -> every lane holds its 16-lane row sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15    -> rows 1,3 += previous row
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31    -> rows 2,3 += lanes 0..31
  return v;
}
__device__ __forceinline__ float wave_sum(float v) { return readlane(wave_sum_lane63(v), 63); }

// ---- transposing wave reduction ----------------------------------------------------------------------------
// in: every lane holds 64 partial values v[0..63].  out: lane l holds sum over all 64 lanes of v[l].
// Six halving steps (lane bit k <-> value-index bit k): at each step a lane keeps the half of the values
// whose index bit equals its own lane bit and receives the partner lane's partials for that half.
//   bit 5: v_permlane32_swap, bit 4: v_permlane16_swap (one swap + one add fold TWO values),
//   bit 3: DPP row_ror:8, bit 2: DPP row_half_mirror (partner l^7: still a bit-2 flip), bits 1,0: quad_perm.
// 141 VALU instructions for 64 values (2.2 per value) against 8 per value for 64 separate DPP reductions.
__device__ __forceinline__ float swap32_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float swap16_add(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ float fold_dpp(float x, float y, bool upper) {
  float keep = upper ? y : x, send = upper ? x : y;
  int t = __builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xF, 0xF, true);
  return keep + __int_as_float(t);
}
// The two fold steps whose lane bit selects whole DPP BANKS (4 lanes) need no select at all: a DPP add writes only
// the banks its bank_mask enables, so "x + partner's x" goes to the banks with the lane bit clear and "y + partner's y"
// to the others -- two instructions instead of two v_cndmask + one DPP add, and no VCC traffic.
//   lane bit 3 (row_ror:8, partner l ^ 8): banks 0,1 <- x, banks 2,3 <- y
//   lane bit 2 (row_half_mirror, partner l ^ 7): banks 0,2 <- x, banks 1,3 <- y
__device__ __forceinline__ float fold_bit3(float x, float y) {
  float r;
  // (s_nop 1: a DPP read needs two wait states behind the VALU write of its source; the compiler inserts them for its
  // own DPP instructions, not in front of inline assembly)
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
               : "=&v"(r)
               : "v"(x), "v"(y));
  return r;
}
__device__ __forceinline__ float fold_bit2(float x, float y) {
  float r;
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa"
               : "=&v"(r)
               : "v"(x), "v"(y));
  return r;
}
// 32-value variant: lane l ends with the total of v[l >> 1] (lanes l and l^1 hold the same value).
// 70 VALU instructions for 32 values; needs only 32 live registers instead of 64.
__device__ __forceinline__ float wave_transpose_reduce32(const float (&v)[32], int lane) {
  float w[16], x[8], y[4], z[2];
#pragma unroll
  for (int i = 0; i < 16; i++) w[i] = swap32_add(v[i], v[i + 16]);   // lane bit 5 <-> index bit 4
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = swap16_add(w[i], w[i + 8]);     // lane bit 4 <-> index bit 3
  const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
  for (int i = 0; i < 4; i++) y[i] = fold_bit3(x[i], x[i + 4]);  // lane bit 3 <-> index bit 2
#pragma unroll
  for (int i = 0; i < 2; i++) z[i] = fold_bit2(y[i], y[i + 2]);  // lane bit 2 <-> index bit 1
  float u = fold_dpp<0x4E>(z[0], z[1], b1);                                 // lane bit 1 <-> index bit 0
  return dpp_add<0xB1>(u);                                                  // lanes l, l^1: plain sum
}

// ---- the 12-of-16 reduction with the CHEAP exchanges first ---------------------------------------------------------
// A transposing step costs (values it leaves) x (price of that lane distance): a bank-masked DPP fold (lane bits 3, 2)
// two plain instructions per value, a select-based fold (bits 1, 0) three, a permlane swap + add (bits 4, 5) ~3.5 (the
// swap issues at ~2.5x).  The order above spends the swaps where MOST values are left (12 + 8 of them); this one folds
// lane bits 3, 2, 1 first (16, 8, 4 values left), then one permlane16 swap pair (2 left), one permlane32 swap (1) and a
// plain quad-perm add: ~64 issue slots for the two Gaussians instead of ~86.  Pairs whose second value is one of the
// unused slots 12..15 take ONE instruction in the first step (the banks that would hold the unused sum keep garbage,
// which only ever flows into lanes that own unused slots).
// out: lane l holds the total of v[transpose12_slot(l)]; lanes l and l ^ 1 the same value.
__host__ __device__ __forceinline__ int transpose12_slot(int lane) {
  return 16 * ((lane >> 5) & 1) + 8 * ((lane >> 3) & 1) + 4 * ((lane >> 2) & 1) + 2 * ((lane >> 1) & 1) + ((lane >> 4) & 1);
}
// One Gaussian's first two steps as two assembly blocks: every DPP source is an INPUT of its block, so one s_nop 1 per
// block covers the "VALU write -> DPP read" wait states of all its instructions (a fold at a time needs one each).
//   fold3: slots k and k + 8 -> a[k] (k = 0..3 both halves, k = 4..7 the lower banks only: slot k + 8 is unused)
//   fold2: a[k] and a[k + 4] -> b[k]
__device__ __forceinline__ void fold3_block(const float *x, float (&a)[8]) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
               "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
               "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
               "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
               "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3"
               : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3]), "=&v"(a[4]), "=&v"(a[5]), "=&v"(a[6]), "=&v"(a[7])
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]),
                 "v"(x[8]), "v"(x[9]), "v"(x[10]), "v"(x[11]));
}
__device__ __forceinline__ void fold2_block(const float (&a)[8], float (&b)[4]) {
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
               "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
               "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
               "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa"
               : "=&v"(b[0]), "=&v"(b[1]), "=&v"(b[2]), "=&v"(b[3])
               : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]));
}
__device__ __forceinline__ float wave_transpose_reduce32_12of16_cheap_first(const float (&v)[32], int lane) {
  float a0[8], a1[8], b0[4], b1_[4], c[4], d[2];
  fold3_block(&v[0], a0);   // lane bit 3 <-> slot bit 3, Gaussian 0
  fold3_block(&v[16], a1);  //                           Gaussian 1
  fold2_block(a0, b0);      // lane bit 2 <-> slot bit 2
  fold2_block(a1, b1_);
  const bool b1 = lane & 2;
  c[0] = fold_dpp<0x4E>(b0[0], b0[2], b1);  // lane bit 1 <-> slot bit 1
  c[1] = fold_dpp<0x4E>(b0[1], b0[3], b1);
  c[2] = fold_dpp<0x4E>(b1_[0], b1_[2], b1);
  c[3] = fold_dpp<0x4E>(b1_[1], b1_[3], b1);
  d[0] = swap16_add(c[0], c[1]);  // lane bit 4 <-> slot bit 0
  d[1] = swap16_add(c[2], c[3]);
  const float e = swap32_add(d[0], d[1]);  // lane bit 5 <-> Gaussian
  return dpp_add<0xB1>(e);                 // lanes l, l ^ 1: plain sum
}

// The pose-only backward's 5-of-8 reduction (two Gaussians x 8 slots, 5 used) in the same cheap-first order: lane bits 3,
// 2, 1 take slot bits 2, 1, 0 (five + four + three DPP / select instructions per Gaussian instead of five + four swap
// pairs for both), lane bit 4 the Gaussian (one permlane16 swap pair); lane bits 5 and 0 are plain sums.
// out: lane l holds the total of v[transpose5_slot(l)]; lanes differing in bits 5 or 0 only hold the same value.
__host__ __device__ __forceinline__ int transpose5_slot(int lane) {
  return 8 * ((lane >> 4) & 1) + 4 * ((lane >> 3) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 1) & 1);
}
__device__ __forceinline__ void fold32_block5(const float *x, float (&b)[2]) {  // x[0..4] -> b[slot bit 0]
  float a0, a1, a2, a3;
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
               "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
               "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3"
               : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3)
               : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]));
  asm volatile("s_nop 1\n\t"
               "v_add_f32_dpp %0, %2, %2 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
               "v_add_f32_dpp %1, %3, %3 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
               "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0xa"
               : "=&v"(b[0]), "=&v"(b[1])
               : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
}
__device__ __forceinline__ float wave_transpose_reduce16_5of8_cheap_first(const float (&v)[16], int lane) {
  float b0[2], b1[2];
  fold32_block5(&v[0], b0);
  fold32_block5(&v[8], b1);
  const bool bit1 = lane & 2;
  const float c0 = fold_dpp<0x4E>(b0[0], b0[1], bit1), c1 = fold_dpp<0x4E>(b1[0], b1[1], bit1);
  const float d = swap16_add(c0, c1);  // lane bit 4 <-> Gaussian
  const float e = swap32_add(d, d);    // lane bit 5: plain sum (both halves end with the total)
  return dpp_add<0xB1>(e);             // lane bit 0: plain sum
}

template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_max_i(int v) {
  int t = __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);
  return max(v, t);
}
__device__ __forceinline__ int wave_max(int v) {
  v = dpp_max_i<0xB1>(v);
  v = dpp_max_i<0x4E>(v);
  v = dpp_max_i<0x141>(v);
  v = dpp_max_i<0x140>(v);
  v = dpp_max_i<0x142, 0xA>(v);
  v = dpp_max_i<0x143, 0xC>(v);
  return __builtin_amdgcn_readlane(v, 63);
}

// ---- the one definition of "does Gaussian g touch pixel p" ---------------------------------------
// Forward and backward must take bit-identical skip decisions, so both call this and nothing
// here may be re-associated or contracted differently between kernels (explicit *_rn ops).
// power = -1/2 (A dx^2 + C dy^2) - B dx dy; skip power > 0; alpha = min(.99, o e^power);
// skip alpha < 1/255.
// The blend kernels stage every record once per batch, so the exponent's coefficients are pre-scaled there:
// (a, b, c) = -log2(e) (A/2, B, C/2), p = a dx^2 + c dy^2 + b dx dy = power * log2(e), and e^power is ONE v_exp_f32
// (2^p) with no scaling multiply in front of it -- 6 VALU for p instead of 8; `power > 0` is `p > 0`.
struct SplatCoef {
  float a, b, c;
};
__device__ __forceinline__ SplatCoef splat_coef(float A, float B, float Cc) {
  // log2(e) in DOUBLE, the product rounded once: the fp32 constant alone is 1.3e-8 (relative) short, which makes every
  // exponent that much smaller and alpha = o 2^p about |power| x 1.3e-8 ~ 4e-8 too large -- a bias, not noise (measured
  // with fsgs_selftest_splat_alpha around the 1/255 threshold: of 122 decisions that differed from float64, 85 said
  // "blend").  Staged once per (tile, Gaussian) record, not per pixel: three f64 multiplies per 64 pairs.
#if defined(FSGS_DIAG_HOOKS) && defined(FSGS_EXP_L2E_FLOAT)  // diagnostics flavour only: the fp32 constant of rounds 1-2
  const float L2Ef = 1.44269504088896340736f;
  SplatCoef kf;
  kf.a = __fmul_rn(-0.5f * L2Ef, A);
  kf.b = __fmul_rn(-L2Ef, B);
  kf.c = __fmul_rn(-0.5f * L2Ef, Cc);
  return kf;
#endif
  const double L2E = 1.44269504088896340736;
  SplatCoef k;
  k.a = (float)(-0.5 * L2E * (double)A);
  k.b = (float)(-L2E * (double)B);
  k.c = (float)(-0.5 * L2E * (double)Cc);
  return k;
}
struct SplatEval {
  float dx, dy;
  float a;      // o * e^power, NOT clamped (the backward's dL/dG = o dL/dalpha rides on it: w o = a dL/dalpha)
  float alpha;  // min(0.99, a)
};
__device__ __forceinline__ float splat_exponent(float ca, float cb, float cc, float dx, float dy) {
  // dx (a dx + b dy) + c dy^2: five instructions (the term-by-term form needs six)
  const float q = __fmul_rn(__fmaf_rn(cb, dy, __fmul_rn(ca, dx)), dx);
  return __fmaf_rn(__fmul_rn(cc, dy), dy, q);
}
// The offset d = mean2D - pixel of the lane's pixel in quadrant k, from the offset to the lane's FIRST pixel (quadrant
// 0): the four pixels of a lane sit 8 apart, so d_k = d_0 - (8 (k & 1), 8 (k >> 1)).  d_0 is formed once per pair; a
// lane then keeps ONE pixel position instead of four (six registers, which is what stood between the mapping
// backward and five waves per SIMD), and the quadrants on the left / top need no subtraction at all.  Forward and
// backward both come through here, so their skip decisions stay bit-identical.  (mean2D - integer is exact in fp32
// whenever the result is not larger than the mean itself, i.e. for every pair a 16x16 tile can see except
// screen-filling footprints, where the second rounding is 1e-7 of the footprint's size.)
__device__ __forceinline__ float quad_offset(float d0, bool second) {
  return second ? __fsub_rn(d0, (float)FSGS_QUAD) : d0;
}
__device__ __forceinline__ bool splat_alpha(float dx, float dy, float ca, float cb, float cc, float o, SplatEval &e) {
  e.dx = dx;
  e.dy = dy;
  const float p = splat_exponent(ca, cb, cc, e.dx, e.dy);
  if (p > 0.0f) return false;
  e.a = __fmul_rn(o, __builtin_amdgcn_exp2f(p));
  e.alpha = fminf(0.99f, e.a);
  return e.a >= (1.0f / 255.0f);
}

// Branch-free variant for the backward replay: a and alpha are forced to ZERO when the pair does not contribute
// (power > 0, alpha < 1/255 or `live` false).  The skip decision is bit-identical to splat_alpha; with
// alpha = a = 0 every gradient term of the pair vanishes and T / gB stay untouched, so no divergent branch is
// needed around the arithmetic (one v_cndmask: alpha = min(0.99, a) follows from the masked a).
__device__ __forceinline__ bool splat_alpha_masked(float dx, float dy, float ca, float cb, float cc, float o, bool live,
                                                   SplatEval &e) {
  e.dx = dx;
  e.dy = dy;
  const float p = splat_exponent(ca, cb, cc, e.dx, e.dy);
  // (p > 0 is masked below: whatever 2^p is there -- even +inf -- only reaches the unselected side of the select)
  const float a = __fmul_rn(o, __builtin_amdgcn_exp2f(p));
  const bool ok = live && !(p > 0.0f) && a >= (1.0f / 255.0f);
  e.a = ok ? a : 0.0f;
  e.alpha = fminf(0.99f, e.a);
  return ok;
}

// ---- exact footprint culling ---------------------------------------------------------------------------
// A Gaussian contributes to a pixel only if alpha = o*exp(-q) >= 1/255, i.e. q(d) <= tau = ln(255 o)
// with q(d) = 1/2 (A dx^2 + C dy^2) + B dx dy, d = centre - pixel (splat_alpha above).  A rectangle of
// pixels can therefore be skipped as a whole when the minimum of q over the (continuous) rectangle is
// above tau.  Skipping never changes a result -- only pairs that every pixel would have skipped anyway
// are dropped -- so the test only has to be CONSERVATIVE (margin for the rounding of exp/log).
// Pixel centres of the rectangle: [x0, x0+w-1] x [y0, y0+h-1].
__device__ __forceinline__ float footprint_tau(float opacity) {
  // o <= 1/255 can never reach alpha >= 1/255: negative tau culls the Gaussian everywhere
  return opacity > (1.0f / 255.0f) ? __logf(255.0f * opacity) * 1.0001f + 1e-3f : -1.0f;
}
__device__ __forceinline__ bool rect_touched(float gx, float gy, float A, float B, float Cc, float tau, float x0,
                                             float y0, float w, float h) {
  if (tau < 0.f) return false;
  // d-space box
  const float dx_lo = gx - (x0 + w - 1.0f), dx_hi = gx - x0;
  const float dy_lo = gy - (y0 + h - 1.0f), dy_hi = gy - y0;
  if (dx_lo <= 0.f && dx_hi >= 0.f && dy_lo <= 0.f && dy_hi >= 0.f) return true;  // centre inside
  if (!(A > 0.f && Cc > 0.f && A * Cc - B * B > 0.f)) return true;                  // not PD: do not cull
  // convex quadratic, unconstrained minimum outside the box -> the minimum sits on the boundary
  const float iA = 1.0f / A, iC = 1.0f / Cc;
  float best;
  {
    float c = dx_lo, dy = fminf(fmaxf(-B * c * iC, dy_lo), dy_hi);
    best = 0.5f * (A * c * c + Cc * dy * dy) + B * c * dy;
    c = dx_hi; dy = fminf(fmaxf(-B * c * iC, dy_lo), dy_hi);
    best = fminf(best, 0.5f * (A * c * c + Cc * dy * dy) + B * c * dy);
    c = dy_lo;
    float dx = fminf(fmaxf(-B * c * iA, dx_lo), dx_hi);
    best = fminf(best, 0.5f * (A * dx * dx + Cc * c * c) + B * dx * c);
    c = dy_hi; dx = fminf(fmaxf(-B * c * iA, dx_lo), dx_hi);
    best = fminf(best, 0.5f * (A * dx * dx + Cc * c * c) + B * dx * c);
  }
  return best <= tau;
}

// ---- per-Gaussian geometry --------------------------------------------------------
struct Mat3 {
  float m[9];
};
// rotation of an UN-normalised quaternion (r,x,y,z), row-major
__device__ __forceinline__ Mat3 quat_to_R(const float4 q) {
  float r = q.x, x = q.y, y = q.z, z = q.w;
  Mat3 R;
  R.m[0] = 1.f - 2.f * (y * y + z * z); R.m[1] = 2.f * (x * y - r * z);       R.m[2] = 2.f * (x * z + r * y);
  R.m[3] = 2.f * (x * y + r * z);       R.m[4] = 1.f - 2.f * (x * x + z * z); R.m[5] = 2.f * (y * z - r * x);
  R.m[6] = 2.f * (x * z - r * y);       R.m[7] = 2.f * (y * z + r * x);       R.m[8] = 1.f - 2.f * (x * x + y * y);
  return R;
}
// Sigma = R diag(s)^2 R^T -> (xx,xy,xz,yy,yz,zz)
__device__ __forceinline__ void cov3d(const float3 s, const Mat3 &R, float *c6) {
  float M[9];
#pragma unroll
  for (int i = 0; i < 3; i++) {
    M[3 * i] = R.m[3 * i] * s.x; M[3 * i + 1] = R.m[3 * i + 1] * s.y; M[3 * i + 2] = R.m[3 * i + 2] * s.z;
  }
  c6[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  c6[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  c6[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  c6[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  c6[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  c6[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

// Everything the EWA projection needs, computed once and reused by forward and backward.
struct Ewa {
  float tx, ty, tz;        // clamped view-space mean (t~)
  float chix, chiy;        // 0 where the 1.3*tanfov clamp was active
  float M0[3], M1[3];      // rows of M = J * Wv
  float SM0[3], SM1[3];    // Sigma * M0, Sigma * M1
  float a, b, c;           // dilated cov2D
};
// V: view matrix in transposed storage (V[4*c + r] = maths V[r][c]); t = view-space mean.
__device__ __forceinline__ void ewa_project(const float *V, const float3 t, const float *c6, float fx, float fy,
                                            float tanfovx, float tanfovy, Ewa &e) {
  float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
  float txtz = t.x / t.z, tytz = t.y / t.z;
  e.tx = fminf(limx, fmaxf(-limx, txtz)) * t.z;
  e.ty = fminf(limy, fmaxf(-limy, tytz)) * t.z;
  e.tz = t.z;
  e.chix = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  e.chiy = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  float J00 = fx / t.z, J02 = -(fx * e.tx) / (t.z * t.z);
  float J11 = fy / t.z, J12 = -(fy * e.ty) / (t.z * t.z);
#pragma unroll
  for (int c = 0; c < 3; c++) {
    e.M0[c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
    e.M1[c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
  }
  const float S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    e.SM0[r] = S[3 * r] * e.M0[0] + S[3 * r + 1] * e.M0[1] + S[3 * r + 2] * e.M0[2];
    e.SM1[r] = S[3 * r] * e.M1[0] + S[3 * r + 1] * e.M1[1] + S[3 * r + 2] * e.M1[2];
  }
  e.a = e.M0[0] * e.SM0[0] + e.M0[1] * e.SM0[1] + e.M0[2] * e.SM0[2] + 0.3f;
  e.b = e.M0[0] * e.SM1[0] + e.M0[1] * e.SM1[1] + e.M0[2] * e.SM1[2];
  e.c = e.M1[0] * e.SM1[0] + e.M1[1] * e.SM1[1] + e.M1[2] * e.SM1[2] + 0.3f;
}

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (observed, speed only).
// Give every XCD one contiguous run of virtual blocks so neighbouring tiles share an L2.
__device__ __forceinline__ int xcd_swizzle(int bid, int nblocks) {
  const int NXCD = 8;
  int per = nblocks / NXCD;
  int lim = per * NXCD;
  if (bid >= lim) return bid;  // ragged tail stays in place
  return (bid % NXCD) * per + bid / NXCD;
}

// ---- the one definition of the Adam element update (optim.hip and the fused backward of render.hip) ----
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float omb1, float b2, float omb2,
                                         float eps, float step_size, float inv_bc2_sqrt) {
  // torch.optim.Adam: lerp_(grad, 1-beta1); mul_(beta2).addcmul_(grad, grad, value=1-beta2); sqrt/bc2 + eps; addcdiv_
  // (1-beta) is formed in DOUBLE on the host like torch does: 1 - 0.999f would be off by 4.7e-5 relative)
  m = fmaf(omb1, g - m, m);
  v = fmaf(omb2, g * g, b2 * v);
  float denom = sqrtf(v) * inv_bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}


}  // namespace fsgs
