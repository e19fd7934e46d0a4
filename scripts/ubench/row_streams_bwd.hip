// row_streams_bwd.hip -- prototype of "four independent 4x4-block streams per wave" for the backward blend (VERDICT r5 #1 b),
// timed per STEP against the shipping pair loop, both built from the product's own per-pixel step (blend_bwd_pixel) and
// reductions (csrc/raster_kernels.h, fsgs_device.h), on records staged in LDS exactly as the product stages them.
//
//   pair  : the shipping one-wave kernel's inner loop -- per (tile, Gaussian) pair 3 broadcast ds_read_b128, NB quadrant bodies
//           over the 64 lanes (NB = 2 and 3: the C2 scene executes 2.16 per pair), and per TWO pairs one 64-lane x 24-value
//           transposing reduction + one atomic per lane.  64-thread workgroups, 5 waves per SIMD (94 VGPRs in the product).
//   rows  : the proposal -- a 256-thread workgroup per tile, wave q = quadrant q, 16-lane DPP row r = 4x4 block r of the
//           quadrant, every row walking ITS OWN list of the batch's records (a byte list in LDS, different per row): per wave
//           step one ds_read_u8, three row-broadcast ds_read_b128, ONE body (each row another Gaussian), a 16-lane x 12-value
//           transposing reduction inside every row (lane bits 3, 2 with bank-masked DPP adds, bits 1, 0 with select-folds:
//           29 VALU) and one ds_add_f32 per lane into per-record accumulators in LDS (flushed once per batch).  8 waves/SIMD.
//
// What the row design saves is bodies: profiles/r06_lane_utilisation.jsonl, "row_streams" (the diagnostics flavour counted,
// per quadrant and 256-record batch, max over the four rows of their alive records): 1.556 wave steps per pair at C2 against
// 2.16 bodies (dense 1.90 / 2.32, C4 1.45 / 1.98), row balance 0.92 / 0.90 / 0.85.  What it pays is a reduction per STEP instead
// of one per two PAIRS.  The run prints ns per pair / per step and the projected kernel-time ratio
//     1.556 x t(rows step)  /  (t(pair, NB=2) + 0.16 (t(pair, NB=3) - t(pair, NB=2)))
// -- a floor for the row design: its per-batch work (16 footprint tests per record for the 4x4 masks, the lists, the flush) is
// left out, and every row is busy on every step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -I free-surgs_amd/csrc \
//         -o scripts/ubench/row_streams_bwd.bin scripts/ubench/row_streams_bwd.hip
#include "raster_kernels.h"

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int CG = 4;       // channels that carry dL/dpixel in the mapping step (RGB + depth)
constexpr int BATCH = 256;  // records staged per batch

// 16 lanes x 12 values -> lane l (of its row) holds the row's total of value l & 15 (values 12..15: garbage, unused)
__device__ __forceinline__ float row_transpose_reduce12(const float *v, int lane) {
  float a[8], b[4];
  fold3_block(v, a);   // lane bit 3 <-> value bit 3 (12 instructions)
  fold2_block(a, b);   // lane bit 2 <-> value bit 2 (8)
  const bool b1 = lane & 2, b0 = lane & 1;
  const float c0 = fold_dpp<0x4E>(b[0], b[2], b1), c1 = fold_dpp<0x4E>(b[1], b[3], b1);  // lane bit 1 (3 + 3)
  return fold_dpp<0xB1>(c0, c1, b0);                                                        // lane bit 0 (3)
}

__global__ __launch_bounds__(64) void check_row_reduce(const float *in, float *out) {  // in[lane][12]
  float v[12];
  for (int i = 0; i < 12; i++) v[i] = in[threadIdx.x * 12 + i];
  out[threadIdx.x] = row_transpose_reduce12(v, threadIdx.x);
}

struct Staged {
  float4 r0, r1, r2;
};
__device__ __forceinline__ void stage(float4 *rec, int t, uint32_t seed) {
  // plausible records: centre within ~12 px of the tile, conic of a 3-6 px Gaussian, opacity 0.3-0.9 (pre-scaled like the product)
  uint32_t h = seed * 747796405u + 2891336453u + (uint32_t)t * 277803737u;
  auto rnd = [&]() { h = h * 1664525u + 1013904223u; return (float)(h >> 8) * (1.0f / 16777216.0f); };
  const float s = 3.f + 3.f * rnd(), A = 1.f / (s * s), C = A * (0.7f + 0.6f * rnd()), B = 0.3f * A * (rnd() - 0.5f);
  const SplatCoef kf = splat_coef(A, B, C);
  rec[t * 3 + 0] = make_float4(-4.f + 24.f * rnd(), -4.f + 24.f * rnd(), kf.a, kf.b);
  rec[t * 3 + 1] = make_float4(kf.c, 0.3f + 0.6f * rnd(), rnd(), rnd());
  rec[t * 3 + 2] = make_float4(rnd(), rnd(), rnd(), rnd());
}

// ---- the shipping pair loop (one wave per tile, 4 pixels per lane) -----------------------------------------------------------
template <int NB>
__global__ __launch_bounds__(64, 5) void pair_loop(float *acc, int npairs) {
  using Slots = BwdSlots<true, false, CG>;
  constexpr int SL = Slots::SL, NV = Slots::NV;
  __shared__ float4 rec[64 * 3];
  const int lane = threadIdx.x;
  stage(rec, lane, blockIdx.x);
  __syncthreads();
  float T[4], gB[4], gBr[4], g[4][CG];
  for (int k = 0; k < 4; k++) {
    T[k] = 0.05f + 0.001f * lane; gB[k] = 0.01f * k; gBr[k] = 0.005f * k;
    for (int c = 0; c < CG; c++) g[k][c] = 1e-3f * (lane + c - 32);
  }
  const float px0 = (float)(lane & 7), py0 = (float)(lane >> 3);
  const Slots slots(lane);
  float *row = acc + (size_t)blockIdx.x * 64 * 16;
  for (int p = 0; p < npairs; p += 2) {
    float v[NV];
#pragma unroll
    for (int i = 0; i < NV; i++) v[i] = 0.f;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int j = (p + u) & 63;
      const float4 r0 = rec[j * 3 + 0], r1 = rec[j * 3 + 1], r2 = rec[j * 3 + 2];
      const float bcol[6] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
      const float dx0 = __fsub_rn(r0.x, px0), dy0 = __fsub_rn(r0.y, py0);
#pragma unroll
      for (int k = 0; k < NB; k++)
        blend_bwd_pixel<CG, true, false>(&v[SL * u], T[k], gB[k], gBr[k], g[k], quad_offset(dx0, k & 1), quad_offset(dy0, k >> 1),
                                         r0.z, r0.w, r1.x, r1.y, bcol, true);
#pragma unroll
      for (int k = 0; k < NB; k++) T[k] = T[k] > 1.0f ? 0.05f : T[k];  // (keeps the replayed transmittance finite; 2 VALU per body)
    }
    const float tot = Slots::reduce(v, lane);
    if (slots.used && tot != 0.f) atomicAdd(row + (((p & 63) + slots.u) * 16 + slots.c), tot);
  }
}

// ---- the row-stream step (four waves per tile, one pixel per lane, every 16-lane row its own record) ------------------------
// ACC: 0 = no accumulation (VALU floor, not a correct kernel), 1 = ds_add_f32 into per-record LDS sums (flushed once per batch),
//      2 = one global atomic per lane and step straight into the per-record rows (5.75 x 12 = 69 atomics per pair where the
//          shipping kernel issues 12-14)
template <int ACC>
__global__ __launch_bounds__(256) void row_loop(float *acc, int nsteps) {
  __shared__ float4 rec[BATCH * 3];
  __shared__ float sums[BATCH * 16];
  __shared__ unsigned char lists[16][BATCH];
  const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6, r = lane >> 4, l = lane & 15;
  stage(rec, tid, blockIdx.x);
  for (int i = tid; i < BATCH * 16; i += 256) sums[i] = 0.f;
  for (int b = 0; b < 16; b++) lists[b][tid] = (unsigned char)((tid * 7 + b * 29) & 255);  // every row walks another order
  __syncthreads();
  float T = 0.05f + 0.001f * lane, gB = 0.01f, gBr = 0.005f, g[CG];
  for (int c = 0; c < CG; c++) g[c] = 1e-3f * (lane + c - 32);
  // pixel of the lane: quadrant q, 4x4 block r of it, pixel l of the block
  const float px = (float)(8 * (q & 1) + 4 * (r & 1) + (l & 3)), py = (float)(8 * (q >> 1) + 4 * (r >> 1) + (l >> 2));
  const unsigned char *my_list = lists[4 * q + r];
  for (int s = 0; s < nsteps; s++) {
    const int j = my_list[s & (BATCH - 1)];  // row-uniform: one LDS byte, then three row-broadcast 16-byte reads
    const float4 r0 = rec[j * 3 + 0], r1 = rec[j * 3 + 1], r2 = rec[j * 3 + 2];
    const float bcol[6] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; i++) v[i] = 0.f;
    blend_bwd_pixel<CG, true, false>(v, T, gB, gBr, g, __fsub_rn(r0.x, px), __fsub_rn(r0.y, py), r0.z, r0.w, r1.x, r1.y, bcol, true);
    T = T > 1.0f ? 0.05f : T;
    const float tot = row_transpose_reduce12(v, lane);
    if constexpr (ACC == 1) {
      if (l < 12 && tot != 0.f) atomicAdd(&sums[j * 16 + l], tot);  // ds_add_f32: four addresses per instruction, one per row
    } else if constexpr (ACC == 2) {
      if (l < 12 && tot != 0.f) atomicAdd(acc + ((size_t)blockIdx.x * BATCH + j) * 16 + l, tot);
    } else {
      // (no accumulation at all -- NOT a correct kernel: the VALU-only floor of a row step, to separate the reduction's
      // instruction cost from the LDS atomics' in the timing)
      if (l < 12 && tot == 12345.f) sums[j * 16 + l] = tot;
    }
  }
  __syncthreads();
  float *row = acc + (size_t)blockIdx.x * BATCH * 16;
  for (int i = tid; i < BATCH * 16; i += 256)
    if (sums[i] != 0.f) atomicAdd(row + i, sums[i]);
}

template <typename F>
static float time_ms(F launch) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep) best = fminf(best, ms);
  }
  return best;
}

int main() {
  // the row reduction's lane -> value map, against plain sums
  {
    std::vector<float> in(64 * 12), out(64);
    srand(5);
    for (auto &x : in) x = (float)rand() / RAND_MAX - 0.5f;
    float *d_in, *d_out;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_out, 256));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    check_row_reduce<<<1, 64>>>(d_in, d_out);
    CK(hipMemcpy(out.data(), d_out, 256, hipMemcpyDeviceToHost));
    float worst = 0.f;
    for (int lane = 0; lane < 64; lane++) {
      if ((lane & 15) >= 12) continue;
      float want = 0.f;
      for (int m = 0; m < 16; m++) want += in[((lane & ~15) + m) * 12 + (lane & 15)];
      worst = fmaxf(worst, fabsf(out[lane] - want));
    }
    printf("row_transpose_reduce12: max |lane total - plain sum over its 16-lane row| = %.3g %s\n", worst, worst < 1e-5f ? "(ok)" : "(MISMATCH)");
  }
  const int tiles = 5120, npairs = 2048, nsteps = 2048;
  float *acc;
  CK(hipMalloc(&acc, (size_t)tiles * BATCH * 16 * 4));
  CK(hipMemset(acc, 0, (size_t)tiles * BATCH * 16 * 4));
  const float t2 = time_ms([&] { pair_loop<2><<<tiles, 64>>>(acc, npairs); });
  const float t3 = time_ms([&] { pair_loop<3><<<tiles, 64>>>(acc, npairs); });
  const float tr = time_ms([&] { row_loop<1><<<tiles, 256>>>(acc, nsteps); });
  const float tr0 = time_ms([&] { row_loop<0><<<tiles, 256>>>(acc, nsteps); });
  const float tr2 = time_ms([&] { row_loop<2><<<tiles, 256>>>(acc, nsteps); });
  // ns per pair (per step) and SIMD-slot: 5120 one-wave workgroups = 5 waves on each of 1024 SIMDs; 5120 x 4 waves = 20 per SIMD in
  // 2.5 generations of 8 -- both normalised to the WHOLE launch, which is what a kernel of 5120 tiles pays
  const double pair2 = t2 * 1e6 / ((double)npairs * tiles), pair3 = t3 * 1e6 / ((double)npairs * tiles), step = tr * 1e6 / ((double)nsteps * tiles * 4);
  printf("pair loop (shipping): NB=2 %.3f ms = %.4f ns per pair and tile-launch   NB=3 %.3f ms = %.4f ns   -> C2 mix (2.16 bodies) %.4f ns per pair\n",
         t2, pair2, t3, pair3, pair2 + 0.16 * (pair3 - pair2));
  const double step0 = tr0 * 1e6 / ((double)nsteps * tiles * 4);
  printf("row streams         : %.3f ms = %.4f ns per wave step (4 waves per tile: %.4f ns per tile step)\n", tr, step, 4 * step);
  const double step2 = tr2 * 1e6 / ((double)nsteps * tiles * 4);
  printf("  ... without the ds_add_f32 into the per-record LDS sums (VALU floor, not a correct kernel): %.3f ms = %.4f ns per wave step\n", tr0, step0);
  printf("  ... with one GLOBAL atomic per lane and step instead (no LDS sums, no flush): %.3f ms = %.4f ns per wave step\n", tr2, step2);
  const struct { const char *name; double steps, bodies; } sc[] = {{"C2", 1.556, 2.16}, {"C2 dense", 1.903, 2.32}, {"C4", 1.450, 1.98}};
  for (auto &c : sc) {
    const double pair = pair2 + (c.bodies - 2.0) * (pair3 - pair2);
    // a pair costs the row design `steps` wave steps IN EACH QUADRANT WAVE THAT HAS IT -- the counter is already per pair over the
    // tile's four quadrants (sum over quadrants of max over rows, divided by pairs)
    printf("  %-9s rows / shipping = %.3f x %.4f / %.4f = %.2f  (blend_bwd would take %.0f %% of today's time, before the per-batch "
           "masks, lists and flush; with the LDS atomics free: %.2f; with global atomics per step: %.2f)\n", c.name, c.steps, step, pair,
           c.steps * step / pair, 100.0 * c.steps * step / pair, c.steps * step0 / pair, c.steps * step2 / pair);
  }
  return 0;
}
