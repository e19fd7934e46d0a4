// raster_kernels.h -- kernels and host helpers shared by raster.hip (operator boundary) and
// render.hip (fused render).  See raster.hip for the design notes.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <cstdlib>
#include <cstring>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

namespace {
using namespace fsgs;


// ------------------------------------------------------------------------------------------------
// device-side parameter block (passed by value)
// ------------------------------------------------------------------------------------------------
// Diagnostics hooks of the blend launches (occupancy throttle by dynamic LDS, per-wave start / end stamps into a buffer
// whose ADDRESS comes from the environment): compiled in only by `FSGS_DIAG=1 python free-surgs_amd/build.py`, which
// scripts/dev/diag_tile_times.py and the occupancy experiments ask for.  The product build never reads these variables.
inline const char *diag_env(const char *name) {
#ifdef FSGS_DIAG_HOOKS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

struct CamParams {
  int W, H, gx, gy, flags;
  int bwd_prio_step;  // one-wave backward blend: entries per issue-priority level (0 = leave s_setprio alone); blend_bwd_prio_step()
  float tanfovx, tanfovy, fx, fy, scale_modifier;
  float V[16];
  float PM[16];
  float bg[FSGS_MAX_CHANNELS];
};

// ------------------------------------------------------------------------------------------------
// R1  per-Gaussian preprocess, shared by the operator boundary and the fused render
// ------------------------------------------------------------------------------------------------
struct Projected {
  int radius;
  uint32_t ntile, key;
  ushort4 rect;
  float2 xy;
  float4 conic_op;
  float tz;
};
// the tiles a Gaussian's 3-sigma square touches: [minx, maxx) x [miny, maxy) (UPSTREAM getRect); also how
// det_gather_kernel finds a Gaussian's pairs again from its stored mean2D and radius
__device__ __forceinline__ void tile_rect(int gx, int gy, float px, float py, int r_, int &minx, int &miny, int &maxx,
                                          int &maxy) {
  minx = min(gx, max(0, (int)((px - r_) / FSGS_TILE)));
  miny = min(gy, max(0, (int)((py - r_) / FSGS_TILE)));
  maxx = min(gx, max(0, (int)((px + r_ + FSGS_TILE - 1) / FSGS_TILE)));
  maxy = min(gy, max(0, (int)((py + r_ + FSGS_TILE - 1) / FSGS_TILE)));
}
// mean: Gaussian centre in the raster camera's world frame; s: activated scale * modifier;
// q: quaternion as handed over (no re-normalisation); opacity: activated.
__device__ __forceinline__ Projected project_gaussian(const CamParams &cam, float mx, float my, float mz, float3 s,
                                                      float4 q, float opacity) {
  Projected o;
  o.radius = 0;
  o.ntile = 0;
  o.key = 0xFFFFFFFFu;
  o.rect = make_ushort4(0, 0, 0, 0);
  o.xy = make_float2(0.f, 0.f);
  o.conic_op = make_float4(0.f, 0.f, 0.f, 0.f);
  o.tz = 0.f;
  const float *V = cam.V, *PM = cam.PM;
  float3 t;
  t.x = V[0] * mx + V[4] * my + V[8] * mz + V[12];
  t.y = V[1] * mx + V[5] * my + V[9] * mz + V[13];
  t.z = V[2] * mx + V[6] * my + V[10] * mz + V[14];
  if (!(t.z > 0.2f)) return o;  // near-plane cull
  float h0 = PM[0] * mx + PM[4] * my + PM[8] * mz + PM[12];
  float h1 = PM[1] * mx + PM[5] * my + PM[9] * mz + PM[13];
  float h3 = PM[3] * mx + PM[7] * my + PM[11] * mz + PM[15];
  float pw = 1.0f / (h3 + 0.0000001f);
  float ndcx = h0 * pw, ndcy = h1 * pw;
  Mat3 R = quat_to_R(q);
  float c6[6];
  cov3d(s, R, c6);
  Ewa e;
  ewa_project(V, t, c6, cam.fx, cam.fy, cam.tanfovx, cam.tanfovy, e);
  float det = e.a * e.c - e.b * e.b;
  if (det == 0.0f) return o;
  float det_inv = 1.0f / det;
  float mid = 0.5f * (e.a + e.c);
  float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
  float lam = fmaxf(mid + sq, mid - sq);
  int r_ = (int)ceilf(3.0f * sqrtf(lam));
  float px = ((ndcx + 1.0f) * cam.W - 1.0f) * 0.5f;
  float py = ((ndcy + 1.0f) * cam.H - 1.0f) * 0.5f;
  int minx, miny, maxx, maxy;
  tile_rect(cam.gx, cam.gy, px, py, r_, minx, miny, maxx, maxy);
  int area = (maxx - minx) * (maxy - miny);
  if (area <= 0) return o;
  const float cA = e.c * det_inv, cB = -e.b * det_inv, cC = e.a * det_inv;
  o.radius = r_;  // radii / visibility keep UPSTREAM's meaning even when no tile survives the footprint test
  o.xy = make_float2(px, py);
  o.conic_op = make_float4(cA, cB, cC, opacity);
  o.tz = t.z;
  if (footprint_tau(opacity) < 0.f) return o;  // alpha can never reach 1/255
  o.ntile = (uint32_t)area;                    // upper bound; the binning kernels apply the exact test per tile
  o.rect = make_ushort4((unsigned short)minx, (unsigned short)miny, (unsigned short)maxx, (unsigned short)maxy);
  o.key = __float_as_uint(t.z);  // positive float: the bit pattern is order preserving
  return o;
}

// What the blend kernels gather per (tile, Gaussian) pair: ONE 64-byte line per Gaussian instead of three or four
// scattered ones (xy 8 B, conic + opacity 16 B, depth 4 B, colours 4 C B in four arrays): float4[4] =
//   [0] mean2D x, y | conic A, B   [1] conic C | opacity | view depth | -   [2] colours 0..3   [3] colours 4..7
// Every pair used to cost up to four cache lines from the fabric whenever the tile's XCD had not seen the Gaussian yet
// (with the longest-first order that is almost always): 2.5x (forward) / 3.2x (backward) the algorithmic bytes.
constexpr int kRecF4 = 4;
struct GeomOut {  // per-Gaussian arrays written by every preprocess kernel
  float2 *xy;
  float4 *conic_op;
  float *depth;
  float4 *rec;  // [P * kRecF4], 64-byte aligned; colours are filled in by the calling kernel
  int32_t *radii;
  uint32_t *tiles;
  ushort4 *rect;
  uint32_t *tile_count;  // [tiles * BIN_SUBS] cursors of the binning pass, followed by {R, overflow report}
  int gx;
  uint32_t clear_words;  // > 0: the preprocess kernel clears that many words of tile_count (see clear_binning_cursors)
};
// The binning cursors must be zero when bin_scatter_kernel starts.  Nothing in front of it reads them, so the
// preprocess kernel -- the launch in front of it on the same stream -- clears them on the side (a few dozen words
// per workgroup) instead of a 5 us fill kernel of its own sitting in the step's critical path.
__device__ __forceinline__ void clear_binning_cursors(const GeomOut &g) {
  if (g.clear_words == 0) return;
  for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < g.clear_words; w += gridDim.x * blockDim.x)
    g.tile_count[w] = 0u;
}
// Same-address atomics serialise at ~46 ns each on MI355X (profiles/r01_atomic_scope_ubench.txt): the longest tile
// list alone would cost > 100 us per binning pass.  Every tile therefore owns BIN_SUBS counters / cursors (bin_slot below
// says which one a key takes), and its list is the concatenation of the BIN_SUBS sub-lists (the per-tile sort restores the
// (depth, index) order anyway).
#ifndef FSGS_BIN_SUBS
#define FSGS_BIN_SUBS 8  // (A/B build switch, a power of two >= 4: scripts/dev/ab_bin_subs.sh)
#endif
constexpr int BIN_SUBS = FSGS_BIN_SUBS;
static_assert(BIN_SUBS >= 4 && (BIN_SUBS & (BIN_SUBS - 1)) == 0, "bin_slot masks with BIN_SUBS - 1; the order pass reads the cursors as uint4");
// Which of a tile's BIN_SUBS sub-lists a key goes to: the Gaussian index (gaussian & 7).  Round 6 tried the XCD the writing
// workgroup runs on instead (HW_REG_XCC_ID; FSGS_BIN_SUB_XCC=1): every 64-byte line of a segment then collects its eight keys in ONE
// L2 and the key stores' write amplification drops (WRITE_SIZE 51.7 -> 45.3 MB; the rest is the cursor atomics, tallied at 32 B a
// request) -- but the 64 Gaussians of a wave are neighbours in the image (clouds are initialised in pixel order), so candidates of
// different Gaussians for the SAME tile then share one cursor instead of spreading over eight, and same-address atomics serialise:
// bin_scatter 27.7 -> 45.1 us at C2 (profiles/r06_bench_kernel_stats_xcc_sublists.csv).  Not kept.  FSGS_BIN_SUB_XCC=2: the same with the cursor atomic executed in the XCD's own
// L2 (workgroup scope, no sc1) -- 29.5 -> 46.6 us, no better: it is the shared cursor that serialises, not where the atomic runs
// (profiles/r06_ab_bin_l2_atomics.txt).  More sub-lists do not help either: FSGS_BIN_SUBS = 16 / 32 -> 50 / 47 us, 4 -> +-0
// (profiles/r06_ab_bin_subs_and_loss_v2.txt).  The per-tile sort orders by
// (depth, index), so the sub-list a key sat in never shows in a result either way.
#ifndef FSGS_BIN_SUB_XCC
#define FSGS_BIN_SUB_XCC 0
#endif
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
  return x;
}
__device__ __forceinline__ int bin_slot(int tile, int sub) { return tile * BIN_SUBS + (sub & (BIN_SUBS - 1)); }

__device__ __forceinline__ void store_projected(const GeomOut &g, int i, const Projected &o) {
  g.radii[i] = o.radius;
  g.tiles[i] = o.ntile;
  g.rect[i] = o.rect;
  g.xy[i] = o.xy;
  g.conic_op[i] = o.conic_op;
  g.depth[i] = o.tz;
  g.rec[(size_t)i * kRecF4 + 0] = make_float4(o.xy.x, o.xy.y, o.conic_op.x, o.conic_op.y);
  g.rec[(size_t)i * kRecF4 + 1] = make_float4(o.conic_op.z, o.conic_op.w, o.tz, 0.f);
}
template <int C>
__device__ __forceinline__ void store_record_colors(float4 *rec, int i, const float *c) {
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; k++) v[k] = k < C ? c[k < C ? k : 0] : 0.f;
  rec[(size_t)i * kRecF4 + 2] = make_float4(v[0], v[1], v[2], v[3]);
  if (C > 4) rec[(size_t)i * kRecF4 + 3] = make_float4(v[4], v[5], v[6], v[7]);
}

template <int C>
__global__ __launch_bounds__(256) void preprocess_fwd_kernel(int P, CamParams cam, const float *__restrict__ means3D,
                                                             const float *__restrict__ colors,
                                                             const float *__restrict__ opac,
                                                             const float *__restrict__ scales,
                                                             const float *__restrict__ rots, GeomOut g) {
  clear_binning_cursors(g);
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  {
    float c[C];
#pragma unroll
    for (int k = 0; k < C; k++) c[k] = colors[(size_t)i * C + k];
    store_record_colors<C>(g.rec, i, c);
  }
  float3 s = make_float3(cam.scale_modifier * scales[3 * i], cam.scale_modifier * scales[3 * i + 1],
                         cam.scale_modifier * scales[3 * i + 2]);
  float4 q = make_float4(rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]);
  Projected o = project_gaussian(cam, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], s, q, opac[i]);
  store_projected(g, i, o);
}

// ------------------------------------------------------------------------------------------------
// R2-R5  binning: per-tile lists built directly -- one pass, no global sort, no count pass, no scan
//   scatter : the (Gaussian, tile) candidates of a wave are flattened over its lanes; a candidate that passes the
//             exact footprint test (rect_touched: some pixel of the tile can reach alpha >= 1/255) claims
//             slot = atomic cursor of its (tile, sub-list) and writes the 64-bit key (depth bits << 32 | index)
//             into that sub-list's FIXED-CAPACITY segment: segment (tile, s) starts at (tile*8 + s) * cap_sub,
//             cap_sub = max_pairs / (8 tiles).  288 GB of HBM buy the slack; a segment that overflows is
//             reported (FSGS_ERR_CAPACITY + the capacity that would have sufficed) and the caller retries.
//   sort    : one workgroup per tile gathers its eight segments into LDS, sorts (normalised bitonic network,
//             virtual +inf padding), leaves the Gaussian indices in `plist` at the tile's base and writes
//             ranges[tile] = [base, base + n)
//   order   : one workgroup: longest-first dispatch order of the tiles, R = sum n, and the mailbox word
// The resulting order inside a tile is (depth, index) ascending = UPSTREAM's order including ties,
// independent of the order in which the atomics landed.
// ------------------------------------------------------------------------------------------------
// A thread-per-Gaussian loop over the tile rect is bound by the LARGEST rect of the launch (one lane walks it
// alone, and every step waits for a returning atomic), so the candidates of the 64 Gaussians of a wave are
// flattened instead: an inclusive scan of the rect areas, then lane l of step s takes candidate 64*s + l, finds
// its Gaussian by binary search in LDS and tests that one tile.  BIN_UNROLL candidates per lane keep several
// atomics in flight (two: more of them per lane only lengthen the wave's critical path -- every step waits for its slowest
// returning atomic).
#ifndef FSGS_BIN_UNROLL
#define FSGS_BIN_UNROLL 2  // (round 5 A/B, profiles/r05_ab_bin_unroll.txt: 2 -> 30.1 us, 4 -> 32.4, 8 -> 34.6 at C2; rounds 1-4 ran 4)
#endif
constexpr int BIN_UNROLL = FSGS_BIN_UNROLL;
__global__ __launch_bounds__(256) void bin_scatter_kernel(int P, int gx, const uint32_t *__restrict__ tiles,
                                                          const ushort4 *__restrict__ rect,
                                                          const float2 *__restrict__ xy,
                                                          const float4 *__restrict__ conic_op,
                                                          const float *__restrict__ depth,
                                                          uint32_t *__restrict__ cursors,
                                                          unsigned long long *__restrict__ keys, uint32_t cap_sub) {
  __shared__ float4 rec_a[256];  // mean2D x,y | conic A,B
  __shared__ float4 rec_b[256];  // conic C | tau | depth bits | Gaussian index
  __shared__ uint4 rec_c[256];   // rect x0,y0 | width | ceil(2^32 / width)
  __shared__ uint32_t pre[256];  // inclusive scan of the rect areas, per wave
  const int lane = threadIdx.x & 63, wbase = threadIdx.x & ~63;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int xcc = FSGS_BIN_SUB_XCC ? (int)xcc_id() : 0;  // wave-uniform
  uint32_t area = 0;
  if (i < P && tiles[i] != 0) {
    const ushort4 rc = rect[i];
    const float2 p = xy[i];
    const float4 co = conic_op[i];
    const float tau = footprint_tau(co.w);
    const uint32_t w = (uint32_t)(rc.z - rc.x), h = (uint32_t)(rc.w - rc.y);
    if (tau >= 0.f) area = w * h;
    rec_a[threadIdx.x] = make_float4(p.x, p.y, co.x, co.y);
    rec_b[threadIdx.x] = make_float4(co.z, tau, depth[i], __uint_as_float((uint32_t)i));
    rec_c[threadIdx.x] = make_uint4(rc.x, rc.y, w, w > 1 ? 0xFFFFFFFFu / w + 1u : 0u);
  }
  uint32_t incl = area;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += up;
  }
  pre[threadIdx.x] = incl;
  const uint32_t total = (uint32_t)readlane((int)incl, 63);
  __syncthreads();
  for (uint32_t base = 0; base < total; base += 64 * BIN_UNROLL) {
    // slot = 0xFFFFFFFF: no pair (a value no capacity reaches: the store below needs no separate "hit" flag -- as a bool
    // array the flag lived in a scalar mask register that one build variant's control flow left stale on a wave-uniform
    // early-out path, profiles/r05_ab_bin_unroll.txt)
    uint32_t slot[BIN_UNROLL], seg[BIN_UNROLL], klo[BIN_UNROLL], khi[BIN_UNROLL];
#pragma unroll
    for (int u = 0; u < BIN_UNROLL; u++) {
      const uint32_t item = base + (uint32_t)(u * 64 + lane);
      slot[u] = 0xFFFFFFFFu;
      seg[u] = klo[u] = khi[u] = 0u;
      if (item < total) {
        int own = 0;  // number of lanes whose inclusive prefix is <= item  ==  the owning lane
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
          if (pre[wbase + own + step - 1] <= item) own += step;
        const uint32_t local = item - (own ? pre[wbase + own - 1] : 0u);
        const float4 a = rec_a[wbase + own], b = rec_b[wbase + own];
        const uint4 c = rec_c[wbase + own];
        const uint32_t ry = c.z > 1 ? __umulhi(local, c.w) : local;  // local / width (exact: local * width < 2^32)
        const uint32_t rx = local - ry * c.z;
        const int tx = (int)(c.x + rx), ty = (int)(c.y + ry);
        if (rect_touched(a.x, a.y, a.z, a.w, b.x, b.y, (float)(tx * FSGS_TILE), (float)(ty * FSGS_TILE),
                         (float)FSGS_TILE, (float)FSGS_TILE)) {
          klo[u] = __float_as_uint(b.w);
          khi[u] = __float_as_uint(b.z);
          seg[u] = (uint32_t)bin_slot(ty * gx + tx, FSGS_BIN_SUB_XCC ? xcc : (int)klo[u]);
#if FSGS_BIN_SUB_XCC == 2
          // the cursor of (tile, XCD) is only ever touched from this XCD: the atomic can execute in the XCD's own L2 (no sc1:
          // workgroup scope), not at the memory side where agent-scope atomics of a multi-XCD part meet
          slot[u] = __hip_atomic_fetch_add(&cursors[seg[u]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
          slot[u] = atomicAdd(&cursors[seg[u]], 1u);
#endif
        }
      }
    }
#pragma unroll
    for (int u = 0; u < BIN_UNROLL; u++)
      if (slot[u] < cap_sub)  // an overflowing segment keeps counting (the sort kernel reports it)
        keys[(size_t)seg[u] * cap_sub + slot[u]] = ((unsigned long long)khi[u] << 32) | klo[u];
  }
}

constexpr int SORT_LDS_KEYS = 2048;  // 16 KB of LDS per workgroup; longer lists sort in global memory
constexpr int SORT_RANK_KEYS = 256;  // up to here: rank sort (one key per thread); above: bitonic network

// compare-exchange network over m (power of two) virtual elements, n real ones; every exchange puts the
// smaller key at the lower index, so the +inf padding (indices >= n) never moves and is never touched.
// Exchange t of a step belongs to chunk t >> 6, and in every step whose partner distance is < 128 (mirror steps
// with k <= 128, shuffle steps with j <= 64) the 64 exchanges of a chunk stay inside elements
// [128 chunk, 128 chunk + 128).  A chunk is always handled by the same wave, so between two such steps a wave
// only needs its own LDS traffic in order (WAVE_LOCAL): of the 36 steps of a 256-key sort only one needs the
// workgroup barrier on either side.  Lists sorted in global memory keep the barrier everywhere.
template <bool WAVE_LOCAL, typename Mem>
__device__ __forceinline__ void bitonic_sort_ascending(Mem keys, int n, int m) {
  bool prev_local = false;
  auto sync_before = [&](bool local) {
    if (WAVE_LOCAL && local && prev_local) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
      __syncthreads();
    }
    prev_local = local;
  };
  const int half = m >> 1;
  for (int k = 2, lk = 1; k <= m; k <<= 1, lk++) {
    const int hk = k >> 1;
    sync_before(k <= 128);
    for (int t = threadIdx.x; t < half; t += blockDim.x) {  // first step of the merge: mirror partner
      const int blk = t >> (lk - 1), off = t & (hk - 1);
      const int i = (blk << lk) + off, l = (blk << lk) + k - 1 - off;
      if (l < n) {
        unsigned long long a = keys[i], b = keys[l];
        if (a > b) { keys[i] = b; keys[l] = a; }
      }
    }
    for (int j = k >> 2; j > 0; j >>= 1) {
      sync_before(j <= 64);
      for (int t = threadIdx.x; t < half; t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), l = i + j;
        if (l < n) {
          unsigned long long a = keys[i], b = keys[l];
          if (a > b) { keys[i] = b; keys[l] = a; }
        }
      }
    }
  }
  __syncthreads();
}

// The dispatch order of the blend kernels (longest list first, see tile_order_kernel below), R and the mailbox word
// depend on the segment fill counts ALONE -- list length of a tile = sum of its eight clamped cursors -- which the
// scatter kernel has finished before the sort kernel starts.  So they need neither a launch of their own behind the
// sort nor its results: ONE extra workgroup of the sort launch (block 0, dispatched first) forms them while the other
// workgroups sort.  That removes a kernel boundary and a single-workgroup kernel (7 us at C2 on a 256-CU chip) from
// the forward's critical path, and the host -- which polls the mailbox before it can hand the blend's launch over --
// hears R a whole sort kernel earlier.  ntiles <= 8192.
// 256 bins of four list lengths each (lists of 1020 entries and more share the first bin: they start first either way);
// the tiles' bins are parked in LDS as bytes, so this path costs the sort kernel no registers (as one more launch-wide
// register array it took the whole kernel from 8 to 5 waves per SIMD).
__host__ __device__ __forceinline__ uint32_t *order_plain(uint32_t *order, int ntiles) { return order + ntiles + 16; }
__host__ __device__ __forceinline__ const uint32_t *order_plain(const uint32_t *order, int ntiles) { return order + ntiles + 16; }
constexpr int ORDER_BINS_FUSED = 256;
constexpr int ORDER_FOLD = 1024;  // SIMDs of the chip = the period of the workgroup -> SIMD placement (see below)
// The order is folded only when the WHOLE launch is resident from the start (blend_bwd holds 5 waves per SIMD: up to 5120
// tiles, C2's 1280 x 1024): with more tiles a freed slot takes the next workgroup, the balance is dynamic and wants the plain
// longest-first order (C4, 8160 tiles: folding cost blend_bwd 0.4 - 2.5 %, blend_fwd 1 %)
constexpr uint32_t ORDER_FOLD_ROUNDS = 5;
// Second half of the order construction: hist[] holds the bin counts, bins[] every tile's bin (0 = longest).  256 threads.  (A
// function of its own since round 6 tried to re-form the backward's order from the quadrant bodies the forward blend counts per tile
// -- weight 35 bodies + 63 pairs, one extra workgroup on the side stream: blend_bwd 232.5 vs 233 us at C2, 309 vs 314 dense,
// profiles/r06_ab_refine_order_and_fwd_pair.txt -- not worth an entry point; dropped.)
__device__ __forceinline__ void tile_order_finish(uint32_t *hist, const uint8_t *bins, int ntiles, uint32_t *__restrict__ order,
                                                  bool write_plain) {
  __shared__ uint32_t wave_tot[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  // exclusive scan of the 256 bins, one per thread
  const uint32_t c = hist[threadIdx.x];
  uint32_t incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
    if (lane >= off) incl += up;
  }
  if (lane == 63) wave_tot[wv] = incl;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < 4; w++)
    if (w < wv) before += wave_tot[w];
  hist[threadIdx.x] = before + incl - c;
  __syncthreads();
  // rank (0 = longest list) -> dispatch position.  Workgroup b of a one-wave-per-workgroup launch lands on the SAME SIMD as
  // b + 1024, b + 2048, ... (1024 SIMDs; measured: scripts/ubench/dispatch_map.hip, profiles/r03_dispatch_map.txt), and at C2
  // all 5120 tiles are resident from the start -- so a strictly descending order hands one SIMD the longest tile of EVERY
  // round of 1024 and another the shortest of every round (20 % above the mean for evenly spread lengths).  Laying some
  // rounds out BACKWARDS pairs a SIMD's long tiles with short ones.
  // WHICH rounds run backwards is chosen from the lengths at hand: a round's lists span `range` (first minus last, in bins of
  // four); the rounds are taken by descending range and each is laid against the slope accumulated so far.  (Plain
  // alternation left the two widest rounds of C2 -- the longest 1024 lists and the shortest 1024 -- running the same way:
  // the sums of a SIMD's list lengths then spread 899 .. 1154 around 1040 and the SIMDs finished 9 % apart.)
  const bool fold = (uint32_t)ntiles <= ORDER_FOLD * ORDER_FOLD_ROUNDS;
  __shared__ uint32_t s_rev;
  __shared__ int s_edge[2 * ORDER_FOLD_ROUNDS];
  const uint32_t nrounds = ((uint32_t)ntiles + ORDER_FOLD - 1) / ORDER_FOLD;
  if (fold) {
    if (threadIdx.x < 2 * nrounds) {
      const uint32_t k = threadIdx.x >> 1;
      const uint32_t r = k * ORDER_FOLD + ((threadIdx.x & 1u) ? min((uint32_t)ORDER_FOLD, (uint32_t)ntiles - k * ORDER_FOLD) - 1u : 0u);
      int b = 0;  // the bin of rank r: the largest b whose first rank is <= r (hist holds the bins' first ranks, ascending)
#pragma unroll
      for (int step = ORDER_BINS_FUSED / 2; step > 0; step >>= 1)
        if (hist[b + step] <= r) b += step;
      s_edge[threadIdx.x] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int range[ORDER_FOLD_ROUNDS];
      for (uint32_t k = 0; k < ORDER_FOLD_ROUNDS; k++) range[k] = k < nrounds ? s_edge[2 * k + 1] - s_edge[2 * k] : -1;
      uint32_t rev = 0;
      int acc = 0;  // > 0: the load so far falls along a round's positions
      for (uint32_t n = 0; n < nrounds; n++) {
        uint32_t best = 0;
        for (uint32_t k = 1; k < ORDER_FOLD_ROUNDS; k++)
          if (range[k] > range[best]) best = k;
        if (acc > 0) { rev |= 1u << best; acc -= range[best]; } else { acc += range[best]; }
        range[best] = -1;
      }
      s_rev = rev;
    }
    __syncthreads();
  }
  const uint32_t rev = fold ? s_rev : 0u;
  for (int i = (int)threadIdx.x; i < ntiles; i += 256) {
    const uint32_t rank = atomicAdd(&hist[bins[i]], 1u);
#if defined(FSGS_DIAG_HOOKS) && defined(FSGS_EXP_NO_FOLD)  // diagnostics flavour only: plain longest-first (a valid order)
    order[rank] = (uint32_t)i;
#else
    const uint32_t round = rank / ORDER_FOLD, idx = rank - round * ORDER_FOLD;
    const uint32_t m = min((uint32_t)ORDER_FOLD, (uint32_t)ntiles - round * ORDER_FOLD);  // the last round may be short
    order[round * ORDER_FOLD + (((rev >> round) & 1u) ? m - 1u - idx : idx)] = (uint32_t)i;
#endif
    // ... and the plain longest-first order behind it, for the four-waves-per-tile launches: their 256-thread workgroups are
    // not placed with the period the fold is built on, and plain LPT is 3 % faster for them (C2 blend_fwd 118.5 -> 115.0 us)
    if (write_plain) order_plain(order, ntiles)[rank] = (uint32_t)i;
  }
}

__device__ __forceinline__ void tile_order_from_cursors(uint32_t *smem /* >= 1 KB + 8 KB of LDS */, int ntiles,
                                                        const uint32_t *__restrict__ cursors, uint32_t cap_sub,
                                                        uint32_t *__restrict__ order, uint32_t *__restrict__ total_out,
                                                        uint32_t *host_word) {
  __shared__ uint32_t wave_len[4], wave_worst[4];
  uint32_t *hist = smem;                                             // [256]
  uint8_t *bins = reinterpret_cast<uint8_t *>(smem + ORDER_BINS_FUSED);  // [ntiles <= 8192]
  uint32_t mine = 0, worst = 0;
  hist[threadIdx.x] = 0;
#pragma unroll 4
  for (int i = (int)threadIdx.x; i < ntiles; i += 256) {  // strided ownership: coalesced 32-byte rows
    uint32_t n = 0;
#pragma unroll
    for (int q = 0; q < BIN_SUBS; q += 4) {
      const uint4 a = *reinterpret_cast<const uint4 *>(cursors + (size_t)i * BIN_SUBS + q);
      worst = max(worst, max(max(a.x, a.y), max(a.z, a.w)));
      n += min(a.x, cap_sub) + min(a.y, cap_sub) + min(a.z, cap_sub) + min(a.w, cap_sub);
    }
    mine += n;
    bins[i] = (uint8_t)(ORDER_BINS_FUSED - 1 - (int)min(n >> 2, (uint32_t)(ORDER_BINS_FUSED - 1)));  // bin 0 = longest
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  {
    uint32_t sm = mine, wm = worst;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      sm += (uint32_t)__shfl_xor((int)sm, off, 64);
      wm = max(wm, (uint32_t)__shfl_xor((int)wm, off, 64));
    }
    if (lane == 0) { wave_len[wv] = sm; wave_worst[wv] = wm; }
  }
  __syncthreads();
  for (int i = (int)threadIdx.x; i < ntiles; i += 256) atomicAdd(&hist[bins[i]], 1u);
  if (threadIdx.x == 0) {
    const uint32_t R = wave_len[0] + wave_len[1] + wave_len[2] + wave_len[3];
    const uint32_t w = max(max(wave_worst[0], wave_worst[1]), max(wave_worst[2], wave_worst[3]));
    const uint32_t need = w > cap_sub ? w : 0u;  // a segment overflowed: the capacity that would have sufficed
    total_out[0] = R;
    total_out[1] = need;
    if (host_word)
      __hip_atomic_store(host_word, need ? (0x80000000u | need) : R, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __syncthreads();
  tile_order_finish(hist, bins, ntiles, order, true);
}

// order != nullptr: the launch has ntiles + 1 workgroups, block 0 forms the dispatch order / R / mailbox word (above) and
// reports overflows itself; otherwise block b sorts tile b and overflows go to *overflow_need (read by tile_order_kernel).
__global__ __launch_bounds__(256) void sort_tiles_kernel(int ntiles, const uint32_t *__restrict__ cursors,
                                                         unsigned long long *__restrict__ keys,
                                                         uint32_t *__restrict__ plist, int2 *__restrict__ ranges,
                                                         uint32_t cap_sub, uint32_t *__restrict__ overflow_need,
                                                         uint32_t *__restrict__ order, uint32_t *__restrict__ total_out,
                                                         uint32_t *host_word) {
  __shared__ __attribute__((aligned(16))) unsigned long long lds[SORT_LDS_KEYS + 2];
  if (order != nullptr && blockIdx.x == 0) {
    tile_order_from_cursors(reinterpret_cast<uint32_t *>(lds), ntiles, cursors, cap_sub, order, total_out, host_word);
    return;
  }
  const int tile = order != nullptr ? (int)blockIdx.x - 1 : (int)blockIdx.x;
  // the eight segment fill counts (wave-uniform loads); a count above the capacity = dropped keys
  uint32_t cnt[BIN_SUBS], off[BIN_SUBS + 1], worst = 0;
  off[0] = 0;
#pragma unroll
  for (int s = 0; s < BIN_SUBS; s++) {
    const uint32_t c = cursors[tile * BIN_SUBS + s];
    worst = max(worst, c);
    cnt[s] = min(c, cap_sub);
    off[s + 1] = off[s] + cnt[s];
  }
  if (order == nullptr && worst > cap_sub && threadIdx.x == 0) atomicMax(overflow_need, worst);
  const int n = (int)off[BIN_SUBS];
  const size_t base = (size_t)tile * BIN_SUBS * cap_sub;  // of the tile's keys AND of its plist slice
  if (threadIdx.x == 0) ranges[tile] = make_int2((int)base, (int)base + n);
  if (n <= 0) return;
  int m = 1;
  while (m < n) m <<= 1;
  unsigned long long *gk = keys + base;
  if (n <= SORT_LDS_KEYS) {
    // the eight sub-lists are staged as ONE flat range [0, n): a loop per sub-list is eight dependent load -> LDS-store
    // round trips in a row (most sub-lists are shorter than the workgroup), this is one (two beyond 256 keys)
    for (int i0 = 0; i0 < n; i0 += 2 * (int)blockDim.x) {
      unsigned long long kv[2];
      int at[2];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        at[u] = i0 + u * (int)blockDim.x + (int)threadIdx.x;
        const uint32_t i = (uint32_t)min(at[u], n - 1);
        int sseg = 0;
#pragma unroll
        for (int q = 1; q < BIN_SUBS; q++) sseg += (i >= off[q]) ? 1 : 0;  // off[] ascends; empty sub-lists are skipped over
        uint32_t o = off[0];
#pragma unroll
        for (int q = 1; q < BIN_SUBS; q++) o = (sseg >= q) ? off[q] : o;
        kv[u] = gk[(size_t)sseg * cap_sub + (i - o)];
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
        if (at[u] < n) lds[at[u]] = kv[u];
    }
    if (n <= SORT_RANK_KEYS) {
      // Short lists (most tiles: mean 214 at C2): RANK sort.  Keys are unique (the Gaussian index sits in the low
      // word), so a key's final position is the number of keys below it; thread t counts that for key t against
      // the whole list, two list keys per broadcast ds_read_b128.  n^2 compares, but the only LDS traffic is
      // broadcast reads -- the bitonic network moves every key through LDS 36 times and is LDS-bandwidth-bound.
      // (Two keys per thread, n <= 512, was measured too: slower than the network from n ~ 300.)
      if (threadIdx.x == 0) { lds[n] = ~0ull; lds[n + 1] = ~0ull; }  // +inf padding for the pairwise reads
      __syncthreads();
      const int t = threadIdx.x;
      const bool mine = t < n;
      const unsigned long long key = mine ? lds[t] : 0ull;
      uint32_t rank = 0;
      const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(lds);
      for (int j = 0; j < (n + 1) >> 1; j++) {
        const ulonglong2 ab = pairs[j];
        rank += (ab.x < key) + (ab.y < key);
      }
      if (mine) plist[base + rank] = (uint32_t)key;
    } else if (n <= 2 * SORT_RANK_KEYS) {
      // 257..512 keys: two rank-sorted halves A = keys [0,256), B = keys [256,n), merged by rank: a key's final
      // position = its rank in its own half + the number of keys of the OTHER half below it (binary search in the
      // sorted other half).  ~n/2 broadcast reads + 9 probes per key instead of 45 network steps through LDS.
      __shared__ __attribute__((aligned(16))) unsigned long long half_sorted[2 * SORT_RANK_KEYS + 2];
      const int nB = n - SORT_RANK_KEYS;
      if (threadIdx.x == 0) { lds[n] = ~0ull; lds[n + 1] = ~0ull; }
      __syncthreads();
      const int t = threadIdx.x;
      const bool hasB = t < nB;
      const unsigned long long kA = lds[t], kB = hasB ? lds[SORT_RANK_KEYS + t] : 0ull;
      uint32_t rA = 0, rB = 0;
      const ulonglong2 *pairs = reinterpret_cast<const ulonglong2 *>(lds);
      for (int j = 0; j < SORT_RANK_KEYS / 2; j++) {  // A: exactly 256 keys
        const ulonglong2 ab = pairs[j];
        rA += (ab.x < kA) + (ab.y < kA);
      }
      for (int j = SORT_RANK_KEYS / 2; j < (n + 1) >> 1; j++) {  // B (+inf padded)
        const ulonglong2 ab = pairs[j];
        rB += (ab.x < kB) + (ab.y < kB);
      }
      half_sorted[rA] = kA;
      if (hasB) half_sorted[SORT_RANK_KEYS + rB] = kB;
      __syncthreads();
      // number of keys of a sorted array below `key` (keys are unique across the halves too)
      auto below = [&](const unsigned long long *arr, int len, unsigned long long key) {
        int lo = 0;
#pragma unroll
        for (int step = 256; step > 0; step >>= 1)
          if (lo + step <= len && arr[lo + step - 1] < key) lo += step;
        return lo;
      };
      plist[base + rA + below(half_sorted + SORT_RANK_KEYS, nB, kA)] = (uint32_t)kA;
      if (hasB) plist[base + rB + below(half_sorted, SORT_RANK_KEYS, kB)] = (uint32_t)kB;
    } else {
      bitonic_sort_ascending<true>(lds, n, m);
      for (int i = threadIdx.x; i < n; i += blockDim.x) plist[base + i] = (uint32_t)lds[i];
    }
  } else {  // rare: a tile with more than 2048 Gaussians sorts in place in global memory (L2 resident)
    // close the gaps between the segments first: segment s moves down to off[s] (destination <= source; a chunk
    // is read by everybody before anybody writes it, chunks ascend)
    for (int s = 1; s < BIN_SUBS; s++) {
      if (off[s] == (uint32_t)s * cap_sub) continue;  // already in place
      for (uint32_t c0 = 0; c0 < cnt[s]; c0 += blockDim.x) {
        const uint32_t i = c0 + threadIdx.x;
        unsigned long long v = 0;
        if (i < cnt[s]) v = gk[(size_t)s * cap_sub + i];
        __syncthreads();
        if (i < cnt[s]) gk[off[s] + i] = v;
        __syncthreads();
      }
    }
    bitonic_sort_ascending<false>(gk, n, m);
    for (int i = threadIdx.x; i < n; i += blockDim.x) plist[base + i] = (uint32_t)gk[i];
  }
}

// Dispatch order of the blend kernels: workgroup b takes tile order[b].
//   * longest-processing-time first: tiles sorted by descending list length, so long lists start first and
//     the tail of the launch is made of short ones (5120 tiles on ~4096-5120 wave slots is barely more than
//     one "round"; the stragglers must be cheap);
//   * XCD-aware: the dispatcher places workgroup b on XCD b % 8 (observed, speed only), and every XCD has its
//     own L2.  The image is cut into 8 bands of consecutive tiles; band x is sorted on its own and its i-th
//     longest tile goes to position 8*i + x, so each L2 only ever sees the Gaussians of one band
//     (without this the longest-first order scatters every XCD over the whole image and the fabric
//     traffic of the blend kernels triples).
// the sort launch's own order workgroup (tile_order_from_cursors) forms the orders: up to 8192 tiles, not with the XCD-banded order
inline bool fused_order_workgroup(int flags, int ntiles) { return !(flags & FSGS_FLAG_XCD_BANDED_ORDER) && ntiles <= 8 * 1024; }

constexpr int ORDER_MAX_TILES = 1 << 20;
constexpr int ORDER_BINS = 1024;
constexpr int ORDER_XCD = 8;
// bands of EQUAL WORK, not equal tile count: with start_i = the prefix sum of the list lengths, tile i
// belongs to band floor(8 * start_i / R).  Bands are interleaved round-robin (position 8*rank + band while
// every band still has tiles, dense afterwards), so the order is a permutation of [0, ntiles).
// nbands = 1: plain longest-first over the whole image (fastest: 568 vs 624 us for both blend kernels at C2,
// because in-order dispatch stalls whenever one XCD is momentarily fuller than the others);
// nbands = 8: XCD-banded (1.7x instead of 3x the algorithmic fabric traffic).  FsgsRasterCfg.flags bit 0.
// Also the end of the binning: R = sum of the list lengths goes to `total_out` and, together with the overflow
// report of the sort kernel, to the pinned host mailbox (bit 31 set = a segment overflowed, low bits = the segment
// capacity that would have sufficed; otherwise R).
__global__ __launch_bounds__(1024) void tile_order_kernel(int ntiles, int nbands, const int2 *__restrict__ ranges,
                                                          uint32_t *__restrict__ order,
                                                          uint32_t *__restrict__ total_out,
                                                          const uint32_t *__restrict__ overflow_need,
                                                          uint32_t *host_word) {
  __shared__ uint32_t hist[ORDER_XCD][ORDER_BINS];
  __shared__ uint32_t part[ORDER_XCD][128];
  __shared__ uint32_t band_size[ORDER_XCD];
  __shared__ uint32_t wave_tot[16];
  // prefix of the list lengths over the tiles (thread t owns tiles [t*per, (t+1)*per)) and their total
  const int per = (ntiles + 1023) / 1024;
  const int t0 = threadIdx.x * per;
  uint32_t mine = 0;
  for (int q = 0; q < per; q++)
    if (t0 + q < ntiles) {
      const int2 rg = ranges[t0 + q];
      mine += (uint32_t)(rg.y - rg.x);
    }
  uint32_t incl = mine;
  {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)incl, off, 64);
      if (lane >= off) incl += up;
    }
    if (lane == 63) wave_tot[threadIdx.x >> 6] = incl;
  }
  for (int i = threadIdx.x; i < ORDER_XCD * ORDER_BINS; i += blockDim.x) (&hist[0][0])[i] = 0;
  if (threadIdx.x < ORDER_XCD) band_size[threadIdx.x] = 0;
  __syncthreads();
  uint32_t start_of_mine = incl - mine, Rtot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const uint32_t v = wave_tot[w];
    if (w < (int)(threadIdx.x >> 6)) start_of_mine += v;
    Rtot += v;
  }
  if (threadIdx.x == 0) {
    *total_out = Rtot;
    const uint32_t need = *overflow_need;
    if (host_word)
      __hip_atomic_store(host_word, need ? (0x80000000u | need) : Rtot, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const unsigned long long R = (unsigned long long)max(Rtot, 1u);
  uint32_t run = start_of_mine;
  for (int q = 0; q < per; q++) {
    const int i = t0 + q;
    if (i >= ntiles) break;
    const int2 rg = ranges[i];
    const int len = rg.y - rg.x;
    const int n = min(len, ORDER_BINS - 1);
    const int band = nbands == 1 ? 0 : (int)min((unsigned long long)(nbands - 1), (unsigned long long)run * nbands / R);
    run += (uint32_t)len;
    atomicAdd(&hist[band][ORDER_BINS - 1 - n], 1u);  // bin 0 = longest lists
    atomicAdd(&band_size[band], 1u);
  }
  __syncthreads();
  {  // exclusive scan of each band's 1024 bins: 128 threads per band, 8 bins each
    const int band = threadIdx.x >> 7, t = threadIdx.x & 127;
    uint32_t loc[8], s = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { loc[q] = hist[band][t * 8 + q]; s += loc[q]; }
    // 128 threads per band = two waves: wave-level inclusive scan + the other wave's total
    const int lane = threadIdx.x & 63;
    uint32_t inc2 = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t up = (uint32_t)__shfl_up((int)inc2, off, 64);
      if (lane >= off) inc2 += up;
    }
    if (lane == 63) part[band][t >> 6] = inc2;
    __syncthreads();
    uint32_t r2 = inc2 - s + ((t >> 6) ? part[band][0] : 0u);
#pragma unroll
    for (int q = 0; q < 8; q++) { hist[band][t * 8 + q] = r2; r2 += loc[q]; }
  }
  __syncthreads();
  run = start_of_mine;
  for (int q = 0; q < per; q++) {
    const int i = t0 + q;
    if (i >= ntiles) break;
    const int2 rg = ranges[i];
    const int len = rg.y - rg.x;
    const int n = min(len, ORDER_BINS - 1);
    const int band = nbands == 1 ? 0 : (int)min((unsigned long long)(nbands - 1), (unsigned long long)run * nbands / R);
    run += (uint32_t)len;
    uint32_t rank = atomicAdd(&hist[band][ORDER_BINS - 1 - n], 1u);  // rank inside the band, longest first
    uint32_t pos = 0;
#pragma unroll
    for (int b2 = 0; b2 < ORDER_XCD; b2++) {
      if (b2 >= nbands) break;
      uint32_t sz = band_size[b2];
      pos += min(rank, sz) + ((b2 < band && sz > rank) ? 1u : 0u);
    }
    order[pos] = (uint32_t)i;
    order_plain(order, ntiles)[pos] = (uint32_t)i;  // (this path has one order for every flavour)
  }
}

// Pixel ownership inside a 16x16 tile: four 8x8 quadrants, lane l owns pixel (l & 7, l >> 3) of each
// quadrant k = 0..3 at offset (8*(k&1), 8*(k>>1)).  Whether a Gaussian can reach a quadrant at all is a
// wave-uniform question answered once per pair by its owning lane (rect_touched, exact), carried as a
// 4-bit mask and tested on the scalar unit -- on the C2 scene only ~2.2 of 4 quadrants survive.
struct TilePix {
  int x[4], y[4];
};
__device__ __forceinline__ TilePix tile_pixels(int tile, int gx, int lane) {
  TilePix t;
  const int x0 = (tile % gx) * FSGS_TILE + (lane & 7), y0 = (tile / gx) * FSGS_TILE + (lane >> 3);
#pragma unroll
  for (int k = 0; k < 4; k++) {
    t.x[k] = x0 + FSGS_QUAD * (k & 1);
    t.y[k] = y0 + FSGS_QUAD * (k >> 1);
  }
  return t;
}
__device__ __forceinline__ uint32_t quadrant_mask(int tile, int gx, float2 gxy, float4 gco) {
  const float tau = footprint_tau(gco.w);
  const float x0 = (float)((tile % gx) * FSGS_TILE), y0 = (float)((tile / gx) * FSGS_TILE);
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 4; k++)
    m |= rect_touched(gxy.x, gxy.y, gco.x, gco.y, gco.z, tau, x0 + FSGS_QUAD * (k & 1), y0 + FSGS_QUAD * (k >> 1),
                      (float)FSGS_QUAD, (float)FSGS_QUAD)
             ? (1u << k)
             : 0u;
  return m;
}

#ifdef FSGS_DIAG_HOOKS
// Lane utilisation of the blend kernels (diagnostics flavour only, scripts/lane_utilisation.py): per executed
// 8x8 quadrant body the number of lanes that really blend (forward) / carry a non-zero alpha (backward), and per
// (tile, Gaussian) pair what a body over 64 lanes chosen by 4x4-pixel blocks could save.  Lane l owns pixel (l & 7, l >> 3)
// of every quadrant, so 4x4 block b = 2 (y >> 2) + (x >> 2) of a quadrant is a fixed set of 16 lanes; a packed body gives
// lane set b the pixels of ONE quadrant, so a pair needs max_b #{quadrants whose block b is alive} bodies instead of one
// per reachable quadrant.  "alive" by the lanes that contributed (the ceiling) and by the exact footprint test on the
// 4x4 block (what a kernel could decide up front).  Wave-uniform counters, summed into a global block of 32 u64 at the end:
//   0 pairs  1 bodies  2 contributing lanes  3..11 bodies by contributing lanes (0 | 1-8 | ... | 57-64)
//   12 packed bodies (contributing blocks)  13 alive blocks (contributing)  14 packed bodies (footprint test)
//   15 alive blocks (footprint test)  16 pairs whose packed count (footprint test) is below their body count
struct DiagLanes {
  uint32_t pairs, bodies, lanes, packed_true, blocks_true, packed_rect, blocks_rect, pairs_gain;
  // round 6 (VERDICT r5 #1 b, "four independent 4x4-block streams per wave"): if each 16-lane row of a quadrant's wave walked
  // the list of ITS 4x4 block (alive by the footprint test, behind nobody's deepest contributor), a batch of 256 records would
  // take max_b n_b wave steps for sum_b n_b row bodies: row_steps / row_alive summed over (quadrant, 256-record batch)
  uint32_t row_steps, row_alive;
  uint32_t cnt[16];  // alive records of the current 256-record batch per (quadrant, block)
};
__device__ __forceinline__ void diag_rows_flush(DiagLanes &d) {
#pragma unroll
  for (int k = 0; k < 4; k++) {
    d.row_steps += max(max(d.cnt[4 * k], d.cnt[4 * k + 1]), max(d.cnt[4 * k + 2], d.cnt[4 * k + 3]));
    d.row_alive += d.cnt[4 * k] + d.cnt[4 * k + 1] + d.cnt[4 * k + 2] + d.cnt[4 * k + 3];
  }
#pragma unroll
  for (int i = 0; i < 16; i++) d.cnt[i] = 0;
}
__device__ __forceinline__ uint32_t diag_block_bits(unsigned long long ballot) {  // 4 bits: 4x4 blocks with a set lane
  const unsigned long long B0 = 0x0F0F0F0Full, B1 = 0xF0F0F0F0ull;
  return ((ballot & B0) ? 1u : 0u) | ((ballot & B1) ? 2u : 0u) | ((ballot & (B0 << 32)) ? 4u : 0u) |
         ((ballot & (B1 << 32)) ? 8u : 0u);
}
__device__ __forceinline__ uint32_t diag_packed(uint32_t m16) {  // bit 4 k + b: block b of quadrant k alive
  uint32_t best = 0;
#pragma unroll
  for (int b = 0; b < 4; b++) best = max(best, (uint32_t)__popc((m16 >> b) & 0x1111u));
  return best;
}
// footprint test of the sixteen 4x4 blocks of a tile for one Gaussian (bit 4 k + b)
__device__ __forceinline__ uint32_t diag_block_mask16(int tile, int gx, float2 gxy, float4 gco) {
  const float tau = footprint_tau(gco.w);
  const float x0 = (float)((tile % gx) * FSGS_TILE), y0 = (float)((tile / gx) * FSGS_TILE);
  uint32_t m = 0;
  for (int k = 0; k < 4; k++)
    for (int b = 0; b < 4; b++)
      m |= rect_touched(gxy.x, gxy.y, gco.x, gco.y, gco.z, tau, x0 + FSGS_QUAD * (k & 1) + 4.f * (b & 1),
                        y0 + FSGS_QUAD * (k >> 1) + 4.f * (b >> 1), 4.f, 4.f)
               ? (1u << (4 * k + b))
               : 0u;
  return m;
}
__device__ __forceinline__ void diag_flush(unsigned long long *out, const DiagLanes &d, const uint32_t *hist, int lane) {
  if (!out || lane != 0) return;
  atomicAdd(out + 0, (unsigned long long)d.pairs);
  atomicAdd(out + 1, (unsigned long long)d.bodies);
  atomicAdd(out + 2, (unsigned long long)d.lanes);
  for (int i = 0; i < 9; i++) atomicAdd(out + 3 + i, (unsigned long long)hist[i]);
  atomicAdd(out + 12, (unsigned long long)d.packed_true);
  atomicAdd(out + 13, (unsigned long long)d.blocks_true);
  atomicAdd(out + 14, (unsigned long long)d.packed_rect);
  atomicAdd(out + 15, (unsigned long long)d.blocks_rect);
  atomicAdd(out + 16, (unsigned long long)d.pairs_gain);
  atomicAdd(out + 17, (unsigned long long)d.row_steps);
  atomicAdd(out + 18, (unsigned long long)d.row_alive);
}
#endif

// The ONE per-(pixel, Gaussian) forward blend step, shared by every flavour of the forward blend (one wave
// per tile, four waves per tile, diagnostics): returns whether the lane blended the Gaussian into its pixel.
// T > 0: transmittance of a pixel that is still blending; a FINISHED pixel keeps its transmittance with the sign flipped.
template <int CP, bool WITH_DEPTH>
__device__ __forceinline__ bool blend_fwd_pixel(float &T, float &D, float2v (&acc)[CP], uint32_t &last, float dx, float dy,
                                                float bA, float bB, float bC, float bo, float bz,
                                                const float2v (&bcol2)[4], uint32_t pos) {
  if (!(T > 0.f)) return false;  // finished
  SplatEval e;
  if (!splat_alpha(dx, dy, bA, bB, bC, bo, e)) return false;
  const float test_T = T * (1.0f - e.alpha);
  if (test_T < 0.0001f) {
    T = -T;
    return false;
  }
  const float w = e.alpha * T;
#pragma unroll
  for (int cp = 0; cp < CP; cp++) acc[cp] = __builtin_elementwise_fma(bcol2[cp], float2v{w, w}, acc[cp]);
  if (WITH_DEPTH) D = fmaf(bz, w, D);
  T = test_T;
  last = pos;
  return true;
}

// Branch-free form of the same step (A/B: FSGS_FWD_BRANCHFREE): 98 % of the executed quadrant bodies have at least one lane that
// blends, so the three nested lane-mask branches of blend_fwd_pixel almost never skip an instruction for the WAVE -- they cost
// ~10 scalar instructions per body and put the colour reads behind a second LDS round trip.  Here every lane runs the whole body
// with w = 0 where it does not blend (acc + 0 c = acc, T (1 - 0) = T: the same bits), the skip decision is splat_alpha_masked's
// (bit-identical to splat_alpha), and all of the record is read in one round.
template <int CP, bool WITH_DEPTH>
__device__ __forceinline__ void blend_fwd_pixel_branchfree(float &T, float &D, float2v (&acc)[CP], uint32_t &last, float dx, float dy,
                                                           float bA, float bB, float bC, float bo, float bz,
                                                           const float2v (&bcol2)[4], uint32_t pos) {
  SplatEval e;
#if FSGS_FWD_BF_LEAN
  // (A/B) the three lane-mask combinations (ok && .., ok && !stop twice) replaced by what the values already say: alpha = 0 where
  // the lane skips the pair, so test_T = T there, and T >= 1e-4 for every pixel still blending (a smaller one was stopped) while a
  // finished pixel's T is negative -- "0 <= test_T < 1e-4" is ONE unsigned compare of the bits and is true exactly where
  // ok && test_T < 1e-4; and the lane blends exactly where its weight is positive.
  (void)splat_alpha_masked(dx, dy, bA, bB, bC, bo, T > 0.f, e);
  const float test_T = T * (1.0f - e.alpha);
  const bool stop = __float_as_uint(test_T) < 0x38D1B717u;  // bits of 1e-4f
  const float w = stop ? 0.0f : e.alpha * T;
#pragma unroll
  for (int cp = 0; cp < CP; cp++) acc[cp] = __builtin_elementwise_fma(bcol2[cp], float2v{w, w}, acc[cp]);
  if (WITH_DEPTH) D = fmaf(bz, w, D);
  last = w > 0.0f ? pos : last;
  T = stop ? -T : test_T;
#else
  const bool ok = splat_alpha_masked(dx, dy, bA, bB, bC, bo, T > 0.f, e);  // alpha = 0 for finished pixels and skipped pairs
  const float test_T = T * (1.0f - e.alpha);                                  // == T where alpha == 0
  const bool stop = ok && test_T < 0.0001f;
  const float w = (ok && !stop) ? e.alpha * T : 0.0f;
#pragma unroll
  for (int cp = 0; cp < CP; cp++) acc[cp] = __builtin_elementwise_fma(bcol2[cp], float2v{w, w}, acc[cp]);
  if (WITH_DEPTH) D = fmaf(bz, w, D);
  last = (ok && !stop) ? pos : last;
  T = stop ? -T : test_T;
#endif
}

// out_color holds channels [0, min(C,3)); channels >= 3 go to out_color2 (the fused render's depth /
// silhouette / depth^2 planes).  WITH_DEPTH: also accumulate the depth-fork's third output.
// DIAG (diagnostics flavour of the library only, `FSGS_DIAG=1 python free-surgs_amd/build.py`): the SAME instruction stream plus
// instrumentation behind `if constexpr`; the product library instantiates DIAG = 0 only.  Levels:
//   1  per-workgroup start / end stamps and the scalar pair / body counts (FSGS_DBG_TILE_TIMES*) -- within noise of the product;
//   2  + the lane-utilisation counters (FSGS_DBG_LANES*): one ballot + popcount per quadrant body and a 16-block footprint mask
//      per staged record -- picked by the launcher only when the counter buffer is set (scripts/lane_utilisation.py).
constexpr int kDiagStampWords = 6;
template <int DIAG>
struct DiagPtrs {
  // FSGS_DBG_TILE_TIMES*: kDiagStampWords u64 per workgroup -- start, end (s_memrealtime, 100 MHz), pairs << 32 | bodies,
  // list << 32 | walked, and (round 6) start, end in s_memtime ticks = shader cycles: delta cycles / delta 100 MHz ticks is the
  // shader clock the tile's wave ran at (VERDICT r5 #4 a: valu_frac had assumed 2.4 GHz)
  unsigned long long *times;
  unsigned long long *lanes;  // FSGS_DBG_LANES*: 32 u64 lane-utilisation counters (diag_flush)
};
template <>
struct DiagPtrs<0> {};
#ifdef FSGS_DIAG_HOOKS
constexpr bool kDiagBuild = true;
#else
constexpr bool kDiagBuild = false;
#endif

template <int C, bool WITH_DEPTH, int DIAG = 0>
__global__ __launch_bounds__(64) void blend_fwd_kernel(
    CamParams cam, int ntiles, const uint32_t *__restrict__ order, const int2 *__restrict__ ranges,
    const uint32_t *__restrict__ plist, const float4 *__restrict__ grec, float *__restrict__ final_T,
    uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, float *__restrict__ out_color2,
    float *__restrict__ out_depth, DiagPtrs<DIAG> dbg) {
  // dbg.times (FSGS_DBG_TILE_TIMES_FWD / FSGS_DBG_TILE_TIMES, scripts/dev/diag_tile_times.py only):
  // 100 MHz wall-clock stamps of this wave's start and end, to measure load balance and the kernel's tail
  unsigned long long dbg_t0 = 0ull, dbg_c0 = 0ull;
  uint32_t dbg_bodies = 0, dbg_pairs = 0;  // quadrant bodies executed / pairs not skipped altogether (scalar counters)
#ifdef FSGS_DIAG_HOOKS
  DiagLanes dl{};
  __shared__ uint32_t dbg_hist[9];
  if constexpr (DIAG) {
    if (dbg.times) {
      dbg_t0 = wall_clock64();
      dbg_c0 = __builtin_readcyclecounter();
    }
    if (DIAG >= 2 && threadIdx.x < 9) dbg_hist[threadIdx.x] = 0;
  }
#else
  static_assert(!DIAG, "the diagnostics flavour needs FSGS_DIAG_HOOKS");
#endif
  // one 64-lane workgroup per tile: the dispatcher refills a SIMD slot as soon as ONE tile is done
  constexpr int REC4 = C > 4 ? 4 : 3;  // float4s per staged record
  __shared__ float4 rec[64 * REC4];
  const int lane = threadIdx.x;
  const uint32_t tile_u = order ? order[blockIdx.x] : (uint32_t)xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile_u >= (uint32_t)ntiles) return;  // also the 0xFFFFFFFF holes of the banded order
  const int tile = (int)tile_u;
  const int W = cam.W, H = cam.H;
  const TilePix tp = tile_pixels(tile, cam.gx, lane);
  const float px0 = (float)tp.x[0], py0 = (float)tp.y[0];  // the lane's pixel in quadrant 0 (the others: + 8, quad_offset)
  // T[k] > 0: transmittance of a pixel that is still blending; a FINISHED pixel (T (1 - alpha) < 1e-4 seen, or outside
  // the image) keeps its transmittance with the sign flipped -- the "done" flag costs no register and one compare
  // (86 -> 80 VGPRs: six waves per SIMD; one VALU less per quadrant body)
  // the colour sums are kept as channel PAIRS: a pair's update is one v_pk_fma_f32 with the record's (x, y) / (z, w) halves
  // as they come out of the broadcast read.  Written out here because the build runs without the SLP vectoriser (which
  // formed these pairs by itself, but cost blend_bwd 5 % with the register shuffles it added there).
  constexpr int CP = (C + 1) / 2;
  float T[4], D[4];
  float2v acc[4][CP];
  uint32_t last[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    T[k] = (tp.x[k] < W && tp.y[k] < H) ? 1.0f : -1.0f;
    D[k] = 0.0f;
    last[k] = 0;
#pragma unroll
    for (int cp = 0; cp < CP; cp++) acc[k][cp] = float2v{0.0f, 0.0f};
  }
  const int2 rg = ranges[tile];
  for (int base = rg.x; base < rg.y; base += 64) {
    // quadrants whose 64 pixels are all finished (or outside the image) need no more work
    uint32_t alive = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) alive |= (__ballot(T[k] > 0.f) != 0ull) ? (1u << k) : 0u;
    if (alive == 0) break;
    const int n = min(64, rg.y - base);
    // lane j gathers record j of this batch
    uint32_t g = plist[base + (lane < n ? lane : 0)];
    const float4 *rp = grec + (size_t)g * kRecF4;  // one 64-byte line
    const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
    const float4 q3 = C > 4 ? rp[3] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float2 gxy = make_float2(q0.x, q0.y);
    const float4 gco = make_float4(q0.z, q0.w, q1.x, q1.y);
    const float gz = WITH_DEPTH ? q1.z : 0.f;
    const float g8[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    float gcol[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) gcol[ch] = g8[ch];
    const uint32_t gmask = quadrant_mask(tile, cam.gx, gxy, gco);
    uint32_t gmask16 = 0;
#ifdef FSGS_DIAG_HOOKS
    if constexpr (DIAG >= 2) gmask16 = diag_block_mask16(tile, cam.gx, gxy, gco);
#endif
    // park the 64 records in LDS: v_readlane costs ~8 cycles each on gfx950 (SGPR write -> VALU read), 13 of them
    // per pair were as expensive as half the blending arithmetic; a same-address ds_read_b128 is a broadcast
    __syncthreads();  // previous batch fully consumed (single-wave workgroup: this is just a wait)
    const SplatCoef kf = splat_coef(gco.x, gco.y, gco.z);
    rec[lane * REC4 + 0] = make_float4(gxy.x, gxy.y, kf.a, kf.b);
    rec[lane * REC4 + 1] = make_float4(kf.c, gco.w, gz, 0.f);
    {
      float c6[8];
#pragma unroll
      for (int ch = 0; ch < 8; ch++) c6[ch] = ch < C ? gcol[ch < C ? ch : 0] : 0.f;
      rec[lane * REC4 + 2] = make_float4(c6[0], c6[1], c6[2], c6[3]);
      if (C > 4) rec[lane * REC4 + 3] = make_float4(c6[4], c6[5], c6[6], c6[7]);
    }
    __syncthreads();
    for (int j = 0; j < n; j++) {
      // broadcast reads of record j (same LDS address in every lane); the other waves of the SIMD hide the latency
      const float4 r0 = rec[j * REC4 + 0], r1 = rec[j * REC4 + 1], r2 = rec[j * REC4 + 2];
      const float4 r3 = C > 4 ? rec[j * REC4 + 3] : make_float4(0.f, 0.f, 0.f, 0.f);
      const uint32_t bm = readlane(gmask, j) & alive;  // scalar
      if (bm == 0) continue;
      if constexpr (DIAG) {
        dbg_bodies += (uint32_t)__popc(bm);
        dbg_pairs += 1;
      }
      const float bx = r0.x, by = r0.y, bA = r0.z, bB = r0.w, bC = r1.x, bo = r1.y, bz = r1.z;
      const float2v bcol2[4] = {float2v{r2.x, r2.y}, float2v{r2.z, r2.w}, float2v{r3.x, r3.y}, float2v{r3.z, r3.w}};
      const uint32_t pos = (uint32_t)(base + j - rg.x + 1);
      const float dx0 = __fsub_rn(bx, px0), dy0 = __fsub_rn(by, py0);
      uint32_t dbg_m16 = 0;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (!((bm >> k) & 1u)) continue;  // wave-uniform
        const bool contributed = blend_fwd_pixel<CP, WITH_DEPTH>(T[k], D[k], acc[k], last[k], quad_offset(dx0, k & 1),
                                                                 quad_offset(dy0, k >> 1), bA, bB, bC, bo, bz, bcol2, pos);
#ifdef FSGS_DIAG_HOOKS
        if constexpr (DIAG >= 2) {
          {
            const unsigned long long bal = __ballot(contributed);
            const int cnt = __popcll(bal);
            dl.lanes += (uint32_t)cnt;
            if (lane == 0) dbg_hist[(cnt + 7) >> 3] += 1;
            dbg_m16 |= diag_block_bits(bal) << (4 * k);
          }
        }
#else
        (void)contributed;
#endif
      }
#ifdef FSGS_DIAG_HOOKS
      if constexpr (DIAG >= 2) {
        {
          uint32_t qsel = 0;  // the 4-bit groups of the quadrants this pair executed
          for (int k = 0; k < 4; k++) qsel |= ((bm >> k) & 1u) ? (0xFu << (4 * k)) : 0u;
          const uint32_t r16 = readlane(gmask16, j) & qsel;
          const uint32_t nb = (uint32_t)__popc(bm), pr = diag_packed(r16);
          dl.pairs += 1; dl.bodies += nb;
          dl.packed_true += diag_packed(dbg_m16); dl.blocks_true += (uint32_t)__popc(dbg_m16);
          dl.packed_rect += pr; dl.blocks_rect += (uint32_t)__popc(r16);
          dl.pairs_gain += pr < nb ? 1u : 0u;
        }
      }
#endif
      (void)dbg_m16; (void)gmask16;
    }
  }
#ifdef FSGS_DIAG_HOOKS
  if constexpr (DIAG >= 2) {
    __syncthreads();
    diag_flush(dbg.lanes, dl, dbg_hist, lane);
  }
#endif
  const size_t HW = (size_t)H * W;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (tp.x[k] < W && tp.y[k] < H) {
      size_t pix = (size_t)tp.y[k] * W + tp.x[k];
      const float Tf = fabsf(T[k]);
      final_T[pix] = Tf;
      n_contrib[pix] = last[k];
#pragma unroll
      for (int ch = 0; ch < C; ch++) {
        float v = fmaf(Tf, cam.bg[ch], acc[k][ch >> 1][ch & 1]);
        if (ch < 3) out_color[ch * HW + pix] = v;
        else out_color2[(ch - 3) * HW + pix] = v;
      }
      if (WITH_DEPTH) out_depth[pix] = D[k];
    }
  }
#ifdef FSGS_DIAG_HOOKS
  if constexpr (DIAG) {
    if (dbg.times && lane == 0) {
      dbg.times[kDiagStampWords * blockIdx.x + 0] = dbg_t0;
      dbg.times[kDiagStampWords * blockIdx.x + 1] = wall_clock64();
      dbg.times[kDiagStampWords * blockIdx.x + 2] = ((unsigned long long)dbg_pairs << 32) | dbg_bodies;
      dbg.times[kDiagStampWords * blockIdx.x + 3] = ((unsigned long long)(uint32_t)(rg.y - rg.x) << 32) |
                                      max(max(last[0], last[1]), max(last[2], last[3]));
      dbg.times[kDiagStampWords * blockIdx.x + 4] = dbg_c0;
      dbg.times[kDiagStampWords * blockIdx.x + 5] = __builtin_readcyclecounter();
    }
  }
#endif
  (void)dbg_t0; (void)dbg_c0; (void)dbg_bodies; (void)dbg_pairs;
}

// ------------------------------------------------------------------------------------------------
// R6 for SMALL tile grids: four waves per tile (FSGS_FLAG_BLEND_QUAD_WAVES / auto below kQuadWavesMaxTiles)
// ------------------------------------------------------------------------------------------------
// One wave per tile leaves the chip idle when the grid is small: C1 (640x512) has 1280 tiles for 1024 SIMDs with 5-6 wave
// slots each, and the kernel lasts as long as its LONGEST list walked by ONE wave doing ~2.2 quadrant bodies per pair
// (profiles/r04_bench_C1.json: 0.6 - 1.0 ns per pair against 0.14 - 0.24 at C2).  Here a tile is a 256-thread workgroup:
//   * wave q owns quadrant q with ONE pixel per lane (same pixel <-> lane map as the one-wave kernel's quadrant q, and the
//     same arithmetic through blend_fwd_pixel / quad_offset: the two flavours' outputs are bit-identical);
//   * a batch is 256 records, thread t gathers and stages record t ONCE for all four waves, its 4-bit quadrant mask beside it;
//   * a wave ballots bit q of the masks of 64 records at a time and walks only the SET bits (s_ff1): a pair that cannot reach
//     its quadrant costs it nothing -- not even the broadcast reads;
//   * a wave whose quadrant is finished keeps gathering / staging for the others; the workgroup stops when all four are.
// Per pair this is MORE issue slots than the one-wave kernel on a full chip (each wave pays its own loop and record reads),
// which is why the big grids keep the one-wave kernel: launch_blend_fwd picks by the number of tiles.
constexpr int QW_BATCH = 256;
#ifndef FSGS_FWD_BF_LEAN
#define FSGS_FWD_BF_LEAN 0
#endif
#ifndef FSGS_FWD_BRANCHFREE
#define FSGS_FWD_BRANCHFREE 1  // 0: the branchy body (A/B: profiles/r06_ab_fwd_branchfree.txt: C2 blend_fwd 115.4 -> 107.5 us, C1 step -2.7 %; a per-body
// wave-level exit for finished quadrants on top of it: 108 -> 113.5 us at C2, 163 -> 160 dense, _branchfree2.txt: not kept)
#endif
// (Two records per trip of the set-bit walk were tried twice in round 6 -- on the branchy body, 115 -> 121 us, and on the branch-free
// one, 108 -> 113 us: profiles/r06_ab_fwd_pair.txt, r06_ab_refine_order_and_fwd_pair.txt -- and are not in the source any more.)
template <int C, bool WITH_DEPTH>
__global__ __launch_bounds__(256) void blend_fwd_quad_kernel(
    CamParams cam, int ntiles, const uint32_t *__restrict__ order, const int2 *__restrict__ ranges,
    const uint32_t *__restrict__ plist, const float4 *__restrict__ grec, float *__restrict__ final_T,
    uint32_t *__restrict__ n_contrib, float *__restrict__ out_color, float *__restrict__ out_color2,
    float *__restrict__ out_depth) {
  constexpr int REC4 = C > 4 ? 4 : 3;
  constexpr int CP = (C + 1) / 2;
  __shared__ float4 rec[QW_BATCH * REC4];
  __shared__ uint32_t reach[QW_BATCH];  // 4-bit quadrant masks of the staged records (0 beyond the batch's end)
  __shared__ uint32_t wave_alive[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave's quadrant (scalar)
  const uint32_t tile_u = order ? order[blockIdx.x] : (uint32_t)xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile_u >= (uint32_t)ntiles) return;
  const int tile = (int)tile_u;
  const int W = cam.W, H = cam.H;
  const int x0 = (tile % cam.gx) * FSGS_TILE + (lane & 7), y0 = (tile / cam.gx) * FSGS_TILE + (lane >> 3);
  const int x = x0 + FSGS_QUAD * (q & 1), y = y0 + FSGS_QUAD * (q >> 1);
  const float px0 = (float)x0, py0 = (float)y0;  // the lane's pixel in quadrant 0, as in the one-wave kernel
  // quad_offset() with the wave's quadrant in scalar registers: d - 8 or d - 0 (exact, -0 included) -- one v_sub with a scalar
  // operand per coordinate instead of add + select
  const float offx = (q & 1) ? (float)FSGS_QUAD : 0.0f, offy = (q >> 1) ? (float)FSGS_QUAD : 0.0f;
  const bool inside = x < W && y < H;
  float T = inside ? 1.0f : -1.0f, D = 0.0f;
  float2v acc[CP];
  uint32_t last = 0;
#pragma unroll
  for (int cp = 0; cp < CP; cp++) acc[cp] = float2v{0.0f, 0.0f};
  const int2 rg = ranges[tile];
  for (int base = rg.x; base < rg.y; base += QW_BATCH) {
    const bool my_alive = __ballot(T > 0.f) != 0ull;
    if (lane == 0) wave_alive[q] = my_alive ? 1u : 0u;
    const int n = min(QW_BATCH, rg.y - base);
    const uint32_t g = plist[base + (tid < n ? tid : 0)];
    const float4 *rp = grec + (size_t)g * kRecF4;
    const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
    const float4 q3 = C > 4 ? rp[3] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float2 gxy = make_float2(q0.x, q0.y);
    const float4 gco = make_float4(q0.z, q0.w, q1.x, q1.y);
    const uint32_t gmask = quadrant_mask(tile, cam.gx, gxy, gco);
    __syncthreads();  // the previous batch is consumed; the four alive flags are in place
    if ((wave_alive[0] | wave_alive[1] | wave_alive[2] | wave_alive[3]) == 0u) break;  // uniform over the workgroup
    const SplatCoef kf = splat_coef(gco.x, gco.y, gco.z);
    rec[tid * REC4 + 0] = make_float4(gxy.x, gxy.y, kf.a, kf.b);
    rec[tid * REC4 + 1] = make_float4(kf.c, gco.w, WITH_DEPTH ? q1.z : 0.f, 0.f);
    {
      const float g8[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      float c6[8];
#pragma unroll
      for (int ch = 0; ch < 8; ch++) c6[ch] = ch < C ? g8[ch] : 0.f;
      rec[tid * REC4 + 2] = make_float4(c6[0], c6[1], c6[2], c6[3]);
      if (C > 4) rec[tid * REC4 + 3] = make_float4(c6[4], c6[5], c6[6], c6[7]);
    }
    reach[tid] = tid < n ? gmask : 0u;
    __syncthreads();
    if (my_alive) {
      const int nsub = (n + 63) >> 6;
      for (int sub = 0; sub < nsub; sub++) {
        unsigned long long bits = __ballot(((reach[sub * 64 + lane] >> q) & 1u) != 0u);
        while (bits) {
          const int j = sub * 64 + (int)__builtin_ctzll(bits);  // scalar
          bits &= bits - 1ull;
          const float4 r0 = rec[j * REC4 + 0], r1 = rec[j * REC4 + 1], r2 = rec[j * REC4 + 2];
          const float4 r3 = C > 4 ? rec[j * REC4 + 3] : make_float4(0.f, 0.f, 0.f, 0.f);
          const float2v bcol2[4] = {float2v{r2.x, r2.y}, float2v{r2.z, r2.w}, float2v{r3.x, r3.y}, float2v{r3.z, r3.w}};
          const float dx0 = __fsub_rn(r0.x, px0), dy0 = __fsub_rn(r0.y, py0);
#if FSGS_FWD_BRANCHFREE
          blend_fwd_pixel_branchfree<CP, WITH_DEPTH>(T, D, acc, last, __fsub_rn(dx0, offx), __fsub_rn(dy0, offy), r0.z, r0.w,
                                                     r1.x, r1.y, r1.z, bcol2, (uint32_t)(base + j - rg.x + 1));
#else
          blend_fwd_pixel<CP, WITH_DEPTH>(T, D, acc, last, __fsub_rn(dx0, offx), __fsub_rn(dy0, offy), r0.z, r0.w,
                                          r1.x, r1.y, r1.z, bcol2, (uint32_t)(base + j - rg.x + 1));
#endif
        }
        if (__ballot(T > 0.f) == 0ull) break;  // the quadrant finished inside this batch
      }
    }
  }
  if (inside) {
    const size_t HW = (size_t)H * W, pix = (size_t)y * W + x;
    const float Tf = fabsf(T);
    final_T[pix] = Tf;
    n_contrib[pix] = last;
#pragma unroll
    for (int ch = 0; ch < C; ch++) {
      const float v = fmaf(Tf, cam.bg[ch], acc[ch >> 1][ch & 1]);
      if (ch < 3) out_color[ch * HW + pix] = v;
      else out_color2[(ch - 3) * HW + pix] = v;
    }
    if (WITH_DEPTH) out_depth[pix] = D;
  }
}

// ------------------------------------------------------------------------------------------------
// R7  backward blend
// ------------------------------------------------------------------------------------------------
// grad_acc layout per Gaussian (floats), with w = o G dL/dalpha summed over every pixel the Gaussian blended into and
// d = mean2D - pixel:  [0,1] sum w d_x, w d_y | [2,3,4] sum w d_x^2, w d_x d_y, w d_y^2 | [5] sum w (= dL/dopacity)
// | [6,7] as [0,1] for the RGB channels only (SPLIT).  8 floats stride; dcolors go straight to the output tensor.
constexpr int kAccStride = 8;
// The fused render keeps moments AND colour sums of a Gaussian in ONE 64-byte row, [0..7] moments | [8..13] colours:
// the up to 12 atomics a (tile, Gaussian) pair issues land in one cache line instead of two or three.
constexpr int kFusedRow = 16;
// FSGS_FLAG_DETERMINISTIC: one row of this many floats per (tile, Gaussian) pair, [0..7] moments | [8..8+C) colour sums
constexpr int kDetRow = 16;

// Moments -> the gradients of SURVEY.md A.4.  The moments arrive multiplied by the opacity (w o = a dL/dalpha, see
// SplatEval): dL/dmean2D (pixel units) = -(A m0 + B m1, C m1 + B m0), dL/dconic (A,B,C) = -(m2/2, m3, m4/2),
// dL/dopacity = m5 / o; [6,7] the RGB-only dL/dmean2D.
__device__ __forceinline__ void unpack_moments(const float *m, float4 co, float ga[8]) {
  const float A = co.x, B = co.y, Cc = co.z;
  ga[0] = -fmaf(A, m[0], B * m[1]);
  ga[1] = -fmaf(Cc, m[1], B * m[0]);
  ga[2] = -0.5f * m[2];
  ga[3] = -m[3];
  ga[4] = -0.5f * m[4];
  ga[5] = m[5] != 0.f ? m[5] / co.w : 0.f;  // a Gaussian that blended anywhere has o > 1/255
  ga[6] = -fmaf(A, m[6], B * m[7]);
  ga[7] = -fmaf(Cc, m[7], B * m[6]);
}

// s_setprio takes an immediate: a wave-uniform switch.  Level = remaining / step, once per 64-record batch.  (Round 6 also tried
// setting it every 16 records and thresholds step/4, step/2, step: nothing over this, profiles/r06_ab_bwd_prio3.txt.)
__device__ __forceinline__ void set_issue_priority(int remaining, int step) {
  const int level = remaining / step;
  if (level >= 3) __builtin_amdgcn_s_setprio(3);
  else if (level == 2) __builtin_amdgcn_s_setprio(2);
  else if (level == 1) __builtin_amdgcn_s_setprio(1);
  else __builtin_amdgcn_s_setprio(0);
}

// SPLIT (fused 6-channel pass): channels 0..2 are the RGB pass, 3..5 the depth/silhouette pass of the
// reference's two calls; the RGB pass's own dL/dmean2D goes to accumulator slots 6,7 because
// `viewspace_points` must not see the depth loss (gaussian_renderer/__init__.py:77,90; SURVEY a1 note i).
// POSE_ONLY (the tracking step: gs_grad = False, no parameter gradients, no dL/d(depth, silhouette)): only the
// camera-pose gradient is wanted, which needs the five moments and nothing else -- the colours do not depend on
// the pose, the opacity gradient and the densification statistic are not used (SURVEY a1 note v).  8 slots per
// Gaussian, so the transposing reduction of a Gaussian pair shrinks to 16 values, and only the RGB channels
// carry a gradient.
// CGRAD < C: only the first CGRAD channels carry dL/dpixel (the fused render's losses never touch the silhouette
// and depth^2 planes: CGRAD = 4 drops their two FMAs in the colour dot product and their two dcolour sums).
// ROW: floats per accumulator row when moments and colour sums share one row per Gaussian (kFusedRow, the fused
// render); 0 = the operator boundary's layout (moments [P,8] in scratch, colour sums straight into dcolors [P,C]).
// The ONE per-(pixel, Gaussian) backward blend step, shared by every flavour of the backward blend: replays the pixel's
// transmittance backwards and adds the lane's terms of the Gaussian's SL gradient slots into s[].  Branch-free: a lane that
// does not contribute runs it with alpha = a = 0 (splat_alpha_masked).  Returns whether the lane contributed.
//   T: transmittance in front of the Gaussians replayed so far; gB (gBr): the running colour-behind sums (all / RGB channels)
template <int CG, bool SPLIT, bool POSE_ONLY>
__device__ __forceinline__ bool blend_bwd_pixel(float *s, float &T, float &gB, float &gBr, const float (&g)[CG], float dx,
                                                float dy, float bA, float bB, float bC, float bo, const float (&bcol)[6],
                                                bool live) {
  SplatEval e;  // alpha = G = 0 for lanes that do not contribute: the arithmetic below is a no-op for them
  const bool contributed = splat_alpha_masked(dx, dy, bA, bB, bC, bo, live, e);
  const float inv1ma = __builtin_amdgcn_rcpf(1.0f - e.alpha);
  T = T * inv1ma;
  const float wgt = e.alpha * T;
  float gc = 0.0f, gc_rgb = 0.0f;
#pragma unroll
  for (int ch = 0; ch < CG; ch++) {
    gc = fmaf(g[ch], bcol[ch], gc);
    if (SPLIT && ch == 2) gc_rgb = gc;
    if (!POSE_ONLY) s[8 + ch] = fmaf(wgt, g[ch], s[8 + ch]);
  }
  const float dL_dalpha = fmaf(T, gc, -inv1ma * gB);
  gB = fmaf(wgt, gc, gB);
  // moments of w = o G dL/dalpha over the pixels; the conic / opacity factors are per-Gaussian constants
  // and are applied once, after the tile and atomic sums, by unpack_moments()
  const float w = e.a * dL_dalpha;
  const float wdx = w * e.dx, wdy = w * e.dy;
  if (!POSE_ONLY) s[5] += w;
  s[0] += wdx;
  s[1] += wdy;
  s[2] = fmaf(wdx, e.dx, s[2]);
  s[3] = fmaf(wdx, e.dy, s[3]);
  s[4] = fmaf(wdy, e.dy, s[4]);
  if (SPLIT) {
    const float wr = e.a * fmaf(T, gc_rgb, -inv1ma * gBr);
    gBr = fmaf(wgt, gc_rgb, gBr);
    s[6] = fmaf(wr, e.dx, s[6]);
    s[7] = fmaf(wr, e.dy, s[7]);
  }
  return contributed;
}

// Slot bookkeeping of the transposing reductions, shared by the flavours of the backward blend.
//   gradient component slots of one Gaussian (SL per Gaussian, GP = 2 Gaussians per transposing reduction):
//   0..4 moments of w (mean2D x,y | conic A,B,C) | 5 opacity | 6,7 RGB-only mean2D (SPLIT) | 8..8+C colours
//   POSE_ONLY: the five moments only
// after the reduction REP neighbouring lanes own (Gaussian u = (l / REP) / SL, component c = (l / REP) % SL)
// (the 12-of-16 reduction folds the cheap lane bits first and leaves its totals in another lane order: transpose12_slot)
template <bool SPLIT, bool POSE_ONLY, int CG>
struct BwdSlots {
  static constexpr int GP = 2;                   // Gaussians per transposing reduction
  static constexpr int SL = POSE_ONLY ? 8 : 16;  // slots per Gaussian
  static constexpr int NV = GP * SL;             // values per lane entering the reduction
  static constexpr int REP = 64 / NV;            // lanes that end up with the same total
  static constexpr bool CHEAP_FIRST = !POSE_ONLY && CG <= 4;
  // The cheap-first reductions leave the DPP banks of their UNUSED slots (12..15 of 16, 5..7 of 8) unwritten: those lanes
  // hold undefined register contents, possibly NaN.  That is safe only while `used` below masks exactly those slots out
  // of the atomics -- tie the slot map to the variants, so a change of SL / CG / GP cannot let garbage through
  // (fsgs_selftest_transpose_reduce_n widths 3212 / 1605 check the used slots, tests/test_raster_gpu.py).
  static_assert(GP == 2, "both cheap-first reductions transpose two Gaussians at a time");
  static_assert(!CHEAP_FIRST || (SL == 16 && 8 + CG <= 12), "12-of-16 reduction: slots 12..15 must be unused");
  static_assert(!POSE_ONLY || SL == 8, "5-of-8 reduction: slots 5..7 must be unused");
  int u, c;   // the (Gaussian, component) this lane ends up owning
  bool used;  // this lane issues the atomic of its slot (one lane of those that hold the same total, used slots only)
  __device__ __forceinline__ explicit BwdSlots(int lane) {
    const int slot = POSE_ONLY ? transpose5_slot(lane) : CHEAP_FIRST ? transpose12_slot(lane) : lane / REP;
    u = slot / SL;
    c = slot % SL;
    const bool owner = POSE_ONLY ? !(lane & 0x21) : !(lane & (REP - 1));
    used = owner && (POSE_ONLY ? c < 5 : (c < 6 || (SPLIT && c < 8) || (c >= 8 && c < 8 + CG)));
  }
  static __device__ __forceinline__ float reduce(const float (&v)[NV], int lane) {
    // 64 x NV transposing reduction: lanes (u, c) receive the wave's total of component c of Gaussian u
    if constexpr (POSE_ONLY) return wave_transpose_reduce16_5of8_cheap_first(v, lane);  // slots 5..7 stay undefined
    else if constexpr (CG <= 4) return wave_transpose_reduce32_12of16_cheap_first(v, lane);  // slots 12..15 stay undefined
    else return wave_transpose_reduce32(v, lane);
  }
};

// DET (FSGS_FLAG_DETERMINISTIC, tests): no atomics -- the wave's total of slot c of the pair at list position p goes to
// pair_rows[p][c] (a plain store into the pair's own 16-float row, passed in `grad_acc`); det_gather_kernel then sums a
// Gaussian's rows in a fixed order.
template <int C, bool SPLIT, bool POSE_ONLY = false, int CGRAD = C, int ROW = 0, int DIAG = 0, bool DET = false>
#ifndef FSGS_BWD_WAVES
#define FSGS_BWD_WAVES 5  // waves per SIMD the mapping backward is compiled for (A/B: free-surgs_amd/build.py FSGS_CFLAGS)
#endif
__global__ __launch_bounds__(64, (!POSE_ONLY && CGRAD > 4) ? 3 : (!POSE_ONLY ? FSGS_BWD_WAVES : 4)) void blend_bwd_kernel(
    CamParams cam, int ntiles, const uint32_t *__restrict__ order, const int2 *__restrict__ ranges,
    const uint32_t *__restrict__ plist, const float4 *__restrict__ grec, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dcolor2,
    float *__restrict__ grad_acc, float *__restrict__ dcolors, float *__restrict__ clear16, DiagPtrs<DIAG> dbg) {
  unsigned long long dbg_t0 = 0ull, dbg_c0 = 0ull;  // see blend_fwd_kernel
  uint32_t dbg_bodies = 0, dbg_pairs = 0;
#ifdef FSGS_DIAG_HOOKS
  DiagLanes dl{};
  __shared__ uint32_t dbg_hist[9];
  if constexpr (DIAG) {
    if (dbg.times) {
      dbg_t0 = wall_clock64();
      dbg_c0 = __builtin_readcyclecounter();
    }
    if (DIAG >= 2 && threadIdx.x < 9) dbg_hist[threadIdx.x] = 0;
  }
#else
  static_assert(!DIAG, "the diagnostics flavour needs FSGS_DIAG_HOOKS");
#endif
  constexpr uint32_t acc_stride = ROW ? ROW : kAccStride, col_stride = ROW ? ROW : C;  // compile-time: shifts, no 64-bit mads
  static_assert(!(SPLIT && POSE_ONLY), "the densification statistic is a mapping-only output");
  // 16 floats the NEXT kernel accumulates into with atomics (dL/dw2c): cleared here instead of by a separate fill
  if (clear16 && blockIdx.x == 0 && threadIdx.x < 16) clear16[threadIdx.x] = 0.f;
  constexpr int REC4 = C > 4 ? 4 : 3;
  constexpr int CG = POSE_ONLY ? (C < 3 ? C : 3) : CGRAD;  // channels that carry dL/dpixel
  using Slots = BwdSlots<SPLIT, POSE_ONLY, CG>;
  constexpr int GP = Slots::GP, SL = Slots::SL, NV = Slots::NV;
  __shared__ float4 rec[64 * REC4];
  const int lane = threadIdx.x;
  const uint32_t tile_u = order ? order[blockIdx.x] : (uint32_t)xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile_u >= (uint32_t)ntiles) return;  // also the 0xFFFFFFFF holes of the banded order
  const int tile = (int)tile_u;
  const int W = cam.W, H = cam.H;
  const size_t HW = (size_t)H * W;
  const TilePix tp = tile_pixels(tile, cam.gx, lane);
  // Colour terms of dL/dalpha, per pixel and Gaussian i (T_i = transmittance in front of i):
  //   sum_ch g_ch (c_ch T_i - B_ch / (1 - alpha_i)),   B_ch = sum_{k behind i} c_k alpha_k T_k
  // which is UPSTREAM's T (c - accum_rec) written with the absolute colour behind.  Only the scalar
  // gB = sum_ch g_ch B_ch is needed, so one register per pixel replaces accum_rec / last_color / last_alpha:
  //   gc = sum_ch g_ch c_ch;  dL/dalpha = T gc - (gB + T_final bg.g) / (1 - alpha);  gB += alpha T gc.
  // gB[k] starts at T_final * (bg . dL/dpixel), so the sum in the bracket is one register; gBr: the same restricted
  // to the RGB channels (SPLIT).
  const float px0 = (float)tp.x[0], py0 = (float)tp.y[0];  // the lane's pixel in quadrant 0 (the others: + 8, quad_offset)
  float T[4], gB[4], gBr[4], g[4][CG];
  int last[4], qlast[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    bool inside = tp.x[k] < W && tp.y[k] < H;
    size_t pix = inside ? (size_t)tp.y[k] * W + tp.x[k] : 0;
    const float Tfin = inside ? final_T[pix] : 0.0f;
    T[k] = Tfin;
    last[k] = inside ? (int)n_contrib[pix] : 0;
    qlast[k] = wave_max(last[k]);  // deepest contributor of quadrant k (scalar)
    float bgdot = 0.0f, bgdot_rgb = 0.0f;
#pragma unroll
    for (int ch = 0; ch < CG; ch++) {
      // channels >= 3 of the fused pass come from their own tensor; a missing tensor is a zero gradient
      const float *gp = ch < 3 ? dL_dcolor : dL_dcolor2;
      g[k][ch] = (inside && gp) ? gp[(ch < 3 ? ch : ch - 3) * HW + pix] : 0.0f;
      bgdot = fmaf(cam.bg[ch], g[k][ch], bgdot);
      if (SPLIT && ch < 3) bgdot_rgb = fmaf(cam.bg[ch], g[k][ch], bgdot_rgb);
    }
    gB[k] = Tfin * bgdot;  // the background term rides in the running sum from the start
    gBr[k] = Tfin * bgdot_rgb;
  }
  const int2 rg = ranges[tile];
  int hi = max(max(qlast[0], qlast[1]), max(qlast[2], qlast[3]));  // nothing deeper matters to anyone in the tile
  const int dbg_walked = hi;
  const Slots slots(lane);
  const int my_u = slots.u, my_c = slots.c;
  const bool c_used = slots.used;
  while (hi > 0) {
    // Issue priority by REMAINING work (round 6).  A SIMD arbitrates between its resident waves by priority, then age.  With the
    // longest-first dispatch order the oldest wave of a SIMD is its longest tile, and age alone keeps it ahead of the others to its
    // very end: it finishes first, the youngest (shortest) tiles last, alone on their SIMD, where a lone wave issues at less than
    // half the rate of five (profiles/r06_tile_times_bwd.txt: lists of 300+ entries done after 180 us, lists of 100..200 after
    // 225-240 us; residency 5 waves for 60 % of the launch, then draining).  With four levels -- a wave drops one each time
    // `bwd_prio_step` fewer entries remain -- the wave with the most left to walk issues first, the waves of a SIMD converge on a
    // common finish and the SIMD keeps its waves to the end: blend_bwd 256 -> 233 us at C2, 357 -> 314 us on the dense scene
    // (profiles/r06_ab_bwd_prio*.txt).  Results do not depend on it (only WHEN a wave issues).
    if (cam.bwd_prio_step > 0) set_issue_priority(hi, cam.bwd_prio_step);  // once per 64 records
    const int lo = max(0, hi - 64);
    const int n = hi - lo;
    uint32_t gid = plist[rg.x + lo + (lane < n ? lane : 0)];
    const float4 *rp = grec + (size_t)gid * kRecF4;  // one 64-byte line
    const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
    const float4 q3 = C > 4 ? rp[3] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float2 gxy = make_float2(q0.x, q0.y);
    const float4 gco = make_float4(q0.z, q0.w, q1.x, q1.y);
    const float g8[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
    float gcol[C];
#pragma unroll
    for (int ch = 0; ch < C; ch++) gcol[ch] = g8[ch];
    const uint32_t gmask = quadrant_mask(tile, cam.gx, gxy, gco);
    uint32_t gmask16 = 0;
#ifdef FSGS_DIAG_HOOKS
    if constexpr (DIAG >= 2) gmask16 = diag_block_mask16(tile, cam.gx, gxy, gco);
#endif
    __syncthreads();  // records of the previous batch fully consumed
    const SplatCoef kf = splat_coef(gco.x, gco.y, gco.z);
    rec[lane * REC4 + 0] = make_float4(gxy.x, gxy.y, kf.a, kf.b);
    {
      float c6[8];
#pragma unroll
      for (int ch = 0; ch < 8; ch++) c6[ch] = ch < C ? gcol[ch < C ? ch : 0] : 0.f;
      rec[lane * REC4 + 1] = make_float4(kf.c, gco.w, c6[0], c6[1]);
      rec[lane * REC4 + 2] = make_float4(c6[2], c6[3], c6[4], c6[5]);
    }
    __syncthreads();
    for (int jj = n - 1; jj >= 0; jj -= GP) {
      float v[NV];
#pragma unroll
      for (int i = 0; i < NV; i++) v[i] = 0.f;
      bool any_group = false;
#pragma unroll
      for (int u = 0; u < GP; u++) {
        const int j = jj - u;
        if (j < 0) continue;       // wave-uniform
        const int pos = lo + j;    // 0-based index in the tile list
        uint32_t bm = readlane(gmask, j);  // scalar: quadrants this Gaussian can reach ...
#pragma unroll
        for (int k = 0; k < 4; k++)
          if (pos >= qlast[k]) bm &= ~(1u << k);  // ... and in which somebody blended it or something behind it
        if (bm == 0) continue;
        if constexpr (DIAG) {
          dbg_bodies += (uint32_t)__popc(bm);
          dbg_pairs += 1;
        }
        // broadcast reads of record j (same LDS address in every lane)
        const float4 r0 = rec[j * REC4 + 0], r1 = rec[j * REC4 + 1], r2 = rec[j * REC4 + 2];
        const float bx = r0.x, by = r0.y, bA = r0.z, bB = r0.w, bC = r1.x, bo = r1.y;
        const float bcol[6] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
        float *s = &v[SL * u];
        bool any = false;
        const float dx0 = __fsub_rn(bx, px0), dy0 = __fsub_rn(by, py0);
        uint32_t dbg_m16 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (!((bm >> k) & 1u)) continue;  // wave-uniform
          const bool contributed = blend_bwd_pixel<CG, SPLIT, POSE_ONLY>(s, T[k], gB[k], gBr[k], g[k], quad_offset(dx0, k & 1),
                                                                         quad_offset(dy0, k >> 1), bA, bB, bC, bo, bcol,
                                                                         pos < last[k]);
          any |= contributed;
#ifdef FSGS_DIAG_HOOKS
          if constexpr (DIAG >= 2) {
            {
              const unsigned long long bal = __ballot(contributed);
              const int cnt = __popcll(bal);
              dl.lanes += (uint32_t)cnt;
              if (lane == 0) dbg_hist[(cnt + 7) >> 3] += 1;
              dbg_m16 |= diag_block_bits(bal) << (4 * k);
            }
          }
#endif
        }
        any_group = any_group || (__ballot(any) != 0ull);
#ifdef FSGS_DIAG_HOOKS
        if constexpr (DIAG >= 2) {
          {
            uint32_t qsel = 0;
            for (int k = 0; k < 4; k++) qsel |= ((bm >> k) & 1u) ? (0xFu << (4 * k)) : 0u;
            const uint32_t r16 = readlane(gmask16, j) & qsel;
            const uint32_t nb = (uint32_t)__popc(bm), pr = diag_packed(r16);
            dl.pairs += 1; dl.bodies += nb;
            dl.packed_true += diag_packed(dbg_m16); dl.blocks_true += (uint32_t)__popc(dbg_m16);
            dl.packed_rect += pr; dl.blocks_rect += (uint32_t)__popc(r16);
            dl.pairs_gain += pr < nb ? 1u : 0u;
#pragma unroll
            for (int i = 0; i < 16; i++) dl.cnt[i] += (r16 >> i) & 1u;
          }
        }
#endif
        (void)dbg_m16;
      }
      if (!any_group) continue;  // wave-uniform: none of the GP Gaussians touched any pixel of the tile
      const float tot = Slots::reduce(v, lane);
      const int j_mine = jj - my_u;
      uint32_t gsel = readlane(gid, jj);
#pragma unroll
      for (int u = 1; u < GP; u++) {
        const uint32_t gu = readlane(gid, max(jj - u, 0));
        gsel = my_u == u ? gu : gsel;
      }
      if constexpr (DET) {
        if (j_mine >= 0 && c_used && tot != 0.f) grad_acc[((size_t)rg.x + (size_t)(lo + j_mine)) * kDetRow + (uint32_t)my_c] = tot;
      } else if (j_mine >= 0 && c_used && tot != 0.f) {
        if constexpr (ROW != 0) {
          // one row per Gaussian holds moments AND colour sums (dcolors = grad_acc + 8, stride ROW: launch_blend_bwd checks
          // it), so slot c of either kind is float c of the row: uniform base + a 32-bit offset, no 64-bit address pair
          // to compute and select per lane
          atomicAdd(grad_acc + (gsel * (uint32_t)ROW + (uint32_t)my_c), tot);
        } else {
          float *dst = (POSE_ONLY || my_c < 8) ? grad_acc + (size_t)gsel * acc_stride + my_c
                                               : dcolors + (size_t)gsel * col_stride + (my_c - 8);
          atomicAdd(dst, tot);
        }
      }
    }
    hi = lo;
#ifdef FSGS_DIAG_HOOKS
    if constexpr (DIAG >= 2) {
      if (((dbg_walked - hi) & 255) == 0 || hi == 0) diag_rows_flush(dl);  // a 256-record batch of a four-waves workgroup ends
    }
#endif
  }
#ifdef FSGS_DIAG_HOOKS
  if constexpr (DIAG >= 2) {
    __syncthreads();
    diag_flush(dbg.lanes, dl, dbg_hist, lane);
  }
  if constexpr (DIAG) {
    if (dbg.times && lane == 0) {
      dbg.times[kDiagStampWords * blockIdx.x + 0] = dbg_t0;
      dbg.times[kDiagStampWords * blockIdx.x + 1] = wall_clock64();
      dbg.times[kDiagStampWords * blockIdx.x + 2] = ((unsigned long long)dbg_pairs << 32) | dbg_bodies;
      dbg.times[kDiagStampWords * blockIdx.x + 3] = ((unsigned long long)(uint32_t)(rg.y - rg.x) << 32) | (uint32_t)dbg_walked;
      dbg.times[kDiagStampWords * blockIdx.x + 4] = dbg_c0;
      dbg.times[kDiagStampWords * blockIdx.x + 5] = __builtin_readcyclecounter();
    }
  }
#endif
  (void)dbg_t0; (void)dbg_c0; (void)dbg_bodies; (void)dbg_pairs; (void)dbg_walked;
}

// ------------------------------------------------------------------------------------------------
// R7 for SMALL tile grids: four waves per tile (see blend_fwd_quad_kernel)
// ------------------------------------------------------------------------------------------------
// Wave q replays quadrant q (one pixel per lane) back to front over the records that reach it AND that somebody in it
// blended or blended behind (bit q of the staged mask, pos < the quadrant's deepest contributor): two such records per
// transposing reduction, exactly as the one-wave kernel pairs them, but every wave reduces and issues the atomics of ITS
// quadrant's share of a Gaussian's sums (the one-wave kernel sums the four quadrants in registers first: ~2.2x fewer
// reductions per pair -- the price of the parallelism, paid only on grids that cannot fill the chip otherwise).
template <int C, bool SPLIT, bool POSE_ONLY = false, int CGRAD = C, int ROW = 0>
__global__ __launch_bounds__(256) void blend_bwd_quad_kernel(
    CamParams cam, int ntiles, const uint32_t *__restrict__ order, const int2 *__restrict__ ranges,
    const uint32_t *__restrict__ plist, const float4 *__restrict__ grec, const float *__restrict__ final_T,
    const uint32_t *__restrict__ n_contrib, const float *__restrict__ dL_dcolor, const float *__restrict__ dL_dcolor2,
    float *__restrict__ grad_acc, float *__restrict__ dcolors, float *__restrict__ clear16) {
  constexpr uint32_t acc_stride = ROW ? ROW : kAccStride, col_stride = ROW ? ROW : C;
  static_assert(!(SPLIT && POSE_ONLY), "the densification statistic is a mapping-only output");
  if (clear16 && blockIdx.x == 0 && threadIdx.x < 16) clear16[threadIdx.x] = 0.f;
  constexpr int CG = POSE_ONLY ? (C < 3 ? C : 3) : CGRAD;
  using Slots = BwdSlots<SPLIT, POSE_ONLY, CG>;
  constexpr int SL = Slots::SL, NV = Slots::NV;
  __shared__ float4 rec[QW_BATCH * 3];
  __shared__ uint32_t reach[QW_BATCH];
  __shared__ uint32_t gids[QW_BATCH];
  __shared__ int wave_last[4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t tile_u = order ? order[blockIdx.x] : (uint32_t)xcd_swizzle(blockIdx.x, gridDim.x);
  if (tile_u >= (uint32_t)ntiles) return;
  const int tile = (int)tile_u;
  const int W = cam.W, H = cam.H;
  const size_t HW = (size_t)H * W;
  const int x0 = (tile % cam.gx) * FSGS_TILE + (lane & 7), y0 = (tile / cam.gx) * FSGS_TILE + (lane >> 3);
  const int x = x0 + FSGS_QUAD * (q & 1), y = y0 + FSGS_QUAD * (q >> 1);
  const float px0 = (float)x0, py0 = (float)y0;
  const float offx = (q & 1) ? (float)FSGS_QUAD : 0.0f, offy = (q >> 1) ? (float)FSGS_QUAD : 0.0f;  // see blend_fwd_quad_kernel
  const bool inside = x < W && y < H;
  const size_t pix = inside ? (size_t)y * W + x : 0;
  float T = inside ? final_T[pix] : 0.0f;
  const int last = inside ? (int)n_contrib[pix] : 0;
  const int qlast = wave_max(last);  // deepest contributor of this wave's quadrant (scalar)
  float g[CG], gB, gBr;
  {
    float bgdot = 0.0f, bgdot_rgb = 0.0f;
#pragma unroll
    for (int ch = 0; ch < CG; ch++) {
      const float *gp = ch < 3 ? dL_dcolor : dL_dcolor2;
      g[ch] = (inside && gp) ? gp[(ch < 3 ? ch : ch - 3) * HW + pix] : 0.0f;
      bgdot = fmaf(cam.bg[ch], g[ch], bgdot);
      if (SPLIT && ch < 3) bgdot_rgb = fmaf(cam.bg[ch], g[ch], bgdot_rgb);
    }
    gB = T * bgdot;
    gBr = T * bgdot_rgb;
  }
  if (lane == 0) wave_last[q] = qlast;
  __syncthreads();
  int hi = max(max(wave_last[0], wave_last[1]), max(wave_last[2], wave_last[3]));  // uniform over the workgroup
  const int2 rg = ranges[tile];
  const Slots slots(lane);
  while (hi > 0) {
    const int lo = max(0, hi - QW_BATCH);
    const int n = hi - lo;
    const uint32_t gid = plist[rg.x + lo + (tid < n ? tid : 0)];
    const float4 *rp = grec + (size_t)gid * kRecF4;
    const float4 q0 = rp[0], q1 = rp[1], q2 = rp[2];
    const float4 q3 = C > 4 ? rp[3] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float2 gxy = make_float2(q0.x, q0.y);
    const float4 gco = make_float4(q0.z, q0.w, q1.x, q1.y);
    const uint32_t gmask = quadrant_mask(tile, cam.gx, gxy, gco);
    __syncthreads();  // records of the previous batch fully consumed
    const SplatCoef kf = splat_coef(gco.x, gco.y, gco.z);
    rec[tid * 3 + 0] = make_float4(gxy.x, gxy.y, kf.a, kf.b);
    {
      const float g8[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      float c6[8];
#pragma unroll
      for (int ch = 0; ch < 8; ch++) c6[ch] = ch < C ? g8[ch] : 0.f;
      rec[tid * 3 + 1] = make_float4(kf.c, gco.w, c6[0], c6[1]);
      rec[tid * 3 + 2] = make_float4(c6[2], c6[3], c6[4], c6[5]);
    }
    reach[tid] = tid < n ? gmask : 0u;
    gids[tid] = gid;
    __syncthreads();
    if (qlast > lo) {  // wave-uniform: something in this batch matters to this quadrant
      for (int sub = (n - 1) >> 6; sub >= 0; sub--) {
        const int idx = sub * 64 + lane;
        unsigned long long bits = __ballot(((reach[idx] >> q) & 1u) != 0u && lo + idx < qlast);
        while (bits) {
          // the two deepest records left: Gaussian u = 0 is replayed first (it lies behind u = 1)
          const int j0 = sub * 64 + 63 - (int)__builtin_clzll(bits);
          bits &= ~(1ull << (j0 & 63));
          int j1 = -1;
          if (bits) {
            j1 = sub * 64 + 63 - (int)__builtin_clzll(bits);
            bits &= ~(1ull << (j1 & 63));
          }
          float v[NV];
#pragma unroll
          for (int i = 0; i < NV; i++) v[i] = 0.f;
          bool any = false;
#pragma unroll
          for (int u = 0; u < 2; u++) {
            const int j = u ? j1 : j0;
            if (j < 0) continue;  // wave-uniform
            const float4 r0 = rec[j * 3 + 0], r1 = rec[j * 3 + 1], r2 = rec[j * 3 + 2];
            const float bcol[6] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
            const float dx0 = __fsub_rn(r0.x, px0), dy0 = __fsub_rn(r0.y, py0);
            any |= blend_bwd_pixel<CG, SPLIT, POSE_ONLY>(&v[SL * u], T, gB, gBr, g, __fsub_rn(dx0, offx), __fsub_rn(dy0, offy),
                                                         r0.z, r0.w, r1.x, r1.y, bcol, lo + j < last);
          }
          if (__ballot(any) == 0ull) continue;  // wave-uniform: neither Gaussian touched a pixel of the quadrant
          const float tot = Slots::reduce(v, lane);
          const int j_mine = slots.u ? j1 : j0;
          if (j_mine >= 0 && slots.used && tot != 0.f) {
            const uint32_t gsel = gids[j_mine];
            if constexpr (ROW != 0) {
              atomicAdd(grad_acc + (gsel * (uint32_t)ROW + (uint32_t)slots.c), tot);
            } else {
              float *dst = (POSE_ONLY || slots.c < 8) ? grad_acc + (size_t)gsel * acc_stride + slots.c
                                                      : dcolors + (size_t)gsel * col_stride + (slots.c - 8);
              atomicAdd(dst, tot);
            }
          }
        }
      }
    }
    hi = lo;
  }
}

// ------------------------------------------------------------------------------------------------
// R8 + R9  per-Gaussian backward chain, shared like project_gaussian
// ------------------------------------------------------------------------------------------------
struct GeomGrad {
  float m2x, m2y;  // dL/dmean2D in NDC units (what add_densification_stats norms)
  float dop;       // dL/d(activated opacity)
  float dm[3];     // dL/dmean3D
  float ds[3];     // dL/d(activated scale)
  float dq[4];     // dL/d(quaternion as handed over)
};
// ga: the unpacked accumulator row of this Gaussian (unpack_moments: pixel-unit mean2D, true conic partials,
// opacity).  Only call for radii > 0.
__device__ __forceinline__ GeomGrad geom_backward(const CamParams &cam, float mx, float my, float mz, float3 s_in,
                                                  float4 q, const float *ga) {
  GeomGrad out;
  {
    out.m2x = ga[0] * (0.5f * cam.W);
    out.m2y = ga[1] * (0.5f * cam.H);
    const float gA = ga[2], gB = ga[3], gC = ga[4];
    out.dop = ga[5];
    const float *V = cam.V, *PM = cam.PM;
    float3 t;
    t.x = V[0] * mx + V[4] * my + V[8] * mz + V[12];
    t.y = V[1] * mx + V[5] * my + V[9] * mz + V[13];
    t.z = V[2] * mx + V[6] * my + V[10] * mz + V[14];
    const float mod = cam.scale_modifier;
    float3 s = make_float3(mod * s_in.x, mod * s_in.y, mod * s_in.z);
    Mat3 R = quat_to_R(q);
    float c6[6];
    cov3d(s, R, c6);
    Ewa e;
    ewa_project(V, t, c6, cam.fx, cam.fy, cam.tanfovx, cam.tanfovy, e);
    // A.5 conic -> cov2D
    float a = e.a, b = e.b, c = e.c;
    float D = a * c - b * b;
    float D2 = 1.0f / (D * D + 0.0000001f);
    float dL_da = D2 * (-c * c * gA + b * c * gB + (D - a * c) * gC);
    float dL_dc = D2 * (-a * a * gC + a * b * gB + (D - a * c) * gA);
    float dL_db = D2 * (2.f * b * c * gA - (D + 2.f * b * b) * gB + 2.f * a * b * gC);
    float hb = 0.5f * dL_db;
    // A.6 dL/dSigma = M^T G2 M
    float G3[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++)
        G3[3 * r + cc] = e.M0[r] * (dL_da * e.M0[cc] + hb * e.M1[cc]) + e.M1[r] * (hb * e.M0[cc] + dL_dc * e.M1[cc]);
    float dM0[3], dM1[3];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      dM0[cc] = 2.f * (dL_da * e.SM0[cc] + hb * e.SM1[cc]);
      dM1[cc] = 2.f * (hb * e.SM0[cc] + dL_dc * e.SM1[cc]);
    }
    // Wv[r][c] = V[4*c + r]
    float dJ00 = dM0[0] * V[0] + dM0[1] * V[4] + dM0[2] * V[8];
    float dJ02 = dM0[0] * V[2] + dM0[1] * V[6] + dM0[2] * V[10];
    float dJ11 = dM1[0] * V[1] + dM1[1] * V[5] + dM1[2] * V[9];
    float dJ12 = dM1[0] * V[2] + dM1[1] * V[6] + dM1[2] * V[10];
    float itz = 1.0f / e.tz, itz2 = itz * itz, itz3 = itz2 * itz;
    float dtx = e.chix * -cam.fx * itz2 * dJ02;
    float dty = e.chiy * -cam.fy * itz2 * dJ12;
    float dtz = -cam.fx * itz2 * dJ00 - cam.fy * itz2 * dJ11 + (2.f * cam.fx * e.tx) * itz3 * dJ02 +
                (2.f * cam.fy * e.ty) * itz3 * dJ12;
#pragma unroll
    for (int cc = 0; cc < 3; cc++) out.dm[cc] = V[4 * cc + 0] * dtx + V[4 * cc + 1] * dty + V[4 * cc + 2] * dtz;
    // A.7 projection
    float h0 = PM[0] * mx + PM[4] * my + PM[8] * mz + PM[12];
    float h1 = PM[1] * mx + PM[5] * my + PM[9] * mz + PM[13];
    float h3 = PM[3] * mx + PM[7] * my + PM[11] * mz + PM[15];
    float mw = 1.0f / (h3 + 0.0000001f);
    float mul1 = h0 * mw * mw, mul2 = h1 * mw * mw;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      float P0k = PM[4 * k + 0], P1k = PM[4 * k + 1], P3k = PM[4 * k + 3];
      out.dm[k] += (P0k * mw - P3k * mul1) * out.m2x + (P1k * mw - P3k * mul2) * out.m2y;
    }
    // A.8 Sigma -> scale, quaternion
    float GR[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++)
        GR[3 * r + cc] = G3[3 * r] * R.m[cc] + G3[3 * r + 1] * R.m[3 + cc] + G3[3 * r + 2] * R.m[6 + cc];
    const float sv[3] = {s.x, s.y, s.z};
#pragma unroll
    for (int j = 0; j < 3; j++) {
      float rtgr = R.m[j] * GR[j] + R.m[3 + j] * GR[3 + j] + R.m[6 + j] * GR[6 + j];
      out.ds[j] = 2.f * sv[j] * rtgr * mod;
    }
    float dR[9];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) dR[3 * r + cc] = 2.f * GR[3 * r + cc] * sv[cc] * sv[cc];
    const float qr = q.x, qx = q.y, qy = q.z, qz = q.w;
    out.dq[0] = 2.f * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
    out.dq[1] = 2.f * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2.f * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] -
                   2.f * qx * dR[8]);
    out.dq[2] = 2.f * (-2.f * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] -
                   2.f * qy * dR[8]);
    out.dq[3] = 2.f * (-2.f * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2.f * qz * dR[4] + qy * dR[5] +
                   qx * dR[6] + qy * dR[7]);
  }
  return out;
}

__global__ __launch_bounds__(256) void preprocess_bwd_kernel(
    int P, CamParams cam, const float *__restrict__ means3D, const float *__restrict__ scales,
    const float *__restrict__ rots, const int32_t *__restrict__ radii, const float4 *__restrict__ conic_op,
    const float *__restrict__ grad_acc, float *__restrict__ dmeans2D, float *__restrict__ dopac, float *__restrict__ dmeans3D,
    float *__restrict__ dscales, float *__restrict__ drots) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  GeomGrad r;
  r.m2x = r.m2y = r.dop = 0.f;
  r.dm[0] = r.dm[1] = r.dm[2] = 0.f;
  r.ds[0] = r.ds[1] = r.ds[2] = 0.f;
  r.dq[0] = r.dq[1] = r.dq[2] = r.dq[3] = 0.f;
  if (radii[i] > 0) {
    float3 s = make_float3(scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]);
    float4 q = make_float4(rots[4 * i], rots[4 * i + 1], rots[4 * i + 2], rots[4 * i + 3]);
    float ga[8];
    unpack_moments(grad_acc + (size_t)i * kAccStride, conic_op[i], ga);
    r = geom_backward(cam, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], s, q, ga);
  }
  dmeans2D[3 * i] = r.m2x; dmeans2D[3 * i + 1] = r.m2y; dmeans2D[3 * i + 2] = 0.f;
  dopac[i] = r.dop;
  dmeans3D[3 * i] = r.dm[0]; dmeans3D[3 * i + 1] = r.dm[1]; dmeans3D[3 * i + 2] = r.dm[2];
  dscales[3 * i] = r.ds[0]; dscales[3 * i + 1] = r.ds[1]; dscales[3 * i + 2] = r.ds[2];
  drots[4 * i] = r.dq[0]; drots[4 * i + 1] = r.dq[1]; drots[4 * i + 2] = r.dq[2]; drots[4 * i + 3] = r.dq[3];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
CamParams make_cam(const FsgsRasterCfg *cfg) {
  CamParams c;
  c.W = cfg->image_width;
  c.H = cfg->image_height;
  c.gx = (c.W + FSGS_TILE - 1) / FSGS_TILE;
  c.gy = (c.H + FSGS_TILE - 1) / FSGS_TILE;
  c.tanfovx = cfg->tanfovx;
  c.tanfovy = cfg->tanfovy;
  c.fx = c.W / (2.0f * cfg->tanfovx);
  c.fy = c.H / (2.0f * cfg->tanfovy);
  c.scale_modifier = cfg->scale_modifier;
  c.flags = cfg->flags;
  c.bwd_prio_step = 0;
  memcpy(c.V, cfg->viewmatrix, sizeof(c.V));
  memcpy(c.PM, cfg->projmatrix, sizeof(c.PM));
  memcpy(c.bg, cfg->bg, sizeof(c.bg));
  return c;
}

int tile_bits(int ntiles) {
  int b = 1;
  while ((1 << b) < ntiles) b++;
  return b;
}

struct StateLayout {
  size_t xy, conic_op, depth, ranges, order, final_T, n_contrib, plist, colors, flags, rec, total;
};
// keep_channels > 0 (fused render): also a clamp-flag word per Gaussian (the colours live in the packed records)
StateLayout state_layout(int P, int W, int H, int64_t cap, int keep_channels = 0) {
  StateLayout L;
  Carver c;
  int ntiles = ((W + FSGS_TILE - 1) / FSGS_TILE) * ((H + FSGS_TILE - 1) / FSGS_TILE);
  L.xy = c.take(sizeof(float2) * (size_t)P);
  L.conic_op = c.take(sizeof(float4) * (size_t)P);
  L.depth = c.take(sizeof(float) * (size_t)P);
  L.ranges = c.take(sizeof(int2) * (size_t)ntiles);
  L.order = c.take(sizeof(uint32_t) * 2 * ((size_t)ntiles + 16));  // the folded order, then the plain one (order_plain)
  L.final_T = c.take(sizeof(float) * (size_t)W * H);
  L.n_contrib = c.take(sizeof(uint32_t) * (size_t)W * H);
  L.plist = c.take(sizeof(uint32_t) * (size_t)cap);
  L.rec = c.take(sizeof(float4) * kRecF4 * (size_t)(P > 0 ? P : 1));
  L.colors = L.flags = 0;
  if (keep_channels > 0) {
    L.colors = 0;
    L.flags = c.take((size_t)P * 4);
  }
  L.total = c.total();
  return L;
}

struct ScratchLayout {
  size_t tiles, rect, tile_count, total, keys, total_bytes;
};
int scratch_layout(int P, int W, int H, int64_t cap, ScratchLayout &L) {
  Carver c;
  size_t Pn = (size_t)(P > 0 ? P : 1), Rn = (size_t)(cap > 0 ? cap : 1);
  size_t ntiles = (size_t)((W + FSGS_TILE - 1) / FSGS_TILE) * ((H + FSGS_TILE - 1) / FSGS_TILE);
  L.tiles = c.take(4 * Pn);
  L.rect = c.take(8 * Pn);
  // cursors [tiles * 8] directly followed by {R, overflow report}: cleared together (clear_binning_cursors)
  L.tile_count = c.take(4 * (size_t)ntiles * BIN_SUBS + 16);
  L.total = L.tile_count + 4 * (size_t)ntiles * BIN_SUBS;
  L.keys = c.take(8 * Rn);
  L.total_bytes = c.total();
  return 0;
}

struct FwdBuffers {
  float2 *xy; float4 *co; float *depth; int2 *ranges; uint32_t *order; float *final_T; uint32_t *n_contrib; uint32_t *plist;
  float *colors; uint32_t *flags; float4 *rec;
  uint32_t *tiles; ushort4 *rect; uint32_t *tile_count; uint32_t *total; unsigned long long *keys;
};
int bind_forward_buffers(int P, int W, int H, int64_t max_pairs, int keep_channels, void *state, size_t state_bytes,
                         void *scratch, size_t scratch_bytes, FwdBuffers &B) {
  StateLayout SL = state_layout(P, W, H, max_pairs, keep_channels);
  ScratchLayout XL;
  scratch_layout(P, W, H, max_pairs, XL);
  if (state_bytes < SL.total || scratch_bytes < XL.total_bytes) return FSGS_ERR_CAPACITY;
  char *sb = (char *)state, *xb = (char *)scratch;
  B.xy = (float2 *)(sb + SL.xy); B.co = (float4 *)(sb + SL.conic_op); B.depth = (float *)(sb + SL.depth);
  B.ranges = (int2 *)(sb + SL.ranges); B.order = (uint32_t *)(sb + SL.order); B.final_T = (float *)(sb + SL.final_T);
  B.n_contrib = (uint32_t *)(sb + SL.n_contrib); B.plist = (uint32_t *)(sb + SL.plist);
  B.rec = (float4 *)(sb + SL.rec);
  B.colors = nullptr;
  B.flags = keep_channels ? (uint32_t *)(sb + SL.flags) : nullptr;
  B.tiles = (uint32_t *)(xb + XL.tiles); B.rect = (ushort4 *)(xb + XL.rect);
  B.tile_count = (uint32_t *)(xb + XL.tile_count); B.total = (uint32_t *)(xb + XL.total);
  B.keys = (unsigned long long *)(xb + XL.keys);
  return FSGS_OK;
}
// Everything between the preprocess kernel and the blend: scatter, per-tile sort + dispatch order -- two
// launches (the cursors were cleared by the preprocess kernel in front), all enqueued without knowing R.  The caller enqueues the forward blend right behind
// them and only then calls finish_binning(), which polls the mailbox: the GPU never waits for the host.  If a
// list segment overflowed, the blend ran on truncated (in-bounds) lists and its output is garbage; the call then
// reports FSGS_ERR_CAPACITY with *num_rendered = the max_pairs that would have sufficed.
struct BinningTicket {
  volatile uint32_t *slot = nullptr;
  uint32_t cap_sub = 0;
};
inline uint32_t binning_clear_words(int ntiles) { return (uint32_t)ntiles * BIN_SUBS + 4u; }
// cursors_cleared: the preprocess kernel in front of this call was given GeomOut.clear_words = binning_clear_words()
inline int enqueue_binning(const CamParams &cam, int P, FwdBuffers &B, int64_t max_pairs, BinningTicket &tk,
                           hipStream_t stream, bool cursors_cleared = false) {
  const int ntiles = cam.gx * cam.gy;
  if (ntiles > 1024 * 64) return FSGS_ERR_INVALID;
  const int64_t cap = max_pairs / ((int64_t)ntiles * BIN_SUBS);
  tk.cap_sub = (uint32_t)(cap > 0x0FFFFFFF ? 0x0FFFFFFF : cap);
  if (tk.cap_sub == 0) return FSGS_ERR_CAPACITY;
  // cursors [tiles * 8] followed by {R, overflow report}
  // (+4 words, not +2: a size that is a multiple of 16 bytes stays ONE fill kernel)
  if (!cursors_cleared || P <= 0)
    FSGS_HIP(hipMemsetAsync(B.tile_count, 0, sizeof(uint32_t) * (size_t)binning_clear_words(ntiles), stream));
  if (P > 0) {
    ProfScope ps(PROF_SORT_DEPTH, stream);  // scatter pass
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((P + 255) / 256), dim3(256), 0, stream, P, cam.gx, B.tiles, B.rect, B.xy,
                       B.co, B.depth, B.tile_count, B.keys, tk.cap_sub);
  }
  tk.slot = mailbox_acquire();
  {
    ProfScope ps(PROF_SORT_TILE, stream);
    if (fused_order_workgroup(cam.flags, ntiles)) {
      // the default: dispatch order, R and the mailbox word come from workgroup 0 of the sort launch itself
      hipLaunchKernelGGL(sort_tiles_kernel, dim3(ntiles + 1), dim3(256), 0, stream, ntiles, B.tile_count, B.keys, B.plist,
                         B.ranges, tk.cap_sub, B.total + 1, B.order, B.total, (uint32_t *)tk.slot);
    } else {
      hipLaunchKernelGGL(sort_tiles_kernel, dim3(ntiles), dim3(256), 0, stream, ntiles, B.tile_count, B.keys, B.plist,
                         B.ranges, tk.cap_sub, B.total + 1, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr);
      hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, ntiles,
                         (cam.flags & FSGS_FLAG_XCD_BANDED_ORDER) ? ORDER_XCD : 1, B.ranges, B.order, B.total,
                         B.total + 1, (uint32_t *)tk.slot);
    }
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}
inline int finish_binning(const CamParams &cam, FwdBuffers &B, int64_t max_pairs, const BinningTicket &tk,
                          int64_t *num_rendered, hipStream_t stream) {
  const int ntiles = cam.gx * cam.gy;
  uint32_t word = 0;
  if (!tk.slot || !mailbox_wait(tk.slot, stream, &word)) {
    // no pinned mailbox (or the stream failed): the classic copy + synchronize (UPSTREAM R2 does the same)
    uint32_t two[2] = {0, 0};
    FSGS_HIP(hipMemcpyAsync(two, B.total, sizeof(two), hipMemcpyDeviceToHost, stream));
    FSGS_HIP(hipStreamSynchronize(stream));
    word = two[1] ? (0x80000000u | two[1]) : two[0];
  }
  if (word & 0x80000000u) {
    const int64_t need_sub = (int64_t)(word & 0x7FFFFFFFu);
    *num_rendered = need_sub * ntiles * BIN_SUBS;  // a max_pairs that fits every segment
    return FSGS_ERR_CAPACITY;
  }
  *num_rendered = (int64_t)word;
  (void)max_pairs;
  return FSGS_OK;
}

// ------------------------------------------------------------------------------------------------
// FSGS_FLAG_DETERMINISTIC: the per-Gaussian sums of the backward blend in a FIXED order (tests only)
// ------------------------------------------------------------------------------------------------
// The product backward adds a pair's totals to its Gaussian's row with float atomics, in the order the workgroups happen
// to get there: two runs differ in the last bits.  Here blend_bwd_kernel<.., DET> has stored every pair's totals in the
// pair's own row (indexed by the pair's position in `plist`), and one thread per Gaussian walks the tiles of the Gaussian's
// rect row by row, finds its entry in each tile's list -- the lists are sorted by (depth bits, index), both stored -- and adds
// that row: same order, same bits, every run, whatever the dispatch order.  Costs a [max_pairs,16] float buffer, its
// clearing and ~8 dependent probes per pair; nothing of it is compiled into the product path's kernels.
template <int C, int ROW>
__global__ __launch_bounds__(256) void det_gather_kernel(int P, int gx, int gy, const int32_t *__restrict__ radii,
                                                         const float2 *__restrict__ xy, const float *__restrict__ depth,
                                                         const int2 *__restrict__ ranges, const uint32_t *__restrict__ plist,
                                                         const float *__restrict__ pair_rows, float *__restrict__ grad_acc,
                                                         float *__restrict__ dcolors) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int r_ = radii[i];
  if (r_ <= 0) return;
  const float2 p = xy[i];
  int minx, miny, maxx, maxy;
  tile_rect(gx, gy, p.x, p.y, r_, minx, miny, maxx, maxy);
  const unsigned long long key = ((unsigned long long)__float_as_uint(depth[i]) << 32) | (uint32_t)i;
  float acc[kDetRow];
#pragma unroll
  for (int c = 0; c < kDetRow; c++) acc[c] = 0.f;
  for (int ty = miny; ty < maxy; ty++)
    for (int tx = minx; tx < maxx; tx++) {
      const int2 rg = ranges[ty * gx + tx];
      int lo = rg.x, hi = rg.y;
      while (lo < hi) {  // first entry whose key is not below this Gaussian's
        const int mid = (lo + hi) >> 1;
        const uint32_t g = plist[mid];
        const unsigned long long k = ((unsigned long long)__float_as_uint(depth[g]) << 32) | g;
        if (k < key) lo = mid + 1; else hi = mid;
      }
      if (lo < rg.y && plist[lo] == (uint32_t)i) {
        const float4 *row = reinterpret_cast<const float4 *>(pair_rows + (size_t)lo * kDetRow);
#pragma unroll
        for (int q = 0; q < kDetRow / 4; q++) {
          const float4 v = row[q];
          acc[4 * q] += v.x; acc[4 * q + 1] += v.y; acc[4 * q + 2] += v.z; acc[4 * q + 3] += v.w;
        }
      }
    }
  if constexpr (ROW != 0) {
#pragma unroll
    for (int c = 0; c < ROW; c++) grad_acc[(size_t)i * ROW + c] = acc[c];
  } else {
#pragma unroll
    for (int c = 0; c < kAccStride; c++) grad_acc[(size_t)i * kAccStride + c] = acc[c];
#pragma unroll
    for (int c = 0; c < C; c++) dcolors[(size_t)i * C + c] = acc[8 + c];
  }
}
// what the deterministic route of launch_blend_bwd needs beyond the blend's own arguments
struct DetGather {
  float *pair_rows;  // [max_pairs * kDetRow] floats, 16-byte aligned
  int64_t max_pairs;
  int P;
  const int32_t *radii;
  const float2 *xy;
  const float *depth;
};
inline size_t det_pair_rows_bytes(int64_t max_pairs) { return (size_t)(max_pairs > 0 ? max_pairs : 1) * kDetRow * sizeof(float); }
// `scratch` of a deterministic backward: [per-Gaussian rows (P x 64 B) | pair rows | per-workgroup dL/dw2c partials]
constexpr int kDetW2cBlock = 256;  // Gaussians per workgroup of the per-Gaussian backward kernels (render.hip RB)
struct DetLayout {
  size_t pair_rows, w2c_partials, total;
};
inline DetLayout det_layout(int P, int64_t max_pairs) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  DetLayout L;
  L.pair_rows = up((size_t)(P > 0 ? P : 1) * kFusedRow * sizeof(float));
  L.w2c_partials = L.pair_rows + up(det_pair_rows_bytes(max_pairs));
  L.total = L.w2c_partials + up((size_t)((P > 0 ? P : 1) + kDetW2cBlock - 1) / kDetW2cBlock * 16 * sizeof(float));
  return L;
}

// Which flavour of the blend kernels a launch takes: one wave per tile or four waves per tile (blend_*_quad_kernel).
// FsgsRasterCfg.flags can force either for both directions; otherwise, measured on MI355X (profiles/r05_blend_flavours.txt,
// blend kernel us, one wave -> four waves, mapping step of bench.py):
//     tiles    1280 (C1)   2080      2880      3808      5120 (C2)   8160 (C4)
//     forward  81 -> 37    118 -> 50 128 -> 71 136 -> 88 162 -> 123  331 -> 286
//     backward 98 -> 59    154 -> 90 178 -> 156 214 -> 201 257 -> 292 592 -> 699
//   * forward: four waves everywhere -- a wave only ever walks the records that reach ITS quadrant and stops with it;
//   * backward: four waves while one wave per tile cannot fill the chip (5 waves x 1024 SIMDs = 5120 one-wave workgroups);
//     from there on the one-wave kernel's single transposing reduction per pair (instead of one per quadrant reached)
//     is worth more than the parallelism.
// The per-kernel event timings above place the backward's crossover between 3808 and 5120 tiles; whole-step A/Bs on one box
// without events (scripts/dev/ab_tracking.sh, threshold 4096 against 2560; profiles/r05_blend_flavours.txt) narrow it down:
//     mapping step     2880 tiles: 0.350 (four waves) vs 0.366 ms    3808 tiles: 0.476 vs 0.468   -> crossover ~3300
//     tracking step    2880 tiles: 0.279 vs 0.316                    3808 tiles: 0.358 vs 0.373; 5120 (C2): 0.485 vs 0.471 -> ~4400
// (the pose-only backward's reduction is the cheap 5-of-8 one, so a reduction per quadrant costs it less)
// The crossovers were measured on an MI355X (1024 SIMDs): 3328 tiles = 3.25 per SIMD (mapping), 4352 = 4.25 per SIMD (pose-only).
// What decides is how many one-wave workgroups a SIMD gets, so the rule is stated per SIMD and scaled by the device the call runs
// on (hipDeviceProp: multiProcessorCount x 4 SIMDs; ADVICE r5 -- a hard-coded tile count is simply wrong on any other part).
// FSGS_QUAD_BWD_MAX_TILES / _POSE_MAX_TILES (A/B builds only) still override it with an absolute count.
constexpr int kQuadBwdQuarterTilesPerSimd = 13, kQuadBwdPoseQuarterTilesPerSimd = 17;  // 3.25 and 4.25 tiles per SIMD
inline int device_simd_count() {
  static int simds[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1024;
  if (simds[dev] == 0) {
    hipDeviceProp_t prop;
    simds[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount * 4 : 1024;
  }
  return simds[dev];
}
inline int quad_bwd_max_tiles(bool pose_only) {
#ifdef FSGS_QUAD_BWD_MAX_TILES
  if (!pose_only) return FSGS_QUAD_BWD_MAX_TILES;
#endif
#ifdef FSGS_QUAD_BWD_POSE_MAX_TILES
  if (pose_only) return FSGS_QUAD_BWD_POSE_MAX_TILES;
#endif
  return device_simd_count() * (pose_only ? kQuadBwdPoseQuarterTilesPerSimd : kQuadBwdQuarterTilesPerSimd) / 4;
}
// Entries per issue-priority level of the one-wave backward blend (CamParams::bwd_prio_step): a sixth of the mean list length --
// measured optimum 32 at C2's 204 entries per tile, ~50 on the dense scene's 341 (profiles/r06_ab_bwd_prio2.txt) -- and only when
// every tile of the launch is resident from the start (5 waves per SIMD): with more tiles than wave slots (C4: 8160 on 5120) the
// launch runs in generations, late waves already start behind the early ones, and the sweep found nothing to gain (+0..1 %).
// The four-waves backward of the small grids (C1, X1, X2) gains nothing from it either (profiles/r06_ab_prio_quad.txt): not wired.
// FSGS_BWD_PRIO_STEP (A/B builds) overrides it; 0 there switches the priorities off.
inline int blend_bwd_prio_step(int ntiles, int64_t num_rendered) {
#ifdef FSGS_BWD_PRIO_STEP
  return FSGS_BWD_PRIO_STEP;
#endif
  if (ntiles <= 0 || num_rendered <= 0 || ntiles > 5 * device_simd_count()) return 0;
  const int64_t step = num_rendered / ntiles / 6;
  return (int)(step < 8 ? 8 : step > 4096 ? 4096 : step);
}
inline bool use_quad_waves(uint32_t flags, int ntiles, bool backward, bool pose_only = false) {
  if (flags & FSGS_FLAG_BLEND_ONE_WAVE) return false;
  if (flags & FSGS_FLAG_BLEND_QUAD_WAVES) return true;
  return backward ? ntiles <= quad_bwd_max_tiles(pose_only) : true;
}
inline bool use_quad_waves(const CamParams &cam, int ntiles, bool backward, bool pose_only = false) {
  return use_quad_waves((uint32_t)cam.flags, ntiles, backward, pose_only);
}

template <int DIAG>
inline DiagPtrs<DIAG> diag_ptrs(const char *times_var, const char *lanes_var) {
  DiagPtrs<DIAG> d{};
  if constexpr (DIAG != 0) {
    // load-balance experiments only: a device buffer of 4 * ntiles uint64 whose ADDRESS comes from the environment;
    // lane-utilisation counters (scripts/lane_utilisation.py): 32 u64 on the device, summed over every launch
    d.times = diag_env(times_var) ? (unsigned long long *)strtoull(diag_env(times_var), nullptr, 0) : nullptr;
    d.lanes = diag_env(lanes_var) ? (unsigned long long *)strtoull(diag_env(lanes_var), nullptr, 0) : nullptr;
  }
  return d;
}

template <int C, bool WITH_DEPTH = true>
int launch_blend_fwd(const CamParams &cam, int ntiles, const uint32_t *order, const int2 *ranges, const uint32_t *plist,
                     const float4 *rec, float *final_T, uint32_t *n_contrib,
                     float *out_color, float *out_color2, float *out_depth, hipStream_t s, hipEvent_t done = nullptr) {
  if (use_quad_waves(cam, ntiles, false)) {
    order = order ? order_plain(order, ntiles) : nullptr;
    if (done)  // the launch's own completion signals the event: no marker packet behind the kernel (fsgs_forward_done_event)
      hipExtLaunchKernelGGL((blend_fwd_quad_kernel<C, WITH_DEPTH>), dim3(ntiles), dim3(256), 0, s, nullptr, done, 0, cam, ntiles,
                            order, ranges, plist, rec, final_T, n_contrib, out_color, out_color2, out_depth);
    else
      hipLaunchKernelGGL((blend_fwd_quad_kernel<C, WITH_DEPTH>), dim3(ntiles), dim3(256), 0, s, cam, ntiles, order, ranges,
                         plist, rec, final_T, n_contrib, out_color, out_color2, out_depth);
    return 0;
  }
  static int dbg_lds = diag_env("FSGS_DBG_LDS_FWD") ? atoi(diag_env("FSGS_DBG_LDS_FWD")) : 0;  // occupancy experiments only
  constexpr int kLvl = kDiagBuild ? 1 : 0;
#ifdef FSGS_DIAG_HOOKS
  constexpr int kLvlLanes = 2;
#endif
  static const DiagPtrs<kLvl> dbg = diag_ptrs<kLvl>("FSGS_DBG_TILE_TIMES_FWD", "FSGS_DBG_LANES_FWD");
  // scheduling experiments only (scripts/dev/order_experiment.py): a dispatch order made on the host, holes (0xFFFFFFFF) allowed
  static const uint32_t *dbg_order = diag_env("FSGS_DBG_ORDER_FWD") ? (const uint32_t *)strtoull(diag_env("FSGS_DBG_ORDER_FWD"), nullptr, 0) : nullptr;
  static int dbg_order_n = diag_env("FSGS_DBG_ORDER_FWD_N") ? atoi(diag_env("FSGS_DBG_ORDER_FWD_N")) : 0;
  const int grid = (dbg_order && dbg_order_n > 0) ? dbg_order_n : ntiles;
#ifdef FSGS_DIAG_HOOKS
  {
    if (dbg.lanes) {  // the lane-utilisation counters were asked for: the level-2 instantiation
      DiagPtrs<kLvlLanes> d2;
      d2.times = dbg.times; d2.lanes = dbg.lanes;
      hipLaunchKernelGGL((blend_fwd_kernel<C, WITH_DEPTH, kLvlLanes>), dim3(grid), dim3(64), dbg_lds, s, cam, ntiles,
                         dbg_order ? dbg_order : order, ranges, plist, rec, final_T, n_contrib, out_color, out_color2, out_depth,
                         d2);
      if (done) return hipEventRecord(done, s) == hipSuccess ? 0 : FSGS_ERR_HIP;
      return 0;
    }
  }
#endif
  if (done)
    hipExtLaunchKernelGGL((blend_fwd_kernel<C, WITH_DEPTH, kLvl>), dim3(grid), dim3(64), dbg_lds, s, nullptr, done, 0, cam,
                          ntiles, dbg_order ? dbg_order : order, ranges, plist, rec, final_T, n_contrib, out_color, out_color2,
                          out_depth, dbg);
  else
    hipLaunchKernelGGL((blend_fwd_kernel<C, WITH_DEPTH, kLvl>), dim3(grid), dim3(64), dbg_lds, s, cam, ntiles,
                       dbg_order ? dbg_order : order, ranges, plist, rec, final_T, n_contrib, out_color, out_color2, out_depth,
                       dbg);
  return 0;
}
template <int C, bool SPLIT = false, bool POSE_ONLY = false, int CGRAD = C, int ROW = 0>
int launch_blend_bwd(const CamParams &cam, int ntiles, const uint32_t *order, const int2 *ranges, const uint32_t *plist,
                     const float4 *rec, const float *final_T, const uint32_t *n_contrib,
                     const float *dL, const float *dL2, float *grad_acc, float *dcolors, hipStream_t s,
                     float *clear16 = nullptr, const DetGather *det = nullptr) {
  if (ROW != 0 && dcolors != grad_acc + 8) return FSGS_ERR_INVALID;  // the one-row layout the kernels' addressing assumes
  if (det && det->pair_rows) {
    // FSGS_FLAG_DETERMINISTIC: pair rows (one-wave kernel, plain stores) + the fixed-order gather; see det_gather_kernel
    if (hipMemsetAsync(det->pair_rows, 0, det_pair_rows_bytes(det->max_pairs), s) != hipSuccess) return FSGS_ERR_HIP;
    hipLaunchKernelGGL((blend_bwd_kernel<C, SPLIT, POSE_ONLY, CGRAD, ROW, 0, true>), dim3(ntiles), dim3(64), 0, s, cam,
                       ntiles, order, ranges, plist, rec, final_T, n_contrib, dL, dL2, det->pair_rows, det->pair_rows + 8,
                       clear16, DiagPtrs<0>{});
    hipLaunchKernelGGL((det_gather_kernel<C, ROW>), dim3((det->P + 255) / 256), dim3(256), 0, s, det->P, cam.gx, cam.gy,
                       det->radii, det->xy, det->depth, ranges, plist, (const float *)det->pair_rows, grad_acc, dcolors);
    return 0;
  }
  if (use_quad_waves(cam, ntiles, true, POSE_ONLY)) {
    hipLaunchKernelGGL((blend_bwd_quad_kernel<C, SPLIT, POSE_ONLY, CGRAD, ROW>), dim3(ntiles), dim3(256), 0, s, cam, ntiles,
                       order ? order_plain(order, ntiles) : nullptr, ranges, plist, rec, final_T, n_contrib, dL, dL2, grad_acc, dcolors, clear16);
    return 0;
  }
  static int dbg_lds = diag_env("FSGS_DBG_LDS") ? atoi(diag_env("FSGS_DBG_LDS")) : 0;  // occupancy experiments only
  constexpr int kLvl = kDiagBuild ? 1 : 0;
#ifdef FSGS_DIAG_HOOKS
  constexpr int kLvlLanes = 2;
#endif
  static const DiagPtrs<kLvl> dbg = diag_ptrs<kLvl>("FSGS_DBG_TILE_TIMES", "FSGS_DBG_LANES");
  static const uint32_t *dbg_order = diag_env("FSGS_DBG_ORDER_BWD") ? (const uint32_t *)strtoull(diag_env("FSGS_DBG_ORDER_BWD"), nullptr, 0) : nullptr;
  static int dbg_order_n = diag_env("FSGS_DBG_ORDER_BWD_N") ? atoi(diag_env("FSGS_DBG_ORDER_BWD_N")) : 0;
  const int grid = (dbg_order && dbg_order_n > 0) ? dbg_order_n : ntiles;
#ifdef FSGS_DIAG_HOOKS
  {
    if (dbg.lanes) {
      DiagPtrs<kLvlLanes> d2;
      d2.times = dbg.times; d2.lanes = dbg.lanes;
      hipLaunchKernelGGL((blend_bwd_kernel<C, SPLIT, POSE_ONLY, CGRAD, ROW, kLvlLanes>), dim3(grid), dim3(64), dbg_lds, s, cam,
                         ntiles, dbg_order ? dbg_order : order, ranges, plist, rec, final_T, n_contrib, dL, dL2, grad_acc, dcolors,
                         clear16, d2);
      return 0;
    }
  }
#endif
  hipLaunchKernelGGL((blend_bwd_kernel<C, SPLIT, POSE_ONLY, CGRAD, ROW, kLvl>), dim3(grid), dim3(64), dbg_lds, s, cam,
                     ntiles, dbg_order ? dbg_order : order, ranges, plist, rec, final_T, n_contrib, dL, dL2, grad_acc, dcolors,
                     clear16, dbg);
  return 0;
}

}  // namespace
