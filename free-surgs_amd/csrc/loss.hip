// loss.hip -- fused photometric and depth-correlation losses for gfx950 (MI355X).
//
// Replaces the ~600 small PyTorch kernels per mapping iteration that
// utils/loss_utils.py:41-127 launches (5 depthwise 11x11 conv2d forward + backward for SSIM,
// 40 x (~12 reductions + 4 host syncs) for local_pearson_loss):
//   * photometric_forward : ONE pass over the images builds the five 11x11 Gaussian moments of
//     every pixel (separable, LDS-tiled 32x32 + 5 halo), the SSIM map, the L1 term, their
//     reductions AND the three d(ssim)/d(moment) maps the backward needs.
//   * photometric_backward: one more separable pass turns those maps into dL/dimg.
//   * pearson_stats / pearson_backward: the global Pearson loss and all random 128x128 patches
//     share one reduction launch (region 0 = whole image) and one elementwise gradient launch;
//     patch corners are read from device memory, so nothing ever syncs with the host.
// All of these kernels are HBM-bound streaming passes (36-40 B/px).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fsgs.h"
#include "fsgs_device.h"
#include "fsgs_host.h"

using namespace fsgs;

namespace {

#if defined(FSGS_DIAG_HOOKS) && defined(FSGS_EXP_LOSS_DYNLDS)  // diagnostics flavour only: unused dynamic LDS = fewer workgroups of the forward kernel per CU
constexpr int kFwdDynLds = FSGS_EXP_LOSS_DYNLDS;
#else
constexpr int kFwdDynLds = 0;
#endif

constexpr int SS_TILE = 32;            // output tile edge
constexpr int SS_HALO = 5;             // 11-tap window
constexpr int SS_IN = SS_TILE + 2 * SS_HALO;  // 42
constexpr int SS_BLK = 4;              // outputs per thread along the filtered axis
constexpr float SS_C1 = 0.01f * 0.01f;
constexpr float SS_C2 = 0.03f * 0.03f;

// exp(-(i-5)^2 / (2*1.5^2)) normalised, i = 0..10 (utils/loss_utils.py:56-58), computed in double
__constant__ float kGauss[11] = {0.0010283801f, 0.0075987581f, 0.0360007721f, 0.1093606895f, 0.2130055377f,
                                 0.2660117249f, 0.2130055377f, 0.1093606895f, 0.0360007721f, 0.0075987581f,
                                 0.0010283801f};

// element idx of a plane through a 32-bit BYTE offset: the load takes the plane's (uniform, scalar) base and one vector
// offset register (global_load_dword v, v_off, s[base]) instead of a 64-bit address pair formed per load
#ifndef FSGS_LOSS_V2
#define FSGS_LOSS_V2 0
#endif
__device__ __forceinline__ float ld_plane(const float *__restrict__ base, uint32_t idx) {
#if FSGS_LOSS_V2
  return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)(uint32_t)(idx << 2));
#else
  return base[idx];
#endif
}
__device__ __forceinline__ void st_plane(float *__restrict__ base, uint32_t idx, float v) {
#if FSGS_LOSS_V2
  *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (size_t)(uint32_t)(idx << 2)) = v;
#else
  base[idx] = v;
#endif
}

__device__ __forceinline__ float block_sum_256(float v, float *red) {
  // 4 waves: DPP wave sums, then one LDS hop
  float w = wave_sum(v);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wid] = w;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// ---------------------------------------------------------------------------------------------------
// photometric forward
// sums[0] += sum |x-y| ; sums[1] += sum ssim_map          (doubles, zeroed by the caller)
// maps: [3][C][H][W] = d ssim / d m1, d ssim / d e11, d ssim / d e12   (m1 = G*x, e11 = G*x^2, e12 = G*xy)
// ---------------------------------------------------------------------------------------------------
// The mask of a pixel: mask[p] (a multiplicative factor, or 1) times [presence[p] > 0] (the tracking step's
// "rendered depth > 0" test of train.py:176-178, evaluated here instead of by an elementwise kernel of its own).
__device__ __forceinline__ float pixel_mask(const float *__restrict__ mask, const float *__restrict__ presence, size_t p) {
  float m = mask ? mask[p] : 1.0f;
  if (presence) m = presence[p] > 0.f ? m : 0.f;
  return m;
}
// (the body of the kernel as a function of the tile id: photometric_fwd_kernel and the fused launch of a view's loss stage,
// view_losses_fwd_kernel, both run it)
__device__ __forceinline__ void photometric_fwd_tile(const int tile_id, int C, int H, int W, const float *__restrict__ img,
                                                     const float *__restrict__ gt, const float *__restrict__ mask,
                                                     const float *__restrict__ presence, float *__restrict__ maps,
                                                     float *__restrict__ partials) {
  // image and target of a pixel side by side, the moments as two pairs: one 8-byte LDS access moves a pair and the 11-tap
  // filters of a pair are one v_pk_fma_f32 per tap, each component still accumulating its taps in the order k = 0..10.
  // SSIM needs the two second moments e11 = G*x^2 and e22 = G*y^2 only as their SUM (sigma1^2 + sigma2^2 =
  // e11 + e22 - m1^2 - m2^2, utils/loss_utils.py:76-81), so FOUR filters are run, not five: (m1, m2) and (e11 + e22, e12).
  // That is a fifth of the filter work and, more to the point, 36.6 KB of LDS instead of 42.2: four workgroups per CU instead
  // of three, for a kernel whose time follows its occupancy (1 / 2 / 3 workgroups per CU: 82 / 57 / 44 us, see DESIGN)
  __shared__ float2v sxy[SS_IN][SS_IN + 1];           // (x, y)
  __shared__ float2v hz_m[SS_IN][SS_TILE + 1];        // horizontal pass of (m1, m2)
  __shared__ float2v hz_s[SS_IN][SS_TILE + 1];        //                    (e11 + e22, e12)
  __shared__ float red[4];
  // XCD-aware placement: the launch is one-dimensional and block b runs on XCD b % 8 (observed; speed only), so every XCD
  // is handed one contiguous run of tiles in (channel, row, column) order -- a band of the image -- and the 5-pixel halos
  // two neighbouring tiles both read meet in ONE L2 instead of being fetched over the fabric by two of them
  const int tiles_x = (W + SS_TILE - 1) / SS_TILE, tiles_y = (H + SS_TILE - 1) / SS_TILE;
  const int ch = tile_id / (tiles_x * tiles_y), in_plane = tile_id - ch * (tiles_x * tiles_y);
  const int tile_y = in_plane / tiles_x, tile_x = in_plane - tile_y * tiles_x;
  const int x0 = tile_x * SS_TILE, y0 = tile_y * SS_TILE;
  const size_t plane = (size_t)H * W;
  const float *ip = img + ch * plane, *gp = gt + ch * plane;
  // every global load of the (tile + halo) window is issued before the first is waited for: a rolled loop with the
  // mask / presence branches inside waits for two memory round trips per round, 14 in a row (measured: 57 -> see DESIGN)
  constexpr int NLD = (SS_IN * SS_IN + 255) / 256;
  const float *mp = mask ? mask : ip, *pp = presence ? presence : ip;  // a valid address either way; unused values are dropped
  float va[NLD], vb[NLD], vm[NLD], vp[NLD];
  bool inb[NLD];
#pragma unroll
  for (int r = 0; r < NLD; r++) {
    const int i = threadIdx.x + 256 * r;
    const int ly = i / SS_IN, lx = i - ly * SS_IN;
    const int gy = y0 + ly - SS_HALO, gx = x0 + lx - SS_HALO;
    inb[r] = i < SS_IN * SS_IN && gy >= 0 && gy < H && gx >= 0 && gx < W;  // outside: zero padding (F.conv2d padding=5)
    const uint32_t p = inb[r] ? (uint32_t)gy * (uint32_t)W + (uint32_t)gx : 0u;  // inside one plane: < 2^30
    va[r] = ld_plane(ip, p);
    vb[r] = ld_plane(gp, p);
    vm[r] = ld_plane(mp, p);
    vp[r] = ld_plane(pp, p);
  }
#pragma unroll
  for (int r = 0; r < NLD; r++) {
    const int i = threadIdx.x + 256 * r;
    const int ly = i / SS_IN, lx = i - ly * SS_IN;
    float m = mask ? vm[r] : 1.0f;  // pixel_mask()
    if (presence) m = vp[r] > 0.f ? m : 0.f;
    if (i < SS_IN * SS_IN) sxy[ly][lx] = inb[r] ? float2v{va[r] * m, vb[r] * m} : float2v{0.f, 0.f};
  }
  __syncthreads();
  // Both 11-tap passes are register-blocked: a thread produces SS_BLK consecutive outputs from SS_BLK + 10 inputs
  // it reads once (3.5 LDS reads per output and map instead of 11).  Every output still accumulates its taps in
  // the order k = 0..10.
  // horizontal pass of the five moments: SS_IN rows x (SS_TILE / SS_BLK) segments
#if FSGS_LOSS_V2
  // 336 tasks for 256 threads: the 80 tasks of the second round go to a different wave from tile to tile (in the plain
  // order below they are always wave 0's and a quarter of wave 1's: if wave w of every workgroup sits on SIMD w, SIMD 0
  // runs two rounds for every tile of the CU and SIMDs 2, 3 one)
  const int rot_t = (int)((threadIdx.x + 64u * ((unsigned)tile_id & 3u)) & 255u);
  for (int round = 0; round < 2; round++) {
    const int task = round == 0 ? (int)threadIdx.x : 256 + rot_t;
    if (task >= SS_IN * (SS_TILE / SS_BLK)) break;
    const int ly = task / (SS_TILE / SS_BLK), lx = (task - ly * (SS_TILE / SS_BLK)) * SS_BLK;
#else
  for (int task = threadIdx.x; task < SS_IN * (SS_TILE / SS_BLK); task += 256) {
    const int ly = task / (SS_TILE / SS_BLK), lx = (task - ly * (SS_TILE / SS_BLK)) * SS_BLK;
#endif
    float2v ab[SS_BLK + 10], sq[SS_BLK + 10];
#pragma unroll
    for (int j = 0; j < SS_BLK + 10; j++) {
      ab[j] = sxy[ly][lx + j];
      sq[j] = float2v{ab[j].x * ab[j].x + ab[j].y * ab[j].y, ab[j].x * ab[j].y};
    }
#pragma unroll
    for (int o = 0; o < SS_BLK; o++) {
      float2v m = {0.f, 0.f}, e = {0.f, 0.f};
#pragma unroll
      for (int k = 0; k < 11; k++) {
        const float2v g2 = {kGauss[k], kGauss[k]};
        m = __builtin_elementwise_fma(g2, ab[o + k], m);
        e = __builtin_elementwise_fma(g2, sq[o + k], e);
      }
      hz_m[ly][lx + o] = m;
      hz_s[ly][lx + o] = e;
    }
  }
  __syncthreads();
  float l1_acc = 0.f, ss_acc = 0.f;
  const size_t cplane = (size_t)C * plane;
  {
    // vertical pass: thread = (column lx, SS_BLK consecutive rows); 32 columns x 8 row groups = 256 threads
    static_assert(SS_TILE * (SS_TILE / SS_BLK) == 256, "one task per thread");
    const int lx = threadIdx.x & (SS_TILE - 1), ly0 = (threadIdx.x / SS_TILE) * SS_BLK;
    float2v mom_m[SS_BLK], mom_s[SS_BLK];
    {
      float2v v[SS_BLK + 10];
#pragma unroll
      for (int j = 0; j < SS_BLK + 10; j++) v[j] = hz_m[ly0 + j][lx];
#pragma unroll
      for (int o = 0; o < SS_BLK; o++) {
        float2v acc = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; k++) acc = __builtin_elementwise_fma(float2v{kGauss[k], kGauss[k]}, v[o + k], acc);
        mom_m[o] = acc;
      }
#pragma unroll
      for (int j = 0; j < SS_BLK + 10; j++) v[j] = hz_s[ly0 + j][lx];
#pragma unroll
      for (int o = 0; o < SS_BLK; o++) {
        float2v acc = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 11; k++) acc = __builtin_elementwise_fma(float2v{kGauss[k], kGauss[k]}, v[o + k], acc);
        mom_s[o] = acc;
      }
    }
#pragma unroll
    for (int o = 0; o < SS_BLK; o++) {
      const int ly = ly0 + o, gy = y0 + ly, gx = x0 + lx;
      if (gy >= H || gx >= W) continue;
      const float m1 = mom_m[o].x, m2 = mom_m[o].y, ess = mom_s[o].x, e12 = mom_s[o].y;  // ess = e11 + e22
      float A1 = 2.f * m1 * m2 + SS_C1;
      float A2 = 2.f * (e12 - m1 * m2) + SS_C2;
      float B1 = m1 * m1 + m2 * m2 + SS_C1;
      float B2 = ((ess - m1 * m1) - m2 * m2) + SS_C2;
      // v_rcp_f32 (1 ulp) instead of four IEEE divisions per output: each of those expands to ~10 instructions
      // (div_scale, rcp, Newton steps, div_fixup) -- a fifth of this kernel's VALU work for a 24th bit the loss never sees
      const float rB1 = __builtin_amdgcn_rcpf(B1), rB2 = __builtin_amdgcn_rcpf(B2);
      float inv = rB1 * rB2;
      float S = A1 * A2 * inv;
      ss_acc += S;
      const float2v c = sxy[ly + SS_HALO][lx + SS_HALO];
      l1_acc += fabsf(c.x - c.y);
      const uint32_t p = (uint32_t)gy * (uint32_t)W + (uint32_t)gx;
      float *mc = maps + (size_t)ch * plane;
      st_plane(mc, p, 2.f * m2 * (A2 - A1) * inv - 2.f * m1 * S * (rB1 - rB2));  // d/dm1
      st_plane(mc + cplane, p, -S * rB2);                                         // d/de11
      st_plane(mc + 2 * cplane, p, 2.f * A1 * inv);                               // d/de12
    }
  }
  float t1 = block_sum_256(l1_acc, red);
  float t2 = block_sum_256(ss_acc, red);
  if (threadIdx.x == 0) {
    // per-workgroup partials, reduced by the finish kernel: thousands of atomics on two addresses would
    // serialise in one L2 channel and dominate this (otherwise streaming) kernel
    const size_t b = (size_t)tile_id;  // (channel, row, column) order whatever the placement: the finish sums in this order
    partials[2 * b] = t1;
    partials[2 * b + 1] = t2;
  }
}

__global__ __launch_bounds__(256) void photometric_fwd_kernel(int C, int H, int W, const float *__restrict__ img,
                                                              const float *__restrict__ gt,
                                                              const float *__restrict__ mask,
                                                              const float *__restrict__ presence,
                                                              float *__restrict__ maps,
                                                              float *__restrict__ partials) {
#if defined(FSGS_DIAG_HOOKS) && defined(FSGS_EXP_LOSS_NO_XCD)  // diagnostics flavour only (same results, other placement)
  const int tile_id = blockIdx.x;
#else
  const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
#endif
  photometric_fwd_tile(tile_id, C, H, W, img, gt, mask, presence, maps, partials);
}

// ---- row-streaming geometry of the backward kernel ----
constexpr int ST_W = 64;                      // columns per wave
constexpr int ST_RS = 16;                     // output rows per wave
constexpr int ST_IN = ST_W + 2 * SS_HALO;     // 74
constexpr int ST_NR = ST_RS + 2 * SS_HALO;    // input rows per wave
#ifndef FSGS_LOSS_PF
#define FSGS_LOSS_PF 2
#endif
constexpr int ST_PF = FSGS_LOSS_PF;           // input rows in flight per wave

// loss = (1-l) * L1mean + l * (1 - SSIMmean);  out[0] = loss, out[1] = L1 mean, out[2] = SSIM mean
// The same reduction by ONE wave (fixed order too): the extra workgroup of the fused forward + backward entry, which
// finishes the loss value beside the backward's row streams instead of in a launch of its own between the two.
struct FinishArgs {
  const float *partials;
  int nblocks;
  double n;
  float *out;  // NULL: this launch has no finishing workgroup
};
__device__ __forceinline__ void photometric_finish_wave(const FinishArgs &f, float lambda_dssim, int lane) {
  double a = 0.0, b = 0.0;
  for (int i = lane; i < f.nblocks; i += 64) {
    a += (double)f.partials[2 * i];
    b += (double)f.partials[2 * i + 1];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_xor(a, off, 64);
    b += __shfl_xor(b, off, 64);
  }
  if (lane == 0) {
    const double l1 = a / f.n, ss = b / f.n;
    f.out[0] = (float)((1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ss));
    f.out[1] = (float)l1;
    f.out[2] = (float)ss;
  }
}

__global__ __launch_bounds__(256) void photometric_finish_kernel(const float *__restrict__ partials, int nblocks,
                                                                 double n, float lambda_dssim, float *out) {
  __shared__ double red[2][256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) {
    a += (double)partials[2 * i];
    b += (double)partials[2 * i + 1];
  }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double l1 = red[0][0] / n, ss = red[1][0] / n;
    out[0] = (float)((1.0 - lambda_dssim) * l1 + lambda_dssim * (1.0 - ss));
    out[1] = (float)l1;
    out[2] = (float)ss;
  }
}

// dL/dimg = upstream * [ (1-l)/N sign(x-y) - l/N ( G*dm1 + 2x G*de11 + y G*de12 ) ] * mask
// Row-streaming, ONE WAVE PER WORKGROUP: a wave owns 64 columns (lane = column) and ST_RS output rows.  Input
// rows of the three maps stream through a 74-float LDS line each (the only cross-lane exchange: the 11
// horizontal taps), the horizontal results of the last 11 rows live in a REGISTER ring and the vertical pass reads
// only registers: no workgroup barrier, < 1 KB of LDS.  (The same scheme was measured for the forward kernel,
// whose five moments need 154 VGPRs and 33 extra multiplies per row: 82 us against 55 us for the tiled kernel.)
// (the body as a function of the strip id: photometric_bwd_kernel and view_losses_bwd_kernel both run it)
// INTERIOR (FSGS_LOSS_V2): the strip and its halo lie inside the image (87 % of the strips at 1280x1024) -- nothing is zero
// padding, the twelve selects per input row that put the zeros in are left out
template <bool INTERIOR>
__device__ __forceinline__ void photometric_bwd_strip_t(const int strip, int C, int H, int W, const float *__restrict__ img,
                                                      const float *__restrict__ gt, const float *__restrict__ mask,
                                                      const float *__restrict__ presence, const float *__restrict__ maps,
                                                      const float *__restrict__ upstream, float lambda_dssim,
                                                      float *__restrict__ dimg) {
  __shared__ float row[3][ST_IN + 6];
  const int lane = threadIdx.x;
  const int strips_x = (W + ST_W - 1) / ST_W, strips_y = (H + ST_RS - 1) / ST_RS;
  const int ch = strip / (strips_x * strips_y), s2 = strip - ch * (strips_x * strips_y);
  const int strip_y = s2 / strips_x, strip_x = s2 - strip_y * strips_x;
  const int x0 = strip_x * ST_W, y0 = strip_y * ST_RS;
  const int gx = x0 + lane;
  const size_t plane = (size_t)H * W, cplane = (size_t)C * plane;
  // Every load of this kernel is UNCONDITIONAL, from coordinates clamped into the image with min / max (no comparison the
  // compiler could turn into a branch around the load), and what lies outside is masked where the value is USED, one or
  // more row iterations later.  With the loads inside conditions (or with an address selected by a condition: the compiler
  // then loads the uniform default with the scalar unit and puts the vector load back under a branch) the values meet
  // their defaults in a register copy right behind the load -- which waits for it: one exposed memory round trip per row.
  const int c0 = x0 - SS_HALO + lane, c1 = x0 - SS_HALO + ST_W + lane;
  const bool in_c0 = c0 >= 0 && c0 < W, in_c1 = lane < 2 * SS_HALO && c1 < W;
  const uint32_t c0c = (uint32_t)min(max(c0, 0), W - 1);
  const uint32_t c1c = lane < 2 * SS_HALO ? (uint32_t)min(c1, W - 1) : c0c;  // the other lanes repeat their first address
  const float *map0 = maps + (size_t)ch * plane, *map1 = map0 + cplane, *map2 = map1 + cplane;
  const float *img_c = img + (size_t)ch * plane, *gt_c = gt + (size_t)ch * plane;
  const float *mask_c = mask ? mask : img_c, *pres_c = presence ? presence : img_c;
  const uint32_t gxc = (uint32_t)min(gx, W - 1);
  auto fetch_row = [&](int gy, float (&v0)[3], float (&v1)[3]) {
    const uint32_t rowoff = (uint32_t)min(max(gy, 0), H - 1) * (uint32_t)W;  // 32-bit offsets inside one plane (H * W < 2^30)
    v0[0] = ld_plane(map0, rowoff + c0c); v0[1] = ld_plane(map1, rowoff + c0c); v0[2] = ld_plane(map2, rowoff + c0c);
    v1[0] = ld_plane(map0, rowoff + c1c); v1[1] = ld_plane(map1, rowoff + c1c); v1[2] = ld_plane(map2, rowoff + c1c);
  };
  const float up = upstream ? upstream[0] : 1.0f;
  const float invN = 1.0f / ((float)C * (float)H * (float)W);
  const float k_l1 = up * (1.0f - lambda_dssim) * invN, k_ss = -up * lambda_dssim * invN;
  float ring[11][3];
  // ST_PF input rows are in flight: iteration r consumes the map row and the output pixel (image, target, mask, presence)
  // requested ST_PF iterations earlier and re-issues its stage for iteration r + ST_PF.  With one row in flight every
  // iteration exposed a memory round trip (26 of them per wave at ~4 waves per SIMD: the kernel's time).  The stage of an
  // iteration is a compile-time index (the loop is unrolled 11 * ST_PF deep): rotating the registers instead would wait
  // for the loads it moves.  Branch-free (clamped address, every load issued, unused values dropped): a load inside a
  // conditional block is waited for at the block's end.
  float n0[ST_PF][3], n1[ST_PF][3], nx[ST_PF], ny[ST_PF], nm[ST_PF], np_[ST_PF];
  auto fetch_pixel = [&](int rr, float &x, float &y, float &m, float &pr) {  // the output pixel of iteration rr
    const uint32_t e = (uint32_t)min(max(y0 + rr - 2 * SS_HALO, 0), H - 1) * (uint32_t)W + gxc;
    x = ld_plane(img_c, e);
    y = ld_plane(gt_c, e);
    m = ld_plane(mask_c, e);
    pr = ld_plane(pres_c, e);
  };
#pragma unroll
  for (int st = 0; st < ST_PF; st++) {
    fetch_row(y0 - SS_HALO + st, n0[st], n1[st]);
    fetch_pixel(st, nx[st], ny[st], nm[st], np_[st]);
  }
  for (int r0 = 0; r0 < ST_NR; r0 += 11 * ST_PF) {
#pragma unroll
    for (int qq = 0; qq < 11 * ST_PF; qq++) {
      const int r = r0 + qq;
      if (r >= ST_NR) break;  // wave-uniform
      constexpr int kRing = 11;
      const int q = qq % kRing, st = qq % ST_PF;
      __syncthreads();
      {
        const int gy = y0 - SS_HALO + r;
        const bool in_row = gy >= 0 && gy < H;  // outside the image: zero padding (F.conv2d padding = 5)
#pragma unroll
        for (int m = 0; m < 3; m++) {
          row[m][lane] = (INTERIOR || (in_row && in_c0)) ? n0[st][m] : 0.f;
          if (lane < 2 * SS_HALO) row[m][ST_W + lane] = (INTERIOR || (in_row && in_c1)) ? n1[st][m] : 0.f;
        }
      }
      __syncthreads();
      const int oy = y0 + r - 2 * SS_HALO;
      const bool out_ok = INTERIOR || (r >= 2 * SS_HALO && oy < H && gx < W);  // (r < 2 * SS_HALO leaves the iteration below)
      const float ex = nx[st], ey = ny[st], em = nm[st], ep = np_[st];
      fetch_row(y0 - SS_HALO + r + ST_PF, n0[st], n1[st]);  // (the last ST_PF requests are never used: no condition, see above)
      fetch_pixel(r + ST_PF, nx[st], ny[st], nm[st], np_[st]);
#pragma unroll
      for (int m = 0; m < 3; m++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = fmaf(kGauss[k], row[m][lane + k], acc);
        ring[q][m] = acc;
      }
      if (r < 2 * SS_HALO) continue;
      float f[3];
#pragma unroll
      for (int m = 0; m < 3; m++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 11; k++) acc = fmaf(kGauss[k], ring[(q + 1 + k) % 11][m], acc);
        f[m] = acc;
      }
      if (out_ok) {
        float mk = mask ? em : 1.0f;  // pixel_mask()
        if (presence) mk = ep > 0.f ? mk : 0.f;
        const float x = ex * mk, y = ey * mk;
        const float d = x - y;
        const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        dimg[ch * plane + (size_t)oy * W + gx] = mk * (k_l1 * sgn + k_ss * (f[0] + 2.f * x * f[1] + y * f[2]));
      }
    }
  }
}

__device__ __forceinline__ void photometric_bwd_strip(const int strip, int C, int H, int W, const float *__restrict__ img,
                                                      const float *__restrict__ gt, const float *__restrict__ mask,
                                                      const float *__restrict__ presence, const float *__restrict__ maps,
                                                      const float *__restrict__ upstream, float lambda_dssim,
                                                      float *__restrict__ dimg) {
#if FSGS_LOSS_V2
  const int strips_x = (W + ST_W - 1) / ST_W, strips_y = (H + ST_RS - 1) / ST_RS;
  const int s2 = strip % (strips_x * strips_y);
  const int x0 = (s2 % strips_x) * ST_W, y0 = (s2 / strips_x) * ST_RS;
  if (x0 >= SS_HALO && x0 + ST_W + SS_HALO <= W && y0 >= SS_HALO && y0 + ST_RS + SS_HALO <= H) {  // wave-uniform
    photometric_bwd_strip_t<true>(strip, C, H, W, img, gt, mask, presence, maps, upstream, lambda_dssim, dimg);
    return;
  }
#endif
  photometric_bwd_strip_t<false>(strip, C, H, W, img, gt, mask, presence, maps, upstream, lambda_dssim, dimg);
}

__global__ __launch_bounds__(64) void photometric_bwd_kernel(int C, int H, int W, const float *__restrict__ img,
                                                             const float *__restrict__ gt,
                                                             const float *__restrict__ mask,
                                                             const float *__restrict__ presence,
                                                             const float *__restrict__ maps,
                                                             const float *__restrict__ upstream, float lambda_dssim,
                                                             float *__restrict__ dimg, FinishArgs fin) {
  const int nstrips = ((W + ST_W - 1) / ST_W) * ((H + ST_RS - 1) / ST_RS) * C;
  if ((int)blockIdx.x >= nstrips) {  // the grid is one workgroup longer: it finishes the loss
    if (fin.out) photometric_finish_wave(fin, lambda_dssim, threadIdx.x);
    return;
  }
  // XCD-aware placement (see the forward kernel): every XCD streams one band of vertically adjacent strips, whose 10 shared
  // halo rows per boundary then come out of its own L2
#if defined(FSGS_DIAG_HOOKS) && defined(FSGS_EXP_LOSS_NO_XCD)  // diagnostics flavour only (same results, other placement)
  const int strip = blockIdx.x;
#else
  const int strip = xcd_swizzle(blockIdx.x, nstrips);
#endif
  photometric_bwd_strip(strip, C, H, W, img, gt, mask, presence, maps, upstream, lambda_dssim, dimg);
}

// ---------------------------------------------------------------------------------------------------
// Pearson correlation losses (utils/loss_utils.py:98-127)
// region 0 = whole image; regions 1..n = box x box patches with top-left (row0[r-1], col0[r-1]).
// stats[r][0..4] = sum s, sum t, sum s^2, sum t^2, sum s*t   (doubles, zeroed by the caller)
// ---------------------------------------------------------------------------------------------------
constexpr int PE_CHUNK = 4096;  // pixels per workgroup

// (the body as a function of (region, chunk): pearson_stats_kernel and view_losses_fwd_kernel both run it)
__device__ __forceinline__ void pearson_stats_chunk(const int r, const int chunk, int H, int W, int box,
                                                    const int64_t *__restrict__ row0, const int64_t *__restrict__ col0,
                                                    const float *__restrict__ src, const float *__restrict__ tgt,
                                                    double *__restrict__ partials, int n0, int nb) {
  __shared__ float red[4];
  int ry = 0, rx = 0, rh = H, rw = W;
  if (r > 0) {
    ry = (int)row0[r - 1]; rx = (int)col0[r - 1]; rh = box; rw = box;
  }
  const int n = rh * rw;
  const int begin = chunk * PE_CHUNK;
  if (begin >= n) return;
  const int end = min(n, begin + PE_CHUNK);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
  // a thread's 16 pixels (i = begin + t + 256 k) in two batches of eight: all loads of a batch are issued before the
  // first is used (one pixel per loop iteration is 16 dependent memory round trips); the sums still grow in the order
  // k = 0, 1, ..  Row / column of consecutive k follow by increments, one division per thread.
  constexpr int PB = 8;
  const int dq = 256 / rw, dr = 256 % rw;
  int i = begin + (int)threadIdx.x;
  int y = i / rw, x = i - y * rw;
  for (; i < end; i += 256 * PB) {
    float sv[PB], tv[PB];
#pragma unroll
    for (int k = 0; k < PB; k++) {
      const bool ok = i + 256 * k < end;
      const size_t p = ok ? (size_t)(ry + y) * W + (rx + x) : 0;
      sv[k] = src[p];
      tv[k] = tgt[p];
      x += dr;
      y += dq;
      if (x >= rw) { x -= rw; y++; }
    }
#pragma unroll
    for (int k = 0; k < PB; k++)
      if (i + 256 * k < end) {
        const float s_ = sv[k], t_ = tv[k];
        a0 += s_; a1 += t_;
        a2 = fmaf(s_, s_, a2); a3 = fmaf(t_, t_, a3); a4 = fmaf(s_, t_, a4);
      }
  }
  float t0 = block_sum_256(a0, red), t1 = block_sum_256(a1, red), t2 = block_sum_256(a2, red);
  float t3 = block_sum_256(a3, red), t4 = block_sum_256(a4, red);
  if (threadIdx.x == 0) {
    // one partial per workgroup, summed in a fixed order by the finish kernel: hundreds of same-address double
    // atomics (~46 ns each) used to take longer than reading the two images, and made the sum order-dependent
    double *st = partials + 5 * (size_t)(r == 0 ? chunk : n0 + (r - 1) * nb + chunk);
    st[0] = (double)t0; st[1] = (double)t1; st[2] = (double)t2; st[3] = (double)t3; st[4] = (double)t4;
  }
}

__global__ __launch_bounds__(256) void pearson_stats_kernel(int H, int W, int box, const int64_t *__restrict__ row0,
                                                            const int64_t *__restrict__ col0,
                                                            const float *__restrict__ src,
                                                            const float *__restrict__ tgt,
                                                            double *__restrict__ partials, int n0, int nb) {
  pearson_stats_chunk(blockIdx.y, blockIdx.x, H, W, box, row0, col0, src, tgt, partials, n0, nb);
}

// the five sums of a region -> its row of coefficients (below); the ONE statement of this arithmetic
__device__ __forceinline__ void pearson_region_coef(const double (&st)[5], const double N, float *c) {
  double ms = st[0] / N, mt = st[1] / N;
  double vs = (st[2] - N * ms * ms) / (N - 1.0), vt = (st[3] - N * mt * mt) / (N - 1.0);
  vs = vs > 0 ? vs : 0; vt = vt > 0 ? vt : 0;
  double sds = sqrt(vs), sdt = sqrt(vt);
  double cov = st[4] / N - ms * mt;
  double D = (sds + 1e-6) * (sdt + 1e-6);
  c[0] = (float)ms; c[1] = (float)mt; c[2] = (float)(1.0 / (N * D));
  c[3] = sdt > 0 ? (float)(cov / (D * (sdt + 1e-6) * (N - 1.0) * sdt)) : 0.f;
  c[4] = sds > 0 ? (float)(cov / (D * (sds + 1e-6) * (N - 1.0) * sds)) : 0.f;
  c[5] = (float)(1.0 - cov / D);
}

// per region: coefficients of the gradient + the loss value
//   co = cov / ((sd_s+eps)(sd_t+eps)), cov = E[st]-E[s]E[t] (biased, mean()), sd unbiased
//   coef[r] = {mean_s, mean_t, 1/(N*D), cov/(D*(sd_t+eps)*(N-1)*sd_t), cov/(D*(sd_s+eps)*(N-1)*sd_s), loss}
// one 64-thread workgroup per region
// ONE workgroup of 16 waves: wave w finishes regions w, w + 16, ..; after a barrier wave 0 adds the patch losses in the
// order of the reference's python loop.  (Two launches -- one workgroup per region, then the mean -- cost 12 us of launch
// latency on the side stream for microseconds of work.)
__global__ __launch_bounds__(1024) void pearson_finish_kernel(int H, int W, int box, int nregions,
                                                              const double *__restrict__ partials, int n0, int nb,
                                                              float *__restrict__ coef, float *__restrict__ out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < nregions; r += 16) {
    const int cnt = r == 0 ? n0 : nb;
    const double *base = partials + 5 * (size_t)(r == 0 ? 0 : n0 + (r - 1) * nb);
    double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    for (int i = lane; i < cnt; i += 64)
#pragma unroll
      for (int q = 0; q < 5; q++) st[q] += base[5 * i + q];
#pragma unroll
    for (int q = 0; q < 5; q++)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) st[q] += __shfl_xor(st[q], off, 64);
    if (lane == 0) {
      float *c = coef + 8 * r;
      pearson_region_coef(st, r == 0 ? (double)H * W : (double)box * box, c);
      if (r == 0) out[0] = c[5];  // global loss
    }
  }
  __syncthreads();  // (the coefficient rows were written to global memory by this workgroup: visible after the barrier)
  if (wave == 0) {
    // lane r fetches region r's loss (parallel loads), lane 0 then adds them in the reference's order (nregions <= 65)
    const float mine = (lane >= 1 && lane < nregions) ? coef[8 * lane + 5] : 0.f;
    const float last = nregions > 64 ? coef[8 * 64 + 5] : 0.f;
    float acc = 0.f;
    for (int r = 1; r < nregions && r < 64; r++) acc += readlane(mine, r);
    if (nregions > 64) acc += last;
    if (lane == 0) out[1] = nregions > 1 ? acc / (float)(nregions - 1) : 0.f;
  }
}

// d(loss_r)/d tgt_i = -(s_i-ms)/(N D) + cov/(D (sd_t+eps) (N-1) sd_t) (t_i - mt), weighted by weight[r]
// (weight[0] = upstream of the global loss, weight[r>0] = upstream_local / n_patches); wrt_src swaps roles.
__global__ __launch_bounds__(256) void pearson_bwd_kernel(int H, int W, int box, int nregions,
                                                          const int64_t *__restrict__ row0,
                                                          const int64_t *__restrict__ col0,
                                                          const float *__restrict__ src, const float *__restrict__ tgt,
                                                          const float *__restrict__ coef,
                                                          const float *__restrict__ weight, int wrt_src,
                                                          float *__restrict__ grad) {
  __shared__ float s_coef[65][8];
  __shared__ int s_rect[65][2];
  for (int i = threadIdx.x; i < nregions * 8; i += 256) s_coef[i / 8][i % 8] = coef[i] * ((i % 8) >= 2 && (i % 8) <= 4 ? weight[i / 8] : 1.f);
  for (int i = threadIdx.x; i < nregions; i += 256) {
    s_rect[i][0] = i == 0 ? 0 : (int)row0[i - 1];
    s_rect[i][1] = i == 0 ? 0 : (int)col0[i - 1];
  }
  __syncthreads();
  const size_t n = (size_t)H * W;
  // A chunk of 256 consecutive pixels meets only a few of the patches (~1.5 of 40 at C2): the workgroup first lists, in
  // ascending order, the patches whose rectangle touches the chunk's rows and columns at all, and every pixel then tests
  // that short list instead of all of them (40 rectangle tests per pixel were 90 % of this kernel's instructions).  The
  // terms of a pixel are still added in ascending region order.
  __shared__ int s_list[65];
  __shared__ int s_nlist;
  for (size_t c0 = (size_t)blockIdx.x * 256; c0 < n; c0 += (size_t)gridDim.x * 256) {
    const size_t c1 = min(n, c0 + 256) - 1;  // last pixel of the chunk
    const int ya = (int)(c0 / W), yb = (int)(c1 / W);
    const int xa = ya == yb ? (int)(c0 - (size_t)ya * W) : 0, xb = ya == yb ? (int)(c1 - (size_t)yb * W) : W - 1;
    __syncthreads();  // the previous chunk's list is no longer read
    if (threadIdx.x < 64) {  // one wave builds the list: lanes test regions l + 1 and l + 65 - 64 .. (nregions <= 65)
      int cnt = 0;
      for (int r0 = 1; r0 < nregions; r0 += 64) {
        const int r = r0 + (int)threadIdx.x;
        bool hit = false;
        if (r < nregions) {
          const int py = s_rect[r][0], px_ = s_rect[r][1];
          hit = py <= yb && py + box > ya && px_ <= xb && px_ + box > xa;
        }
        const unsigned long long m = __ballot(hit);
        if (hit) s_list[cnt + __popcll(m & ((1ull << threadIdx.x) - 1ull))] = r;
        cnt += __popcll(m);
      }
      if (threadIdx.x == 0) s_nlist = cnt;
    }
    __syncthreads();
    const size_t p = c0 + threadIdx.x;
    if (p < n) {
      const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
      const float s_ = src[p], t = tgt[p];
      float g = 0.f;
      {
        const float *c = s_coef[0];
        float a = s_ - c[0], b = t - c[1];
        g += wrt_src ? (-b * c[2] + c[4] * a) : (-a * c[2] + c[3] * b);
      }
      const int nl = s_nlist;
      for (int q = 0; q < nl; q++) {
        const int r = s_list[q];
        int dy = y - s_rect[r][0], dx = x - s_rect[r][1];
        if ((unsigned)dy < (unsigned)box && (unsigned)dx < (unsigned)box) {
          const float *c = s_coef[r];
          float a = s_ - c[0], b = t - c[1];
          g += wrt_src ? (-b * c[2] + c[4] * a) : (-a * c[2] + c[3] * b);
        }
      }
      grad[p] = g;
    }
  }
}


// ---------------------------------------------------------------------------------------------------
// One view's loss stage in TWO launches on ONE stream (round 6).
// The photometric pair and the Pearson chain used to run on two streams: the fork behind the forward blend and the join
// in front of the backward blend each cost a signal round trip between two hardware queues -- 10 and 15 us of idle GPU
// per mapping iteration at 1280x1024 (profiles/r06_step_timeline.txt: blend_fwd ends at 173 us, photometric_fwd starts
// at 183; photometric_bwd ends at 249, blend_bwd starts at 264), and the single-workgroup pearson_finish was a 22 us link of
// the side chain.  Here the Pearson work rides in the photometric launches as extra workgroups at the FRONT of the grid
// (dispatched first, they finish under the photometric tiles):
//   launch 1 = pearson_stats chunks | photometric_fwd tiles
//   launch 2 = pearson gradient chunks | photometric_bwd strips | one finishing wave
// and no launch of its own finishes the Pearson regions: every gradient wave sums the per-chunk partials of launch 1
// itself (region 0 by the whole wave, patch r by lane r -- 45 loads and one round of double arithmetic per wave, the same
// order of additions as pearson_finish_kernel), so the stage needs nothing but the stream's own launch order.
// Same results, bit for bit, as fsgs_photometric_loss_forward_backward + fsgs_pearson_forward + fsgs_pearson_backward.
// ---------------------------------------------------------------------------------------------------
struct PearsonView {
  int box, nregions;          // nregions = n_patches + 1
  const int64_t *row0, *col0;
  const float *src, *tgt;     // mono-depth, rendered depth
  double *partials;
  int n0, nb;                 // chunks of region 0 / of a patch
};

constexpr int PV_PIX = 2048;   // pixels per gradient wave (8 chunks of 256)
constexpr int PV_MAXNB = 4;    // a patch's partials are added by ONE lane in the butterfly's order: up to 4 of them (box <= 128;
                               // 8 would cost the launch 40 more VGPRs and the photometric strips beside it a third of their waves)

// the wave's copy of pearson_finish_kernel: lane r ends up with region r's coefficient row in c[0..5] (lanes >= nregions: zeros)
__device__ __forceinline__ void pearson_coefs_wave(const PearsonView &pv, int H, int W, int lane, float (&c)[6]) {
  // region 0: lane-strided sums + xor butterfly (every lane ends with the same bits)
  double st[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  for (int i = lane; i < pv.n0; i += 64)
#pragma unroll
    for (int q = 0; q < 5; q++) st[q] += pv.partials[5 * i + q];
  // patch `lane`: its nb <= 4 partials, added as the butterfly of 64 lanes adds them when lanes >= nb hold zero:
  // (p0 + p2) + (p1 + p3)
  double t[5][PV_MAXNB];
  const bool patch = lane >= 1 && lane < pv.nregions;
  const double *base = pv.partials + 5 * (size_t)(pv.n0 + (patch ? lane - 1 : 0) * pv.nb);
#pragma unroll
  for (int i = 0; i < PV_MAXNB; i++)
#pragma unroll
    for (int q = 0; q < 5; q++) t[q][i] = (patch && i < pv.nb) ? base[5 * i + q] : 0.0;
#pragma unroll
  for (int q = 0; q < 5; q++)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) st[q] += __shfl_xor(st[q], off, 64);
  if (patch) {
#pragma unroll
    for (int q = 0; q < 5; q++) {
      // (0.0 + p = p: the accumulators of pearson_finish_kernel start at zero)
      st[q] = (t[q][0] + t[q][2]) + (t[q][1] + t[q][3]);
    }
  }
#pragma unroll
  for (int q = 0; q < 6; q++) c[q] = 0.f;
  if (lane < pv.nregions) pearson_region_coef(st, lane == 0 ? (double)H * W : (double)pv.box * pv.box, c);
}

// launch 1: blocks [0, n_pe) = Pearson statistics (region 0's n0 chunks, then nb per patch), the rest = photometric tiles
__global__ __launch_bounds__(256) void view_losses_fwd_kernel(int C, int H, int W, const float *__restrict__ img,
                                                              const float *__restrict__ gt,
                                                              const float *__restrict__ mask,
                                                              const float *__restrict__ presence,
                                                              float *__restrict__ maps, float *__restrict__ partials,
                                                              PearsonView pv, int n_pe /* multiple of 8 */) {
  const int b = blockIdx.x;
  if (b < n_pe) {
    int r = 0, chunk = b;
    if (b >= pv.n0) {
      r = 1 + (b - pv.n0) / pv.nb;
      chunk = (b - pv.n0) - (r - 1) * pv.nb;
    }
    if (r < pv.nregions) pearson_stats_chunk(r, chunk, H, W, pv.box, pv.row0, pv.col0, pv.src, pv.tgt, pv.partials, pv.n0, pv.nb);
    return;
  }
  // n_pe is a multiple of 8: block b still runs on XCD (b - n_pe) % 8 and the bands of photometric_fwd_kernel stay whole
  photometric_fwd_tile(xcd_swizzle(b - n_pe, (int)gridDim.x - n_pe), C, H, W, img, gt, mask, presence, maps, partials);
}

// launch 2: blocks [0, n_pg) = Pearson gradient (one wave per PV_PIX pixels), then the photometric strips, then one
// finishing wave (photometric loss value, Pearson loss values, the coefficient rows for callers that keep them)
__global__ __launch_bounds__(64) void view_losses_bwd_kernel(int C, int H, int W, const float *__restrict__ img,
                                                             const float *__restrict__ gt,
                                                             const float *__restrict__ mask,
                                                             const float *__restrict__ presence,
                                                             const float *__restrict__ maps,
                                                             const float *__restrict__ upstream, float lambda_dssim,
                                                             float *__restrict__ dimg, FinishArgs fin, PearsonView pv,
                                                             const float *__restrict__ weight, float *__restrict__ grad,
                                                             float *__restrict__ coef, float *__restrict__ out2,
                                                             int n_pg /* multiple of 8 */) {
  const int lane = threadIdx.x;
  const int nstrips = ((W + ST_W - 1) / ST_W) * ((H + ST_RS - 1) / ST_RS) * C;
  const int b = blockIdx.x;
  if (b >= n_pg && b < n_pg + nstrips) {
    photometric_bwd_strip(xcd_swizzle(b - n_pg, nstrips), C, H, W, img, gt, mask, presence, maps, upstream, lambda_dssim, dimg);
    return;
  }
  const size_t n = (size_t)H * W;
  const bool finishing = b >= n_pg;
  if (!finishing && (size_t)b * PV_PIX >= n) return;  // padding of n_pg
  float c[6];
  pearson_coefs_wave(pv, H, W, lane, c);
  if (finishing) {
    photometric_finish_wave(fin, lambda_dssim, lane);
    if (lane < pv.nregions) {
#pragma unroll
      for (int q = 0; q < 6; q++) coef[8 * lane + q] = c[q];
    }
    // the patch losses in the order of the reference's python loop (pearson_finish_kernel's last step)
    float acc = 0.f;
    for (int r = 1; r < pv.nregions && r < 64; r++) acc += readlane(c[5], r);
    if (lane == 0) {
      out2[0] = c[5];
      out2[1] = pv.nregions > 1 ? acc / (float)(pv.nregions - 1) : 0.f;
    }
    return;
  }
  // ---- pearson_bwd_kernel for one wave: d/d tgt (the rendered depth), 4 pixels of a 256-pixel chunk per lane ----
  __shared__ float s_coef[64][8];
  __shared__ int s_rect[64][2];
  __shared__ int s_list[64];
  {
    const float wgt = lane < pv.nregions ? weight[lane] : 0.f;
    s_coef[lane][0] = c[0]; s_coef[lane][1] = c[1];
    s_coef[lane][2] = c[2] * wgt; s_coef[lane][3] = c[3] * wgt; s_coef[lane][4] = c[4] * wgt;
    s_rect[lane][0] = (lane >= 1 && lane < pv.nregions) ? (int)pv.row0[lane - 1] : 0;
    s_rect[lane][1] = (lane >= 1 && lane < pv.nregions) ? (int)pv.col0[lane - 1] : 0;
  }
  __syncthreads();
  const int box = pv.box;
  const size_t first = (size_t)b * PV_PIX, last_ = min(n, first + PV_PIX);
  for (size_t c0 = first; c0 < last_; c0 += 256) {
    const size_t c1 = min(n, c0 + 256) - 1;  // last pixel of the chunk
    const int ya = (int)(c0 / W), yb = (int)(c1 / W);
    const int xa = ya == yb ? (int)(c0 - (size_t)ya * W) : 0, xb = ya == yb ? (int)(c1 - (size_t)yb * W) : W - 1;
    // the loads of the chunk go out before the list is built
    float sv[4], tv[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t p = min(c0 + lane + 64 * k, n - 1);
      sv[k] = pv.src[p];
      tv[k] = pv.tgt[p];
    }
    __syncthreads();  // the previous chunk's list is no longer read
    int nl;
    {
      const int r = 1 + lane;  // nregions <= 64 on this route
      bool hit = false;
      if (r < pv.nregions) {
        const int py = s_rect[r][0], px_ = s_rect[r][1];
        hit = py <= yb && py + box > ya && px_ <= xb && px_ + box > xa;
      }
      const unsigned long long m = __ballot(hit);
      if (hit) s_list[__popcll(m & ((1ull << lane) - 1ull))] = r;
      nl = __popcll(m);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const size_t p = c0 + lane + 64 * k;
      if (p >= n) continue;
      const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
      const float s_ = sv[k], t = tv[k];
      float g = 0.f;
      {
        const float *cc = s_coef[0];
        float a = s_ - cc[0], bb = t - cc[1];
        g += -a * cc[2] + cc[3] * bb;
      }
      for (int q = 0; q < nl; q++) {
        const int r = s_list[q];
        int dy = y - s_rect[r][0], dx = x - s_rect[r][1];
        if ((unsigned)dy < (unsigned)box && (unsigned)dx < (unsigned)box) {
          const float *cc = s_coef[r];
          float a = s_ - cc[0], bb = t - cc[1];
          g += -a * cc[2] + cc[3] * bb;
        }
      }
      grad[p] = g;
    }
  }
}

}  // namespace

extern "C" {

size_t fsgs_photometric_scratch_bytes(int C, int H, int W) {
  if (C <= 0 || H <= 0 || W <= 0) return 0;
  size_t nb = (size_t)((W + SS_TILE - 1) / SS_TILE) * ((H + SS_TILE - 1) / SS_TILE) * C;
  return nb * 2 * sizeof(float) + 64;
}

int fsgs_photometric_loss_forward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                  const float *presence, float lambda_dssim, float *maps, void *sums2, float *out3,
                                  fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !maps || !sums2 || !out3) return FSGS_ERR_INVALID;
  const int nblocks = ((W + SS_TILE - 1) / SS_TILE) * ((H + SS_TILE - 1) / SS_TILE) * C;
  dim3 grid(nblocks);
  float *partials = (float *)sums2;  // caller-sized by fsgs_photometric_scratch_bytes
  {
    ProfScope ps(PROF_LOSS_RGB_FWD, stream);
    hipLaunchKernelGGL(photometric_fwd_kernel, grid, dim3(256), kFwdDynLds, stream, C, H, W, img, gt, mask, presence, maps, partials);
    hipLaunchKernelGGL(photometric_finish_kernel, dim3(1), dim3(256), 0, stream, partials, nblocks, (double)C * H * W,
                       lambda_dssim, out3);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_photometric_loss_backward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                   const float *presence, const float *maps, const float *upstream, float lambda_dssim, float *dimg,
                                   fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !maps || !dimg) return FSGS_ERR_INVALID;
  dim3 grid(((W + ST_W - 1) / ST_W) * ((H + ST_RS - 1) / ST_RS) * C);
  {
    ProfScope ps(PROF_LOSS_RGB_BWD, stream);
    hipLaunchKernelGGL(photometric_bwd_kernel, grid, dim3(64), 0, stream, C, H, W, img, gt, mask, presence, maps,
                       upstream, lambda_dssim, dimg, FinishArgs{nullptr, 0, 1.0, nullptr});
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_photometric_loss_forward_backward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                           const float *presence, float lambda_dssim, float *maps, void *sums2,
                                           float *out3, const float *upstream, float *dimg, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !maps || !sums2 || !out3 || !dimg) return FSGS_ERR_INVALID;
  const int nblocks = ((W + SS_TILE - 1) / SS_TILE) * ((H + SS_TILE - 1) / SS_TILE) * C;
  dim3 gf(nblocks);
  float *partials = (float *)sums2;
  {
    ProfScope ps(PROF_LOSS_RGB_FWD, stream);
    hipLaunchKernelGGL(photometric_fwd_kernel, gf, dim3(256), kFwdDynLds, stream, C, H, W, img, gt, mask, presence, maps, partials);
  }
  dim3 gb(((W + ST_W - 1) / ST_W) * ((H + ST_RS - 1) / ST_RS) * C + 1);  // + 1: the finishing workgroup
  {
    ProfScope ps(PROF_LOSS_RGB_BWD, stream);
    hipLaunchKernelGGL(photometric_bwd_kernel, gb, dim3(64), 0, stream, C, H, W, img, gt, mask, presence, maps,
                       upstream, lambda_dssim, dimg, FinishArgs{partials, nblocks, (double)C * H * W, out3});
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_view_losses_forward_backward(int C, int H, int W, const float *img, const float *gt, const float *mask,
                                      const float *presence, float lambda_dssim, float *maps, void *photo_scratch,
                                      float *out3, const float *upstream, float *dimg, int n_patches, int box,
                                      const int64_t *patch_row0, const int64_t *patch_col0, const float *src,
                                      const float *tgt, void *pearson_scratch, float *coef, float *out2,
                                      const float *region_weight, float *grad_tgt, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (C <= 0 || H <= 0 || W <= 0 || !img || !gt || !maps || !photo_scratch || !out3 || !dimg) return FSGS_ERR_INVALID;
  if (n_patches < 0 || n_patches > 63 || !src || !tgt || !pearson_scratch || !coef || !out2 || !region_weight || !grad_tgt)
    return FSGS_ERR_INVALID;
  if (n_patches > 0 && (!patch_row0 || !patch_col0 || box <= 1 || box > H || box > W)) return FSGS_ERR_INVALID;
  const int n0 = (int)(((size_t)H * W + PE_CHUNK - 1) / PE_CHUNK);
  const int nb = n_patches > 0 ? (int)(((size_t)box * box + PE_CHUNK - 1) / PE_CHUNK) : 0;
  if (nb > PV_MAXNB) return FSGS_ERR_INVALID;  // (boxes beyond 128 x 128: the three-call route)
  PearsonView pv{box, n_patches + 1, patch_row0, patch_col0, src, tgt, (double *)pearson_scratch, n0, nb};
  const int ntiles = ((W + SS_TILE - 1) / SS_TILE) * ((H + SS_TILE - 1) / SS_TILE) * C;
  const int nstrips = ((W + ST_W - 1) / ST_W) * ((H + ST_RS - 1) / ST_RS) * C;
  const int n_pe = (n0 + n_patches * nb + 7) & ~7;
  const int n_pg = ((int)(((size_t)H * W + PV_PIX - 1) / PV_PIX) + 7) & ~7;
  float *partials = (float *)photo_scratch;
  {
    ProfScope ps(PROF_LOSS_RGB_FWD, stream);
    hipLaunchKernelGGL(view_losses_fwd_kernel, dim3(n_pe + ntiles), dim3(256), kFwdDynLds, stream, C, H, W, img, gt, mask,
                       presence, maps, partials, pv, n_pe);
  }
  {
    ProfScope ps(PROF_LOSS_RGB_BWD, stream);
    hipLaunchKernelGGL(view_losses_bwd_kernel, dim3(n_pg + nstrips + 1), dim3(64), 0, stream, C, H, W, img, gt, mask,
                       presence, maps, upstream, lambda_dssim, dimg, FinishArgs{partials, ntiles, (double)C * H * W, out3},
                       pv, region_weight, grad_tgt, coef, out2, n_pg);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

size_t fsgs_pearson_scratch_bytes(int H, int W, int n_patches, int box) {
  if (H <= 0 || W <= 0 || n_patches < 0 || box < 0) return 0;
  const size_t n0 = ((size_t)H * W + PE_CHUNK - 1) / PE_CHUNK, nb = ((size_t)box * box + PE_CHUNK - 1) / PE_CHUNK;
  return (n0 + (size_t)n_patches * nb) * 5 * sizeof(double) + 64;
}

int fsgs_pearson_forward(int H, int W, int n_patches, int box, const int64_t *patch_row0, const int64_t *patch_col0,
                         const float *src, const float *tgt, void *scratch, float *coef, float *out2,
                         fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (H <= 0 || W <= 0 || n_patches < 0 || n_patches > 64 || !src || !tgt || !scratch || !coef || !out2)
    return FSGS_ERR_INVALID;
  if (n_patches > 0 && (!patch_row0 || !patch_col0 || box <= 1 || box > H || box > W)) return FSGS_ERR_INVALID;
  const int nreg = n_patches + 1;
  const int n0 = (int)(((size_t)H * W + PE_CHUNK - 1) / PE_CHUNK);
  const int nb = n_patches > 0 ? (int)(((size_t)box * box + PE_CHUNK - 1) / PE_CHUNK) : 0;
  double *partials = (double *)scratch;  // caller-sized by fsgs_pearson_scratch_bytes; every slot is written
  {
    ProfScope ps(PROF_PEARSON, stream);
    dim3 grid(n0 > nb ? n0 : nb, nreg);
    hipLaunchKernelGGL(pearson_stats_kernel, grid, dim3(256), 0, stream, H, W, box, patch_row0, patch_col0, src, tgt,
                       partials, n0, nb);
    hipLaunchKernelGGL(pearson_finish_kernel, dim3(1), dim3(1024), 0, stream, H, W, box, nreg, partials, n0, nb, coef,
                       out2);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

int fsgs_pearson_backward(int H, int W, int n_patches, int box, const int64_t *patch_row0, const int64_t *patch_col0,
                          const float *src, const float *tgt, const float *coef, const float *region_weight,
                          int wrt_src, float *grad, fsgs_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (H <= 0 || W <= 0 || n_patches < 0 || n_patches > 64 || !src || !tgt || !coef || !region_weight || !grad)
    return FSGS_ERR_INVALID;
  {
    ProfScope ps(PROF_PEARSON, stream);
    int blocks = (int)(((size_t)H * W + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pearson_bwd_kernel, dim3(blocks), dim3(256), 0, stream, H, W, box, n_patches + 1, patch_row0,
                       patch_col0, src, tgt, coef, region_weight, wrt_src, grad);
  }
  FSGS_HIP(hipGetLastError());
  return FSGS_OK;
}

}  // extern "C"
