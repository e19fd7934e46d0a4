/*
 * raster_oracle.c -- CPU restatement of the tile-based differentiable
 * Gaussian-splat rasteriser that Free-SurGS calls through
 * `diff_gaussian_rasterization.GaussianRasterizer`.
 *
 * THIS FILE IS TEST INFRASTRUCTURE.  It is the checker for the HIP path, never
 * the product: only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it.
 *
 * PARITY UNPINNED: the reference tree does not contain the rasteriser source
 * (un-vendored third-party module, requirements.txt:26 /
 * .gitmodules:4-6, no commit pin) and holds no tests or golden vectors for it
 * (SURVEY.md s0.2, s8c).  The algorithm restated here is the published 3DGS
 * tile rasteriser (+ the depth-fork's third output) as itemised in SURVEY.md
 * s2.1 R1-R9 / Appendix A, anchored on the reference's call sites:
 *   gaussian_renderer/__init__.py:56-92   (two passes, 3 return values)
 *   scene/gaussian_model.py:277-333       (colors_precomp, no shs, no cov3D_precomp)
 *   scene/pose_optimizer.py:600-633       (settings tuple; matrices transposed)
 *   utils/general_utils.py:204-236        (quaternion -> R, Sigma = R S S^T R^T)
 * It is self-pinned by tests/test_oracle_raster.py: analytic single-Gaussian
 * image, occlusion order, the alpha/T thresholds, and fp64 central-difference
 * checks of every gradient (build with -DORACLE_F64); and the two statements of
 * its maths that ARE in the reference tree are pinned to vectors captured from
 * the imported reference (tests/golden/make_golden.py): the settings tuple of
 * PoseModel.setup_camera (camera.npz -> tests/test_golden_host.py) and
 * Sigma = R S S^T R^T of build_covariance_from_scaling_rotation
 * (covariance.npz -> oracle_state_cov3D).  The rasteriser itself stays unpinned.
 *
 * Build:  see oracle/Makefile  (liboracle_f32.so / liboracle_f64.so)
 *
 * Conventions: matrices are handed over exactly as the reference builds them
 * (scene/pose_optimizer.py:604,617-618): a [4,4] tensor that is the TRANSPOSE
 * of the maths matrix, i.e. m[4*k + j] = M[j][k].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifdef ORACLE_F64
typedef double real;
#define R_EXP exp
#define R_SQRT sqrt
#define R_CEIL ceil
#define R_FMAX fmax
#define R_FMIN fmin
#else
typedef float real;
#define R_EXP expf
#define R_SQRT sqrtf
#define R_CEIL ceilf
#define R_FMAX fmaxf
#define R_FMIN fminf
#endif

#define TILE 16
#define MAXC 8

typedef struct {
  int image_height;
  int image_width;
  int channels;          /* colour channels C (3 upstream; 6 = fused rgb+depth/sil/depth^2) */
  int reserved;
  real tanfovx, tanfovy;
  real scale_modifier;
  real bg[MAXC];
  real viewmatrix[16];   /* transposed storage, see header comment */
  real projmatrix[16];
} OracleCfg;

typedef struct {
  int P, C, H, W, gx, gy;
  int64_t R;             /* num_rendered = sum of tiles_touched */
  real *xy;              /* [P,2] pixel centre                         */
  real *conic_op;        /* [P,4] conic A,B,C + opacity                */
  real *depth;           /* [P]   view-space z                         */
  real *cov3D;           /* [P,6] upper triangle of Sigma              */
  real *tvec;            /* [P,3] view-space mean                      */
  real *kappa;           /* [P] a c / det of the dilated 2D covariance: how far det = a c - b^2 cancels.  The conic =
                            (c, -b, a) / det of a needle-shaped footprint carries ~6 kappa eps of relative error in ANY
                            fp32 evaluation (found by the 3000-seed soak: kappa = 3e3 -> the conic of the oracle's fp32
                            build and of the HIP path 1.1e-3 / 0.8e-3 off the fp64 value, to opposite sides) */
  int *radii;            /* [P]                                        */
  int *tiles;            /* [P]   tiles_touched                        */
  int *rect;             /* [P,4] minx,miny,maxx,maxy                  */
  uint32_t *plist;       /* [R]   Gaussian index, sorted by (tile, depth, index) */
  int *range;            /* [tiles,2]                                  */
  real *final_T;         /* [H*W]                                      */
  int *n_contrib;        /* [H*W]                                      */
} OracleState;

/* ---- small helpers ------------------------------------------------------ */

static inline void xform4x3(const real *m, const real *p, real *o) {
  /* o = M[:3,:] * [p;1] with m in transposed storage */
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static inline void xform4x4(const real *m, const real *p, real *o) {
  o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
  o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
  o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
  o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* Rotation matrix of an UN-normalised quaternion (r,x,y,z); row-major R[3*i+j].
 * Same polynomial as utils/general_utils.py:204-226 minus the normalisation,
 * which the caller has already applied (scene/gaussian_model.py:46,124). */
static inline void quat_to_R(const real *q, real *R) {
  real r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - r * z);     R[2] = 2 * (x * z + r * y);
  R[3] = 2 * (x * y + r * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - r * x);
  R[6] = 2 * (x * z - r * y);     R[7] = 2 * (y * z + r * x);     R[8] = 1 - 2 * (x * x + y * y);
}

/* Sigma = R diag(s)^2 R^T, upper triangle (xx,xy,xz,yy,yz,zz). SURVEY A.1-3 */
static inline void cov3d_from_scale_rot(const real *s, real mod, const real *q, real *c6) {
  real R[9];
  quat_to_R(q, R);
  real s0 = mod * s[0], s1 = mod * s[1], s2 = mod * s[2];
  real M[9]; /* M = R * S */
  for (int i = 0; i < 3; i++) { M[3*i] = R[3*i] * s0; M[3*i+1] = R[3*i+1] * s1; M[3*i+2] = R[3*i+2] * s2; }
  c6[0] = M[0]*M[0] + M[1]*M[1] + M[2]*M[2];
  c6[1] = M[0]*M[3] + M[1]*M[4] + M[2]*M[5];
  c6[2] = M[0]*M[6] + M[1]*M[7] + M[2]*M[8];
  c6[3] = M[3]*M[3] + M[4]*M[4] + M[5]*M[5];
  c6[4] = M[3]*M[6] + M[4]*M[7] + M[5]*M[8];
  c6[5] = M[6]*M[6] + M[7]*M[7] + M[8]*M[8];
}

/* ---- sort helper --------------------------------------------------------- */
typedef struct { uint32_t tile; real depth; uint32_t idx; } PairKey;
static int pair_cmp(const void *a, const void *b) {
  const PairKey *x = (const PairKey *)a, *y = (const PairKey *)b;
  if (x->tile != y->tile) return x->tile < y->tile ? -1 : 1;
  if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
  if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1; /* = stable radix sort of index-ordered emission */
  return 0;
}

/* ---- the DECISION thresholds of the algorithm ---------------------------------
 * Nominal values are the published constants.  The parity tests move them by a hair in both
 * directions (oracle_set_thresholds) to find every output element whose value hinges on a comparison that two
 * correct fp32 implementations may resolve differently ("flip attribution", tests/util.py): an element that is
 * identical under the loose and the tight setting cannot owe a HIP-vs-oracle difference to such a flip.
 * Process-global, like the thread count: the tests set, run, and restore. */
typedef struct {
  double alpha_min;     /* 1/255: skip below                       */
  double alpha_max;     /* 0.99: clamp                             */
  double T_min;         /* 1e-4: stop when T (1 - alpha) < T_min   */
  double power_max;     /* 0: skip when power > power_max          */
  double radius_scale;  /* 1: radius = ceil(3 sigma * scale)       */
  double near_plane;    /* 0.2: cull t_z <= near_plane             */
  double rect_shift;    /* 0: added to (p -+ radius) before the division by the tile size */
  double alpha_cond;    /* 0: alpha_min is additionally scaled by (1 - alpha_cond * sum |terms of the exponent|): the
                           exponent's three products cancel on elongated footprints and carry rounding in proportion
                           to their magnitudes, not to their sum */
  double alpha_kappa;   /* 0: the alpha_cond term is additionally multiplied by (1 + alpha_kappa * kappa_g): the conic
                           itself is only good to ~6 kappa eps (OracleState.kappa), and the exponent inherits that */
} OracleThresholds;
static OracleThresholds g_thr = {1.0 / 255.0, 0.99, 0.0001, 0.0, 1.0, 0.2, 0.0, 0.0, 0.0};
void oracle_set_thresholds(double alpha_min, double alpha_max, double T_min, double power_max, double radius_scale,
                           double near_plane, double rect_shift, double alpha_cond, double alpha_kappa) {
  g_thr.alpha_min = alpha_min; g_thr.alpha_max = alpha_max; g_thr.T_min = T_min; g_thr.power_max = power_max;
  g_thr.radius_scale = radius_scale; g_thr.near_plane = near_plane; g_thr.rect_shift = rect_shift;
  g_thr.alpha_cond = alpha_cond; g_thr.alpha_kappa = alpha_kappa;
}
void oracle_reset_thresholds(void) { oracle_set_thresholds(1.0 / 255.0, 0.99, 0.0001, 0.0, 1.0, 0.2, 0.0, 0.0, 0.0); }
/* The other discontinuity: the depth ORDER inside a tile.  Two implementations whose view-space depths differ in the
 * last bit (different glue in front of the rasteriser) may sort a near-tie either way.  The tests find the pairs of
 * list neighbours whose depths are within a few ulp and hand over a per-Gaussian relative shift of the SORT KEY only
 * (the depth that is blended is untouched) that swaps exactly those pairs.  NULL = none. */
static const double *g_depth_shift = 0;
static int g_depth_shift_n = 0;
void oracle_set_depth_shift(const double *rel_shift, int n) { g_depth_shift = rel_shift; g_depth_shift_n = n; }
/* nominal: exactly (real)(1/255); perturbed: moved further in proportion to the exponent's conditioning */
#define THR_ALPHA_MIN_AT(co, dx, dy, kap)                                                                             \
  (g_thr.alpha_cond == 0.0 ? (real)g_thr.alpha_min                                                                    \
                           : (real)(g_thr.alpha_min * (1.0 - g_thr.alpha_cond * (1.0 + g_thr.alpha_kappa * (double)(kap)) *   \
                                                                    (0.5 * (fabs((double)((co)[0] * (dx) * (dx))) +   \
                                                                            fabs((double)((co)[2] * (dy) * (dy)))) +  \
                                                                     fabs((double)((co)[1] * (dx) * (dy)))))))
#define THR_ALPHA_MIN ((real)g_thr.alpha_min)
#define THR_ALPHA_MAX ((real)g_thr.alpha_max)
#define THR_T_MIN ((real)g_thr.T_min)
#define THR_POWER_MAX ((real)g_thr.power_max)

/* ---- API ----------------------------------------------------------------- */

void oracle_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
int oracle_real_bytes(void) { return (int)sizeof(real); }

void oracle_raster_free(OracleState *st) {
  if (!st) return;
  free(st->xy); free(st->conic_op); free(st->depth); free(st->cov3D); free(st->tvec); free(st->kappa);
  free(st->radii); free(st->tiles); free(st->rect); free(st->plist); free(st->range);
  free(st->final_T); free(st->n_contrib);
  free(st);
}

int64_t oracle_state_num_rendered(const OracleState *st) { return st->R; }
const real *oracle_state_xy(const OracleState *st) { return st->xy; }
const real *oracle_state_conic_op(const OracleState *st) { return st->conic_op; }
const real *oracle_state_depth(const OracleState *st) { return st->depth; }
/* [P,6] upper triangle (xx, xy, xz, yy, yz, zz) of Sigma = R S^2 R^T for every Gaussian in front of the near plane --
 * the order strip_symmetric keeps (utils/general_utils.py:191-202); pinned against the reference's own
 * build_covariance_from_scaling_rotation (scene/gaussian_model.py:32-36) by tests/golden/covariance.npz */
const real *oracle_state_cov3D(const OracleState *st) { return st->cov3D; }
const real *oracle_state_kappa(const OracleState *st) { return st->kappa; }
const real *oracle_state_final_T(const OracleState *st) { return st->final_T; }
const int *oracle_state_n_contrib(const OracleState *st) { return st->n_contrib; }
const int *oracle_state_tiles(const OracleState *st) { return st->tiles; }
const uint32_t *oracle_state_point_list(const OracleState *st) { return st->plist; }
const int *oracle_state_ranges(const OracleState *st) { return st->range; }

/*
 * Forward.  SURVEY.md Appendix A.1 (preprocess), A.2 (binning), A.3 (blend).
 * out_color [C,H,W] planar, out_depth [H,W] (the depth-fork's third output,
 * discarded by gaussian_renderer/__init__.py:68-70), radii [P].
 */
OracleState *oracle_raster_forward(const OracleCfg *cfg, int P, const real *means3D, const real *colors,
                                   const real *opac, const real *scales, const real *rots,
                                   real *out_color, real *out_depth, int *out_radii) {
  const int H = cfg->image_height, W = cfg->image_width, C = cfg->channels;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  OracleState *st = (OracleState *)calloc(1, sizeof(OracleState));
  st->P = P; st->C = C; st->H = H; st->W = W; st->gx = gx; st->gy = gy;
  size_t Pn = P > 0 ? (size_t)P : 1;
  st->xy = (real *)calloc(Pn * 2, sizeof(real));
  st->conic_op = (real *)calloc(Pn * 4, sizeof(real));
  st->depth = (real *)calloc(Pn, sizeof(real));
  st->cov3D = (real *)calloc(Pn * 6, sizeof(real));
  st->tvec = (real *)calloc(Pn * 3, sizeof(real));
  st->kappa = (real *)calloc(Pn, sizeof(real));
  st->radii = (int *)calloc(Pn, sizeof(int));
  st->tiles = (int *)calloc(Pn, sizeof(int));
  st->rect = (int *)calloc(Pn * 4, sizeof(int));
  st->range = (int *)calloc((size_t)gx * gy * 2 + 2, sizeof(int));
  st->final_T = (real *)calloc((size_t)H * W + 1, sizeof(real));
  st->n_contrib = (int *)calloc((size_t)H * W + 1, sizeof(int));

  const real fx = W / (2 * cfg->tanfovx), fy = H / (2 * cfg->tanfovy);
  const real *V = cfg->viewmatrix, *PM = cfg->projmatrix;

  /* ---- A.1 per-Gaussian preprocess ---- */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    st->radii[i] = 0; st->tiles[i] = 0;
    real t[3];
    xform4x3(V, means3D + 3 * i, t);
    if (t[2] <= (real)g_thr.near_plane) continue;          /* near-plane cull (0.2) */
    real h[4];
    xform4x4(PM, means3D + 3 * i, h);
    real pw = 1 / (h[3] + (real)0.0000001);
    real ndcx = h[0] * pw, ndcy = h[1] * pw;
    real *c6 = st->cov3D + 6 * i;
    cov3d_from_scale_rot(scales + 3 * i, cfg->scale_modifier, rots + 4 * i, c6);
    /* EWA 2D covariance */
    real limx = (real)1.3 * cfg->tanfovx, limy = (real)1.3 * cfg->tanfovy;
    real txtz = t[0] / t[2], tytz = t[1] / t[2];
    real tx = R_FMIN(limx, R_FMAX(-limx, txtz)) * t[2];
    real ty = R_FMIN(limy, R_FMAX(-limy, tytz)) * t[2];
    real tz = t[2];
    real J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    real J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    /* Wv = V[:3,:3] (maths, row-major): Wv[r][c] = V[4*c + r] */
    real M0[3], M1[3]; /* rows of M = J * Wv */
    for (int c = 0; c < 3; c++) {
      M0[c] = J00 * V[4 * c + 0] + J02 * V[4 * c + 2];
      M1[c] = J11 * V[4 * c + 1] + J12 * V[4 * c + 2];
    }
    real S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    real SM0[3], SM1[3];
    for (int r = 0; r < 3; r++) {
      SM0[r] = S[3*r] * M0[0] + S[3*r+1] * M0[1] + S[3*r+2] * M0[2];
      SM1[r] = S[3*r] * M1[0] + S[3*r+1] * M1[1] + S[3*r+2] * M1[2];
    }
    real a = M0[0]*SM0[0] + M0[1]*SM0[1] + M0[2]*SM0[2] + (real)0.3;
    real b = M0[0]*SM1[0] + M0[1]*SM1[1] + M0[2]*SM1[2];
    real c = M1[0]*SM1[0] + M1[1]*SM1[1] + M1[2]*SM1[2] + (real)0.3;
    real det = a * c - b * b;
    if (det == 0) continue;
    real det_inv = 1 / det;
    real mid = (real)0.5 * (a + c);
    real sq = R_SQRT(R_FMAX((real)0.1, mid * mid - det));
    real lam = R_FMAX(mid + sq, mid - sq);
    int radius = g_thr.radius_scale == 1.0 ? (int)R_CEIL(3 * R_SQRT(lam))
                                           : (int)ceil(3 * sqrt((double)lam) * g_thr.radius_scale);
    real px = ((ndcx + 1) * W - 1) * (real)0.5;
    real py = ((ndcy + 1) * H - 1) * (real)0.5;
    const real rs = (real)g_thr.rect_shift; /* 0 nominally; widens (> 0) or narrows (< 0) the rect by a hair */
    int minx = (int)((px - radius - rs) / TILE), miny = (int)((py - radius - rs) / TILE);
    int maxx = (int)((px + radius + TILE - 1 + rs) / TILE), maxy = (int)((py + radius + TILE - 1 + rs) / TILE);
    minx = minx < 0 ? 0 : (minx > gx ? gx : minx); maxx = maxx < 0 ? 0 : (maxx > gx ? gx : maxx);
    miny = miny < 0 ? 0 : (miny > gy ? gy : miny); maxy = maxy < 0 ? 0 : (maxy > gy ? gy : maxy);
    int area = (maxx - minx) * (maxy - miny);
    if (area <= 0) continue;
    st->depth[i] = t[2];
    st->tvec[3*i] = t[0]; st->tvec[3*i+1] = t[1]; st->tvec[3*i+2] = t[2];
    st->radii[i] = radius;
    st->xy[2*i] = px; st->xy[2*i+1] = py;
    st->conic_op[4*i] = c * det_inv; st->conic_op[4*i+1] = -b * det_inv;
    st->conic_op[4*i+2] = a * det_inv; st->conic_op[4*i+3] = opac[i];
    st->kappa[i] = (a * c) * det_inv;
    st->rect[4*i] = minx; st->rect[4*i+1] = miny; st->rect[4*i+2] = maxx; st->rect[4*i+3] = maxy;
    st->tiles[i] = area;
  }
  for (int i = 0; i < P; i++) out_radii[i] = st->radii[i];

  /* ---- A.2 binning: emit (tile, depth, idx), sort, tile ranges ---- */
  int64_t R = 0;
  for (int i = 0; i < P; i++) R += st->tiles[i];
  st->R = R;
  PairKey *keys = (PairKey *)malloc(sizeof(PairKey) * (size_t)(R > 0 ? R : 1));
  int64_t off = 0;
  for (int i = 0; i < P; i++) {
    if (st->tiles[i] == 0) continue;
    const int *rc = st->rect + 4 * i;
    for (int y = rc[1]; y < rc[3]; y++)
      for (int x = rc[0]; x < rc[2]; x++) {
        keys[off].tile = (uint32_t)(y * gx + x); keys[off].idx = (uint32_t)i;
        keys[off].depth = (g_depth_shift && i < g_depth_shift_n) ? (real)((double)st->depth[i] * (1.0 + g_depth_shift[i]))
                                                                 : st->depth[i];
        off++;
      }
  }
  qsort(keys, (size_t)R, sizeof(PairKey), pair_cmp);
  st->plist = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(R > 0 ? R : 1));
  for (int64_t k = 0; k < R; k++) {
    st->plist[k] = keys[k].idx;
    uint32_t tcur = keys[k].tile;
    if (k == 0 || keys[k - 1].tile != tcur) st->range[2 * tcur] = (int)k;
    if (k == R - 1 || keys[k + 1].tile != tcur) st->range[2 * tcur + 1] = (int)(k + 1);
  }
  free(keys);

  /* ---- A.3 per-pixel front-to-back blend ---- */
  const int ntiles = gx * gy;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < ntiles; tile++) {
    int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
    int r0 = st->range[2 * tile], r1 = st->range[2 * tile + 1];
    for (int py_ = ty0; py_ < ty0 + TILE && py_ < H; py_++)
      for (int px_ = tx0; px_ < tx0 + TILE && px_ < W; px_++) {
        real T = 1, Cacc[MAXC] = {0}, D = 0;
        int contributor = 0, last = 0;
        for (int k = r0; k < r1; k++) {
          uint32_t g = st->plist[k];
          contributor++;
          real dx = st->xy[2*g] - (real)px_, dy = st->xy[2*g+1] - (real)py_;
          const real *co = st->conic_op + 4 * g;
          real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > THR_POWER_MAX) continue;
          real alpha = R_FMIN(THR_ALPHA_MAX, co[3] * R_EXP(power));
          if (alpha < THR_ALPHA_MIN_AT(co, dx, dy, st->kappa[g])) continue;
          real test_T = T * (1 - alpha);
          if (test_T < THR_T_MIN) break;
          real w = alpha * T;
          for (int ch = 0; ch < C; ch++) Cacc[ch] += colors[(size_t)g * C + ch] * w;
          D += st->depth[g] * w;
          T = test_T;
          last = contributor;
        }
        size_t pix = (size_t)py_ * W + px_;
        st->final_T[pix] = T;
        st->n_contrib[pix] = last;
        for (int ch = 0; ch < C; ch++) out_color[(size_t)ch * H * W + pix] = Cacc[ch] + T * cfg->bg[ch];
        out_depth[pix] = D;
      }
  }
  return st;
}

static inline void atomic_add(real *p, real v) {
#pragma omp atomic
  *p += v;
}

/*
 * Backward.  SURVEY.md Appendix A.4 (blend), A.5 (conic->cov2D), A.6 (cov2D->Sigma,t),
 * A.7 (projection), A.8 (Sigma->scale,quaternion).  dL_dout_color [C,H,W].
 * Outputs (all caller-allocated, overwritten): dmeans2D [P,3] (z = 0, NDC-scaled as
 * A.4 says), dcolors [P,C], dopac [P], dmeans3D [P,3], dscales [P,3], drots [P,4].
 * The gradient of the depth-fork's third output is not propagated (a1 note ii).
 */
void oracle_raster_backward(const OracleCfg *cfg, const OracleState *st, const real *means3D, const real *colors,
                            const real *scales, const real *rots, const real *dL_dcolor,
                            real *dmeans2D, real *dcolors, real *dopac, real *dmeans3D,
                            real *dscales, real *drots) {
  const int P = st->P, C = st->C, H = st->H, W = st->W, gx = st->gx, gy = st->gy;
  real *dconic = (real *)calloc((size_t)(P > 0 ? P : 1) * 3, sizeof(real)); /* true partials gA,gB,gC */
  memset(dmeans2D, 0, sizeof(real) * 3 * (size_t)P);
  memset(dcolors, 0, sizeof(real) * (size_t)C * P);
  memset(dopac, 0, sizeof(real) * (size_t)P);
  memset(dmeans3D, 0, sizeof(real) * 3 * (size_t)P);
  memset(dscales, 0, sizeof(real) * 3 * (size_t)P);
  memset(drots, 0, sizeof(real) * 4 * (size_t)P);

  /* ---- A.4 blend backward: back-to-front replay per pixel ---- */
  const int ntiles = gx * gy;
#pragma omp parallel for schedule(dynamic, 4)
  for (int tile = 0; tile < ntiles; tile++) {
    int tx0 = (tile % gx) * TILE, ty0 = (tile / gx) * TILE;
    int r0 = st->range[2 * tile];
    for (int py_ = ty0; py_ < ty0 + TILE && py_ < H; py_++)
      for (int px_ = tx0; px_ < tx0 + TILE && px_ < W; px_++) {
        size_t pix = (size_t)py_ * W + px_;
        const real T_final = st->final_T[pix];
        real T = T_final;
        int last = st->n_contrib[pix];
        real g[MAXC], acc[MAXC] = {0}, cprev[MAXC] = {0};
        real aprev = 0, bgdot = 0;
        for (int ch = 0; ch < C; ch++) { g[ch] = dL_dcolor[(size_t)ch * H * W + pix]; bgdot += cfg->bg[ch] * g[ch]; }
        for (int k = r0 + last - 1; k >= r0; k--) {
          uint32_t id = st->plist[k];
          real dx = st->xy[2*id] - (real)px_, dy = st->xy[2*id+1] - (real)py_;
          const real *co = st->conic_op + 4 * id;
          real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
          if (power > THR_POWER_MAX) continue;
          real G = R_EXP(power);
          real alpha = R_FMIN(THR_ALPHA_MAX, co[3] * G);
          if (alpha < THR_ALPHA_MIN_AT(co, dx, dy, st->kappa[id])) continue;
          T = T / (1 - alpha);
          real wgt = alpha * T, dL_dalpha = 0;
          for (int ch = 0; ch < C; ch++) {
            real cc = colors[(size_t)id * C + ch];
            acc[ch] = aprev * cprev[ch] + (1 - aprev) * acc[ch];
            cprev[ch] = cc;
            dL_dalpha += (cc - acc[ch]) * g[ch];
            atomic_add(&dcolors[(size_t)id * C + ch], wgt * g[ch]);
          }
          dL_dalpha *= T;
          aprev = alpha;
          dL_dalpha += (-T_final / (1 - alpha)) * bgdot;
          real dL_dG = co[3] * dL_dalpha;
          real gdx = G * dx, gdy = G * dy;
          real dG_ddx = -gdx * co[0] - gdy * co[1];
          real dG_ddy = -gdy * co[2] - gdx * co[1];
          atomic_add(&dmeans2D[3 * id + 0], dL_dG * dG_ddx * (real)0.5 * W);
          atomic_add(&dmeans2D[3 * id + 1], dL_dG * dG_ddy * (real)0.5 * H);
          atomic_add(&dconic[3 * id + 0], (real)-0.5 * gdx * dx * dL_dG);
          atomic_add(&dconic[3 * id + 1], -gdx * dy * dL_dG);
          atomic_add(&dconic[3 * id + 2], (real)-0.5 * gdy * dy * dL_dG);
          atomic_add(&dopac[id], G * dL_dalpha);
        }
      }
  }

  /* ---- A.5 - A.8 per-Gaussian chain ---- */
  const real fx = W / (2 * cfg->tanfovx), fy = H / (2 * cfg->tanfovy);
  const real *V = cfg->viewmatrix, *PM = cfg->projmatrix;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < P; i++) {
    if (st->radii[i] <= 0) continue;
    const real *c6 = st->cov3D + 6 * i;
    const real *m = means3D + 3 * i;
    real t[3];
    xform4x3(V, m, t);
    real limx = (real)1.3 * cfg->tanfovx, limy = (real)1.3 * cfg->tanfovy;
    real txtz = t[0] / t[2], tytz = t[1] / t[2];
    real tx = R_FMIN(limx, R_FMAX(-limx, txtz)) * t[2];
    real ty = R_FMIN(limy, R_FMAX(-limy, tytz)) * t[2];
    real chix = (txtz < -limx || txtz > limx) ? 0 : 1;
    real chiy = (tytz < -limy || tytz > limy) ? 0 : 1;
    real tz = t[2];
    real J00 = fx / tz, J02 = -(fx * tx) / (tz * tz);
    real J11 = fy / tz, J12 = -(fy * ty) / (tz * tz);
    real Wv[9]; /* row-major view rotation */
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) Wv[3*r+c] = V[4*c + r];
    real M0[3], M1[3];
    for (int c = 0; c < 3; c++) {
      M0[c] = J00 * Wv[c] + J02 * Wv[6 + c];
      M1[c] = J11 * Wv[3 + c] + J12 * Wv[6 + c];
    }
    real S[9] = {c6[0], c6[1], c6[2], c6[1], c6[3], c6[4], c6[2], c6[4], c6[5]};
    real SM0[3], SM1[3];
    for (int r = 0; r < 3; r++) {
      SM0[r] = S[3*r] * M0[0] + S[3*r+1] * M0[1] + S[3*r+2] * M0[2];
      SM1[r] = S[3*r] * M1[0] + S[3*r+1] * M1[1] + S[3*r+2] * M1[2];
    }
    real a = M0[0]*SM0[0] + M0[1]*SM0[1] + M0[2]*SM0[2] + (real)0.3;
    real b = M0[0]*SM1[0] + M0[1]*SM1[1] + M0[2]*SM1[2];
    real c = M1[0]*SM1[0] + M1[1]*SM1[1] + M1[2]*SM1[2] + (real)0.3;
    /* A.5 */
    real D = a * c - b * b;
    real D2 = 1 / (D * D + (real)0.0000001);
    real gA = dconic[3*i], gB = dconic[3*i+1], gC = dconic[3*i+2];
    real dL_da = D2 * (-c * c * gA + b * c * gB + (D - a * c) * gC);
    real dL_dc = D2 * (-a * a * gC + a * b * gB + (D - a * c) * gA);
    real dL_db = D2 * (2 * b * c * gA - (D + 2 * b * b) * gB + 2 * a * b * gC);
    /* A.6: dL/dSigma = M^T G2 M with G2 = [[da, db/2],[db/2, dc]] */
    real hb = (real)0.5 * dL_db;
    real G3[9];
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
      G3[3*r+cc] = M0[r] * (dL_da * M0[cc] + hb * M1[cc]) + M1[r] * (hb * M0[cc] + dL_dc * M1[cc]);
    /* dL/dM = 2 G2 M Sigma  (rows) */
    real dM0[3], dM1[3];
    for (int cc = 0; cc < 3; cc++) {
      dM0[cc] = 2 * (dL_da * SM0[cc] + hb * SM1[cc]);
      dM1[cc] = 2 * (hb * SM0[cc] + dL_dc * SM1[cc]);
    }
    /* dL/dJ = dL/dM * Wv^T */
    real dJ00 = dM0[0]*Wv[0] + dM0[1]*Wv[1] + dM0[2]*Wv[2];
    real dJ02 = dM0[0]*Wv[6] + dM0[1]*Wv[7] + dM0[2]*Wv[8];
    real dJ11 = dM1[0]*Wv[3] + dM1[1]*Wv[4] + dM1[2]*Wv[5];
    real dJ12 = dM1[0]*Wv[6] + dM1[1]*Wv[7] + dM1[2]*Wv[8];
    real itz = 1 / tz, itz2 = itz * itz, itz3 = itz2 * itz;
    real dtx = chix * -fx * itz2 * dJ02;
    real dty = chiy * -fy * itz2 * dJ12;
    real dtz = -fx * itz2 * dJ00 - fy * itz2 * dJ11 + (2 * fx * tx) * itz3 * dJ02 + (2 * fy * ty) * itz3 * dJ12;
    /* dL/dmean (cov part) = Wv^T dt */
    real dm[3];
    for (int cc = 0; cc < 3; cc++) dm[cc] = Wv[cc] * dtx + Wv[3 + cc] * dty + Wv[6 + cc] * dtz;
    /* A.7 projection */
    real h[4];
    xform4x4(PM, m, h);
    real mw = 1 / (h[3] + (real)0.0000001);
    real mul1 = h[0] * mw * mw, mul2 = h[1] * mw * mw;
    real g2x = dmeans2D[3*i], g2y = dmeans2D[3*i+1];
    for (int k = 0; k < 3; k++) {
      real P0k = PM[4*k + 0], P1k = PM[4*k + 1], P3k = PM[4*k + 3];
      dm[k] += (P0k * mw - P3k * mul1) * g2x + (P1k * mw - P3k * mul2) * g2y;
    }
    dmeans3D[3*i] = dm[0]; dmeans3D[3*i+1] = dm[1]; dmeans3D[3*i+2] = dm[2];
    /* A.8 Sigma -> scale, quaternion */
    real Rm[9];
    quat_to_R(rots + 4 * i, Rm);
    real mod = cfg->scale_modifier;
    real s[3] = {mod * scales[3*i], mod * scales[3*i+1], mod * scales[3*i+2]};
    real GR[9]; /* G3 * R */
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++)
      GR[3*r+cc] = G3[3*r] * Rm[cc] + G3[3*r+1] * Rm[3+cc] + G3[3*r+2] * Rm[6+cc];
    for (int j = 0; j < 3; j++) {
      real rtgr = Rm[j] * GR[j] + Rm[3+j] * GR[3+j] + Rm[6+j] * GR[6+j];
      dscales[3*i+j] = 2 * s[j] * rtgr * mod;
    }
    real dR[9]; /* dL/dR = 2 G3 R S^2 */
    for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) dR[3*r+cc] = 2 * GR[3*r+cc] * s[cc] * s[cc];
    real qr = rots[4*i], qx = rots[4*i+1], qy = rots[4*i+2], qz = rots[4*i+3];
    drots[4*i+0] = 2 * (-qz * dR[1] + qy * dR[2] + qz * dR[3] - qx * dR[5] - qy * dR[6] + qx * dR[7]);
    drots[4*i+1] = 2 * (qy * dR[1] + qz * dR[2] + qy * dR[3] - 2 * qx * dR[4] - qr * dR[5] + qz * dR[6] + qr * dR[7] - 2 * qx * dR[8]);
    drots[4*i+2] = 2 * (-2 * qy * dR[0] + qx * dR[1] + qr * dR[2] + qx * dR[3] + qz * dR[5] - qr * dR[6] + qz * dR[7] - 2 * qy * dR[8]);
    drots[4*i+3] = 2 * (-2 * qz * dR[0] - qr * dR[1] + qx * dR[2] + qr * dR[3] - 2 * qz * dR[4] + qy * dR[5] + qx * dR[6] + qy * dR[7]);
  }
  free(dconic);
}
