/*
 * knn_oracle.c -- CPU restatement of simple-knn's distCUDA2
 * (submodules/simple-knn/simple_knn.cu:147-183, spatial.cu:15-26):
 * for every point, the mean of the squared distances to its 3 nearest
 * neighbours.  The reference's Morton-box search is an exact search
 * (simple_knn.cu:163-181 only skips boxes that cannot beat the current
 * 3rd-best), so the oracle is the brute-force O(P^2) definition:
 *   - "self" is excluded by POSITION, not by value (simple_knn.cu:158,177):
 *     duplicated points give distance 0;
 *   - fewer than 3 neighbours leaves FLT_MAX terms in the mean
 *     (simple_knn.cu:154,182) -> +inf in fp32.
 *
 * THIS FILE IS TEST INFRASTRUCTURE (checker only; see raster_oracle.c header).
 * Pinning: the reference holds no tests for this path; the known-answer cases
 * of SURVEY.md s8c (regular grid -> h^2, duplicates -> 0, P<4) are in
 * tests/test_oracle_knn.py.  The CUDA source cannot be compiled here (nvcc
 * absent, cub/thrust CUDA-only), so there is no oracle/_ref for it.
 */
#include <float.h>
#include <stdint.h>

#ifdef ORACLE_F64
typedef double real;
#define REAL_MAX DBL_MAX
#else
typedef float real;
#define REAL_MAX FLT_MAX
#endif

void oracle_knn_meandist2(int P, const real *pts, real *out) {
#pragma omp parallel for schedule(dynamic, 64)
  for (int i = 0; i < P; i++) {
    real px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    real b0 = REAL_MAX, b1 = REAL_MAX, b2 = REAL_MAX;
    for (int j = 0; j < P; j++) {
      if (j == i) continue;
      real dx = pts[3 * j] - px, dy = pts[3 * j + 1] - py, dz = pts[3 * j + 2] - pz;
      real d = dx * dx + dy * dy + dz * dz;
      if (d < b2) { /* insertion into the sorted 3-best, simple_knn.cu:132-145 */
        if (d < b0) { b2 = b1; b1 = b0; b0 = d; }
        else if (d < b1) { b2 = b1; b1 = d; }
        else b2 = d;
      }
    }
    out[i] = (b0 + b1 + b2) / (real)3.0;
  }
}
