/* oracle/oracle.c -- TEST INFRASTRUCTURE ONLY.  NOT the product path.
 *
 * A plain-C, single-threaded CPU restatement of the reference hot path
 * (r-barnes/richdem v2.2.11): Priority-Flood fill, D8 flow directions, Barnes
 * flat resolution, D8 flow accumulation (d8_flow_accum) and FM_D8 + generic
 * FlowAccumulation (FA_D8).  Every function cites the reference file:line it
 * follows (relative to /root/reference/include/richdem/).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product (librdgpu.so)
 * never links, loads or falls back to it.
 *
 * PARITY PINNING: this restatement is checked, cell for cell, against
 *   (1) the reference's own golden vectors (tests/depressions/testdem1.*,
 *       tests/flow_accum/*.d8/.out -- committed under tests/golden/), and
 *   (2) the unmodified reference headers compiled in place into
 *       oracle/_ref/libref.so (oracle/ref_wrap.cpp) on seeded random inputs,
 * by tests/test_oracle_pinning.py.  Flat resolution, d8_flow_directions and
 * FA_D8 have no golden file in the reference; they are pinned by (2) and by the
 * fixtures generated from (2) in tests/golden/ (tests/golden/make_golden.py).
 *
 * Build: gcc -O2 -std=c11 -shared -fPIC oracle.c -o liboracle.so  (oracle/Makefile)
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* D8 neighbour numbering 234/105/876, common/constants.hpp:44-45 */
static const int D8X[9] = {0, -1, -1, 0, 1, 1, 1, 0, -1};
static const int D8Y[9] = {0, 0, -1, -1, -1, 0, 1, 1, 1};
/* D4 numbering, common/constants.hpp:54-55 */
static const int D4X[5] = {0, -1, 0, 1, 0};
static const int D4Y[5] = {0, 0, -1, 0, 1};

#define T uint8_t
#define SUF u8
#include "oracle_impl.h"
#undef T
#undef SUF
#define T int8_t
#define SUF i8
#include "oracle_impl.h"
#undef T
#undef SUF
#define T int16_t
#define SUF i16
#include "oracle_impl.h"
#undef T
#undef SUF
#define T uint16_t
#define SUF u16
#include "oracle_impl.h"
#undef T
#undef SUF
#define T int32_t
#define SUF i32
#include "oracle_impl.h"
#undef T
#undef SUF
#define T uint32_t
#define SUF u32
#include "oracle_impl.h"
#undef T
#undef SUF
#define T float
#define SUF f32
#define ORC_IS_FLOAT
#define ORC_NEXTUP(v) nextafterf((v), INFINITY)
#include "oracle_impl.h"
#undef ORC_NEXTUP
#undef T
#undef SUF
#define T double
#define SUF f64
#define ORC_NEXTUP(v) nextafter((v), (double)INFINITY)
#include "oracle_impl.h"
#undef ORC_NEXTUP
#undef ORC_IS_FLOAT
#undef T
#undef SUF
#define T int64_t
#define SUF i64
#include "oracle_impl.h"
#undef T
#undef SUF
#define T uint64_t
#define SUF u64
#include "oracle_impl.h"
#undef T
#undef SUF

/* d8_masked_FlowDir, flats/flat_resolution.hpp:42-65, applied as in
 * d8_flow_flats :96-116 (interior cells whose direction is NO_FLOW). */
static void orc_d8_flow_flats(const int32_t *mask, const int32_t *labels, int w, int h, uint8_t *dirs) {
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      size_t i = (size_t)y * w + x;
      if (mask[i] == -1) continue;          /* flat_mask.noData() == -1, :470, :110 */
      if (dirs[i] != 0) continue;           /* :112 */
      int minimum = mask[i], flowdir = 0;
      for (int n = 1; n <= 8; n++) {
        size_t ni = (size_t)(y + D8Y[n]) * w + (x + D8X[n]);
        if (labels[ni] != labels[i]) continue;                       /* :56-57 */
        if (mask[ni] < minimum || (mask[ni] == minimum && flowdir > 0 && flowdir % 2 == 0 && n % 2 == 1)) {
          minimum = mask[ni];
          flowdir = n;
        }
      }
      dirs[i] = (uint8_t)flowdir;
    }
}

void orc_d8_flow_flats_apply(const int32_t *mask, const int32_t *labels, int w, int h, uint8_t *dirs) {
  orc_d8_flow_flats(mask, labels, w, h, dirs);
}

/* d8_flow_accum, methods/d8_methods.hpp:47-139.  FIFO Kahn order. */
#define ORC_D8_ACCUM(SUF, A)                                                                    \
  void orc_d8_flow_accum_##SUF(const uint8_t *dirs, uint8_t nodata, int w, int h, A *area) {    \
    size_t N = (size_t)w * h;                                                                   \
    int8_t *dep = (int8_t *)calloc(N, 1);                                     /* :60 */         \
    for (size_t i = 0; i < N; i++) area[i] = 0;                               /* :63 */         \
    for (int y = 0; y < h; y++)                                               /* :69-91 */      \
      for (int x = 0; x < w; x++) {                                                             \
        size_t i = (size_t)y * w + x;                                                           \
        if (dirs[i] == nodata) { area[i] = (A)-1; continue; }                                   \
        int n = dirs[i];                                                                        \
        if (n == 0) continue;                                                                   \
        int nx = x + D8X[n], ny = y + D8Y[n];                                                   \
        if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;                                   \
        ++dep[(size_t)ny * w + nx];                                                             \
      }                                                                                         \
    int32_t *q = (int32_t *)malloc(N * 4);                                                      \
    size_t qh = 0, qt = 0;                                                                      \
    for (size_t i = 0; i < N; i++)                                            /* :96-99 */      \
      if (dep[i] == 0 && dirs[i] != nodata) q[qt++] = (int32_t)i;                               \
    while (qh < qt) {                                                         /* :104-131 */    \
      size_t c = (size_t)q[qh++];                                                               \
      area[c]++;                                                                                \
      int n = dirs[c];                                                                          \
      if (n == 0) continue;                                                                     \
      int nx = (int)(c % w) + D8X[n], ny = (int)(c / w) + D8Y[n];                               \
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;                                     \
      size_t ni = (size_t)ny * w + nx;                                                          \
      if (dirs[ni] == nodata) continue;                                                         \
      area[ni] += area[c];                                                                      \
      if (--dep[ni] == 0) q[qt++] = (int32_t)ni;                                                \
    }                                                                                           \
    free(dep); free(q);                                                                         \
  }
ORC_D8_ACCUM(i32, int32_t)
ORC_D8_ACCUM(f32, float)
ORC_D8_ACCUM(f64, double)

/* FlowAccumulation, methods/flow_accumulation_generic.hpp:33-100.
 * accum is in/out (pre-loaded with per-cell generated flow). */
void orc_flow_accumulation_f64(const float *props9, int w, int h, double *accum) {
  size_t N = (size_t)w * h;
  const int nshift[9] = {0, -1, -w - 1, -w, -w + 1, 1, w + 1, w, w - 1};   /* Array2D.hpp:858 */
  int8_t *deps = (int8_t *)calloc(N, 1);
  for (int y = 1; y < h - 1; y++)                                        /* :48-58 */
    for (int x = 1; x < w - 1; x++) {
      size_t ci = (size_t)y * w + x;
      if (props9[9 * ci] == -2.0f) continue;
      for (int n = 1; n <= 8; n++)
        if (props9[9 * ci + n] > 0) deps[ci + nshift[n]]++;
    }
  int32_t *q = (int32_t *)malloc(N * 4);
  size_t qh = 0, qt = 0;
  for (size_t i = 0; i < N; i++)                                         /* :61-64 */
    if (deps[i] == 0 && props9[9 * i] != -2.0f) q[qt++] = (int32_t)i;
  while (qh < qt) {                                                      /* :71-92 */
    size_t ci = (size_t)q[qh++];
    double c_accum = accum[ci];
    for (int n = 1; n <= 8; n++) {
      if (props9[9 * ci + n] <= 0) continue;
      size_t ni = ci + nshift[n];
      if (props9[9 * ni] == -2.0f) continue;
      accum[ni] += props9[9 * ci + n] * c_accum;                         /* :87 */
      if (--deps[ni] == 0) q[qt++] = (int32_t)ni;
    }
  }
  for (size_t i = 0; i < N; i++)                                         /* :95-97 */
    if (props9[9 * i] == -2.0f) accum[i] = -1.0;
  free(deps); free(q);
}
