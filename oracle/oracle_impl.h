/* oracle/oracle_impl.h -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.c).
 *
 * Included once per elevation type with T (C type) and SUF (name suffix)
 * defined.  Plain-C restatement of the reference algorithms; every function
 * cites the reference file:line it follows (paths relative to
 * /root/reference/include/richdem/).
 */

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUF)

/* ------------------------------------------------------------------------- */
/* Priority-Flood (improved), depressions/Barnes2014.hpp:230-304.            */
/* Min-heap on z + plain FIFO for pit cells.  topo = 8 or 4.                 */
/* The filled surface is algorithm independent, so the */
/* same function is the oracle for FillDepressions<D8> (Zhou2016.hpp:126-191)*/
/* and FillDepressions<D4> (depressions.hpp:13-21).                          */
/* ------------------------------------------------------------------------- */
typedef struct { T z; int32_t x, y; } FN(hcell);

static void FN(heap_push)(FN(hcell) **heap, size_t *n, size_t *cap, FN(hcell) c) {
  if (*n == *cap) {
    *cap = *cap ? *cap * 2 : 1024;
    *heap = (FN(hcell) *)realloc(*heap, *cap * sizeof(FN(hcell)));
  }
  size_t i = (*n)++;
  FN(hcell) *h = *heap;
  while (i > 0) {
    size_t p = (i - 1) / 2;
    if (!(h[p].z > c.z)) break;  /* GridCellZ::operator> (common/grid_cell.hpp:29-38) */
    h[i] = h[p];
    i = p;
  }
  h[i] = c;
}

static FN(hcell) FN(heap_pop)(FN(hcell) *h, size_t *n) {
  FN(hcell) top = h[0];
  FN(hcell) last = h[--(*n)];
  size_t i = 0, sz = *n;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m;
    if (l >= sz) break;
    m = (r < sz && h[l].z > h[r].z) ? r : l;
    if (!(last.z > h[m].z)) break;
    h[i] = h[m];
    i = m;
  }
  if (sz) h[i] = last;
  return top;
}

void FN(orc_fill)(T *dem, int w, int h, int topo) {
  const int *dx = topo == 4 ? D4X : D8X, *dy = topo == 4 ? D4Y : D8Y;
  const int nmax = topo == 4 ? 4 : 8;
  size_t N = (size_t)w * h;
  int8_t *closed = (int8_t *)calloc(N, 1);                     /* Barnes2014.hpp:248 */
  FN(hcell) *heap = NULL; size_t hn = 0, hcap = 0;
  FN(hcell) *pit = (FN(hcell) *)malloc(N * sizeof(FN(hcell))); /* each cell enters at most once */
  size_t ph = 0, pt = 0;
  for (int x = 0; x < w; x++) {                                /* :255-260 */
    FN(hcell) a = {dem[x], x, 0}, b = {dem[(size_t)(h - 1) * w + x], x, h - 1};
    FN(heap_push)(&heap, &hn, &hcap, a);
    FN(heap_push)(&heap, &hn, &hcap, b);
    closed[x] = 1; closed[(size_t)(h - 1) * w + x] = 1;
  }
  for (int y = 1; y < h - 1; y++) {                            /* :261-266 */
    FN(hcell) a = {dem[(size_t)y * w], 0, y}, b = {dem[(size_t)y * w + w - 1], w - 1, y};
    FN(heap_push)(&heap, &hn, &hcap, a);
    FN(heap_push)(&heap, &hn, &hcap, b);
    closed[(size_t)y * w] = 1; closed[(size_t)y * w + w - 1] = 1;
  }
  while (hn > 0 || ph < pt) {                                  /* :270-297 */
    FN(hcell) c;
    if (ph < pt) c = pit[ph++];
    else c = FN(heap_pop)(heap, &hn);
    for (int n = 1; n <= nmax; n++) {
      int nx = c.x + dx[n], ny = c.y + dy[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (closed[ni]) continue;
      closed[ni] = 1;
      if (dem[ni] <= c.z) {
        if (dem[ni] < c.z) dem[ni] = c.z;                      /* overwrite only when strictly lower, :290-294 */
        FN(hcell) p = {c.z, nx, ny};
        pit[pt++] = p;
      } else {
        FN(hcell) o = {dem[ni], nx, ny};
        FN(heap_push)(&heap, &hn, &hcap, o);
      }
    }
  }
  free(closed); free(heap); free(pit);
}

/* ------------------------------------------------------------------------- */
/* d8_FlowDir + d8_flow_directions, flowmet/d8_flowdirs.hpp:32-74, :96-123.  */
/* ------------------------------------------------------------------------- */
static int FN(d8_flowdir_cell)(const T *dem, int w, int h, int x, int y) {
  T minimum = dem[(size_t)y * w + x];
  int flowdir = 0; /* NO_FLOW, common/constants.hpp:80 */
  if (x == 0 || y == 0 || x == w - 1 || y == h - 1) {          /* :37-54 */
    if (x == 0 && y == 0) return 2;
    else if (x == 0 && y == h - 1) return 8;
    else if (x == w - 1 && y == 0) return 4;
    else if (x == w - 1 && y == h - 1) return 6;
    else if (x == 0) return 1;
    else if (x == w - 1) return 5;
    else if (y == 0) return 3;
    else if (y == h - 1) return 7;
  }
  for (int n = 1; n <= 8; n++) {                               /* :63-71 */
    T e = dem[(size_t)(y + D8Y[n]) * w + (x + D8X[n])];
    if (e < minimum || (e == minimum && flowdir > 0 && flowdir % 2 == 0 && n % 2 == 1)) {
      minimum = e;
      flowdir = n;
    }
  }
  return flowdir;
}

void FN(orc_d8_flowdirs)(const T *dem, T nodata, int w, int h, uint8_t *dirs) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (dem[i] == nodata) dirs[i] = 255;                     /* FLOWDIR_NO_DATA, :116-117 */
      else dirs[i] = (uint8_t)FN(d8_flowdir_cell)(dem, w, h, x, y);
    }
}

/* ------------------------------------------------------------------------- */
/* resolve_flats_barnes, flats/flat_resolution.hpp:447-517 with              */
/* find_flat_edges :381-418, label_this :331-355, BuildAwayGradient :152-198,*/
/* BuildTowardsCombinedGradient :241-298.  dirs must hold d8_flow_directions */
/* output.  mask/labels are int32 w*h, written in full.                      */
/* ------------------------------------------------------------------------- */
void FN(orc_resolve_flats)(const T *dem, int w, int h, const uint8_t *dirs,
                           int32_t *mask, int32_t *labels) {
  size_t N = (size_t)w * h;
  memset(mask, 0, N * 4);
  memset(labels, 0, N * 4);
  /* find_flat_edges: column-major scan (x outer, y inner), :392-414 */
  int32_t *low = (int32_t *)malloc(N * 4), *high = (int32_t *)malloc(N * 4);
  size_t nlow = 0, nhigh = 0;
  for (int x = 0; x < w; x++)
    for (int y = 0; y < h; y++) {
      size_t i = (size_t)y * w + x;
      if (dirs[i] == 255) continue;
      for (int n = 1; n <= 8; n++) {
        int nx = x + D8X[n], ny = y + D8Y[n];
        if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
        size_t ni = (size_t)ny * w + nx;
        if (dirs[ni] == 255) continue;
        if (dirs[i] != 0 && dirs[ni] == 0 && dem[ni] == dem[i]) { low[nlow++] = (int32_t)i; break; }
        else if (dirs[i] == 0 && dem[i] < dem[ni]) { high[nhigh++] = (int32_t)i; break; }
      }
    }
  if (nlow == 0) { free(low); free(high); return; }            /* :475-481 */

  /* label_this for every still-unlabelled low edge, :483-487 / :331-355 */
  int32_t *q = (int32_t *)malloc(N * 4);
  int group = 1;
  for (size_t k = 0; k < nlow; k++) {
    size_t s = (size_t)low[k];
    if (labels[s] != 0) continue;
    T target = dem[s];
    size_t qh = 0, qt = 0;
    labels[s] = group; q[qt++] = (int32_t)s;
    while (qh < qt) {
      size_t c = (size_t)q[qh++];
      int cx = (int)(c % w), cy = (int)(c / w);
      for (int n = 1; n <= 8; n++) {
        int nx = cx + D8X[n], ny = cy + D8Y[n];
        if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
        size_t ni = (size_t)ny * w + nx;
        if (labels[ni] > 0 || dem[ni] != target) continue;     /* same partition as the reference's
                                                                  label-on-pop flood; labelling on
                                                                  push only bounds the queue */
        labels[ni] = group; q[qt++] = (int32_t)ni;
      }
    }
    group++;
  }
  /* drop high edges of flats without outlets, :491-500 */
  size_t nh2 = 0;
  for (size_t k = 0; k < nhigh; k++) if (labels[high[k]] != 0) high[nh2++] = high[k];
  nhigh = nh2;
  int32_t *flat_height = (int32_t *)calloc((size_t)group, 4);

  /* BuildAwayGradient :152-198 and BuildTowardsCombinedGradient :241-298.
     The reference runs one FIFO with an iteration marker; all sources enter at
     level 1, so a cell is first popped at its BFS level and later duplicates
     are skipped (:178, :279).  A frontier array per level is equivalent. */
  for (int pass = 0; pass < 2; pass++) {
    const int32_t *src = pass == 0 ? high : low;
    size_t ncur = pass == 0 ? nhigh : nlow;
    if (pass == 1)
      for (size_t i = 0; i < N; i++) mask[i] = -mask[i];       /* :259-262 */
    size_t fcap = ncur > 16 ? ncur : 16, ncap = 1024;
    int32_t *cur = (int32_t *)malloc(fcap * 4), *nxt = (int32_t *)malloc(ncap * 4);
    memcpy(cur, src, ncur * 4);
    int loops = 1;
    while (ncur > 0) {
      size_t nn = 0;
      for (size_t k = 0; k < ncur; k++) {
        size_t c = (size_t)cur[k];
        if (mask[c] > 0) continue;                             /* :178 / :279 */
        if (pass == 0) {
          mask[c] = loops;                                     /* :180 */
          flat_height[labels[c]] = loops;                      /* :181 */
        } else if (mask[c] != 0) {
          mask[c] = (flat_height[labels[c]] + mask[c]) + 2 * loops;  /* :281-282 */
        } else {
          mask[c] = 2 * loops;                                 /* :284 */
        }
        int cx = (int)(c % w), cy = (int)(c / w);
        for (int n = 1; n <= 8; n++) {
          int nx = cx + D8X[n], ny = cy + D8Y[n];
          if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
          size_t ni = (size_t)ny * w + nx;
          if (labels[ni] == labels[c] && dirs[ni] == 0) {      /* :189-192 / :289-292 */
            if (nn == ncap) { ncap *= 2; nxt = (int32_t *)realloc(nxt, ncap * 4); }
            nxt[nn++] = (int32_t)ni;
          }
        }
      }
      if (nn > fcap) { fcap = nn; cur = (int32_t *)realloc(cur, fcap * 4); }
      memcpy(cur, nxt, nn * 4);
      ncur = nn;
      loops++;
    }
    free(cur); free(nxt);
  }
  free(low); free(high); free(q); free(flat_height);
}

/* d8_masked_FlowDir :42-65 + d8_flow_flats :96-116 */
static void orc_d8_flow_flats(const int32_t *mask, const int32_t *labels, int w, int h, uint8_t *dirs);

/* barnes_flat_resolution_d8(alter=false), flats/flat_resolution.hpp:587-605 */
void FN(orc_flat_resolution)(const T *dem, T nodata, int w, int h, uint8_t *dirs) {
  size_t N = (size_t)w * h;
  int32_t *mask = (int32_t *)malloc(N * 4), *labels = (int32_t *)malloc(N * 4);
  FN(orc_d8_flowdirs)(dem, nodata, w, h, dirs);
  FN(orc_resolve_flats)(dem, w, h, dirs, mask, labels);
  orc_d8_flow_flats(mask, labels, w, h, dirs);
  free(mask); free(labels);
}

/* barnes_flat_resolution_d8(alter=true), flats/flat_resolution.hpp:597-600 with
 * d8_flats_alter_dem :545-582.  The DEM is altered in place (nextafterf towards numeric_limits<U>::infinity(), which is
 * 0 for integer U: those walk towards zero, as the reference's do). */
void FN(orc_flat_resolution_alter)(T *dem, T nodata, int w, int h, uint8_t *dirs) {
  size_t N = (size_t)w * h;
  int32_t *mask = (int32_t *)malloc(N * 4), *labels = (int32_t *)malloc(N * 4);
  FN(orc_d8_flowdirs)(dem, nodata, w, h, dirs);
  FN(orc_resolve_flats)(dem, w, h, dirs, mask, labels);
  for (int y = 1; y < h - 1; y++)                                            /* :556-558 */
    for (int x = 1; x < w - 1; x++) {
      size_t i = (size_t)y * w + x;
      if (labels[i] == 0) continue;                                          /* :559-560 */
      /* :567-568: towards numeric_limits<T>::infinity(), which is 0 for integer T */
      for (int k = 0; k < mask[i]; ++k) dem[i] = (T)nextafterf((float)dem[i], _Generic((T)0, float: INFINITY, double: INFINITY, default: 0.0f));
    }
  FN(orc_d8_flowdirs)(dem, nodata, w, h, dirs);                             /* :600 */
  free(mask); free(labels);
}

/* ------------------------------------------------------------------------- */
/* FM_OCallaghan<D8> = FM_D8, flowmet/OCallaghan1984.hpp:13-77, :81-84.       */
/* props9: 9 floats per cell, index 9*i+n (common/Array3D.hpp:203-206).       */
/* ------------------------------------------------------------------------- */
void FN(orc_fm_d8)(const T *dem, T nodata, int w, int h, float *props9) {
  size_t N = (size_t)w * h;
  for (size_t i = 0; i < N * 9; i++) props9[i] = -1.0f;        /* NO_FLOW_GEN, :26 */
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (dem[i] == nodata) { props9[9 * i] = -2.0f; continue; }     /* NO_DATA_GEN, :37-40 */
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;    /* :42-43 */
      T e = dem[i];
      int lowest_n = 0;
      T lowest = 0; int have = 0;
      for (int n = 1; n <= 8; n++) {                                 /* :49-65 */
        size_t ni = (size_t)(y + D8Y[n]) * w + (x + D8X[n]);
        if (dem[ni] == nodata) continue;
        T ne = dem[ni];
        if (ne >= e) continue;
        /* reference compares against numeric_limits<T>::max() initially, so a
           neighbour equal to max() can never be chosen; it cannot be < e either
           unless e > max(), i.e. never -- equivalent to "first strictly lowest" */
        if (!have || ne < lowest) { lowest = ne; lowest_n = n; have = 1; }
      }
      if (lowest_n == 0) continue;
      props9[9 * i] = 0.0f;                                          /* HAS_FLOW_GEN, :70 */
      props9[9 * i + lowest_n] = 1.0f;                               /* :74 */
    }
}

/* FA_D8, methods/flow_accumulation.hpp:27 */
void orc_flow_accumulation_f64(const float *props9, int w, int h, double *accum);
void FN(orc_fa_d8)(const T *dem, T nodata, int w, int h, double *accum) {
  float *props = (float *)malloc((size_t)w * h * 9 * sizeof(float));
  FN(orc_fm_d8)(dem, nodata, w, h, props);
  orc_flow_accumulation_f64(props, w, h, accum);
  free(props);
}

/* FA_D8 with one receiver byte per cell instead of the 36 B/cell proportions array -- the same
 * FM_D8 rule (OCallaghan1984.hpp:37-74) and the same Kahn order (flow_accumulation_generic.hpp:48-97),
 * written so that a 40000 x 40000 DEM fits a 64 GB host (FA_D8 proper needs ~78 GB there).  Checked
 * equal to orc_fa_d8 and to the compiled reference's FA_D8 by tests/test_oracle_pinning.py; used for
 * the S3 digests of tests/golden/make_golden.py --s3-digests. */
void FN(orc_fa_d8_lean)(const T *dem, T nodata, int w, int h, double *accum) {
  size_t N = (size_t)w * h;
  const int nshift[9] = {0, -1, -w - 1, -w, -w + 1, 1, w + 1, w, w - 1};
  uint8_t *recv = (uint8_t *)calloc(N, 1);                     /* 0 no flow, 1..8, 255 NoData */
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (dem[i] == nodata) { recv[i] = 255; continue; }
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;
      T e = dem[i], lowest = 0;
      int lowest_n = 0;
      for (int n = 1; n <= 8; n++) {
        T ne = dem[i + nshift[n]];
        if (ne == nodata || ne >= e) continue;
        if (!lowest_n || ne < lowest) { lowest = ne; lowest_n = n; }
      }
      recv[i] = (uint8_t)lowest_n;
    }
  int8_t *deps = (int8_t *)calloc(N, 1);
  for (int y = 1; y < h - 1; y++)
    for (int x = 1; x < w - 1; x++) {
      size_t ci = (size_t)y * w + x;
      if (recv[ci] >= 1 && recv[ci] <= 8) deps[ci + nshift[recv[ci]]]++;
    }
  int32_t *q = (int32_t *)malloc(N * 4);
  size_t qh = 0, qt = 0;
  for (size_t i = 0; i < N; i++)
    if (deps[i] == 0 && recv[i] != 255) q[qt++] = (int32_t)i;
  while (qh < qt) {
    size_t ci = (size_t)q[qh++];
    if (recv[ci] < 1 || recv[ci] > 8) continue;
    size_t ni = ci + nshift[recv[ci]];
    if (recv[ni] == 255) continue;
    accum[ni] += 1.0f * accum[ci];
    if (--deps[ni] == 0) q[qt++] = (int32_t)ni;
  }
  for (size_t i = 0; i < N; i++)
    if (recv[i] == 255) accum[i] = -1.0;
  free(recv); free(deps); free(q);
}

/* ------------------------------------------------------------------------- */
/* dinf_FlowDir + dinf_flow_directions, flowmet/dinf_flowdirs.hpp:45-115,    */
/* :128-152 (facet tables :21-27).                                           */
/* ------------------------------------------------------------------------- */
void FN(orc_dinf_flowdirs)(const T *dem, T nodata, int w, int h, float *out) {
  static const int dy_e1[8] = {0, -1, -1, 0, 0, 1, 1, 0}, dx_e1[8] = {1, 0, 0, -1, -1, 0, 0, 1};
  static const int dy_e2[8] = {-1, -1, -1, -1, 1, 1, 1, 1}, dx_e2[8] = {1, 1, -1, -1, -1, -1, 1, 1};
  static const double ac[8] = {0., 1., 1., 2., 2., 3., 3., 4.}, af[8] = {1., -1., 1., -1., 1., -1., 1., -1.};
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (dem[i] == nodata) { out[i] = -1.0f; continue; }                  /* :147-148 */
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) {                  /* :46-63 */
        double a;
        if (x == 0 && y == 0) a = 3 * M_PI / 4;
        else if (x == 0 && y == h - 1) a = 5 * M_PI / 4;
        else if (x == w - 1 && y == 0) a = 1 * M_PI / 4;
        else if (x == w - 1 && y == h - 1) a = 7 * M_PI / 4;
        else if (x == 0) a = 4 * M_PI / 4;
        else if (x == w - 1) a = 0 * M_PI / 4;
        else if (y == 0) a = 2 * M_PI / 4;
        else a = 6 * M_PI / 4;
        out[i] = (float)a;
        continue;
      }
      int nmax = -1;
      double smax = 0, rmax = 0;
      for (int n = 0; n < 8; n++) {                                        /* :68-96 */
        const double e0 = (double)dem[i];
        const double e1 = (double)dem[(size_t)(y + dy_e1[n]) * w + (x + dx_e1[n])];
        const double e2 = (double)dem[(size_t)(y + dy_e2[n]) * w + (x + dx_e2[n])];
        const double d1 = 1, d2 = 1;
        const double s1 = (e0 - e1) / d1, s2 = (e1 - e2) / d2;
        double r = atan2(s2, s1), s;
        if (r < 0) { r = 0; s = s1; }
        else if (r > atan2(d2, d1)) { r = atan2(d2, d1); s = (e0 - e2) / sqrt(d1 * d1 + d2 * d2); }
        else s = sqrt(s1 * s1 + s2 * s2);
        if (s > smax) { smax = s; nmax = n; rmax = r; }
      }
      double rg = 0;                                                       /* NO_FLOW */
      if (nmax != -1) rg = af[nmax] * rmax + ac[nmax] * M_PI / 2;
      out[i] = (float)rg;
    }
}

/* ------------------------------------------------------------------------- */
/* FM_Tarboton = FM_Dinfinity, flowmet/Tarboton1997.hpp:14-144.              */
/* ------------------------------------------------------------------------- */
void FN(orc_fm_tarboton)(const T *dem, T nodata, int w, int h, float *props9) {
  static const int dy_e1[9] = {0, 0, -1, -1, 0, 0, 1, 1, 0}, dx_e1[9] = {0, -1, 0, 0, 1, 1, 0, 0, -1};
  static const int dy_e2[9] = {0, -1, -1, -1, -1, 1, 1, 1, 1}, dx_e2[9] = {0, -1, -1, 1, 1, 1, 1, -1, -1};
  static const double af[9] = {0, -1., 1., -1., 1., -1., 1., -1., 1.};
  const double d1 = 1, d2 = 1;
  const float dang = (float)atan2(d2, d1);                                 /* :27 */
  size_t N = (size_t)w * h;
  for (size_t i = 0; i < N * 9; i++) props9[i] = -1.0f;                    /* :25 */
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (dem[i] == nodata) { props9[9 * i] = -2.0f; continue; }           /* :44-47 */
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;          /* :49-50 */
      int nmax = -1;
      double smax = 0;
      float rmax = 0;
      for (int n = 1; n <= 8; n++) {                                       /* :56-92; neighbours of interior cells are in grid */
        const T v1 = dem[(size_t)(y + dy_e1[n]) * w + (x + dx_e1[n])], v2 = dem[(size_t)(y + dy_e2[n]) * w + (x + dx_e2[n])];
        if (v1 == nodata || v2 == nodata) continue;
        const double e0 = (double)dem[i], e1 = (double)v1, e2 = (double)v2;
        const double s1 = (e0 - e1) / d1, s2 = (e1 - e2) / d2;
        double r = atan2(s2, s1), s;
        if (r < 1e-7) { r = 0; s = s1; }
        else if (r > dang - 1e-7) { r = dang; s = (e0 - e2) / sqrt(d1 * d1 + d2 * d2); }
        else s = sqrt(s1 * s1 + s2 * s2);
        if (s > smax) { smax = s; nmax = n; rmax = (float)r; }
      }
      if (nmax == -1) continue;
      props9[9 * i] = 0.0f;                                                /* :97 */
      if (af[nmax] == 1 && rmax == 0) rmax = dang;                         /* :99-104 */
      else if (af[nmax] == 1 && rmax == dang) rmax = 0;
      else if (af[nmax] == 1) rmax = (float)(M_PI / 4 - rmax);
      const int nxt = nmax + 1 == 9 ? 1 : nmax + 1;
      if (rmax == 0) props9[9 * i + nmax] = 1;                             /* :106-113 */
      else if (rmax == dang) props9[9 * i + nxt] = 1;
      else {
        props9[9 * i + nmax] = (float)(rmax / (M_PI / 4.));
        props9[9 * i + nxt] = (float)(1 - rmax / (M_PI / 4.));
      }
    }
}

/* FA_Tarboton, methods/flow_accumulation.hpp:16 */
void FN(orc_fa_tarboton)(const T *dem, T nodata, int w, int h, double *accum) {
  float *props = (float *)malloc((size_t)w * h * 9 * sizeof(float));
  FN(orc_fm_tarboton)(dem, nodata, w, h, props);
  orc_flow_accumulation_f64(props, w, h, accum);
  free(props);
}


/* ------------------------------------------------------------------------- */
/* FM_Holmgren (flowmet/Holmgren1994.hpp:14-88), FM_Quinn = Holmgren with     */
/* x = 1 (flowmet/Quinn1991.hpp:13-17), FM_Freeman (flowmet/Freeman1991.hpp:  */
/* 14-86), FM_OCallaghan<D4> = FM_D4 (flowmet/OCallaghan1984.hpp:13-77, :86). */
/* method: 0 Holmgren, 1 Freeman, 2 Quinn, 3 D4.                              */
/* ------------------------------------------------------------------------- */
void FN(orc_fm_mfd)(const T *dem, T nodata, int w, int h, int method, double xparam, float *props9) {
  static const double DR[9] = {0, 1, 1.414213562373095048801688724209698078569671875376948, 1,
                               1.414213562373095048801688724209698078569671875376948, 1,
                               1.414213562373095048801688724209698078569671875376948, 1,
                               1.414213562373095048801688724209698078569671875376948};   /* constants.hpp:34,70 */
  static const double HL[9] = {0, 0.5, 0.354, 0.5, 0.354, 0.5, 0.354, 0.5, 0.354};   /* Holmgren1994.hpp:26-28 */
  size_t N = (size_t)w * h;
  if (method == 2) { method = 0; xparam = 1.0; }
  for (size_t i = 0; i < N * 9; i++) props9[i] = -1.0f;
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      float *p = props9 + 9 * i;
      if (dem[i] == nodata) { p[0] = -2.0f; continue; }
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) continue;
      const T e = dem[i];
      if (method == 3) {          /* D4 neighbours (constants.hpp:54-55) stored in slots 1..4, OCallaghan1984.hpp:46-74 */
        int lowest_n = 0, have = 0;
        T lowest = 0;
        for (int n = 1; n <= 4; n++) {
          size_t ni = (size_t)(y + D4Y[n]) * w + (x + D4X[n]);
          if (dem[ni] == nodata) continue;
          T ne = dem[ni];
          if (ne >= e) continue;
          if (!have || ne < lowest) { lowest = ne; lowest_n = n; have = 1; }
        }
        if (lowest_n == 0) continue;
        p[0] = 0.0f;
        p[lowest_n] = 1.0f;
        continue;
      }
      double C = 0;
      for (int n = 1; n <= 8; n++) {
        size_t ni = (size_t)(y + D8Y[n]) * w + (x + D8X[n]);
        if (dem[ni] == nodata) continue;
        const T ne = dem[ni];
        if (ne < e) {
          const double rise = e - ne;
          const double run = DR[n];
          const double grad = rise / run;
          if (method == 0) {
            p[n] = (float)pow(grad * HL[n], xparam);     /* Holmgren1994.hpp:66-67: C sums the stored floats */
            C += p[n];
          } else {
            const double cval = pow(grad, xparam);       /* Freeman1991.hpp:63-65: C sums the doubles */
            p[n] = (float)cval;
            C += cval;
          }
        }
      }
      if (C > 0) {
        p[0] = 0.0f;
        C = 1 / C;
        for (int n = 1; n <= 8; n++) {
          if (p[n] > 0) p[n] = (float)(p[n] * C);
          else p[n] = 0;
        }
      }
    }
}

void FN(orc_fa_mfd)(const T *dem, T nodata, int w, int h, int method, double xparam, double *accum) {
  float *props = (float *)malloc((size_t)w * h * 9 * sizeof(float));
  FN(orc_fm_mfd)(dem, nodata, w, h, method, xparam, props);
  orc_flow_accumulation_f64(props, w, h, accum);
  free(props);
}


/* ------------------------------------------------------------------------- */
/* ResolveFlatsEpsilon, flats/flats.hpp:21-28 = FindFlats (flats/find_flats.hpp:29-69) + GetFlatMask          */
/* (flats/Barnes2014.hpp:398-467) + ResolveFlatsEpsilon_Barnes2014 (:496-550).  The flat mask of GetFlatMask   */
/* is the BFS construction restated in orc_resolve_flats, applied to FindFlats' notion of "flat" (no lower    */
/* neighbour AND no NoData neighbour; edge cells never flat); equivalence with the compiled reference is      */
/* checked in tests/test_oracle_pinning.py.  Every labelled interior cell is raised by flat_mask increments   */
/* of std::nextafter(e, numeric_limits<T>::infinity()) IN TYPE T; for integer T that infinity() is 0 and the  */
/* arguments promote to double, so a step moves the value by one TOWARDS ZERO -- reproduced as is.            */
/* ------------------------------------------------------------------------- */
#define ORC_NEXT_UP(x) _Generic((x), float: nextafterf((float)(x), INFINITY), double: nextafter((double)(x), (double)INFINITY), \
                                default: nextafter((double)(x), 0.0))
void FN(orc_find_flats)(const T *dem, T nodata, int w, int h, uint8_t *flats /* 0 flat, 1 not, 255 NoData */) {
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (dem[i] == nodata) { flats[i] = 255; continue; }                    /* find_flats.hpp:43-46 */
      if (x == 0 || y == 0 || x == w - 1 || y == h - 1) { flats[i] = 1; continue; }   /* :48-51 */
      uint8_t f = 0;
      for (int n = 1; n <= 8; n++) {                                          /* :56-63 */
        size_t ni = (size_t)(y + D8Y[n]) * w + (x + D8X[n]);
        if (dem[ni] < dem[i] || dem[ni] == nodata) { f = 1; break; }
      }
      flats[i] = f;
    }
}

void FN(orc_resolve_flats_epsilon)(T *dem, T nodata, int w, int h) {
  size_t N = (size_t)w * h;
  uint8_t *flats = (uint8_t *)malloc(N);
  int32_t *mask = (int32_t *)malloc(N * 4), *labels = (int32_t *)malloc(N * 4);
  FN(orc_find_flats)(dem, nodata, w, h, flats);
  FN(orc_resolve_flats)(dem, w, h, flats, mask, labels);
  for (int y = 1; y < h - 1; y++)                                            /* Barnes2014.hpp:511-512 */
    for (int x = 1; x < w - 1; x++) {
      size_t i = (size_t)y * w + x;
      if (labels[i] == 0) continue;                                          /* :516-517 */
      for (int k = 0; k < mask[i]; ++k) dem[i] = (T)ORC_NEXT_UP(dem[i]);     /* :527-528 */
    }
  free(flats); free(mask); free(labels);
}
#undef ORC_NEXT_UP


/* pit_mask<topo>, depressions/Barnes2014.hpp:593-676: the Priority-Flood of orc_fill with a mask instead of a raised
 * DEM: a cell is 1 iff it is strictly below the level it is reached at (:653-656), i.e. iff the fill raises it;
 * cells reached at their own level or above stay / become 0 (:651, :659); every NoData cell ends as 3 (:668-669). */
void FN(orc_pit_mask)(const T *dem, T nodata, int w, int h, int topo, uint8_t *mask) {
  size_t N = (size_t)w * h;
  T *filled = (T *)malloc(N * sizeof(T));
  memcpy(filled, dem, N * sizeof(T));
  FN(orc_fill)(filled, w, h, topo);
  for (size_t i = 0; i < N; i++) mask[i] = dem[i] == nodata ? 3 : (filled[i] > dem[i] ? 1 : 0);
  free(filled);
}


/* ------------------------------------------------------------------------- */
/* PriorityFloodFlowdirs_Barnes2014, depressions/Barnes2014.hpp:483-555.      */
/* A Priority-Flood that raises nothing: cells leave a STABLE queue -- lowest  */
/* elevation first, equal elevations in insertion order (GridCellZk_low_pq,    */
/* common/grid_cell.hpp:75-81,:114-122) -- and every cell flows to the cell    */
/* that closed it (neighbours are visited cardinals first, :524).  Because the */
/* order is total, the output does not depend on the heap implementation.      */
/* ------------------------------------------------------------------------- */
typedef struct { T z; int32_t k, x, y; } FN(kcell);
static int FN(kcell_gt)(const FN(kcell) *a, const FN(kcell) *b) { return a->z > b->z || (a->z == b->z && a->k > b->k); }
void FN(orc_pf_flowdirs)(const T *dem, T nodata, int w, int h, uint8_t *dirs) {
  static const int order[9] = {0, 1, 3, 5, 7, 2, 4, 6, 8}, inverse[9] = {0, 5, 6, 7, 8, 1, 2, 3, 4};
  size_t N = (size_t)w * h, hn = 0, hcap = 2 * ((size_t)w + h) + 16;
  FN(kcell) *heap = (FN(kcell) *)malloc(hcap * sizeof(FN(kcell)));
  int8_t *closed = (int8_t *)calloc(N, 1);
  int32_t count = 0;
  memset(dirs, 0, N);                                                        /* flowdirs.setNoData(NO_FLOW = 0) */
#define ORC_PUSH(X, Y, Z) do { FN(kcell) c_ = {(Z), ++count, (X), (Y)}; size_t i_ = hn++;                      \
    if (hn > hcap) { hcap *= 2; heap = (FN(kcell) *)realloc(heap, hcap * sizeof(FN(kcell))); }                 \
    while (i_ > 0) { size_t p_ = (i_ - 1) / 2; if (!FN(kcell_gt)(&heap[p_], &c_)) break; heap[i_] = heap[p_]; i_ = p_; } \
    heap[i_] = c_; } while (0)
  for (int x = 0; x < w; x++) {                                              /* :508-515 */
    ORC_PUSH(x, 0, dem[x]);
    ORC_PUSH(x, h - 1, dem[(size_t)(h - 1) * w + x]);
    dirs[x] = 3; dirs[(size_t)(h - 1) * w + x] = 7;
    closed[x] = 1; closed[(size_t)(h - 1) * w + x] = 1;
  }
  for (int y = 1; y < h - 1; y++) {                                          /* :516-523 */
    ORC_PUSH(0, y, dem[(size_t)y * w]);
    ORC_PUSH(w - 1, y, dem[(size_t)y * w + w - 1]);
    dirs[(size_t)y * w] = 1; dirs[(size_t)y * w + w - 1] = 5;
    closed[(size_t)y * w] = 1; closed[(size_t)y * w + w - 1] = 1;
  }
  dirs[0] = 2; dirs[w - 1] = 4; dirs[(size_t)(h - 1) * w] = 8; dirs[(size_t)(h - 1) * w + w - 1] = 6;   /* :525-528 */
  while (hn > 0) {                                                           /* :533-553 */
    FN(kcell) c = heap[0], last = heap[--hn];
    size_t i = 0;
    for (;;) {
      size_t l = 2 * i + 1, r = l + 1, m;
      if (l >= hn) break;
      m = (r < hn && FN(kcell_gt)(&heap[l], &heap[r])) ? r : l;
      if (!FN(kcell_gt)(&last, &heap[m])) break;
      heap[i] = heap[m];
      i = m;
    }
    if (hn) heap[i] = last;
    for (int no = 1; no <= 8; no++) {
      int n = order[no], nx = c.x + D8X[n], ny = c.y + D8Y[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (closed[ni]) continue;
      closed[ni] = 1;
      dirs[ni] = dem[ni] == nodata ? 0 : (uint8_t)inverse[n];               /* :545-548 */
      ORC_PUSH(nx, ny, dem[ni]);
    }
  }
#undef ORC_PUSH
  free(heap); free(closed);
}

/* ------------------------------------------------------------------------- */
/* the other outputs of the Priority-Flood sweep.               */
/* Ties: the reference pops equal elevations in the order libstdc++'s         */
/* std::priority_queue happens to hold them; this heap has its own order.  On */
/* DEMs without equal elevations among the heap's cells the results agree     */
/* (tests/test_oracle_pinning.py); DESIGN.md section 3b says what is defined  */
/* when they do not.                                                          */
/* ------------------------------------------------------------------------- */
static void FN(seed_border)(const T *dem, int w, int h, int8_t *closed, FN(hcell) **heap, size_t *hn, size_t *hcap) {
  for (int x = 0; x < w; x++) {                                /* e.g. Barnes2014.hpp:358-363 */
    FN(hcell) a = {dem[x], x, 0}, b = {dem[(size_t)(h - 1) * w + x], x, h - 1};
    FN(heap_push)(heap, hn, hcap, a);
    FN(heap_push)(heap, hn, hcap, b);
    closed[x] = 1; closed[(size_t)(h - 1) * w + x] = 1;
  }
  for (int y = 1; y < h - 1; y++) {                            /* :364-369 */
    FN(hcell) a = {dem[(size_t)y * w], 0, y}, b = {dem[(size_t)y * w + w - 1], w - 1, y};
    FN(heap_push)(heap, hn, hcap, a);
    FN(heap_push)(heap, hn, hcap, b);
    closed[(size_t)y * w] = 1; closed[(size_t)y * w + w - 1] = 1;
  }
}

#ifdef ORC_IS_FLOAT
/* PriorityFloodEpsilon_Barnes2014<topo>, depressions/Barnes2014.hpp:335-420 (floating point only, :424-451). */
void FN(orc_fill_epsilon)(T *dem, T nodata, int w, int h, int topo) {
  const int *dx = topo == 4 ? D4X : D8X, *dy = topo == 4 ? D4Y : D8Y;
  const int nmax = topo == 4 ? 4 : 8;
  size_t N = (size_t)w * h;
  int8_t *closed = (int8_t *)calloc(N, 1);                     /* :355 */
  FN(hcell) *heap = NULL; size_t hn = 0, hcap = 0;
  FN(hcell) *pit = (FN(hcell) *)malloc(N * sizeof(FN(hcell)));
  size_t ph = 0, pt = 0;
  FN(seed_border)(dem, w, h, closed, &heap, &hn, &hcap);
  while (hn > 0 || ph < pt) {                                  /* :373-414 */
    FN(hcell) c;
    if (ph < pt && hn > 0 && heap[0].z == pit[ph].z) c = FN(heap_pop)(heap, &hn);   /* :375-378 */
    else if (ph < pt) c = pit[ph++];                                                /* :379-383 */
    else c = FN(heap_pop)(heap, &hn);                                               /* :384-388 */
    const T up = ORC_NEXTUP(c.z);                              /* std::nextafter(c.z, +inf) */
    for (int n = 1; n <= nmax; n++) {
      int nx = c.x + dx[n], ny = c.y + dy[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (closed[ni]) continue;
      closed[ni] = 1;
      if (dem[ni] == nodata) {                                 /* :401-402: queued with z = NoData, not altered */
        FN(hcell) p = {nodata, nx, ny};
        pit[pt++] = p;
      } else if (dem[ni] <= up) {                              /* :404-409 */
        dem[ni] = up;
        FN(hcell) p = {up, nx, ny};
        pit[pt++] = p;
      } else {
        FN(hcell) o = {dem[ni], nx, ny};
        FN(heap_push)(&heap, &hn, &hcap, o);
      }
    }
  }
  free(closed); free(heap); free(pit);
}
#endif

/* PriorityFloodWatersheds_Barnes2014<topo>(elevations, labels, alter_elevations), depressions/Barnes2014.hpp:713-807. */
void FN(orc_watersheds)(T *dem, T nodata, int w, int h, int topo, int alter, int32_t *labels) {
  const int *dx = topo == 4 ? D4X : D8X, *dy = topo == 4 ? D4Y : D8Y;
  const int nmax = topo == 4 ? 4 : 8;
  size_t N = (size_t)w * h;
  int8_t *closed = (int8_t *)calloc(N, 1);
  FN(hcell) *heap = NULL; size_t hn = 0, hcap = 0;
  FN(hcell) *pit = (FN(hcell) *)malloc(N * sizeof(FN(hcell)));
  size_t ph = 0, pt = 0;
  int32_t clabel = 1;                                          /* :721 */
  for (size_t i = 0; i < N; i++) labels[i] = -1;               /* :737-738 */
  FN(seed_border)(dem, w, h, closed, &heap, &hn, &hcap);
  while (hn > 0 || ph < pt) {                                  /* :760-800 */
    FN(hcell) c;
    if (ph < pt) c = pit[ph++];
    else c = FN(heap_pop)(heap, &hn);
    size_t ci = (size_t)c.y * w + c.x;
    if (labels[ci] == -1 && dem[ci] != nodata) labels[ci] = clabel++;   /* :777-778 */
    for (int n = 1; n <= nmax; n++) {
      int nx = c.x + dx[n], ny = c.y + dy[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (closed[ni]) continue;
      labels[ni] = labels[ci];                                 /* :789 */
      closed[ni] = 1;
      if (dem[ni] <= c.z) {                                    /* :792-796 */
        if (alter) dem[ni] = c.z;
        FN(hcell) p = {c.z, nx, ny};
        pit[pt++] = p;
      } else {
        FN(hcell) o = {dem[ni], nx, ny};
        FN(heap_push)(&heap, &hn, &hcap, o);
      }
    }
  }
  free(closed); free(heap); free(pit);
}

/* PriorityFlood_Barnes2014_max_dep<topo>(elevations, max_dep_size), depressions/Barnes2014.hpp:844-931. */
void FN(orc_fill_max_dep)(T *dem, int w, int h, int topo, uint64_t max_dep_size) {
  const int *dx = topo == 4 ? D4X : D8X, *dy = topo == 4 ? D4Y : D8Y;
  const int nmax = topo == 4 ? 4 : 8;
  size_t N = (size_t)w * h;
  int8_t *closed = (int8_t *)calloc(N, 1);
  FN(hcell) *heap = NULL; size_t hn = 0, hcap = 0;
  FN(hcell) *pit = (FN(hcell) *)malloc(N * sizeof(FN(hcell)));
  size_t ph = 0, pt = 0;
  size_t *dep = (size_t *)malloc(N * sizeof(size_t));          /* dep_cells, :887 */
  size_t ndep = 0;
  T dep_elev = 0;                                              /* :886 */
  FN(seed_border)(dem, w, h, closed, &heap, &hn, &hcap);
  while (hn > 0 || ph < pt) {                                  /* :891-927 */
    FN(hcell) c;
    if (ph < pt) {
      c = pit[ph++];
      dep[ndep++] = (size_t)c.y * w + c.x;                     /* :896 */
    } else {
      c = FN(heap_pop)(heap, &hn);
      if (ndep <= max_dep_size)                                /* :900-903 */
        for (size_t k = 0; k < ndep; k++) dem[dep[k]] = dep_elev;
      ndep = 0;                                                /* :903 / :905 */
    }
    for (int n = 1; n <= nmax; n++) {
      int nx = c.x + dx[n], ny = c.y + dy[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (closed[ni]) continue;
      closed[ni] = 1;
      if (dem[ni] < c.z) {                                     /* :918-921 */
        FN(hcell) p = {c.z, nx, ny};
        pit[pt++] = p;
        dep_elev = c.z;
      } else {
        FN(hcell) o = {dem[ni], nx, ny};
        FN(heap_push)(&heap, &hn, &hcap, o);
      }
    }
  }
  free(closed); free(heap); free(pit); free(dep);
}

/* HasDepressions<topo>, depressions/Barnes2014.hpp:44-103: the flood of PriorityFlood_Original (:136-198) on the
 * UNRAISED elevations; the first cell discovered from a strictly higher one ends it with "true" (:91-95). */
int FN(orc_has_depressions)(const T *dem, int w, int h, int topo) {
  const int *dx = topo == 4 ? D4X : D8X, *dy = topo == 4 ? D4Y : D8Y;
  const int nmax = topo == 4 ? 4 : 8;
  size_t N = (size_t)w * h;
  int8_t *closed = (int8_t *)calloc(N, 1);                     /* :62 */
  FN(hcell) *heap = NULL; size_t hn = 0, hcap = 0;
  int found = 0;
  FN(seed_border)(dem, w, h, closed, &heap, &hn, &hcap);       /* :69-80 */
  while (hn > 0 && !found) {                                   /* :84-99 */
    FN(hcell) c = FN(heap_pop)(heap, &hn);
    for (int n = 1; n <= nmax; n++) {
      int nx = c.x + dx[n], ny = c.y + dy[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (closed[ni]) continue;
      closed[ni] = 1;
      if (dem[ni] < dem[(size_t)c.y * w + c.x]) { found = 1; break; }   /* :91-95 */
      FN(hcell) o = {dem[ni], nx, ny};
      FN(heap_push)(&heap, &hn, &hcap, o);
    }
  }
  free(closed); free(heap);
  return found;
}

/* PriorityFlood_Wei2018, depressions/Wei2018.hpp:154-202.  What distinguishes it from the other fills is its seeding
 * (InitPriorityQue, :14-50): NoData cells are flagged up front and never altered, and every data cell next to one is a seed
 * at its own elevation beside the raster's edge cells.  The flood itself (ProcessPit :124-150 raising cells <= the spill
 * level, ProcessTraceQue :54-120 deciding which slope cells need the heap at all) produces the Priority-Flood surface over
 * those seeds; it is restated here as the plain flood of orc_fill (heap + pit queue) with Wei's seeds and flags, which
 * tests/test_oracle_pinning.py holds against the compiled reference on rasters with NoData holes. */
void FN(orc_fill_wei2018)(T *dem, T nodata, int w, int h) {
  size_t N = (size_t)w * h;
  int8_t *flag = (int8_t *)calloc(N, 1);                       /* :166 */
  FN(hcell) *heap = NULL; size_t hn = 0, hcap = 0;
  FN(hcell) *pit = (FN(hcell) *)malloc(N * sizeof(FN(hcell)));
  size_t ph = 0, pt = 0;
  for (int y = 0; y < h; y++)                                  /* InitPriorityQue :23-49 */
    for (int x = 0; x < w; x++) {
      size_t i = (size_t)y * w + x;
      if (flag[i]) continue;
      if (dem[i] == nodata) {
        flag[i] = 1;
        for (int n = 1; n <= 8; n++) {
          int nx = x + D8X[n], ny = y + D8Y[n];
          if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
          size_t ni = (size_t)ny * w + nx;
          if (flag[ni] || dem[ni] == nodata) continue;
          FN(hcell) o = {dem[ni], nx, ny};
          FN(heap_push)(&heap, &hn, &hcap, o);
          flag[ni] = 1;
        }
      } else if (x == 0 || y == 0 || x == w - 1 || y == h - 1) {
        FN(hcell) o = {dem[i], x, y};
        FN(heap_push)(&heap, &hn, &hcap, o);
        flag[i] = 1;
      }
    }
  while (hn > 0 || ph < pt) {                                  /* :171-197 */
    FN(hcell) c;
    if (ph < pt) c = pit[ph++];
    else c = FN(heap_pop)(heap, &hn);
    for (int n = 1; n <= 8; n++) {
      int nx = c.x + D8X[n], ny = c.y + D8Y[n];
      if (nx < 0 || ny < 0 || nx >= w || ny >= h) continue;
      size_t ni = (size_t)ny * w + nx;
      if (flag[ni]) continue;
      flag[ni] = 1;
      if (dem[ni] <= c.z) {                                    /* depression cell, :184-189 / ProcessPit :143-146 */
        dem[ni] = c.z;
        FN(hcell) p = {c.z, nx, ny};
        pit[pt++] = p;
      } else {                                                 /* slope cell, :190-194 */
        FN(hcell) o = {dem[ni], nx, ny};
        FN(heap_push)(&heap, &hn, &hcap, o);
      }
    }
  }
  free(flag); free(heap); free(pit);
}

#undef CAT_
#undef CAT
#undef FN
