/* Address/UB-sanitised self test of the oracle (CPU only; GPU sanitizers are not available on the
 * pool).  Exercises every allocation path: kd-tree growth, range lists, ghost iterator, polygon and
 * Dubins code.  Build+run: make -C oracle selftest   (tests/test_oracle_sanitizer.py) */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rrtx_oracle.h"

static double frand(unsigned *s) { *s = *s * 1664525u + 1013904223u; return (double)(*s >> 8) / 16777216.0; }

int main(void) {
  unsigned seed = 12345;
  int bad = 0;
  /* 3-D tree: kd == naive */
  orc_kd *t = orc_kd_create(3);
  for (int i = 0; i < 5000; ++i) { double p[3] = {frand(&seed), frand(&seed), frand(&seed)}; orc_kd_insert(t, p); }
  for (int k = 0; k < 200; ++k) {
    double q[3] = {frand(&seed), frand(&seed), frand(&seed)};
    int64_t a, b; double da, db;
    orc_kd_nearest(t, q, &a, &da); orc_kd_nearest_naive(t, q, &b, &db);
    bad += (a != b) || (da != db);
    orc_list *l = orc_kd_find_within_range(t, 0.15, q);
    orc_kd_find_more_within_range(t, 0.1, q, l);
    int32_t idx[4096]; double key[4096];
    int64_t n = orc_list_read(l, 4096, idx, key);
    int64_t m = orc_range_naive(t, 0.15, q, 0, NULL, NULL);
    bad += (n != m);
    orc_kd_empty_range_list(t, l);
  }
  orc_kd_destroy(t);
  /* wrapped 4-D tree */
  orc_kd *w = orc_kd_create(4);
  int wd[1] = {3}; double wp[1] = {2.0 * 3.141592653589793};
  orc_kd_set_wraps(w, 1, wd, wp);
  for (int i = 0; i < 3000; ++i) { double p[4] = {10 * frand(&seed), 10 * frand(&seed), 0.0, wp[0] * frand(&seed)}; orc_kd_insert(w, p); }
  for (int k = 0; k < 100; ++k) {
    double q[4] = {10 * frand(&seed), 10 * frand(&seed), 0.0, wp[0] * frand(&seed)};
    orc_list *l = orc_kd_find_within_range(w, 3.5, q);
    int64_t n = orc_list_length(l);
    int64_t m = orc_range_naive(w, 3.5, q, 0, NULL, NULL);
    bad += (n != m);
    orc_kd_empty_range_list(w, l);
    double g[8 * 4];
    (void)orc_ghost_points(w, q, 3.5, 8, g);
  }
  orc_kd_destroy(w);
  /* spheres, polygons, dubins */
  orc_sphere sp[4];
  for (int i = 0; i < 4; ++i) { sp[i].c[0] = 3.0 * i; sp[i].c[1] = 0; sp[i].c[2] = 0; sp[i].radius = 1.0; sp[i].life_span = INFINITY; sp[i].unused = (i == 2); sp[i].pad = 0; }
  double p0[3] = {-2, 1.4, 0}, p1[3] = {2, 1.4, 0}, z[3] = {10, 10, 10};
  int32_t fh;
  bad += orc_edge_check_spheres(sp, 4, p0, p1, 0.5, &fh) != 0;
  bad += orc_edge_check_spheres(sp, 4, z, z, 0.5, &fh) != 1;
  double clr; (void)orc_point_check_spheres(sp, 4, p0, 0.5, 1, &clr);
  double sq[8] = {0, 0, 1, 0, 1, 1, 0, 1};
  orc_polygon pg; pg.kind = 3; pg.nverts = 4; pg.verts = sq; pg.life_span = INFINITY; pg.unused = 0; pg.npath = 0; pg.path = 0;
  orc_polygon_ctor(sq, 4, &pg.cx, &pg.cy, &pg.radius);
  double e0[2] = {-1, .5}, e1[2] = {2, .5};
  bad += orc_edge_check_polygons(&pg, 1, e0, e1, 0.1, &fh) != 1;
  (void)orc_point_check_polygons(&pg, 1, e0, 0.1, &clr);
  /* moving obstacle (kind 6) + k nearest under the sanitizers */
  double mpath[6] = {0, 0, 0, 10, 0, 10};
  orc_polygon mv = pg; mv.kind = 6; mv.npath = 2; mv.path = mpath;
  double m0[3] = {5, -5, 0}, m1[3] = {5, 5, 10}, m2[3] = {5, -5, 20}, m3[3] = {5, 5, 30};
  bad += orc_edge_check_polygons(&mv, 1, m0, m1, 0.1, &fh) != 1;
  bad += orc_edge_check_polygons(&mv, 1, m2, m3, 6.0, &fh) != 0;
  double mp[3] = {5.5, 0.5, 5};
  bad += orc_point_check_polygons(&mv, 1, mp, 0.5, &clr) != 1;
  {
    orc_kd *kt = orc_kd_create(3);
    for (int i = 0; i < 500; ++i) { double p[3] = {frand(&seed), frand(&seed), frand(&seed)}; orc_kd_insert(kt, p); }
    int32_t ki[40]; double kd[40];
    for (int i = 0; i < 50; ++i) {
      double q[3] = {frand(&seed), frand(&seed), frand(&seed)};
      bad += orc_kd_knearest(kt, 1, q, 40, ki, kd) != 2;
      bad += orc_kd_knearest(kt, 33, q, 40, ki, kd) != 33;
      bad += orc_kd_knearest_naive(kt, 7, q, 40, ki, kd) != 7;
    }
    orc_kd_destroy(kt);
  }
  double traj[2 * 1024]; int tl; double cost; char word[4];
  for (int k = 0; k < 2000; ++k) {
    double s[4] = {20 * frand(&seed), 20 * frand(&seed), 0, 6.28 * frand(&seed)};
    double g[4] = {s[0] + 6 * (frand(&seed) - .5), s[1] + 6 * (frand(&seed) - .5), 0, 6.28 * frand(&seed)};
    orc_dubins_steer(s, g, 1.0, &cost, word, traj, 1024, &tl);
    bad += !(cost > 0) || tl > 1024;
    (void)orc_dubins_edge_check_polygons(&pg, 1, s, g, traj, tl, 0.5, 1.0, &fh);
  }
  printf(bad ? "selftest FAILED (%d)\n" : "selftest ok\n", bad);
  return bad != 0;
}
