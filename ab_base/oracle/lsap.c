/* oracle/lsap.c -- CPU restatement (TEST INFRASTRUCTURE) of the rectangular linear sum
 * assignment solver behind scipy.optimize.linear_sum_assignment (scipy 1.15.3,
 * scipy/optimize/rectangular_lsap/rectangular_lsap.cpp: D. F. Crouse, "On implementing 2D
 * rectangular assignment algorithms", IEEE TAES 52(4), 2016), the function the reference
 * calls at mmdet/core/bbox/assigners/gfl_hungarian_assigner.py:143-147 on a float32 CPU
 * tensor.  scipy's source is not in /root/reference; this file restates the published
 * algorithm and is pinned by differential tests against the installed scipy
 * (tests/test_oracle.py) and by frozen vectors (tests/golden/lsap_cases.npz).
 *
 * Plain C, no dependencies.  Build: `make -C oracle`  ->  oracle/_build/liboracle_lsap.so
 *
 * int oracle_lsap(const float* cost, int nr, int nc, long long* row, long long* col)
 *   returns 0, -3 (NaN / -inf entry: scipy "matrix contains invalid numeric entries")
 *   or -4 (scipy "cost matrix is infeasible"); writes min(nr,nc) pairs, rows ascending.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;

/* One shortest-augmenting-path search from row `start`.  Returns the sink column or -1. */
static int find_path(int nc, const double* c, const double* u, const double* v, int* pred,
                     const int* row_of_col, double* dist, int start, char* row_seen,
                     char* col_done, int* todo, double* out_min) {
  double best = 0.0;
  int n_todo = nc, sink = -1, i = start, k;
  for (k = 0; k < nc; ++k) { todo[k] = nc - 1 - k; col_done[k] = 0; dist[k] = INFINITY; }
  while (sink < 0) {
    int pick = -1;
    double low = INFINITY;
    row_seen[i] = 1;
    for (k = 0; k < n_todo; ++k) {
      const int j = todo[k];
      const double r = best + c[(size_t)i * nc + j] - u[i] - v[j];
      if (r < dist[j]) { dist[j] = r; pred[j] = i; }
      /* ties: an unassigned column replaces the current pick, an assigned one never does */
      if (dist[j] < low || (dist[j] == low && row_of_col[j] < 0)) { low = dist[j]; pick = k; }
    }
    best = low;
    if (best == INFINITY) return -1;
    {
      const int j = todo[pick];
      if (row_of_col[j] < 0) sink = j; else i = row_of_col[j];
      col_done[j] = 1;
      todo[pick] = todo[--n_todo];
    }
  }
  *out_min = best;
  return sink;
}

static const int* g_keys;
static int cmp_by_key(const void* a, const void* b) {
  const int x = g_keys[*(const int*)a], y = g_keys[*(const int*)b];
  return (x > y) - (x < y);
}

int oracle_lsap(const float* cost, int nr0, int nc0, i64* row, i64* col) {
  const int flip = nc0 < nr0;
  const int nr = flip ? nc0 : nr0, nc = flip ? nr0 : nc0;
  int i, j, cur, rc = 0;
  double *c, *u, *v, *dist;
  int *pred, *col_of_row, *row_of_col, *todo;
  char *row_seen, *col_done;
  if (nr0 == 0 || nc0 == 0) return 0;
  c = (double*)malloc(sizeof(double) * (size_t)nr * nc);
  for (i = 0; i < nr0; ++i)
    for (j = 0; j < nc0; ++j) {
      const double x = (double)cost[(size_t)i * nc0 + j];
      if (x != x || x == -INFINITY) { free(c); return -3; }
      if (flip) c[(size_t)j * nc + i] = x; else c[(size_t)i * nc + j] = x;
    }
  u = (double*)calloc(nr, sizeof(double));
  v = (double*)calloc(nc, sizeof(double));
  dist = (double*)malloc(sizeof(double) * nc);
  pred = (int*)malloc(sizeof(int) * nc);
  col_of_row = (int*)malloc(sizeof(int) * nr);
  row_of_col = (int*)malloc(sizeof(int) * nc);
  todo = (int*)malloc(sizeof(int) * nc);
  row_seen = (char*)malloc(nr);
  col_done = (char*)malloc(nc);
  for (i = 0; i < nr; ++i) col_of_row[i] = -1;
  for (j = 0; j < nc; ++j) { row_of_col[j] = -1; pred[j] = -1; }

  for (cur = 0; cur < nr && rc == 0; ++cur) {
    double m = 0.0;
    int sink;
    memset(row_seen, 0, nr);
    sink = find_path(nc, c, u, v, pred, row_of_col, dist, cur, row_seen, col_done, todo, &m);
    if (sink < 0) { rc = -4; break; }
    u[cur] += m;
    for (i = 0; i < nr; ++i)
      if (row_seen[i] && i != cur) u[i] += m - dist[col_of_row[i]];
    for (j = 0; j < nc; ++j)
      if (col_done[j]) v[j] -= m - dist[j];
    j = sink;
    for (;;) {
      const int r = pred[j];
      const int next = col_of_row[r];
      row_of_col[j] = r;
      col_of_row[r] = j;
      j = next;
      if (r == cur) break;
    }
  }
  if (rc == 0) {
    if (flip) {
      int* order = (int*)malloc(sizeof(int) * nr);
      for (i = 0; i < nr; ++i) order[i] = i;
      g_keys = col_of_row;
      qsort(order, nr, sizeof(int), cmp_by_key);
      for (i = 0; i < nr; ++i) { row[i] = col_of_row[order[i]]; col[i] = order[i]; }
      free(order);
    } else {
      for (i = 0; i < nr; ++i) { row[i] = i; col[i] = col_of_row[i]; }
    }
  }
  free(c); free(u); free(v); free(dist); free(pred); free(col_of_row); free(row_of_col);
  free(todo); free(row_seen); free(col_done);
  return rc;
}
