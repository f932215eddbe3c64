/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py) — plain-C, fp64 evaluation of the pairwise
 * manifold-distance path, used as the full-size checker of the GPU kernels.
 *
 * It restates the MATHEMATICS of the reference path (citations relative to
 * /root/reference/graphembed/graphembed/), not its eps-fudged fp shortcuts:
 *   SPD      d2_ij = sum_k log^2 lambda_k(L_i^-1 X_j L_i^-T)          manifolds/spd.py:175-181,163-169
 *            d d2/dX_j =  2 L_i^-T V diag(log w / w) V^T L_i^-1,
 *            d d2/dX_i = -2 L_i^-T V diag(log w)     V^T L_i^-1       (what autograd yields, symmetrised)
 *   Lorentz  d = acosh(max(-<x,y>_L, 1))                               manifolds/lorentz.py:72-77
 *   Sphere   d = acos(clamp <x,y>)                                     manifolds/sphere.py:68-74
 *   Euclid   d2 = sum (y-x)^2                                          manifolds/base.py:29-33,56-57
 * with the reference's value-only clamps (w in [wmin,wmax], d2 >= wmin / d >= 1e-8).
 * Pinned by tests/test_oracle_exact.py against oracle/ref_port.py in fp64 (itself pinned against
 * the reference's golden vectors) — agreement to the reference's own eps bias.
 *
 * Eigen-decomposition: cyclic Jacobi to machine precision.  OpenMP over rows.
 * Build: make -C oracle   ->  oracle/_build/liboracle_exact.so
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DMAX 9

static void jacobi(int d, double a[DMAX][DMAX], double v[DMAX][DMAX], double w[DMAX]) {
  for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) v[r][c] = (r == c);
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0;
    for (int p = 0; p < d; ++p) for (int q = p + 1; q < d; ++q) off += fabs(a[p][q]);
    if (off == 0.0) break;
    int rotated = 0;
    for (int p = 0; p < d - 1; ++p) for (int q = p + 1; q < d; ++q) {
      const double apq = a[p][q];
      if (fabs(apq) <= 1e-300 || fabs(apq) <= 1e-18 * sqrt(fabs(a[p][p] * a[q][q]))) { a[p][q] = a[q][p] = 0; continue; }
      rotated = 1;
      const double theta = (a[q][q] - a[p][p]) / (2 * apq);
      const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
      const double c = 1 / sqrt(t * t + 1), s = t * c;
      a[p][p] -= t * apq; a[q][q] += t * apq; a[p][q] = a[q][p] = 0;
      for (int r = 0; r < d; ++r) if (r != p && r != q) {
        const double arp = a[r][p], arq = a[r][q];
        a[r][p] = a[p][r] = c * arp - s * arq;
        a[r][q] = a[q][r] = s * arp + c * arq;
      }
      for (int r = 0; r < d; ++r) {
        const double vrp = v[r][p], vrq = v[r][q];
        v[r][p] = c * vrp - s * vrq; v[r][q] = s * vrp + c * vrq;
      }
    }
    if (!rotated) break;
  }
  for (int k = 0; k < d; ++k) w[k] = a[k][k];
}

/* Linv (lower) of sym(X); returns 0 if not positive definite */
static int inv_chol(int d, const double* x, double li[DMAX][DMAX], double xs[DMAX][DMAX]) {
  double l[DMAX][DMAX];
  memset(l, 0, sizeof(l)); memset(li, 0, sizeof(double) * DMAX * DMAX);
  for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) xs[r][c] = 0.5 * (x[r * d + c] + x[c * d + r]);
  for (int j = 0; j < d; ++j) {
    double s = xs[j][j];
    for (int k = 0; k < j; ++k) s -= l[j][k] * l[j][k];
    if (!(s > 0)) return 0;
    l[j][j] = sqrt(s);
    for (int i = j + 1; i < d; ++i) {
      double t = xs[i][j];
      for (int k = 0; k < j; ++k) t -= l[i][k] * l[j][k];
      l[i][j] = t / l[j][j];
    }
  }
  for (int j = 0; j < d; ++j) {
    li[j][j] = 1 / l[j][j];
    for (int i = j + 1; i < d; ++i) {
      double s = 0;
      for (int k = j; k < i; ++k) s += l[i][k] * li[k][j];
      li[i][j] = -s / l[i][i];
    }
  }
  return 1;
}

static long pair_off(long n, long i) { return i * (2 * n - i - 1) / 2; }

/* per-pair decomposition of A = Li Xj Li^T */
static double pair_eig(int d, double li[DMAX][DMAX], double xj[DMAX][DMAX], double wmin, double wmax,
                       double v[DMAX][DMAX], double w[DMAX], double lw[DMAX], double* rho) {
  double b[DMAX][DMAX], a[DMAX][DMAX];
  for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) { double s = 0; for (int k = 0; k <= r; ++k) s += li[r][k] * xj[k][c]; b[r][c] = s; }
  for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) { double s = 0; for (int k = 0; k <= c; ++k) s += b[r][k] * li[c][k]; a[r][c] = s; }
  for (int r = 0; r < d; ++r) for (int c = r + 1; c < d; ++c) a[r][c] = a[c][r] = 0.5 * (a[r][c] + a[c][r]);
  jacobi(d, a, v, w);
  double s = 0;
  /* the reference clamps the VALUES of the eigenvalues in place (w.data.clamp_, spd.py:163-169): log's backward then divides by the
   * clamped value while the eigenvalue decomposition's backward runs on the true A.  rho = true / clamped (1 unless a clamp binds)
   * is what that leaves on the X_i side of the gradient; the X_j side has 1 / clamped. */
  for (int k = 0; k < d; ++k) {
    const double wt = w[k];
    w[k] = fmin(fmax(w[k], wmin), wmax); lw[k] = log(w[k]); s += lw[k] * lw[k];
    if (rho) rho[k] = (w[k] == wt) ? 1.0 : wt / w[k];
  }
  return s;
}

int oracle_spd_pdist(const double* x, long n, int d, int squared, double wmin, double wmax, double* out) {
  if (d > DMAX) return -2;
  double (*li)[DMAX][DMAX] = malloc(sizeof(double[DMAX][DMAX]) * n);
  double (*xs)[DMAX][DMAX] = malloc(sizeof(double[DMAX][DMAX]) * n);
  int bad = 0;
  for (long i = 0; i < n; ++i) if (!inv_chol(d, x + i * d * d, li[i], xs[i])) bad = 1;
  if (!bad) {
#pragma omp parallel for schedule(dynamic, 4)
    for (long i = 0; i < n; ++i) {
      double v[DMAX][DMAX], w[DMAX], lw[DMAX];
      for (long j = i + 1; j < n; ++j) {
        double s = fmax(pair_eig(d, li[i], xs[j], wmin, wmax, v, w, lw, NULL), wmin);
        out[pair_off(n, i) + (j - i - 1)] = squared ? s : sqrt(s);
      }
    }
  }
  free(li); free(xs);
  return bad ? -3 : 0;
}

int oracle_spd_pdist_grad(const double* x, const double* g, long n, int d, int squared, double wmin, double wmax,
                          double* grad) {
  if (d > DMAX) return -2;
  double (*li)[DMAX][DMAX] = malloc(sizeof(double[DMAX][DMAX]) * n);
  double (*xs)[DMAX][DMAX] = malloc(sizeof(double[DMAX][DMAX]) * n);
  int bad = 0;
  for (long i = 0; i < n; ++i) if (!inv_chol(d, x + i * d * d, li[i], xs[i])) bad = 1;
  memset(grad, 0, sizeof(double) * n * d * d);
  if (!bad) {
#pragma omp parallel
    {
      double* loc = calloc((size_t)n * d * d, sizeof(double));
#pragma omp for schedule(dynamic, 4)
      for (long i = 0; i < n; ++i) {
        double v[DMAX][DMAX], w[DMAX], lw[DMAX], rho[DMAX], m[DMAX][DMAX], nn[DMAX][DMAX], t[DMAX][DMAX];
        for (long j = i + 1; j < n; ++j) {
          const double s = pair_eig(d, li[i], xs[j], wmin, wmax, v, w, lw, rho);
          double gs = g[pair_off(n, i) + (j - i - 1)];
          if (!squared) gs *= 0.5 / sqrt(fmax(s, wmin));
          for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
            double a = 0, b = 0;
            for (int k = 0; k < d; ++k) { a += v[r][k] * (2 * gs * lw[k] * rho[k]) * v[c][k]; b += v[r][k] * (2 * gs * lw[k] / w[k]) * v[c][k]; }
            m[r][c] = a; nn[r][c] = b;
          }
          /* grad_i -= Li^T M Li ; grad_j += Li^T N Li */
          for (int pass = 0; pass < 2; ++pass) {
            double (*src)[DMAX] = pass ? nn : m;
            for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) { double a = 0; for (int k = r; k < d; ++k) a += li[i][k][r] * src[k][c]; t[r][c] = a; }
            double* dst = loc + (pass ? j : i) * d * d;
            for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
              double a = 0; for (int k = c; k < d; ++k) a += t[r][k] * li[i][k][c];
              dst[r * d + c] += pass ? a : -a;
            }
          }
        }
      }
#pragma omp critical
      for (size_t k = 0; k < (size_t)n * d * d; ++k) grad[k] += loc[k];
      free(loc);
    }
  }
  free(li); free(xs);
  return bad ? -3 : 0;
}

/* kind: 0 Euclidean, 1 Lorentz, 2 Sphere */
static double vec_q(int kind, const double* a, const double* b, int m) {
  double q = 0;
  if (kind == 0) { for (int k = 0; k < m; ++k) q += (b[k] - a[k]) * (b[k] - a[k]); }
  else if (kind == 1) { for (int k = 1; k < m; ++k) q += a[k] * b[k]; q = a[0] * b[0] - q; }
  else { for (int k = 0; k < m; ++k) q += a[k] * b[k]; }
  return q;
}
static double vec_val(int kind, double q, int squared, double* dq) {
  if (kind == 0) { const double s = fmax(q, 1e-8); *dq = squared ? 1 : 0.5 / sqrt(s); return squared ? s : sqrt(s); }
  if (kind == 1) {
    const double t = fmax(q, 1), z = sqrt(t * t - 1), d = fmax(log(t + z), 1e-8);
    *dq = (squared ? 2 * d : 1) / fmax(z, 1e-8);
    return squared ? d * d : d;
  }
  const double c = fmin(fmax(q, -1 + 1e-16), 1 - 1e-16), th = fmax(acos(c), 1e-8);
  *dq = -(squared ? 2 * th : 1) / fmax(sqrt(1 - c * c), 1e-8);
  return squared ? th * th : th;
}

int oracle_vec_pdist(int kind, const double* x, long n, int m, int squared, double* out) {
#pragma omp parallel for schedule(dynamic, 16)
  for (long i = 0; i < n; ++i)
    for (long j = i + 1; j < n; ++j) { double dq; out[pair_off(n, i) + (j - i - 1)] = vec_val(kind, vec_q(kind, x + i * m, x + j * m, m), squared, &dq); }
  return 0;
}

int oracle_vec_pdist_grad(int kind, const double* x, const double* g, long n, int m, int squared, double* grad) {
  memset(grad, 0, sizeof(double) * n * m);
#pragma omp parallel
  {
    double* loc = calloc((size_t)n * m, sizeof(double));
#pragma omp for schedule(dynamic, 16)
    for (long i = 0; i < n; ++i)
      for (long j = i + 1; j < n; ++j) {
        double dq;
        vec_val(kind, vec_q(kind, x + i * m, x + j * m, m), squared, &dq);
        const double w = g[pair_off(n, i) + (j - i - 1)] * dq;
        for (int k = 0; k < m; ++k) {
          const double xi = x[i * m + k], xj = x[j * m + k];
          double ai, aj;
          if (kind == 0) { ai = 2 * (xi - xj); aj = -ai; }
          else if (kind == 1) { ai = k ? -xj : xj; aj = k ? -xi : xi; }
          else { ai = xj; aj = xi; }
          loc[i * m + k] += w * ai; loc[j * m + k] += w * aj;
        }
      }
#pragma omp critical
    for (size_t k = 0; k < (size_t)n * m; ++k) grad[k] += loc[k];
    free(loc);
  }
  return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
