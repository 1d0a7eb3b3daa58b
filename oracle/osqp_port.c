/* ORACLE (test infrastructure, never shipped, never on the product path).
 *
 * Plain-C restatement of
 *   (1) the QP assembly of the reference's MPC._init_problem  (/root/reference/src/MPC.py:61-155,
 *       /root/reference/src/spatial_bicycle_models.py:391-417), and
 *   (2) the OSQP algorithm the reference calls at /root/reference/src/MPC.py:158-159,183
 *       (third-party, un-vendored, unpinned: README.md:62-68; restated from the published
 *       algorithm, Stellato et al., Math. Prog. Comp. 2020, with the 0.6.x defaults): Ruiz
 *       equilibration, sparse quasi-definite LDL' of the KKT matrix, ADMM with relaxation and
 *       residual-balancing rho adaptation, termination / infeasibility tests,
 *   plus the certified polish (interior-point refinement, iterated active-set solve with
 *   extended-precision residuals, KKT certificate) described in oracle/osqp_np.py, whose
 *   sparse twin this file is.
 *
 * PARITY UNPINNED at the solver boundary (no OSQP here, no reference tests); the KKT certificate
 * is what pins results.  Used by tests/ as the checker and by bench.py as the timed CPU baseline
 * ("port", OpenMP over instances).  The product library never links or loads this.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define OSQP_INFTY 1e30
#define MIN_SCALING 1e-4
#define MAX_SCALING 1e4
#define RHO_MIN 1e-6
#define RHO_MAX 1e6
#define RHO_TOL 1e-4
#define RHO_EQ_OVER_RHO_INEQ 1e3
#define INF_BOUND (OSQP_INFTY * MIN_SCALING)

enum { SOLVED = 1, SOLVED_INACCURATE = 2, MAX_ITER_REACHED = -2, PRIMAL_INFEASIBLE = -3, DUAL_INFEASIBLE = -4, UNSOLVED = -10 };

typedef struct {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int32_t max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval;
  double adaptive_rho_tolerance;
  int32_t polish, ipm_max_iter;
  double ipm_tol, ipm_reg, as_delta;
  int32_t as_refine, as_rounds;
  double cert_tol;
  int32_t early_polish, early_scaling, phase1;
  double ipm_diverged, phase1_theta, phase1_eps;
  double ipm_start_slack, ipm_start_mu;    /* centred start of the early interior-point attempt (see osqp_np.Settings) */
  double ipm_start_dual;                   /* mu0 = max(ipm_start_mu, ipm_start_dual * ipm_start_slack * |P x + q|_inf) */
  double as_add_fraction;                  /* active-set rounds add only violations >= this fraction of the worst */
} oracle_settings;

typedef struct {
  int32_t status, iters, ipm_iters, as_rounds, polished, rho_updates;
  double pri_res, dua_res, obj;
} oracle_info;

/* ------------------------------------------------------------------ small helpers */
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }
static double ninf(const double* v, int n) { double m = 0; for (int i = 0; i < n; ++i) m = dmax(m, fabs(v[i])); return m; }
static double ninf_s(const double* s, const double* v, int n) { double m = 0; for (int i = 0; i < n; ++i) m = dmax(m, fabs(s[i] * v[i])); return m; }
static double limit_scaling(double v) { v = v < MIN_SCALING ? 1.0 : v; return v > MAX_SCALING ? MAX_SCALING : v; }

/* CSC general matrix (m x n) */
typedef struct { int m, n; int* p; int* i; double* x; } csc;

static void csc_mul(const csc* A, const double* x, double* y) { /* y = A x */
  memset(y, 0, sizeof(double) * A->m);
  for (int j = 0; j < A->n; ++j) for (int k = A->p[j]; k < A->p[j + 1]; ++k) y[A->i[k]] += A->x[k] * x[j];
}
static void csc_tmul(const csc* A, const double* y, double* x) { /* x = A' y */
  for (int j = 0; j < A->n; ++j) { double s = 0; for (int k = A->p[j]; k < A->p[j + 1]; ++k) s += A->x[k] * y[A->i[k]]; x[j] = s; }
}
/* symmetric, upper triangle stored */
static void sym_mul(const csc* P, const double* x, double* y) {
  memset(y, 0, sizeof(double) * P->n);
  for (int j = 0; j < P->n; ++j)
    for (int k = P->p[j]; k < P->p[j + 1]; ++k) {
      int i = P->i[k];
      y[i] += P->x[k] * x[j];
      if (i != j) y[j] += P->x[k] * x[i];
    }
}

/* ------------------------------------------------------------------ sparse LDL' (own code)
 * K symmetric quasi-definite, upper triangle in CSC (after permutation).  Up-looking
 * factorisation driven by the elimination tree. */
typedef struct {
  int n;
  int *perm, *iperm;        /* perm[new] = old */
  int *Kp, *Ki; double* Kx; /* permuted upper triangle */
  int* map;                 /* entry k of the caller's upper-triangular K -> position in Kx */
  int *parent, *Lp, *Li; double *Lx, *D;
  int *flag, *pattern, *lnz; double* y; double* tmp;
} ldl_t;

static void ldl_free(ldl_t* f) {
  if (!f) return;
  free(f->perm); free(f->iperm); free(f->Kp); free(f->Ki); free(f->Kx); free(f->map); free(f->parent);
  free(f->Lp); free(f->Li); free(f->Lx); free(f->D); free(f->flag); free(f->pattern); free(f->lnz); free(f->y); free(f->tmp);
  free(f);
}

/* greedy minimum-degree ordering on a dense adjacency bitmap (N <= a few thousand) */
static void min_degree(int N, const int* Kp, const int* Ki, int* perm) {
  unsigned char* adj = (unsigned char*)calloc((size_t)N * N, 1);
  int* deg = (int*)calloc(N, sizeof(int));
  unsigned char* gone = (unsigned char*)calloc(N, 1);
  int* nb = (int*)malloc(sizeof(int) * N);
  for (int j = 0; j < N; ++j)
    for (int k = Kp[j]; k < Kp[j + 1]; ++k) {
      int i = Ki[k];
      if (i != j && !adj[(size_t)i * N + j]) { adj[(size_t)i * N + j] = adj[(size_t)j * N + i] = 1; deg[i]++; deg[j]++; }
    }
  for (int step = 0; step < N; ++step) {
    int best = -1;
    for (int v = 0; v < N; ++v) if (!gone[v] && (best < 0 || deg[v] < deg[best])) best = v;
    perm[step] = best;
    gone[best] = 1;
    int cnt = 0;
    for (int v = 0; v < N; ++v) if (!gone[v] && adj[(size_t)best * N + v]) nb[cnt++] = v;
    for (int a = 0; a < cnt; ++a) {
      int u = nb[a];
      adj[(size_t)u * N + best] = 0; deg[u]--;
      for (int b = a + 1; b < cnt; ++b) {
        int w = nb[b];
        if (!adj[(size_t)u * N + w]) { adj[(size_t)u * N + w] = adj[(size_t)w * N + u] = 1; deg[u]++; deg[w]++; }
      }
    }
  }
  free(adj); free(deg); free(gone); free(nb);
}

/* symbolic analysis of upper-triangular K (pattern only); perm_in may be NULL (computed) */
static ldl_t* ldl_analyse(int N, const int* Kp, const int* Ki, const int* perm_in) {
  ldl_t* f = (ldl_t*)calloc(1, sizeof(ldl_t));
  f->n = N;
  f->perm = (int*)malloc(sizeof(int) * N);
  f->iperm = (int*)malloc(sizeof(int) * N);
  if (perm_in) memcpy(f->perm, perm_in, sizeof(int) * N); else min_degree(N, Kp, Ki, f->perm);
  for (int k = 0; k < N; ++k) f->iperm[f->perm[k]] = k;
  int nnz = Kp[N];
  /* permuted upper triangle: entry (i,j) -> (min(pi,pj), max(pi,pj)) */
  f->Kp = (int*)calloc(N + 1, sizeof(int));
  f->Ki = (int*)malloc(sizeof(int) * (nnz > 0 ? nnz : 1));
  f->Kx = (double*)calloc(nnz > 0 ? nnz : 1, sizeof(double));
  f->map = (int*)malloc(sizeof(int) * (nnz > 0 ? nnz : 1));
  int* cnt = (int*)calloc(N + 1, sizeof(int));
  for (int j = 0; j < N; ++j)
    for (int k = Kp[j]; k < Kp[j + 1]; ++k) {
      int a = f->iperm[Ki[k]], b = f->iperm[j];
      cnt[(a > b ? a : b) + 1]++;
    }
  for (int j = 0; j < N; ++j) f->Kp[j + 1] = f->Kp[j] + cnt[j + 1];
  memcpy(cnt, f->Kp, sizeof(int) * (N + 1));
  for (int j = 0; j < N; ++j)
    for (int k = Kp[j]; k < Kp[j + 1]; ++k) {
      int a = f->iperm[Ki[k]], b = f->iperm[j];
      int col = a > b ? a : b, row = a > b ? b : a;
      int pos = cnt[col]++;
      f->Ki[pos] = row;
      f->map[k] = pos;
    }
  free(cnt);
  /* elimination tree and column counts */
  f->parent = (int*)malloc(sizeof(int) * N);
  f->lnz = (int*)calloc(N, sizeof(int));
  f->flag = (int*)malloc(sizeof(int) * N);
  for (int k = 0; k < N; ++k) {
    f->parent[k] = -1;
    f->flag[k] = k;
    for (int p = f->Kp[k]; p < f->Kp[k + 1]; ++p) {
      int i = f->Ki[p];
      while (i < k && f->flag[i] != k) {
        if (f->parent[i] == -1) f->parent[i] = k;
        f->lnz[i]++;
        f->flag[i] = k;
        i = f->parent[i];
      }
    }
  }
  f->Lp = (int*)malloc(sizeof(int) * (N + 1));
  f->Lp[0] = 0;
  for (int k = 0; k < N; ++k) f->Lp[k + 1] = f->Lp[k] + f->lnz[k];
  int lnnz = f->Lp[N];
  f->Li = (int*)malloc(sizeof(int) * (lnnz > 0 ? lnnz : 1));
  f->Lx = (double*)malloc(sizeof(double) * (lnnz > 0 ? lnnz : 1));
  f->D = (double*)malloc(sizeof(double) * N);
  f->pattern = (int*)malloc(sizeof(int) * N);
  f->y = (double*)calloc(N, sizeof(double));
  f->tmp = (double*)malloc(sizeof(double) * N);
  return f;
}

/* numeric factorisation; Kx_user = values of the caller's upper-triangular K in its own order */
static int ldl_factor(ldl_t* f, const double* Kx_user, int nnz) {
  int N = f->n;
  for (int k = 0; k < nnz; ++k) f->Kx[f->map[k]] = 0.0;
  for (int k = 0; k < nnz; ++k) f->Kx[f->map[k]] += Kx_user[k];
  for (int k = 0; k < N; ++k) { f->lnz[k] = 0; f->y[k] = 0.0; f->flag[k] = -1; }
  for (int k = 0; k < N; ++k) {
    int top = N;
    f->flag[k] = k;
    double dk = 0.0;
    for (int p = f->Kp[k]; p < f->Kp[k + 1]; ++p) {
      int i = f->Ki[p];
      if (i == k) { dk += f->Kx[p]; continue; }
      f->y[i] += f->Kx[p];
      int len = 0;
      for (; f->flag[i] != k; i = f->parent[i]) { f->pattern[len++] = i; f->flag[i] = k; }
      while (len > 0) f->pattern[--top] = f->pattern[--len];
    }
    for (; top < N; ++top) {
      int i = f->pattern[top];
      double yi = f->y[i];
      f->y[i] = 0.0;
      int p2 = f->Lp[i] + f->lnz[i];
      for (int p = f->Lp[i]; p < p2; ++p) f->y[f->Li[p]] -= f->Lx[p] * yi;
      double lki = yi / f->D[i];
      dk -= lki * yi;
      f->Li[p2] = k;
      f->Lx[p2] = lki;
      f->lnz[i]++;
    }
    if (dk == 0.0) return -1;
    f->D[k] = dk;
  }
  return 0;
}

static void ldl_solve(const ldl_t* f, const double* b, double* x) {
  int N = f->n;
  double* t = f->tmp;
  for (int k = 0; k < N; ++k) t[k] = b[f->perm[k]];
  for (int j = 0; j < N; ++j) for (int p = f->Lp[j]; p < f->Lp[j] + f->lnz[j]; ++p) t[f->Li[p]] -= f->Lx[p] * t[j];
  for (int j = 0; j < N; ++j) t[j] /= f->D[j];
  for (int j = N - 1; j >= 0; --j) for (int p = f->Lp[j]; p < f->Lp[j] + f->lnz[j]; ++p) t[j] -= f->Lx[p] * t[f->Li[p]];
  for (int k = 0; k < N; ++k) x[f->perm[k]] = t[k];
}

/* ------------------------------------------------------------------ KKT  [P + s I, A'; A, -diag(d)]
 * upper triangle, CSC, columns: n of x then m of rows.  Positions of the diagonal entries are
 * remembered so sigma / rho changes only rewrite values. */
typedef struct {
  int N, nnz; int *p, *i; double* x;
  int* Pdiag_pos;   /* position of (j,j) for j < n */
  int* Ddiag_pos;   /* position of (n+r, n+r) */
  int* Apos;        /* KKT position of A entry k */
  int* Ppos;        /* KKT position of P entry k */
  int *At_p, *At_i, *At_k; /* row-wise view of A: for row r entries (col, k) */
} kkt_t;

static void kkt_free(kkt_t* K) { if (!K) return; free(K->p); free(K->i); free(K->x); free(K->Pdiag_pos); free(K->Ddiag_pos); free(K->Apos); free(K->Ppos); free(K->At_p); free(K->At_i); free(K->At_k); free(K); }

static kkt_t* kkt_build(const csc* P, const csc* A, const int* rows, int nrows) {
  /* rows: subset of A's rows used (NULL = all); KKT dimension n + nrows */
  int n = P->n, m = A->m;
  int* rmap = (int*)malloc(sizeof(int) * m);
  for (int r = 0; r < m; ++r) rmap[r] = rows ? -1 : r;
  if (rows) for (int k = 0; k < nrows; ++k) rmap[rows[k]] = k; else nrows = m;
  kkt_t* K = (kkt_t*)calloc(1, sizeof(kkt_t));
  K->N = n + nrows;
  /* row-wise A */
  K->At_p = (int*)calloc(m + 1, sizeof(int));
  int annz = A->p[n];
  K->At_i = (int*)malloc(sizeof(int) * (annz > 0 ? annz : 1));
  K->At_k = (int*)malloc(sizeof(int) * (annz > 0 ? annz : 1));
  for (int k = 0; k < annz; ++k) K->At_p[A->i[k] + 1]++;
  for (int r = 0; r < m; ++r) K->At_p[r + 1] += K->At_p[r];
  int* fill = (int*)malloc(sizeof(int) * (m + 1));
  memcpy(fill, K->At_p, sizeof(int) * (m + 1));
  for (int j = 0; j < n; ++j) for (int k = A->p[j]; k < A->p[j + 1]; ++k) { int pos = fill[A->i[k]]++; K->At_i[pos] = j; K->At_k[pos] = k; }
  free(fill);
  int pnnz = P->p[n];
  int cap = pnnz + n + annz + nrows;
  K->p = (int*)calloc(K->N + 1, sizeof(int));
  K->i = (int*)malloc(sizeof(int) * cap);
  K->x = (double*)calloc(cap, sizeof(double));
  K->Pdiag_pos = (int*)malloc(sizeof(int) * n);
  K->Ddiag_pos = (int*)malloc(sizeof(int) * (nrows > 0 ? nrows : 1));
  K->Apos = (int*)malloc(sizeof(int) * (annz > 0 ? annz : 1));
  K->Ppos = (int*)malloc(sizeof(int) * (pnnz > 0 ? pnnz : 1));
  for (int k = 0; k < annz; ++k) K->Apos[k] = -1;
  int pos = 0;
  for (int j = 0; j < n; ++j) {
    int have_diag = 0;
    for (int k = P->p[j]; k < P->p[j + 1]; ++k) {
      K->i[pos] = P->i[k];
      K->Ppos[k] = pos;
      if (P->i[k] == j) { have_diag = 1; K->Pdiag_pos[j] = pos; }
      pos++;
    }
    if (!have_diag) { K->i[pos] = j; K->Pdiag_pos[j] = pos; pos++; }
    K->p[j + 1] = pos;
  }
  for (int r = 0; r < m; ++r) {
    if (rmap[r] < 0) continue;
    int c = n + rmap[r];
    for (int q = K->At_p[r]; q < K->At_p[r + 1]; ++q) { K->i[pos] = K->At_i[q]; K->Apos[K->At_k[q]] = pos; pos++; }
    K->i[pos] = c; K->Ddiag_pos[rmap[r]] = pos; pos++;
    K->p[c + 1] = pos;
  }
  /* columns of the row block must be filled in increasing c: rows subset given in order */
  K->nnz = pos;
  free(rmap);
  return K;
}

/* ------------------------------------------------------------------ workspace */
typedef struct {
  int n, m;
  csc P, A;                 /* scaled copies */
  double *q, *l, *u;        /* scaled */
  const csc *P0, *A0; const double *q0, *l0, *u0; /* unscaled originals (bounds clipped into l0c/u0c) */
  double *l0c, *u0c;
  double *D, *E, *Dinv, *Einv, c, cinv;
  double rho, *rho_vec, *rho_inv;
  int* ctype;
  kkt_t* K; ldl_t* F;
  const oracle_settings* st;
} work_t;

static void copy_csc(csc* dst, const csc* src) {
  int nnz = src->p[src->n];
  dst->m = src->m; dst->n = src->n;
  dst->p = (int*)malloc(sizeof(int) * (src->n + 1)); memcpy(dst->p, src->p, sizeof(int) * (src->n + 1));
  dst->i = (int*)malloc(sizeof(int) * (nnz > 0 ? nnz : 1)); memcpy(dst->i, src->i, sizeof(int) * nnz);
  dst->x = (double*)malloc(sizeof(double) * (nnz > 0 ? nnz : 1)); memcpy(dst->x, src->x, sizeof(double) * nnz);
}

/* `passes` Ruiz sweeps; resumable: the bounds are always recomputed from the unscaled copies */
static void scale_data(work_t* w, int passes) {
  int n = w->n, m = w->m;
  double* Dt = (double*)malloc(sizeof(double) * n);
  double* Et = (double*)malloc(sizeof(double) * m);
  for (int it = 0; it < passes; ++it) {
    for (int j = 0; j < n; ++j) Dt[j] = 0;
    for (int r = 0; r < m; ++r) Et[r] = 0;
    for (int j = 0; j < n; ++j)
      for (int k = w->P.p[j]; k < w->P.p[j + 1]; ++k) {
        double a = fabs(w->P.x[k]); int i = w->P.i[k];
        Dt[j] = dmax(Dt[j], a); if (i != j) Dt[i] = dmax(Dt[i], a);
      }
    for (int j = 0; j < n; ++j)
      for (int k = w->A.p[j]; k < w->A.p[j + 1]; ++k) {
        double a = fabs(w->A.x[k]);
        Dt[j] = dmax(Dt[j], a); Et[w->A.i[k]] = dmax(Et[w->A.i[k]], a);
      }
    for (int j = 0; j < n; ++j) Dt[j] = 1.0 / sqrt(limit_scaling(Dt[j]));
    for (int r = 0; r < m; ++r) Et[r] = 1.0 / sqrt(limit_scaling(Et[r]));
    for (int j = 0; j < n; ++j) for (int k = w->P.p[j]; k < w->P.p[j + 1]; ++k) w->P.x[k] = (Dt[w->P.i[k]] * w->P.x[k]) * Dt[j];
    for (int j = 0; j < n; ++j) for (int k = w->A.p[j]; k < w->A.p[j + 1]; ++k) w->A.x[k] = (Et[w->A.i[k]] * w->A.x[k]) * Dt[j];
    for (int j = 0; j < n; ++j) { w->q[j] *= Dt[j]; w->D[j] *= Dt[j]; }
    for (int r = 0; r < m; ++r) w->E[r] *= Et[r];
    /* cost normalisation: mean column inf-norm of P vs ||q||_inf */
    for (int j = 0; j < n; ++j) Dt[j] = 0;
    for (int j = 0; j < n; ++j)
      for (int k = w->P.p[j]; k < w->P.p[j + 1]; ++k) {
        double a = fabs(w->P.x[k]); int i = w->P.i[k];
        Dt[j] = dmax(Dt[j], a); if (i != j) Dt[i] = dmax(Dt[i], a);
      }
    double mean = 0; for (int j = 0; j < n; ++j) mean += Dt[j]; mean /= n;
    double ct = dmax(mean, limit_scaling(ninf(w->q, n)));
    ct = 1.0 / limit_scaling(ct);
    int pnnz = w->P.p[n];
    for (int k = 0; k < pnnz; ++k) w->P.x[k] *= ct;
    for (int j = 0; j < n; ++j) w->q[j] *= ct;
    w->c *= ct;
  }
  for (int r = 0; r < m; ++r) { w->l[r] = w->l0c[r] * w->E[r]; w->u[r] = w->u0c[r] * w->E[r]; }
  free(Dt); free(Et);
}

static void set_rho_vec(work_t* w) {
  for (int r = 0; r < w->m; ++r) {
    int lo_inf = w->l[r] < -INF_BOUND, up_inf = w->u[r] > INF_BOUND;
    if (lo_inf && up_inf) { w->ctype[r] = -1; w->rho_vec[r] = RHO_MIN; }
    else if (w->u[r] - w->l[r] < RHO_TOL) { w->ctype[r] = 1; w->rho_vec[r] = RHO_EQ_OVER_RHO_INEQ * w->rho; }
    else { w->ctype[r] = 0; w->rho_vec[r] = w->rho; }
    w->rho_inv[r] = 1.0 / w->rho_vec[r];
  }
}

static int kkt_fill_and_factor(work_t* w, kkt_t* K, ldl_t* F, double sigma, const double* dvals /* per KKT row block */, int nrows) {
  int n = w->n;
  memset(K->x, 0, sizeof(double) * K->nnz);
  int pnnz = w->P.p[n];
  for (int k = 0; k < pnnz; ++k) K->x[K->Ppos[k]] += w->P.x[k];
  for (int j = 0; j < n; ++j) K->x[K->Pdiag_pos[j]] += sigma;
  int annz = w->A.p[n];
  for (int k = 0; k < annz; ++k) if (K->Apos[k] >= 0) K->x[K->Apos[k]] = w->A.x[k];
  for (int r = 0; r < nrows; ++r) K->x[K->Ddiag_pos[r]] = -dvals[r];
  return ldl_factor(F, K->x, K->nnz);
}

typedef struct { double *Ax, *Px, *Aty, *rp, *rd; double pri, dua; } info_t;

static void compute_info(const work_t* w, const double* x, const double* z, const double* y, info_t* o) {
  int n = w->n, m = w->m;
  csc_mul(&w->A, x, o->Ax); sym_mul(&w->P, x, o->Px); csc_tmul(&w->A, y, o->Aty);
  for (int r = 0; r < m; ++r) o->rp[r] = o->Ax[r] - z[r];
  for (int j = 0; j < n; ++j) o->rd[j] = o->Px[j] + w->q[j] + o->Aty[j];
  o->pri = ninf_s(w->Einv, o->rp, m);
  o->dua = w->cinv * ninf_s(w->Dinv, o->rd, n);
}

static int primal_infeasible(const work_t* w, const double* dy, double eps, double* tn, double* tm) {
  int n = w->n, m = w->m;
  double nrm = 0, lhs = 0;
  for (int r = 0; r < m; ++r) {
    int lo_inf = w->l[r] < -INF_BOUND, up_inf = w->u[r] > INF_BOUND;
    double d = dy[r];
    if (up_inf && lo_inf) d = 0; else if (up_inf) d = dmin(d, 0); else if (lo_inf) d = dmax(d, 0);
    tm[r] = d;
    nrm = dmax(nrm, fabs(w->E[r] * d));
  }
  if (nrm > eps) {
    for (int r = 0; r < m; ++r) lhs += w->u[r] * dmax(tm[r], 0) + w->l[r] * dmin(tm[r], 0);
    if (lhs < -eps * nrm) { csc_tmul(&w->A, tm, tn); return ninf_s(w->Dinv, tn, n) < eps * nrm; }
  }
  return 0;
}

static int dual_infeasible(const work_t* w, const double* dx, double eps, double* tn, double* tm) {
  int n = w->n, m = w->m;
  double nrm = ninf_s(w->D, dx, n);
  if (nrm > eps) {
    double qdx = 0; for (int j = 0; j < n; ++j) qdx += w->q[j] * dx[j];
    if (qdx < -w->c * eps * nrm) {
      sym_mul(&w->P, dx, tn);
      if (ninf_s(w->Dinv, tn, n) < w->c * eps * nrm) {
        csc_mul(&w->A, dx, tm);
        for (int r = 0; r < m; ++r) {
          double v = w->Einv[r] * tm[r];
          int lo_inf = w->l[r] < -INF_BOUND, up_inf = w->u[r] > INF_BOUND;
          if ((!up_inf && v > eps * nrm) || (!lo_inf && v < -eps * nrm)) return 0;
        }
        return 1;
      }
    }
  }
  return 0;
}

static int check_termination(const work_t* w, const info_t* o, const double* z, const double* dx, const double* dy, int approx, double* tn, double* tm) {
  const oracle_settings* st = w->st;
  double k = approx ? 10.0 : 1.0;
  double eps_prim = st->eps_abs * k + st->eps_rel * k * dmax(ninf_s(w->Einv, z, w->m), ninf_s(w->Einv, o->Ax, w->m));
  double eps_dual = st->eps_abs * k + st->eps_rel * k * w->cinv * dmax(dmax(ninf_s(w->Dinv, w->q, w->n), ninf_s(w->Dinv, o->Aty, w->n)), ninf_s(w->Dinv, o->Px, w->n));
  int prim_ok = o->pri < eps_prim, dual_ok = o->dua < eps_dual;
  int pinf = !prim_ok && primal_infeasible(w, dy, st->eps_prim_inf * k, tn, tm);
  int dinf = !dual_ok && dual_infeasible(w, dx, st->eps_dual_inf * k, tn, tm);
  if (prim_ok && dual_ok) return approx ? SOLVED_INACCURATE : SOLVED;
  if (pinf) return PRIMAL_INFEASIBLE;
  if (dinf) return DUAL_INFEASIBLE;
  return UNSOLVED;
}

/* KKT certificate in the unscaled problem */
static int certificate(const work_t* w, const double* xs, const double* ys, double tol, double* prim, double* stat) {
  int n = w->n, m = w->m;
  double* Ax = (double*)malloc(sizeof(double) * m);
  double* g = (double*)malloc(sizeof(double) * n);
  double* t = (double*)malloc(sizeof(double) * n);
  csc_mul(w->A0, xs, Ax);
  sym_mul(w->P0, xs, g);
  csc_tmul(w->A0, ys, t);
  double pv = 0, sv = 0, cv = 0;
  for (int j = 0; j < n; ++j) sv = dmax(sv, fabs(g[j] + w->q0[j] + t[j]));
  for (int r = 0; r < m; ++r) {
    double l = w->l0c[r], u = w->u0c[r];
    pv = dmax(pv, dmax(dmax(l - Ax[r], Ax[r] - u), 0));
    double yp = dmax(ys[r], 0), ym = dmin(ys[r], 0);
    double cu = u < INF_BOUND ? yp * fabs(u - Ax[r]) : (yp > 0 ? 1e300 : 0);
    double cl = l > -INF_BOUND ? -ym * fabs(Ax[r] - l) : (ym < 0 ? 1e300 : 0);
    cv = dmax(cv, dmax(cu, cl));
  }
  free(Ax); free(g); free(t);
  *prim = pv; *stat = sv;
  return pv <= tol && sv <= tol && cv <= tol;
}

/* ------------------------------------------------------------------ polish stage 1: interior point */
typedef struct { int *eq, *L, *U; } classes_t;

/* soft (phase 1, see phase1()): per-row gamma^2 or NULL.  A soft row reads  l <= (Ax)_r + gamma_r w_r <= u  with the cost
 * 1/2 w_r^2 (the caller has zeroed P and q); w_r = gamma_r (zl_r - zu_r) is eliminated: (Ax)_r - gamma_r^2 y_r replaces
 * (Ax)_r in the slack equations and gamma_r^2 is added to the row's diagonal entry of the reduced KKT matrix.  In that
 * mode the loop also ends as soon as the multipliers pass OSQP's primal-infeasibility test. */
static int primal_infeasible(const work_t* w, const double* dy, double eps, double* tn, double* tm);
static int ipm_refine(work_t* w, kkt_t* K, ldl_t* F, const classes_t* cl, double* x, double* y, double tol, double theta, int* iters_out, int* low, int* upp, const double* soft, double mu0, double* keep, int resume) {
  /* keep (4 m doubles or NULL): the slacks and multipliers (sl, su, zl, zu) at exit; resume != 0: start from them and
   * from (x, y) instead of flooring afresh - the retry at a tighter tolerance CONTINUES the iteration */
  const oracle_settings* st = w->st;
  int n = w->n, m = w->m, N = n + m;
  const double reg = st->ipm_reg;
  double *Ax = (double*)malloc(sizeof(double) * m), *nu = (double*)calloc(m, sizeof(double));
  double *sl = (double*)malloc(sizeof(double) * m), *su = (double*)malloc(sizeof(double) * m);
  double *zl = (double*)malloc(sizeof(double) * m), *zu = (double*)malloc(sizeof(double) * m);
  double *rd = (double*)malloc(sizeof(double) * n), *req = (double*)malloc(sizeof(double) * m);
  double *rl = (double*)malloc(sizeof(double) * m), *ru = (double*)malloc(sizeof(double) * m);
  double *d = (double*)malloc(sizeof(double) * m), *rhs = (double*)malloc(sizeof(double) * N), *sol = (double*)malloc(sizeof(double) * N);
  double *res2 = (double*)malloc(sizeof(double) * N), *cor = (double*)malloc(sizeof(double) * N);
  double *rcl = (double*)malloc(sizeof(double) * m), *rcu = (double*)malloc(sizeof(double) * m);
  double *dsl = (double*)malloc(sizeof(double) * m), *dsu = (double*)malloc(sizeof(double) * m);
  double *dzl = (double*)malloc(sizeof(double) * m), *dzu = (double*)malloc(sizeof(double) * m);
  double *tn = (double*)malloc(sizeof(double) * n), *tm = (double*)malloc(sizeof(double) * m), *yy = (double*)malloc(sizeof(double) * m);
  csc_mul(&w->A, x, Ax);
  if (mu0 > 0.0 && st->ipm_start_dual > 0.0) {      /* multipliers commensurate with the dual residual of the start */
    sym_mul(&w->P, x, tn);
    double r0 = 0.0;
    for (int j = 0; j < n; ++j) r0 = dmax(r0, fabs(tn[j] + w->q[j]));
    mu0 = dmax(mu0, st->ipm_start_dual * theta * r0);
  }
  int nb = 0;
  for (int r = 0; r < m; ++r) {
    nu[r] = cl->eq[r] ? y[r] : 0.0;
    sl[r] = cl->L[r] ? dmax(Ax[r] - w->l[r], theta) : 1.0;
    su[r] = cl->U[r] ? dmax(w->u[r] - Ax[r], theta) : 1.0;
    zl[r] = cl->L[r] ? dmax(-y[r], theta) : 0.0;
    zu[r] = cl->U[r] ? dmax(y[r], theta) : 0.0;
    if (mu0 > 0.0) { nu[r] = 0.0; zl[r] = cl->L[r] ? mu0 / sl[r] : 0.0; zu[r] = cl->U[r] ? mu0 / su[r] : 0.0; }
    if (resume && keep) { sl[r] = keep[r]; su[r] = keep[m + r]; zl[r] = keep[2 * m + r]; zu[r] = keep[3 * m + r]; }
    else { low[r] = cl->L[r] && zl[r] > sl[r]; upp[r] = cl->U[r] && zu[r] > su[r]; }   /* (before any step) */
    nb += cl->L[r] + cl->U[r];
  }
  if (nb < 1) nb = 1;
  int conv = 0, it = 0, stalled = 0;
  double mu_min = 1e300;
  double* g2 = (double*)calloc(m, sizeof(double));
  double* dk = (double*)malloc(sizeof(double) * m);
  if (soft) for (int r = 0; r < m; ++r) g2[r] = (cl->L[r] || cl->U[r]) ? soft[r] : 0.0;
  /* phase 1: the equality rows are as soft as OSQP's ADMM makes them - RHO_EQ_OVER_RHO_INEQ times the weight of an inequality
     row's violation (osqp_np._ipm_refine; mpmpc_core.hpp: P1_EQ_SOFT) */
  const double eqs = soft ? 1.0 / RHO_EQ_OVER_RHO_INEQ : 0.0;
  for (it = 0; it <= st->ipm_max_iter; ++it) {
    csc_mul(&w->A, x, Ax);
    for (int r = 0; r < m; ++r) yy[r] = nu[r] + zu[r] - zl[r];
    sym_mul(&w->P, x, rd); csc_tmul(&w->A, yy, tn);
    double res = 0, mu = 0;
    for (int j = 0; j < n; ++j) { rd[j] += w->q[j] + tn[j]; res = dmax(res, fabs(rd[j])); }
    for (int r = 0; r < m; ++r) {
      req[r] = cl->eq[r] ? Ax[r] - w->l[r] - eqs * nu[r] : 0.0;
      rl[r] = cl->L[r] ? Ax[r] - g2[r] * (zu[r] - zl[r]) - w->l[r] - sl[r] : 0.0;
      ru[r] = cl->U[r] ? w->u[r] - Ax[r] + g2[r] * (zu[r] - zl[r]) - su[r] : 0.0;
      res = dmax(res, dmax(fabs(req[r]), dmax(fabs(rl[r]), fabs(ru[r]))));
      if (cl->L[r]) mu += sl[r] * zl[r];
      if (cl->U[r]) mu += su[r] * zu[r];
    }
    mu /= nb;
    if (res < dmax(tol, 1e-11) && mu < tol) { conv = 1; break; }
    if (soft && primal_infeasible(w, yy, st->phase1_eps, tn, tm)) { conv = 1; break; }
    if (it == st->ipm_max_iter) break;
    /* mu of a feasible problem falls (nearly) monotonically; on an infeasible one the multipliers blow up within a few
       iterations: give up at once, phase 1 is what can decide such an instance */
    if (!soft && (mu > st->ipm_diverged * mu_min || (mu < tol * 1e-3 && res > 1e-5))) break;   /* ... or mu collapsed while the residual stands */
    mu_min = dmin(mu_min, mu);
    for (int r = 0; r < m; ++r) {
      double wt = (cl->L[r] ? zl[r] / sl[r] : 0.0) + (cl->U[r] ? zu[r] / su[r] : 0.0);
      d[r] = cl->eq[r] ? reg + eqs : ((cl->L[r] || cl->U[r]) ? 1.0 / dmax(wt, 1e-300) : 1e30);
      dk[r] = d[r] + g2[r];
    }
    if (kkt_fill_and_factor(w, K, F, reg, dk, m) != 0) break;
    for (int r = 0; r < m; ++r) { rcl[r] = sl[r] * zl[r]; rcu[r] = su[r] * zu[r]; }
    double a = 1.0;
    for (int pass = 0; pass < 2; ++pass) {
      for (int j = 0; j < n; ++j) rhs[j] = -rd[j];
      for (int r = 0; r < m; ++r) {
        double t = (cl->L[r] ? (rcl[r] + zl[r] * rl[r]) / sl[r] : 0.0) - (cl->U[r] ? (rcu[r] + zu[r] * ru[r]) / su[r] : 0.0);
        rhs[n + r] = cl->eq[r] ? -req[r] : ((cl->L[r] || cl->U[r]) ? -t * d[r] : 0.0);
      }
      ldl_solve(F, rhs, sol);
      /* one refinement step against the un-regularised Newton matrix */
      sym_mul(&w->P, sol, tn); csc_tmul(&w->A, sol + n, res2);
      for (int j = 0; j < n; ++j) res2[j] = rhs[j] - (tn[j] + res2[j]);
      csc_mul(&w->A, sol, tm);
      for (int r = 0; r < m; ++r) res2[n + r] = rhs[n + r] - (tm[r] - (cl->eq[r] ? eqs : dk[r]) * sol[n + r]);
      ldl_solve(F, res2, cor);
      for (int k = 0; k < N; ++k) sol[k] += cor[k];
      csc_mul(&w->A, sol, tm);
      for (int r = 0; r < m; ++r) tm[r] -= g2[r] * sol[n + r];
      double ratio = 1e300;
      for (int r = 0; r < m; ++r) {
        dsl[r] = cl->L[r] ? tm[r] + rl[r] : 0.0;
        dsu[r] = cl->U[r] ? -tm[r] + ru[r] : 0.0;
        dzl[r] = cl->L[r] ? (-rcl[r] - zl[r] * dsl[r]) / sl[r] : 0.0;
        dzu[r] = cl->U[r] ? (-rcu[r] - zu[r] * dsu[r]) / su[r] : 0.0;
        if (cl->L[r] && dsl[r] < 0) ratio = dmin(ratio, -sl[r] / dsl[r]);
        if (cl->U[r] && dsu[r] < 0) ratio = dmin(ratio, -su[r] / dsu[r]);
        if (cl->L[r] && dzl[r] < 0) ratio = dmin(ratio, -zl[r] / dzl[r]);
        if (cl->U[r] && dzu[r] < 0) ratio = dmin(ratio, -zu[r] / dzu[r]);
      }
      if (pass == 0) {
        a = dmin(1.0, ratio);
        double mu_aff = 0;
        for (int r = 0; r < m; ++r) {
          if (cl->L[r]) mu_aff += (sl[r] + a * dsl[r]) * (zl[r] + a * dzl[r]);
          if (cl->U[r]) mu_aff += (su[r] + a * dsu[r]) * (zu[r] + a * dzu[r]);
        }
        mu_aff /= nb;
        double sg = mu > 0 ? mu_aff / mu : 0.0; sg = sg * sg * sg;
        for (int r = 0; r < m; ++r) { rcl[r] = sl[r] * zl[r] - sg * mu + dsl[r] * dzl[r]; rcu[r] = su[r] * zu[r] - sg * mu + dsu[r] * dzu[r]; }
      } else {
        a = dmin(1.0, 0.995 * ratio);
        stalled = a < 1e-6 ? stalled + 1 : 0;
        for (int j = 0; j < n; ++j) x[j] += a * sol[j];
        for (int r = 0; r < m; ++r) {
          if (cl->eq[r]) nu[r] += a * sol[n + r];
          /* active-set indicators of this step (Tapia): the slack of an active bound shrinks faster than its multiplier */
          low[r] = cl->L[r] && dsl[r] * zl[r] < dzl[r] * sl[r];
          upp[r] = cl->U[r] && dsu[r] * zu[r] < dzu[r] * su[r];
          sl[r] += a * dsl[r]; su[r] += a * dsu[r]; zl[r] += a * dzl[r]; zu[r] += a * dzu[r];
        }
      }
    }
    if (stalled >= 3) { ++it; break; }
  }
  for (int r = 0; r < m; ++r) {
    y[r] = nu[r] + zu[r] - zl[r];
    upp[r] = upp[r] && !low[r];
    if (keep) { keep[r] = sl[r]; keep[m + r] = su[r]; keep[2 * m + r] = zl[r]; keep[3 * m + r] = zu[r]; }
  }
  *iters_out = it;
  free(Ax); free(nu); free(sl); free(su); free(zl); free(zu); free(rd); free(req); free(rl); free(ru); free(d); free(rhs); free(sol);
  free(res2); free(cor); free(rcl); free(rcu); free(dsl); free(dsu); free(dzl); free(dzu); free(tn); free(tm); free(yy);
  free(g2); free(dk);
  return conv;
}

/* ------------------------------------------------------------------ polish stage 2: iterated active-set solve */
static int active_set_polish(work_t* w, const classes_t* cl, int* low, int* upp, double* x, double* y, int* rounds_out, double add_fraction) {
  const oracle_settings* st = w->st;
  int n = w->n, m = w->m;
  const double tol = 1e-9;
  int* rows = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
  double* Ax = (double*)malloc(sizeof(double) * m);
  double* rowmax = (double*)calloc(m > 0 ? m : 1, sizeof(double));
  for (int j = 0; j < n; ++j)
    for (int p = w->A.p[j]; p < w->A.p[j + 1]; ++p) rowmax[w->A.i[p]] = dmax(rowmax[w->A.i[p]], fabs(w->A.x[p]));
  for (int r = 0; r < m; ++r) if (!(rowmax[r] > 0.0)) rowmax[r] = 1e-300;
  int ok = 0, rnd = 0;
  for (rnd = 1; rnd <= st->as_rounds; ++rnd) {
    int k = 0;
    for (int r = 0; r < m; ++r) if (cl->eq[r] || low[r] || upp[r]) rows[k++] = r;
    kkt_t* K = kkt_build(&w->P, &w->A, rows, k);
    ldl_t* F = ldl_analyse(K->N, K->p, K->i, NULL);
    double* dv = (double*)malloc(sizeof(double) * (k > 0 ? k : 1));
    int N = n + k;
    double *rhs = (double*)malloc(sizeof(double) * N), *cor = (double*)malloc(sizeof(double) * N), *rr = (double*)malloc(sizeof(double) * N);
    long double* sol = (long double*)calloc(N, sizeof(long double));
    long double* acc = (long double*)malloc(sizeof(long double) * N);
    for (int j = 0; j < n; ++j) rhs[j] = -w->q[j];
    for (int r = 0; r < k; ++r) rhs[n + r] = upp[rows[r]] ? w->u[rows[r]] : w->l[rows[r]];
    int bad = 0;
    /* LDL' without pivoting of a quasi-definite matrix: if the refinement does not contract with the
     * nominal regularisation, retry with a larger one (the refinement removes its bias anyway) */
    double delta = st->as_delta;
    for (int tries = 0; tries < 4; ++tries, delta *= 30.0) {
      for (int r = 0; r < k; ++r) dv[r] = delta;
      for (int i = 0; i < N; ++i) sol[i] = 0.0L;
      bad = kkt_fill_and_factor(w, K, F, delta, dv, k);
      long double resn = 0.0L;
      for (int rf = 0; rf <= st->as_refine + 3 && !bad; ++rf) {
        /* residual rhs - K0 sol in extended precision (K0 = un-regularised reduced KKT) */
        for (int i = 0; i < N; ++i) acc[i] = rhs[i];
        for (int j = 0; j < n; ++j)
          for (int p = w->P.p[j]; p < w->P.p[j + 1]; ++p) {
            int i = w->P.i[p];
            acc[i] -= (long double)w->P.x[p] * sol[j];
            if (i != j) acc[j] -= (long double)w->P.x[p] * sol[i];
          }
        for (int r = 0; r < k; ++r)
          for (int p = K->At_p[rows[r]]; p < K->At_p[rows[r] + 1]; ++p) {
            int j = K->At_i[p]; double a = w->A.x[K->At_k[p]];
            acc[j] -= (long double)a * sol[n + r];
            acc[n + r] -= (long double)a * sol[j];
          }
        resn = 0.0L;
        for (int i = 0; i < N; ++i) { rr[i] = (double)acc[i]; long double t = acc[i] < 0 ? -acc[i] : acc[i]; if (t > resn) resn = t; }
        if (resn < 1e-14L) break;
        ldl_solve(F, rr, cor);
        for (int i = 0; i < N; ++i) sol[i] += cor[i];
      }
      if (!bad && resn < 1e-11L) break;
    }
    free(acc);
    for (int j = 0; j < n; ++j) x[j] = (double)sol[j];
    for (int r = 0; r < m; ++r) y[r] = 0;
    for (int r = 0; r < k; ++r) y[rows[r]] = (double)sol[n + r];
    free(rhs); free(cor); free(rr); free(sol); free(dv); kkt_free(K); ldl_free(F);
    if (bad) break;
    csc_mul(&w->A, x, Ax);
    int any = 0;
    int *nl = (int*)malloc(sizeof(int) * (m > 0 ? m : 1)), *nu_ = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
    double worst = 0.0;           /* worst violation of a scaled variable: row violation / max |A_r.| */
    for (int r = 0; r < m; ++r) {
      if (cl->L[r] && !low[r] && Ax[r] < w->l[r] - tol) worst = dmax(worst, (w->l[r] - Ax[r]) / rowmax[r]);
      if (cl->U[r] && !upp[r] && Ax[r] > w->u[r] + tol) worst = dmax(worst, (Ax[r] - w->u[r]) / rowmax[r]);
    }
    const double thr = add_fraction * worst;
    for (int r = 0; r < m; ++r) {
      int vl = cl->L[r] && !low[r] && Ax[r] < w->l[r] - tol;
      int vu = cl->U[r] && !upp[r] && Ax[r] > w->u[r] + tol;
      int bl = low[r] && y[r] > tol, bu = upp[r] && y[r] < -tol;
      if (vl || vu || bl || bu) any = 1;
      vl = vl && !((w->l[r] - Ax[r]) / rowmax[r] < thr);       /* only the violations near the worst one are added */
      vu = vu && !((Ax[r] - w->u[r]) / rowmax[r] < thr);
      nl[r] = (low[r] && !bl) || vl;
      nu_[r] = ((upp[r] && !bu) || vu) && !nl[r];
    }
    if (any) for (int r = 0; r < m; ++r) { low[r] = nl[r]; upp[r] = nu_[r]; }
    free(nl); free(nu_);
    if (!any) { ok = 1; break; }
  }
  *rounds_out = rnd > st->as_rounds ? st->as_rounds : rnd;
  free(rows); free(Ax); free(rowmax);
  return ok;
}

/* polish = 2: interior-point refinement from the scaled point (x, y), iterated active-set solve,
 * KKT certificate.  On success writes the certified (unscaled) point and returns 1. */
/* floor of the warm-started slacks / multipliers: pri_res / 80 in [3e-4, 3e-3] (closer ADMM point, smaller floor) */
static double warm_start_floor(double pri_res) { return dmin(3e-3, dmax(3e-4, 0.0125 * pri_res)); }

static int certified_polish(work_t* w, const double* x, const double* y, double theta, double mu0, double* x_out, double* y_out, oracle_info* info) {
  const oracle_settings* st = w->st;
  int n = w->n, m = w->m;
  classes_t cl; cl.eq = (int*)malloc(sizeof(int) * m); cl.L = (int*)malloc(sizeof(int) * m); cl.U = (int*)malloc(sizeof(int) * m);
  for (int r = 0; r < m; ++r) {
    int fl = w->l[r] > -INF_BOUND, fu = w->u[r] < INF_BOUND;
    cl.eq[r] = fl && fu && (w->u[r] - w->l[r] <= 1e-12 * dmax(1.0, fabs(w->l[r])));
    cl.L[r] = fl && !cl.eq[r]; cl.U[r] = fu && !cl.eq[r];
  }
  int *low = (int*)calloc(m, sizeof(int)), *upp = (int*)calloc(m, sizeof(int));
  double *xi = (double*)malloc(sizeof(double) * n), *yi = (double*)malloc(sizeof(double) * m);
  double *xa = (double*)malloc(sizeof(double) * n), *ya = (double*)malloc(sizeof(double) * m);
  double *xs = (double*)malloc(sizeof(double) * n), *ys = (double*)malloc(sizeof(double) * m);
  memcpy(xi, x, sizeof(double) * n); memcpy(yi, y, sizeof(double) * m);
  double tol = st->ipm_tol;
  double* keep = (double*)malloc(sizeof(double) * 4 * (m > 0 ? m : 1));
  int good = 0;
  for (int attempt = 0; attempt < 2 && !good; ++attempt) {
    int nit = 0;
    int conv = ipm_refine(w, w->K, w->F, &cl, xi, yi, tol, theta, &nit, low, upp, NULL, attempt == 0 ? mu0 : 0.0, keep, attempt);
    info->ipm_iters += nit;
    if (!conv) break;
    int rounds = 0;
    /* (the retry is more careful: only the upper half of the violations) */
    int okm = active_set_polish(w, &cl, low, upp, xa, ya, &rounds, attempt == 0 ? st->as_add_fraction : dmax(st->as_add_fraction, 0.5));
    info->as_rounds += rounds;
    if (okm) {
      for (int j = 0; j < n; ++j) xs[j] = w->D[j] * xa[j];
      for (int r = 0; r < m; ++r) ys[r] = w->E[r] * ya[r] * w->cinv;
      double pv, sv;
      if (certificate(w, xs, ys, st->cert_tol, &pv, &sv)) {
        memcpy(x_out, xs, sizeof(double) * n); memcpy(y_out, ys, sizeof(double) * m);
        info->status = SOLVED; info->polished = 1; info->pri_res = pv; info->dua_res = sv;
        good = 1;
      }
    }
    tol *= 1e-4;
  }
  free(cl.eq); free(cl.L); free(cl.U); free(low); free(upp); free(xi); free(yi); free(xa); free(ya); free(xs); free(ys); free(keep);
  return good;
}

/* ------------------------------------------------------------------ phase 1: is the QP infeasible?
 *     min 1/2 |w|^2   s.t.  equality rows as they are,  l <= (Ax)_r + gamma_r w_r <= u  on every other finite row
 * Always feasible when the equality rows are; optimum 0 iff the QP is feasible; at its optimum the multipliers y satisfy
 * A'y = 0 and u'max(y,0) + l'min(y,0) = -|w|^2 < 0: a Farkas ray, put to OSQP's own primal-infeasibility test.
 * gamma_r = 1 (the violation of the scaled row counts).  Returns 1 when certified; x_out: least-violation point, y_out: the ray. */
static int certified_polish(work_t* w, const double* x, const double* y, double theta, double mu0, double* x_out, double* y_out, oracle_info* info);
/* returns 1: certified infeasible, 2: found FEASIBLE and certified optimal by a second polish attempt from phase 1's
 * point (inside every box, well centred: the warm-started interior point of the first attempt occasionally jams next to
 * a degenerate vertex), 0: neither */
static int phase1(work_t* w, double* x_out, double* y_out, oracle_info* info) {
  const oracle_settings* st = w->st;
  int n = w->n, m = w->m;
  classes_t cl; cl.eq = (int*)malloc(sizeof(int) * m); cl.L = (int*)malloc(sizeof(int) * m); cl.U = (int*)malloc(sizeof(int) * m);
  for (int r = 0; r < m; ++r) {
    int fl = w->l[r] > -INF_BOUND, fu = w->u[r] < INF_BOUND;
    cl.eq[r] = fl && fu && (w->u[r] - w->l[r] <= 1e-12 * dmax(1.0, fabs(w->l[r])));
    cl.L[r] = fl && !cl.eq[r]; cl.U[r] = fu && !cl.eq[r];
  }
  double* soft = (double*)calloc(m, sizeof(double));
  /* gamma_r = 1: unit weight on the violation of the SCALED ROW - OSQP's own measure, the one its ADMM iteration minimises
     on an infeasible QP (osqp_np._phase1; rounds 2 - 4: gamma_r = max |A_r.|, the violation of the scaled variable) */
  for (int r = 0; r < m; ++r) soft[r] = 1.0;
  int pnnz = w->P.p[n];
  double* Psave = (double*)malloc(sizeof(double) * (pnnz > 0 ? pnnz : 1)); memcpy(Psave, w->P.x, sizeof(double) * pnnz);
  double* qsave = (double*)malloc(sizeof(double) * n); memcpy(qsave, w->q, sizeof(double) * n);
  memset(w->P.x, 0, sizeof(double) * pnnz); memset(w->q, 0, sizeof(double) * n);
  double *x = (double*)calloc(n, sizeof(double)), *y = (double*)calloc(m, sizeof(double));
  double *tn = (double*)malloc(sizeof(double) * n), *tm = (double*)malloc(sizeof(double) * m);
  int *low = (int*)calloc(m, sizeof(int)), *upp = (int*)calloc(m, sizeof(int));
  int nit = 0;
  /* (two digits beyond the polish's tolerance: the quantities of a marginal verdict are themselves at the 1e-9 level) */
  int conv = ipm_refine(w, w->K, w->F, &cl, x, y, dmin(st->ipm_tol * 1e-2, 1e-11), st->phase1_theta, &nit, low, upp, soft, 0.0, NULL, 0);
  info->ipm_iters += nit;
  /* (A) OSQP's test at phase1_eps: any iterate whose ray passes is a certificate */
  int cert = primal_infeasible(w, y, st->phase1_eps, tn, tm);
  memcpy(w->P.x, Psave, sizeof(double) * pnnz); memcpy(w->q, qsave, sizeof(double) * n);
  double *xs = (double*)malloc(sizeof(double) * n), *ys = (double*)malloc(sizeof(double) * m);
  for (int j = 0; j < n; ++j) xs[j] = w->D[j] * x[j];
  for (int r = 0; r < m; ++r) ys[r] = w->E[r] * y[r] * w->cinv;
  double pv, sv;
  certificate(w, xs, ys, st->cert_tol, &pv, &sv);
  if (!cert && conv) {
    /* (B) the iteration ran to its converged optimum (it did not stop at the ray test) and that optimum still violates
       a bound by more than cert_tol: infeasible however small the margin - taken when the ray's support is negative by
       at least a hundred times its own residual |A'y| */
    double nrm = 0, lhs = 0;
    for (int r = 0; r < m; ++r) {
      int lo_inf = w->l[r] < -INF_BOUND, up_inf = w->u[r] > INF_BOUND;
      double d = y[r];
      if (up_inf && lo_inf) d = 0; else if (up_inf) d = dmin(d, 0); else if (lo_inf) d = dmax(d, 0);
      tm[r] = d;
      nrm = dmax(nrm, fabs(w->E[r] * d));
      lhs += w->u[r] * dmax(d, 0) + w->l[r] * dmin(d, 0);
    }
    csc_tmul(&w->A, tm, tn);
    double res = ninf_s(w->Dinv, tn, n);
    if (pv > st->cert_tol && nrm > 0.0 && lhs < 0.0 && lhs < -100.0 * res) cert = 1;
  }
  if (cert) {
    memcpy(x_out, xs, sizeof(double) * n); memcpy(y_out, ys, sizeof(double) * m);
    info->status = PRIMAL_INFEASIBLE; info->polished = 0; info->pri_res = pv; info->dua_res = 0.0;
  } else if (conv && !(pv > st->cert_tol)) {
    double* y0 = (double*)calloc(m, sizeof(double));
    if (certified_polish(w, x, y0, st->ipm_start_mu > 0.0 ? st->ipm_start_slack : 3e-3, st->ipm_start_mu, x_out, y_out, info)) cert = 2;
    free(y0);
  }
  free(xs); free(ys);
  free(cl.eq); free(cl.L); free(cl.U); free(soft); free(Psave); free(qsave); free(x); free(y); free(tn); free(tm); free(low); free(upp);
  return cert;
}

/* ------------------------------------------------------------------ the solver */
int oracle_solve_csc(int n, int m, const int* Pp, const int* Pi, const double* Px, const double* q, const int* Ap, const int* Ai,
                     const double* Ax_, const double* l, const double* u, const oracle_settings* st, const int* perm_in,
                     double* x_out, double* y_out, oracle_info* info) {
  {
    /* an empty interval row: trivially infeasible (stock OSQP refuses such data at setup); zero ray, the gap as violation */
    double gap = 0;
    for (int r = 0; r < m; ++r) if (l[r] > u[r]) gap = dmax(gap, l[r] - u[r]);
    if (gap > 0) {
      memset(x_out, 0, sizeof(double) * n); memset(y_out, 0, sizeof(double) * m);
      memset(info, 0, sizeof(*info));
      info->status = PRIMAL_INFEASIBLE; info->pri_res = gap;
      return 0;
    }
  }
  csc P0 = {n, n, (int*)Pp, (int*)Pi, (double*)Px}, A0 = {m, n, (int*)Ap, (int*)Ai, (double*)Ax_};
  work_t w; memset(&w, 0, sizeof(w));
  w.n = n; w.m = m; w.st = st; w.P0 = &P0; w.A0 = &A0; w.q0 = q;
  copy_csc(&w.P, &P0); copy_csc(&w.A, &A0);
  w.q = (double*)malloc(sizeof(double) * n); memcpy(w.q, q, sizeof(double) * n);
  w.l = (double*)malloc(sizeof(double) * m); w.u = (double*)malloc(sizeof(double) * m);
  w.l0c = (double*)malloc(sizeof(double) * m); w.u0c = (double*)malloc(sizeof(double) * m);
  for (int r = 0; r < m; ++r) { w.l0c[r] = w.l[r] = dmax(l[r], -OSQP_INFTY); w.u0c[r] = w.u[r] = dmin(u[r], OSQP_INFTY); }
  w.D = (double*)malloc(sizeof(double) * n); w.Dinv = (double*)malloc(sizeof(double) * n);
  w.E = (double*)malloc(sizeof(double) * m); w.Einv = (double*)malloc(sizeof(double) * m);
  for (int j = 0; j < n; ++j) w.D[j] = 1; for (int r = 0; r < m; ++r) w.E[r] = 1;
  w.c = 1.0;
  /* with an early polish attempt only early_scaling of the Ruiz passes come first; an instance the attempt
     cannot certify gets the rest and restarts ADMM from cold */
  const int early_on = st->polish == 2 && st->early_polish > 0 && st->early_polish < st->max_iter;
  int passes_done = early_on && st->early_scaling > 0 && st->early_scaling < st->scaling ? st->early_scaling : st->scaling;
  if (passes_done) scale_data(&w, passes_done);
  for (int j = 0; j < n; ++j) w.Dinv[j] = 1.0 / w.D[j];
  for (int r = 0; r < m; ++r) w.Einv[r] = 1.0 / w.E[r];
  w.cinv = 1.0 / w.c;
  w.rho = st->rho;
  w.rho_vec = (double*)malloc(sizeof(double) * m); w.rho_inv = (double*)malloc(sizeof(double) * m); w.ctype = (int*)malloc(sizeof(int) * m);
  set_rho_vec(&w);
  w.K = kkt_build(&w.P, &w.A, NULL, 0);
  w.F = ldl_analyse(w.K->N, w.K->p, w.K->i, perm_in);
  kkt_fill_and_factor(&w, w.K, w.F, st->sigma, w.rho_inv, m);

  int N = n + m;
  double *x = (double*)calloc(n, sizeof(double)), *z = (double*)calloc(m, sizeof(double)), *y = (double*)calloc(m, sizeof(double));
  double *xp = (double*)calloc(n, sizeof(double)), *zp = (double*)calloc(m, sizeof(double));
  double *dx = (double*)calloc(n, sizeof(double)), *dy = (double*)calloc(m, sizeof(double));
  double *rhs = (double*)malloc(sizeof(double) * N), *sol = (double*)malloc(sizeof(double) * N);
  double *tn = (double*)malloc(sizeof(double) * n), *tm = (double*)malloc(sizeof(double) * m);
  info_t o; o.Ax = (double*)malloc(sizeof(double) * m); o.Px = (double*)malloc(sizeof(double) * n); o.Aty = (double*)malloc(sizeof(double) * n);
  o.rp = (double*)malloc(sizeof(double) * m); o.rd = (double*)malloc(sizeof(double) * n);
  int status = UNSOLVED, it = 0, rho_updates = 0, early_done = 0, early_ipm = 0, early_as = 0, early_failed = 0;
  compute_info(&w, x, z, y, &o);
  const double alpha = st->alpha;
  while (it < st->max_iter) {
    ++it;
    memcpy(xp, x, sizeof(double) * n); memcpy(zp, z, sizeof(double) * m);
    for (int j = 0; j < n; ++j) rhs[j] = st->sigma * xp[j] - w.q[j];
    for (int r = 0; r < m; ++r) rhs[n + r] = zp[r] - y[r] * w.rho_inv[r];
    ldl_solve(w.F, rhs, sol);
    for (int j = 0; j < n; ++j) { x[j] = alpha * sol[j] + (1 - alpha) * xp[j]; dx[j] = x[j] - xp[j]; }
    for (int r = 0; r < m; ++r) {
      double zt = zp[r] + (sol[n + r] - y[r]) * w.rho_inv[r];
      double zr = alpha * zt + (1 - alpha) * zp[r];
      double zn = dmin(dmax(zr + y[r] * w.rho_inv[r], w.l[r]), w.u[r]);
      dy[r] = w.rho_vec[r] * (zr - zn);
      z[r] = zn; y[r] += dy[r];
    }
    int can_check = st->check_termination > 0 && it % st->check_termination == 0;
    int can_adapt = st->adaptive_rho && st->adaptive_rho_interval > 0 && it % st->adaptive_rho_interval == 0;
    if (can_check || can_adapt) compute_info(&w, x, z, y, &o);
    if (can_check) { status = check_termination(&w, &o, z, dx, dy, 0, tn, tm); if (status != UNSOLVED) break; }
    if (early_on && !early_failed && it == st->early_polish) {
      /* the polish only needs a reasonable starting point: try it now; if it cannot certify, ADMM goes on */
      info->status = UNSOLVED; info->iters = it; info->rho_updates = rho_updates; info->ipm_iters = 0; info->as_rounds = 0; info->polished = 0;
      compute_info(&w, x, z, y, &o);
      if (certified_polish(&w, x, y, st->ipm_start_mu > 0.0 ? st->ipm_start_slack : warm_start_floor(o.pri), st->ipm_start_mu, x_out, y_out, info)) { early_done = 1; break; }
      /* not certified: before any long ADMM run, ask whether the problem is infeasible at all */
      if (st->phase1 && phase1(&w, x_out, y_out, info)) { early_done = 1; break; }
      if (passes_done < st->scaling) {
        /* not certified: the remaining Ruiz passes, then the full OSQP iteration from a cold start */
        scale_data(&w, st->scaling - passes_done);
        passes_done = st->scaling;
        for (int j = 0; j < n; ++j) w.Dinv[j] = 1.0 / w.D[j];
        for (int r = 0; r < m; ++r) w.Einv[r] = 1.0 / w.E[r];
        w.cinv = 1.0 / w.c;
        w.rho = st->rho; rho_updates = 0;
        set_rho_vec(&w);
        memset(x, 0, sizeof(double) * n); memset(z, 0, sizeof(double) * m); memset(y, 0, sizeof(double) * m);
        memset(dx, 0, sizeof(double) * n); memset(dy, 0, sizeof(double) * m);
        early_ipm = info->ipm_iters; early_as = info->as_rounds; early_failed = 1;
        it = 0;
        kkt_fill_and_factor(&w, w.K, w.F, st->sigma, w.rho_inv, m);
        continue;
      }
      /* the interior point re-used the KKT workspace: restore the ADMM factorisation */
      kkt_fill_and_factor(&w, w.K, w.F, st->sigma, w.rho_inv, m);
    }
    if (can_adapt) {
      double pr = ninf(o.rp, m) / (dmax(ninf(z, m), ninf(o.Ax, m)) + 1e-10);
      double du = ninf(o.rd, n) / (dmax(dmax(ninf(w.q, n), ninf(o.Aty, n)), ninf(o.Px, n)) + 1e-10);
      double est = dmin(dmax(w.rho * sqrt(pr / (du + 1e-10)), RHO_MIN), RHO_MAX);
      if (est > w.rho * st->adaptive_rho_tolerance || est < w.rho / st->adaptive_rho_tolerance) {
        w.rho = est; set_rho_vec(&w);
        kkt_fill_and_factor(&w, w.K, w.F, st->sigma, w.rho_inv, m);
        rho_updates++;
      }
    }
  }
  if (early_done) goto finish;
  if (!early_failed) { early_ipm = info->ipm_iters; early_as = info->as_rounds; }
  if (status == UNSOLVED) {
    compute_info(&w, x, z, y, &o);
    status = check_termination(&w, &o, z, dx, dy, 0, tn, tm);
    if (status == UNSOLVED) status = check_termination(&w, &o, z, dx, dy, 1, tn, tm);
    if (status == UNSOLVED) status = MAX_ITER_REACHED;
  }
  compute_info(&w, x, z, y, &o);
  for (int j = 0; j < n; ++j) x_out[j] = w.D[j] * x[j];
  for (int r = 0; r < m; ++r) y_out[r] = w.E[r] * y[r] * w.cinv;
  info->status = status; info->iters = it; info->rho_updates = rho_updates; info->pri_res = o.pri; info->dua_res = o.dua;
  info->ipm_iters = st->polish == 2 && st->early_polish > 0 && it > st->early_polish ? early_ipm : 0;
  info->as_rounds = st->polish == 2 && st->early_polish > 0 && it > st->early_polish ? early_as : 0;
  info->polished = 0;

  if (st->polish == 2 && (status == SOLVED || status == SOLVED_INACCURATE || status == MAX_ITER_REACHED)) {
    if (!certified_polish(&w, x, y, warm_start_floor(o.pri), 0.0, x_out, y_out, info)) { info->status = SOLVED_INACCURATE; info->polished = -1; }
  }
finish:
  /* objective of the returned point */
  { double* g = (double*)malloc(sizeof(double) * n); sym_mul(&P0, x_out, g); double ob = 0; for (int j = 0; j < n; ++j) ob += 0.5 * x_out[j] * g[j] + q[j] * x_out[j]; info->obj = ob; free(g); }
  free(x); free(z); free(y); free(xp); free(zp); free(dx); free(dy); free(rhs); free(sol); free(tn); free(tm);
  free(o.Ax); free(o.Px); free(o.Aty); free(o.rp); free(o.rd);
  kkt_free(w.K); ldl_free(w.F);
  free(w.P.p); free(w.P.i); free(w.P.x); free(w.A.p); free(w.A.i); free(w.A.x); free(w.q); free(w.l); free(w.u); free(w.l0c); free(w.u0c);
  free(w.D); free(w.Dinv); free(w.E); free(w.Einv); free(w.rho_vec); free(w.rho_inv); free(w.ctype);
  return 0;
}

/* ------------------------------------------------------------------ MPC assembly (src/MPC.py:61-155) */
typedef struct {
  int32_t N, circular;
  double Q[3], R[2], QN[3], xmin[3], xmax[3], umin[2], umax[2], ay_max, wheelbase;
} oracle_mpc_cfg;

/* builds q, l, u and CSC A (exact zeros dropped like scipy) + diagonal P for one instance */
static void mpc_assemble(const oracle_mpc_cfg* c, int n_wp, const double* kappa, const double* v_ref, const double* ds_next,
                         int wp, const double* x0, const double* cc, const double* lb, const double* ub,
                         int* Pp, int* Pi, double* Px, double* q, int* Ap, int* Ai, double* Ax, double* l, double* u) {
  const int N = c->N, nx = 3, nu = 2, ne = nx * (N + 1), n = ne + nu * N;
  /* column-wise construction of A = [Ax Bu; I] */
  double *a10 = (double*)malloc(sizeof(double) * N), *a20 = (double*)malloc(sizeof(double) * N), *b20 = (double*)malloc(sizeof(double) * N);
  double *ds = (double*)malloc(sizeof(double) * N), *kp = (double*)malloc(sizeof(double) * N), *vv = (double*)malloc(sizeof(double) * N);
  for (int k = 0; k < N; ++k) {
    int idx = wp + k; if (idx >= n_wp) idx = c->circular ? idx % n_wp : n_wp - 1;
    double kap = kappa[idx], v = v_ref[idx], d = ds_next[idx];
    kp[k] = kap; vv[k] = v; ds[k] = d;
    a10[k] = (-(kap * kap)) * d; a20[k] = ((-kap) / v) * d; b20[k] = ((-1.0) / (v * v)) * d;
    double f2 = (1.0 / v) * d;
    l[nx * (k + 1) + 0] = 0.0; l[nx * (k + 1) + 1] = d * kap; l[nx * (k + 1) + 2] = b20[k] * v - f2;
  }
  for (int i = 0; i < nx; ++i) l[i] = -x0[i];
  for (int r = 0; r < ne; ++r) u[r] = l[r];
  int pos = 0;
  for (int k = 0; k <= N; ++k) {
    for (int j = 0; j < nx; ++j) {
      int col = nx * k + j;
      Ap[col] = pos;
      Ai[pos] = col; Ax[pos++] = -1.0;                         /* -I */
      if (k < N) {
        int r0 = nx * (k + 1);
        double e0 = j == 0 ? 1.0 : (j == 1 ? ds[k] : 0.0);
        double e1 = j == 0 ? a10[k] : (j == 1 ? 1.0 : 0.0);
        double e2 = j == 0 ? a20[k] : (j == 2 ? 1.0 : 0.0);
        if (e0 != 0.0) { Ai[pos] = r0; Ax[pos++] = e0; }
        if (e1 != 0.0) { Ai[pos] = r0 + 1; Ax[pos++] = e1; }
        if (e2 != 0.0) { Ai[pos] = r0 + 2; Ax[pos++] = e2; }
      }
      Ai[pos] = ne + col; Ax[pos++] = 1.0;                     /* identity row */
    }
  }
  for (int k = 0; k < N; ++k) {
    int r0 = nx * (k + 1);
    int col = ne + nu * k;
    Ap[col] = pos;                                             /* v column: B[2,0] */
    if (b20[k] != 0.0) { Ai[pos] = r0 + 2; Ax[pos++] = b20[k]; }
    Ai[pos] = ne + col; Ax[pos++] = 1.0;
    Ap[col + 1] = pos;                                         /* kappa column: B[1,1] */
    if (ds[k] != 0.0) { Ai[pos] = r0 + 1; Ax[pos++] = ds[k]; }
    Ai[pos] = ne + col + 1; Ax[pos++] = 1.0;
  }
  Ap[n] = pos;
  /* boxes */
  for (int k = 0; k <= N; ++k)
    for (int j = 0; j < nx; ++j) { l[ne + nx * k + j] = c->xmin[j]; u[ne + nx * k + j] = c->xmax[j]; }
  l[ne] = x0[0]; u[ne] = x0[0];
  for (int k = 1; k <= N; ++k) { l[ne + nx * k] = lb[k - 1]; u[ne + nx * k] = ub[k - 1]; }
  double cc_last = cc[2 * N - 1];
  for (int k = 0; k < N; ++k) {
    double kpred = tan(cc[3 + k] + cc_last) / c->wheelbase;
    double vmax = sqrt(c->ay_max / (fabs(kpred) + 1e-12));
    int r = 2 * ne + nu * k;
    l[r] = c->umin[0]; u[r] = vmax < c->umax[0] ? vmax : c->umax[0];
    l[r + 1] = c->umin[1]; u[r + 1] = c->umax[1];
  }
  /* cost */
  for (int j = 0; j < n; ++j) { Pp[j] = j; Pi[j] = j; }
  Pp[n] = n;
  for (int k = 0; k <= N; ++k) {
    double xr0 = k == 0 ? 0.0 : (lb[k - 1] + ub[k - 1]) / 2;
    for (int j = 0; j < nx; ++j) {
      double xr = j == 0 ? xr0 : 0.0;
      Px[nx * k + j] = k < N ? c->Q[j] : c->QN[j];
      q[nx * k + j] = k < N ? (-c->Q[j]) * xr : -(c->QN[j] * xr);
    }
  }
  for (int k = 0; k < N; ++k) {
    Px[ne + nu * k] = c->R[0]; Px[ne + nu * k + 1] = c->R[1];
    q[ne + nu * k] = (-c->R[0]) * vv[k]; q[ne + nu * k + 1] = (-c->R[1]) * kp[k];
  }
  free(a10); free(a20); free(b20); free(ds); free(kp); free(vv);
}

/* dense export for tests: same arrays the numpy oracle produces */
int oracle_mpc_assemble_dense(const oracle_mpc_cfg* c, int n_wp, const double* kappa, const double* v_ref, const double* ds_next,
                              int wp, const double* x0, const double* cc, const double* lb, const double* ub,
                              double* Pdiag, double* q, double* A_dense /* m x n row-major */, double* l, double* u, int* nnzA) {
  const int N = c->N, n = 5 * N + 3, m = 8 * N + 6;
  int *Pp = (int*)malloc(sizeof(int) * (n + 1)), *Pi = (int*)malloc(sizeof(int) * n), *Ap = (int*)malloc(sizeof(int) * (n + 1)), *Ai = (int*)malloc(sizeof(int) * (8 * n));
  double* Ax = (double*)malloc(sizeof(double) * (8 * n));
  mpc_assemble(c, n_wp, kappa, v_ref, ds_next, wp, x0, cc, lb, ub, Pp, Pi, Pdiag, q, Ap, Ai, Ax, l, u);
  memset(A_dense, 0, sizeof(double) * (size_t)m * n);
  for (int j = 0; j < n; ++j) for (int k = Ap[j]; k < Ap[j + 1]; ++k) A_dense[(size_t)Ai[k] * n + j] = Ax[k];
  *nnzA = Ap[n];
  free(Pp); free(Pi); free(Ap); free(Ai); free(Ax);
  return 0;
}

/* The reference-equivalent CPU path for B instances: assembly + fresh setup + solve per instance
 * (src/MPC.py:158-159 builds a new OSQP workspace every step).  OpenMP over instances.  The fill
 * reducing ordering is computed once per KKT pattern and shared (favours the CPU). */
int oracle_mpc_batch(const oracle_mpc_cfg* c, const oracle_settings* st, int n_wp, const double* kappa, const double* v_ref,
                     const double* ds_next, int B, const int* wp_id, const double* x0, const double* cc, const double* lb,
                     const double* ub, int nthreads, double* z, double* u0, int* status, int* iters /* [B*3] */, double* resid,
                     double* y) {
  const int N = c->N, n = 5 * N + 3, m = 8 * N + 6;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
  {
    int *Pp = (int*)malloc(sizeof(int) * (n + 1)), *Pi = (int*)malloc(sizeof(int) * n), *Ap = (int*)malloc(sizeof(int) * (n + 1)), *Ai = (int*)malloc(sizeof(int) * (8 * n));
    double *Px = (double*)malloc(sizeof(double) * n), *q = (double*)malloc(sizeof(double) * n), *Ax = (double*)malloc(sizeof(double) * (8 * n));
    double *l = (double*)malloc(sizeof(double) * m), *u = (double*)malloc(sizeof(double) * m), *xs = (double*)malloc(sizeof(double) * n), *ys = (double*)malloc(sizeof(double) * m);
    int* perm = NULL; int perm_nnz = -1;
#pragma omp for schedule(dynamic, 4)
    for (int b = 0; b < B; ++b) {
      mpc_assemble(c, n_wp, kappa, v_ref, ds_next, wp_id[b], x0 + 3 * b, cc + 2 * N * b, lb + N * b, ub + N * b, Pp, Pi, Px, q, Ap, Ai, Ax, l, u);
      if (perm == NULL || perm_nnz != Ap[n]) {   /* ordering cache keyed on the pattern size */
        csc Pm = {n, n, Pp, Pi, Px}, Am = {m, n, Ap, Ai, Ax};
        kkt_t* K = kkt_build(&Pm, &Am, NULL, 0);
        free(perm); perm = (int*)malloc(sizeof(int) * K->N);
        min_degree(K->N, K->p, K->i, perm);
        perm_nnz = Ap[n];
        kkt_free(K);
      }
      oracle_info inf;
      oracle_solve_csc(n, m, Pp, Pi, Px, q, Ap, Ai, Ax, l, u, st, perm, xs, ys, &inf);
      if (z) memcpy(z + (size_t)n * b, xs, sizeof(double) * n);
      if (y) memcpy(y + (size_t)m * b, ys, sizeof(double) * m);
      if (u0) { u0[2 * b] = xs[3 * (N + 1)]; u0[2 * b + 1] = atan(xs[3 * (N + 1) + 1] * c->wheelbase); }
      if (status) status[b] = inf.status;
      if (iters) { iters[3 * b] = inf.iters; iters[3 * b + 1] = inf.ipm_iters; iters[3 * b + 2] = inf.as_rounds; }
      if (resid) { resid[2 * b] = inf.pri_res; resid[2 * b + 1] = inf.dua_res; }
    }
    free(Pp); free(Pi); free(Ap); free(Ai); free(Px); free(q); free(Ax); free(l); free(u); free(xs); free(ys); free(perm);
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
