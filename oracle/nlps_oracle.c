/*
 * nlps_oracle.c — CPU ORACLE (test infrastructure only; see nlps_oracle.h header note).
 * PARITY UNPINNED: plain-C restatement of the reference algorithm, pinned by no reference-run
 * output (the reference needs <lapacke.h>/LAPACK, absent here) — see DESIGN.md.  The two exceptions are single
 * functions: the trial b_e (tests/Constitutive/test.py) and the spectral stiffness density, checked against the
 * reference's own numpy script (tests/golden/ref_etm2d.npz).
 *
 * Every function cites the reference file:line (relative to /root/reference/nl-partsol/src) it
 * follows.  Loop orders, operand orders and comparison operators follow the reference so that
 * integer results (I0, neighbour lists, masks) are what the reference would produce.
 */
#include "nlps_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TOL_NR 10E-6 /* Macros.h:40 */
#define PI_MATRIXLIB 3.14159265358979323846 /* Macros.h:42 */

/* DSQR, Macros.h:49-50: (a == 0 ? 0 : a*a) */
static inline double dsqr(double a) { return a == 0.0 ? 0.0 : a * a; }

void orc_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ======================================================================================
 * Mesh: structured Q4/H8 grid whose neighbour tables are produced with the list semantics of
 * InOutFun/Read_GramsBox.c:293-507, Nodes/Read-GID-Mesh.c:406-416 and Matlib/ChainOp.c:163-293.
 * Canonical numbering (the build's synthetic meshes): nodes x-fastest, elements x-fastest,
 * GiD connectivity order (Q4 counter-clockwise; H8 bottom face ccw then top face ccw).
 * ====================================================================================== */

typedef struct {
  int ndim, n[3], nc[3];
} grid_t;

static inline int node_id(const grid_t *g, int i, int j, int k) {
  return i + g->n[0] * (j + g->n[1] * k);
}

/* Connectivity chain of element (ci,cj,ck): file order pushed one by one => chain is the reverse
 * (Read-GID-Mesh.c:411-413, ChainOp.c:163-182). */
static int element_chain(const grid_t *g, int ci, int cj, int ck, int *out) {
  int file[8], nn;
  if (g->ndim == 2) {
    file[0] = node_id(g, ci, cj, 0);
    file[1] = node_id(g, ci + 1, cj, 0);
    file[2] = node_id(g, ci + 1, cj + 1, 0);
    file[3] = node_id(g, ci, cj + 1, 0);
    nn = 4;
  } else {
    for (int t = 0; t < 2; t++) {
      file[4 * t + 0] = node_id(g, ci, cj, ck + t);
      file[4 * t + 1] = node_id(g, ci + 1, cj, ck + t);
      file[4 * t + 2] = node_id(g, ci + 1, cj + 1, ck + t);
      file[4 * t + 3] = node_id(g, ci, cj + 1, ck + t);
    }
    nn = 8;
  }
  for (int a = 0; a < nn; a++) out[a] = file[nn - 1 - a];
  return nn;
}

static int in_list(const int *v, int n, int x) {
  for (int i = 0; i < n; i++)
    if (v[i] == x) return 1;
  return 0;
}

/* node_I_locality, Read_GramsBox.c:371-400: union (ChainOp.c:275-293) of the connectivity chains
 * of the elements around node I, the elements taken in NodeNeighbour chain order.  NodeNeighbour[I]
 * is filled by pushing element indices in ascending order (Read_GramsBox.c:293-330), so its chain
 * order is DESCENDING element index.  Returns the result in chain order (front first). */
static int node_locality(const grid_t *g, int I, int *out) {
  int i = I % g->n[0], j = (I / g->n[0]) % g->n[1], k = I / (g->n[0] * g->n[1]);
  int push[32], np = 0;
  int klo = g->ndim == 3 ? k : 0, khi = g->ndim == 3 ? k - 1 : 0;
  /* descending element index: ck from high to low, cj high to low, ci high to low */
  for (int ck = klo; ck >= khi; ck--) {
    if (g->ndim == 3 && (ck < 0 || ck >= g->nc[2])) continue;
    for (int cj = j; cj >= j - 1; cj--) {
      if (cj < 0 || cj >= g->nc[1]) continue;
      for (int ci = i; ci >= i - 1; ci--) {
        if (ci < 0 || ci >= g->nc[0]) continue;
        int ch[8];
        int nn = element_chain(g, ci, cj, ck, ch);
        for (int a = 0; a < nn; a++)
          if (!in_list(push, np, ch[a])) push[np++] = ch[a];
      }
    }
  }
  for (int a = 0; a < np; a++) out[a] = push[np - 1 - a];
  return np;
}

/* fill_nodal_locality with 2 rings + ring_search_nodal_locality, Read_GramsBox.c:334-456. */
static int node_two_ring(const grid_t *g, int I, int *out) {
  int S[ORC_MAXNB], ns = 0;
  int search[ORC_MAXNB], nsearch = 1;
  search[0] = I;
  for (int ring = 0; ring < 2; ring++) {
    int newp[ORC_MAXNB], nnew = 0;
    for (int s = 0; s < nsearch; s++) {
      int aux[32];
      int na = node_locality(g, search[s], aux);
      for (int a = 0; a < na; a++) {
        if (!in_list(S, ns, aux[a])) {
          S[ns++] = aux[a];
          newp[nnew++] = aux[a];
        }
      }
    }
    for (int a = 0; a < nnew; a++) search[a] = newp[nnew - 1 - a]; /* chain order of new set */
    nsearch = nnew;
  }
  for (int a = 0; a < ns; a++) out[a] = S[ns - 1 - a];
  return ns;
}

static inline int axis_class(int i, int n) {
  if (i < 2) return i;
  if (i > n - 3) return 4 - (n - 1 - i);
  return 2;
}

orc_mesh *orc_mesh_build(int ndim, const int n[3], const double origin[3], double h) {
  orc_mesh *m = (orc_mesh *)calloc(1, sizeof(orc_mesh));
  grid_t g;
  g.ndim = ndim;
  for (int a = 0; a < 3; a++) {
    g.n[a] = (a < ndim) ? n[a] : 1;
    g.nc[a] = (a < ndim) ? n[a] - 1 : 1;
    m->n[a] = g.n[a];
    m->origin[a] = (a < ndim) ? origin[a] : 0.0;
  }
  m->ndim = ndim;
  m->h = h;
  m->nnodes = g.n[0] * g.n[1] * g.n[2];
  int nn = m->nnodes;
  m->coords = (double *)malloc(sizeof(double) * nn * ndim);
  for (int I = 0; I < nn; I++) {
    int ijk[3] = {I % g.n[0], (I / g.n[0]) % g.n[1], I / (g.n[0] * g.n[1])};
    for (int a = 0; a < ndim; a++) m->coords[I * ndim + a] = origin[a] + h * (double)ijk[a];
  }
  m->r1_ptr = (int *)malloc(sizeof(int) * (nn + 1));
  m->r2_ptr = (int *)malloc(sizeof(int) * (nn + 1));
  int cap1 = ndim == 2 ? 9 : 27, cap2 = ndim == 2 ? 25 : 125;
  m->r1 = (int *)malloc(sizeof(int) * (size_t)nn * cap1);
  m->r2 = (int *)malloc(sizeof(int) * (size_t)nn * cap2);
  m->h_avg = (double *)malloc(sizeof(double) * nn);
  m->active = (unsigned char *)calloc(nn, 1);

  /* Stencils are translation invariant inside one boundary class; cache them as ijk offsets. */
  int cacheable = 1;
  for (int a = 0; a < ndim; a++)
    if (g.n[a] < 5) cacheable = 0;
  typedef struct {
    int have, n1, n2;
    signed char o1[27][3], o2[125][3];
  } cls_t;
  cls_t *cache = (cls_t *)calloc(125, sizeof(cls_t));

  int p1 = 0, p2 = 0;
  for (int I = 0; I < nn; I++) {
    int ijk[3] = {I % g.n[0], (I / g.n[0]) % g.n[1], I / (g.n[0] * g.n[1])};
    m->r1_ptr[I] = p1;
    m->r2_ptr[I] = p2;
    int key = 0;
    if (cacheable) {
      int c[3] = {2, 2, 2};
      for (int a = 0; a < ndim; a++) c[a] = axis_class(ijk[a], g.n[a]);
      key = c[0] + 5 * (c[1] + 5 * c[2]);
    }
    if (cacheable && cache[key].have) {
      cls_t *c = &cache[key];
      for (int a = 0; a < c->n1; a++)
        m->r1[p1++] = node_id(&g, ijk[0] + c->o1[a][0], ijk[1] + c->o1[a][1], ijk[2] + c->o1[a][2]);
      for (int a = 0; a < c->n2; a++)
        m->r2[p2++] = node_id(&g, ijk[0] + c->o2[a][0], ijk[1] + c->o2[a][1], ijk[2] + c->o2[a][2]);
    } else {
      int l1[32], l2[ORC_MAXNB];
      int n1 = node_locality(&g, I, l1);
      int n2 = node_two_ring(&g, I, l2);
      for (int a = 0; a < n1; a++) m->r1[p1++] = l1[a];
      for (int a = 0; a < n2; a++) m->r2[p2++] = l2[a];
      if (cacheable) {
        cls_t *c = &cache[key];
        c->have = 1;
        c->n1 = n1;
        c->n2 = n2;
        for (int a = 0; a < n1; a++) {
          int J = l1[a];
          c->o1[a][0] = (signed char)(J % g.n[0] - ijk[0]);
          c->o1[a][1] = (signed char)((J / g.n[0]) % g.n[1] - ijk[1]);
          c->o1[a][2] = (signed char)(J / (g.n[0] * g.n[1]) - ijk[2]);
        }
        for (int a = 0; a < n2; a++) {
          int J = l2[a];
          c->o2[a][0] = (signed char)(J % g.n[0] - ijk[0]);
          c->o2[a][1] = (signed char)((J / g.n[0]) % g.n[1] - ijk[1]);
          c->o2[a][2] = (signed char)(J / (g.n[0] * g.n[1]) - ijk[2]);
        }
      }
    }
  }
  m->r1_ptr[nn] = p1;
  m->r2_ptr[nn] = p2;
  free(cache);

  /* compute_nodal_distance_local, Read_GramsBox.c:460-507: mean distance to the 1-ring
   * neighbours (self excluded), norm = pow(sum DSQR, 0.5) (MatrixOp.c:843-870). */
  for (int A = 0; A < nn; A++) {
    double avg = 0.0;
    int cnt = 0;
    for (int q = m->r1_ptr[A]; q < m->r1_ptr[A + 1]; q++) {
      int B = m->r1[q];
      if (A != B) {
        double aux = 0.0;
        for (int a = 0; a < ndim; a++)
          aux += dsqr(m->coords[B * ndim + a] - m->coords[A * ndim + a]);
        avg += pow(aux, 0.5);
        cnt++;
      }
    }
    m->h_avg[A] = avg / (double)cnt;
  }
  return m;
}

void orc_mesh_free(orc_mesh *m) {
  if (!m) return;
  free(m->coords);
  free(m->r1_ptr);
  free(m->r1);
  free(m->r2_ptr);
  free(m->r2);
  free(m->h_avg);
  free(m->active);
  free(m);
}

/* ======================================================================================
 * Small dense algebra replacing LAPACK on <=3x3 (Matlib/MatrixOp.c:320-382 dgetrf/dgetri,
 * Matlib/TensorLib.c:172-228 dsyev, :829-905 adjunt, :966-990 rcond).
 * ====================================================================================== */

/* inverse__MatrixLib__ (MatrixOp.c:320): reference = LAPACK LU + dgetri; here closed-form
 * adjugate/determinant (identical up to rounding).  Row-major n x n, n = 1..3. */
int orc_inverse(double *Am1, const double *A, int n) {
  if (n == 1) {
    if (A[0] == 0.0) return 1;
    Am1[0] = 1.0 / A[0];
    return 0;
  }
  if (n == 2) {
    double det = A[0] * A[3] - A[1] * A[2];
    if (det == 0.0) return 1;
    double id = 1.0 / det;
    Am1[0] = A[3] * id;
    Am1[1] = -A[1] * id;
    Am1[2] = -A[2] * id;
    Am1[3] = A[0] * id;
    return 0;
  }
  double c00 = A[4] * A[8] - A[5] * A[7];
  double c01 = A[5] * A[6] - A[3] * A[8];
  double c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (det == 0.0) return 1;
  double id = 1.0 / det;
  Am1[0] = c00 * id;
  Am1[1] = (A[2] * A[7] - A[1] * A[8]) * id;
  Am1[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Am1[3] = c01 * id;
  Am1[4] = (A[0] * A[8] - A[2] * A[6]) * id;
  Am1[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Am1[6] = c02 * id;
  Am1[7] = (A[1] * A[6] - A[0] * A[7]) * id;
  Am1[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return 0;
}

/* rcond__TensorLib__ (TensorLib.c:966-990) hands the UNFACTORED matrix to dgecon, which reads it
 * as LU factors (unit-lower L = strict lower triangle, U = upper triangle) and returns
 * 1 / (ANORM * ||(L U)^-1||_1) with ANORM = ||A||_1 from dlange (:981).  Restated literally with
 * the exact 1-norm of the small inverse (dgecon's dlacn2 estimate is exact at these sizes in
 * practice).  E.g. [[2,1],[1,3]] -> 0.25 (SURVEY.md §7 hard part 6).  Only the <1e-8 gate uses it. */
double orc_rcond_ref(const double *A, int n) {
  double LU[9], inv[9];
  double anorm = 0.0;
  for (int j = 0; j < n; j++) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += fabs(A[i * n + j]);
    if (s > anorm) anorm = s;
  }
  /* LU = L * U */
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int k = 0; k < n; k++) {
        double l = (k < i) ? A[i * n + k] : (k == i ? 1.0 : 0.0);
        double u = (k <= j) ? A[k * n + j] : 0.0;
        s += l * u;
      }
      LU[i * n + j] = s;
    }
  if (anorm == 0.0) return 0.0;
  if (orc_inverse(inv, LU, n)) return 0.0;
  double inorm = 0.0;
  for (int j = 0; j < n; j++) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += fabs(inv[i * n + j]);
    if (s > inorm) inorm = s;
  }
  if (!(inorm > 0.0) || isinf(inorm) || isnan(inorm)) return 0.0;
  return (1.0 / inorm) / anorm;
}

/* sym_eigen_analysis__TensorLib__ (TensorLib.c:172-228) = LAPACKE_dsyev(ROW_MAJOR,'V','U'):
 * eigenvalues ascending, eigenvector A in COLUMN A of the row-major matrix (eigvec[i*n+A]).
 * Restated as cyclic Jacobi (orthonormal to rounding also for repeated eigenvalues).  Vector
 * sign is LAPACK-implementation defined; every use on the path is sign-independent
 * (sum_A f_A n_A (x) n_A).  'U': only the upper triangle of A is referenced. */
int orc_sym_eigen(double *eigval, double *eigvec, const double *A, int n) {
  double a[9], v[9];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      a[i * n + j] = (j >= i) ? A[i * n + j] : A[j * n + i];
      v[i * n + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int sweep = 0; sweep < 50; sweep++) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        if (i != j) off += a[i * n + j] * a[i * n + j];
        else diag += a[i * n + j] * a[i * n + j];
      }
    if (off <= 1e-32 * diag || off == 0.0) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = a[p * n + q];
        if (apq == 0.0) continue;
        double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          double akp = a[k * n + p], akq = a[k * n + q];
          a[k * n + p] = c * akp - s * akq;
          a[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          double apk = a[p * n + k], aqk = a[q * n + k];
          a[p * n + k] = c * apk - s * aqk;
          a[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          double vkp = v[k * n + p], vkq = v[k * n + q];
          v[k * n + p] = c * vkp - s * vkq;
          v[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < n; i++) eigval[i] = a[i * n + i];
  for (int i = 0; i < n - 1; i++)
    for (int j = i + 1; j < n; j++)
      if (eigval[idx[j]] < eigval[idx[i]]) {
        int t = idx[i];
        idx[i] = idx[j];
        idx[j] = t;
      }
  double w[3];
  for (int i = 0; i < n; i++) w[i] = eigval[idx[i]];
  for (int A2 = 0; A2 < n; A2++) {
    eigval[A2] = w[A2];
    for (int i = 0; i < n; i++) eigvec[i * n + A2] = v[i * n + idx[A2]];
  }
  for (int i = 0; i < n; i++)
    if (isnan(eigval[i])) return 1;
  return 0;
}

/* I3__TensorLib__, TensorLib.c:154-168 (same term order). */
static double I3(const double *A, int ndim) {
  if (ndim == 2) return A[0] * A[3] - A[1] * A[2];
  return A[0] * A[4] * A[8] - A[0] * A[5] * A[7] + A[1] * A[5] * A[6] - A[1] * A[3] * A[8] +
         A[2] * A[3] * A[7] - A[2] * A[4] * A[6];
}

/* compute_adjunt__TensorLib__, TensorLib.c:829-905: A^{-T} (transpose then LAPACK inverse). */
static int adjunt(double *A_mT, const double *A, int ndim) {
  double At[9];
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) At[i * ndim + j] = A[j * ndim + i];
  return orc_inverse(A_mT, At, ndim);
}

/* ======================================================================================
 * LME shape functions, Nodes/LME.c
 * ====================================================================================== */

/* fa__LME__, LME.c:676-696 */
static inline double fa_lme(const double *la, const double *lambda, double beta, int ndim) {
  double la_x_la = 0.0, la_x_lambda = 0.0;
  for (int i = 0; i < ndim; i++) {
    la_x_la += la[i] * la[i];
    la_x_lambda += la[i] * lambda[i];
  }
  return -beta * la_x_la + la_x_lambda;
}

/* p__LME__, LME.c:700-737 */
void orc_p_lme(double *p, const double *l, int na, int ndim, const double *lambda, double beta) {
  double Z = 0.0;
  for (int a = 0; a < na; a++) {
    p[a] = exp(fa_lme(&l[a * ndim], lambda, beta, ndim));
    Z += p[a];
  }
  double Z_m1 = (double)1 / Z;
  for (int a = 0; a < na; a++) p[a] *= Z_m1;
}

/* r__LME__, LME.c:766-791 */
static void r_lme(double *r, const double *l, const double *p, int na, int ndim) {
  for (int i = 0; i < ndim; i++) {
    r[i] = 0.0;
    for (int a = 0; a < na; a++) r[i] += p[a] * l[a * ndim + i];
  }
}

/* J__LME__, LME.c:795-832 */
static void J_lme(double *J, const double *l, const double *p, const double *r, int na, int ndim) {
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) {
      double s = 0.0;
      for (int a = 0; a < na; a++) s += p[a] * l[a * ndim + i] * l[a * ndim + j];
      s -= r[i] * r[j];
      J[i * ndim + j] = s;
    }
}

/* dp__LME__, LME.c:836-891: dp_a = -p_a J^-1 l_a */
int orc_dp_lme(double *dp, const double *l, const double *p, int na, int ndim) {
  double r[3], J[9], Jm1[9];
  r_lme(r, l, p, na, ndim);
  J_lme(J, l, p, r, na, ndim);
  if (orc_inverse(Jm1, J, ndim)) return 1;
  for (int a = 0; a < na; a++) {
    for (int i = 0; i < ndim; i++) {
      double s = 0.0; /* get_A_x_b_Mat, MatrixOp.c:464-481 */
      for (int j = 0; j < ndim; j++) s += Jm1[i * ndim + j] * l[a * ndim + j];
      dp[a * ndim + i] = -p[a] * s;
    }
  }
  return 0;
}

/* __lambda_Newton_Rapson, LME.c:272-353 */
int orc_lambda_newton(const double *l, int na, int ndim, double *lambda, double beta,
                      const orc_params *prm, int *iters) {
  int MaxIter = prm->max_iter_lme;
  int NumIter = 0;
  double p[ORC_MAXNB], r[3], J[9], Jm1[9];
  double norm_r = 10;
  while (NumIter <= MaxIter) {
    orc_p_lme(p, l, na, ndim, lambda, beta);
    r_lme(r, l, p, na, ndim);
    double aux = 0.0; /* norm__MatrixLib__(r,2), MatrixOp.c:843-870 */
    for (int i = 0; i < ndim; i++) aux += dsqr(r[i]);
    norm_r = pow(aux, 0.5);
    if (norm_r > prm->tol_wrapper_lme) {
      J_lme(J, l, p, r, na, ndim);
      if (orc_rcond_ref(J, ndim) < 1E-8) {
        if (iters) *iters = NumIter;
        return 1;
      }
      /* solve__MatrixLib__, MatrixOp.c:1014-1032: inverse then product */
      if (orc_inverse(Jm1, J, ndim)) return 1;
      for (int i = 0; i < ndim; i++) {
        double d = 0.0;
        for (int j = 0; j < ndim; j++) d += Jm1[i * ndim + j] * r[j];
        lambda[i] -= d;
      }
      NumIter++;
    } else {
      break;
    }
  }
  if (iters) *iters = NumIter;
  if (NumIter >= MaxIter) return 1;
  (void)norm_r;
  return 0;
}

/* point_distance__MeshTools__, Nodes-Tools.c:397-420: sqrt(sum pow(d,2)) */
static inline double point_distance(const double *a, const double *b, int ndim) {
  double D = 0;
  for (int i = 0; i < ndim; i++) D += pow(a[i] - b[i], 2);
  return sqrt(D);
}

/* get_closest_node__MeshTools__, Nodes-Tools.c:476-538: strict '<' => first minimum in chain order */
static int closest_node(const double *x, const int *chain, int n, const double *coords, int ndim) {
  int I = chain[0];
  double DistMin = point_distance(x, &coords[I * ndim], ndim);
  int I_DistMin = I;
  for (int q = 1; q < n; q++) {
    I = chain[q];
    double d = point_distance(x, &coords[I * ndim], ndim);
    if (d < DistMin) {
      DistMin = d;
      I_DistMin = I;
    }
  }
  return I_DistMin;
}


/* tributary__LME__, LME.c:1019-1099.  Walks NodalLocality[I0] in array (= chain) order, keeps
 * active nodes with generalised_Euclidean_distance (MatrixOp.c:895-920, identity metric) <= Ra,
 * and PUSHES each one (prepend, ChainOp.c:163-182) => the output chain is in REVERSE walk order.
 * Returns the count, or -1 when < ndim+1 nodes (the reference exit()s, LME.c:1087-1092). */
static int tributary(int *list, const double *x, double beta_p, int I0, const orc_mesh *M,
                     const orc_params *prm) {
  int ndim = M->ndim;
  int tmp[ORC_MAXNB], nt = 0;
  double Ra = sqrt(-log(prm->tol_zero_lme) / beta_p);
  for (int q = M->r2_ptr[I0]; q < M->r2_ptr[I0 + 1]; q++) {
    int Node0 = M->r2[q];
    if (M->active[Node0]) {
      double sqr_distance = 0;
      for (int i = 0; i < ndim; i++) {
        double la = x[i] - M->coords[Node0 * ndim + i]; /* substraction__MatrixLib__(X_p, X_I) */
        sqr_distance += la * la;                        /* identity metric: la_i * (1*la_i) */
      }
      if (sqrt(sqr_distance) <= Ra) tmp[nt++] = Node0;
    }
  }
  if (nt < ndim + 1) return -1;
  for (int a = 0; a < nt; a++) list[a] = tmp[nt - 1 - a];
  return nt;
}

/* compute_distance__MeshTools__, Nodes-Tools.c:424-446: l_a = x_p - x_a in list order */
static void compute_distance(double *l, const int *list, int nn, const double *x, const orc_mesh *M) {
  int ndim = M->ndim;
  for (int a = 0; a < nn; a++)
    for (int i = 0; i < ndim; i++) l[a * ndim + i] = x[i] - M->coords[list[a] * ndim + i];
}

/* beta__LME__, LME.c:177-185 */
static inline double beta_lme(double gamma, double h_avg) { return gamma / (h_avg * h_avg); }

/* activation loop shared by initialize__LME__ (LME.c:122-141) and local_search__LME__ (:949-965) */
static void activate_one_rings(const orc_particles *P, orc_mesh *M) {
  for (int p = 0; p < P->np; p++) {
    int I0 = P->I0[p];
    for (int q = M->r1_ptr[I0]; q < M->r1_ptr[I0 + 1]; q++) M->active[M->r1[q]] = 1;
  }
}

/* third loop of initialize__LME__ (LME.c:143-173) and local_search__LME__ (:969-1006) */
static int lists_and_lambda(orc_particles *P, orc_mesh *M, const orc_params *prm) {
  int STATUS = 0;
  int ndim = M->ndim;
#pragma omp parallel for schedule(static) reduction(| : STATUS)
  for (int p = 0; p < P->np; p++) {
    double l[ORC_MAXNB * 3];
    const double *x = &P->x[p * ndim];
    double Beta_p = P->beta[p]; /* previous beta (0 at initialisation => Ra = +inf) */
    int nn = tributary(&P->list[(size_t)p * ORC_MAXNB], x, Beta_p, P->I0[p], M, prm);
    if (nn < 0) {
      P->nn[p] = 0;
      if (P->status) P->status[p] |= 2;
      STATUS |= 1;
      continue;
    }
    P->nn[p] = nn;
    compute_distance(l, &P->list[(size_t)p * ORC_MAXNB], nn, x, M);
    Beta_p = beta_lme(prm->gamma_lme, M->h_avg[P->I0[p]]);
    P->beta[p] = Beta_p;
    int st = orc_lambda_newton(l, nn, ndim, &P->lambda[p * ndim], Beta_p, prm, NULL);
    if (st) {
      if (P->status) P->status[p] |= 1;
      STATUS |= 1;
    }
  }
  return STATUS;
}

/* initialize__LME__, LME.c:45-173 (wrapper_LME = Newton-Raphson).  The element search (:73-108)
 * takes the FIRST element in index order whose closed box contains the particle (Q4.c:305-338);
 * on the structured grid that is the lowest cell index per axis.  I0 = closest node of that
 * element in connectivity-chain order (:90-91). */
int orc_initialize_lme(orc_particles *P, orc_mesh *M, const orc_params *prm) {
  int ndim = M->ndim;
  grid_t g;
  g.ndim = ndim;
  for (int a = 0; a < 3; a++) {
    g.n[a] = M->n[a];
    g.nc[a] = (a < ndim) ? M->n[a] - 1 : 1;
  }
  for (int p = 0; p < P->np; p++) {
    const double *x = &P->x[p * ndim];
    int c[3] = {0, 0, 0};
    for (int a = 0; a < ndim; a++) {
      int ci = (int)floor((x[a] - M->origin[a]) / M->h);
      if (ci < 0) ci = 0;
      if (ci > g.nc[a] - 1) ci = g.nc[a] - 1;
      /* lowest-index cell whose closed interval holds x */
      while (ci > 0 && x[a] <= M->origin[a] + M->h * (double)ci) ci--;
      while (ci < g.nc[a] - 1 && x[a] > M->origin[a] + M->h * (double)(ci + 1)) ci++;
      if (x[a] < M->origin[a] + M->h * (double)ci || x[a] > M->origin[a] + M->h * (double)(ci + 1))
        return 1; /* LME.c:110-114: particle not found */
      c[a] = ci;
    }
    int ch[8];
    int nn = element_chain(&g, c[0], c[1], c[2], ch);
    P->I0[p] = closest_node(x, ch, nn, M->coords, ndim);
  }
  activate_one_rings(P, M);
  return lists_and_lambda(P, M, prm);
}

/* local_search__MeshTools__ (Shape-Functions.c:31-90) + local_search__LME__ (LME.c:895-1015) */
int orc_local_search(orc_particles *P, orc_mesh *M, const orc_params *prm) {
  int ndim = M->ndim;
  memset(M->active, 0, (size_t)M->nnodes); /* Shape-Functions.c:38-46 */
#pragma omp parallel for schedule(static)
  for (int p = 0; p < P->np; p++) {
    double aux = 0.0; /* norm__MatrixLib__(dis_p,2) > 0, LME.c:924 */
    for (int i = 0; i < ndim; i++) aux += dsqr(P->dis[p * ndim + i]);
    if (pow(aux, 0.5) > 0.0) {
      int I0 = P->I0[p];
      P->I0[p] = closest_node(&P->x[p * ndim], &M->r1[M->r1_ptr[I0]], M->r1_ptr[I0 + 1] - M->r1_ptr[I0],
                              M->coords, ndim);
    }
  }
  activate_one_rings(P, M);
  return lists_and_lambda(P, M, prm);
}

/* The two halves of orc_local_search, exposed so that multi-rank tests can exchange ActiveNode[]
 * between them (the build's slab partition ORs the flags of the ghost layers at that point). */
int orc_search_phase1(orc_particles *P, orc_mesh *M) {
  int ndim = M->ndim;
  memset(M->active, 0, (size_t)M->nnodes);
  for (int p = 0; p < P->np; p++) {
    double aux = 0.0;
    for (int i = 0; i < ndim; i++) aux += dsqr(P->dis[p * ndim + i]);
    if (pow(aux, 0.5) > 0.0) {
      int I0 = P->I0[p];
      P->I0[p] = closest_node(&P->x[p * ndim], &M->r1[M->r1_ptr[I0]], M->r1_ptr[I0 + 1] - M->r1_ptr[I0],
                              M->coords, ndim);
    }
  }
  activate_one_rings(P, M);
  return 0;
}
int orc_search_phase2(orc_particles *P, orc_mesh *M, const orc_params *prm) { return lists_and_lambda(P, M, prm); }

/* compute_N__MeshTools__, Shape-Functions.c:163-195 (LME branch) */
int orc_compute_N(double *N, const orc_particles *P, const orc_mesh *M, int p) {
  double l[ORC_MAXNB * 3];
  int nn = P->nn[p];
  compute_distance(l, &P->list[(size_t)p * ORC_MAXNB], nn, &P->x[p * M->ndim], M);
  orc_p_lme(N, l, nn, M->ndim, &P->lambda[p * M->ndim], P->beta[p]);
  return nn;
}

/* compute_dN__MeshTools__, Shape-Functions.c:319-354 (LME branch) */
int orc_compute_dN(double *dN, const orc_particles *P, const orc_mesh *M, int p) {
  double l[ORC_MAXNB * 3], N[ORC_MAXNB];
  int nn = P->nn[p];
  compute_distance(l, &P->list[(size_t)p * ORC_MAXNB], nn, &P->x[p * M->ndim], M);
  orc_p_lme(N, l, nn, M->ndim, &P->lambda[p * M->ndim], P->beta[p]);
  if (orc_dp_lme(dN, l, N, nn, M->ndim)) return -1;
  return nn;
}

/* ======================================================================================
 * Masks, Nodes/Nodes-Tools.c:46-156
 * ====================================================================================== */

/* get_active_nodes__MeshTools__, Nodes-Tools.c:46-66 */
int orc_active_nodes(int *nodes2mask, const orc_mesh *M) {
  int Nactivenodes = 0;
  for (int A = 0; A < M->nnodes; A++) {
    if (M->active[A]) {
      nodes2mask[A] = Nactivenodes;
      Nactivenodes++;
    } else {
      nodes2mask[A] = -1;
    }
  }
  return Nactivenodes;
}

/* get_active_dofs__MeshTools__, Nodes-Tools.c:70-156 */
int orc_active_dofs(int *dofs2mask, const int *nodes2mask, int nactive, int ndof, const orc_bcc *bcc,
                    int nbcc, int step, int nsteps) {
  int Order = nactive * ndof;
  memset(dofs2mask, 0, sizeof(int) * (size_t)Order);
  for (int i = 0; i < nbcc; i++)
    for (int j = 0; j < bcc[i].nnodes; j++) {
      int Id_BCC_mask = nodes2mask[bcc[i].nodes[j]];
      if (Id_BCC_mask != -1)
        for (int k = 0; k < bcc[i].dim; k++)
          if (bcc[i].dir[k * nsteps + step] == 1) dofs2mask[Id_BCC_mask * ndof + k] = -1;
    }
  int Nactive = 0;
  for (int A_i = 0; A_i < Order; A_i++)
    if (dofs2mask[A_i] != -1) {
      dofs2mask[A_i] = Nactive;
      Nactive++;
    }
  return Nactive;
}

/* ======================================================================================
 * Stage functions of U_Newmark_Beta (Formulations/Displacements/U-Newmark-beta.c).  The OpenMP
 * structure (parallel for over particles, omp critical around every nodal +=) is the reference's.
 * ====================================================================================== */

/* __compute_nodal_lumped_mass, U-Newmark-beta.c:528-597 */
int orc_lumped_mass(double *Mv, const orc_particles *P, const orc_mesh *M, const int *nodes2mask) {
  int ndim = M->ndim;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < P->np; p++) {
    double N[ORC_MAXNB];
    int nn = orc_compute_N(N, P, M, p);
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    double m_p = P->mass[p];
    for (int A = 0; A < nn; A++) {
      int Mask_node_A = nodes2mask[conn[A]];
      double M_AB_p = N[A] * m_p;
#pragma omp critical
      {
        for (int i = 0; i < ndim; i++) Mv[Mask_node_A * ndim + i] += M_AB_p;
      }
    }
  }
  return 0;
}

/* __get_nodal_field_n, U-Newmark-beta.c:615-696 (the Dirichlet post-loop :698-770 is a no-op:
 * its VecSetValues are commented out) */
int orc_nodal_field_n(double *V, double *Av, const double *Mv, const orc_particles *P,
                      const orc_mesh *M, const int *nodes2mask, const int *dofs2mask, int nactive) {
  int ndim = M->ndim;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < P->np; p++) {
    double N[ORC_MAXNB];
    int nn = orc_compute_N(N, P, M, p);
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    double m_p = P->mass[p];
    const double *vel_p = &P->vel[p * ndim];
    const double *acc_p = &P->acc[p * ndim];
    for (int A = 0; A < nn; A++) {
      int Mask_node_A = nodes2mask[conn[A]];
      double m__x__N = m_p * N[A];
#pragma omp critical
      {
        for (int i = 0; i < ndim; i++) {
          int idx = Mask_node_A * ndim + i;
          if (dofs2mask[idx] != -1) { /* VEC_IGNORE_NEGATIVE_INDICES */
            V[idx] += m__x__N * vel_p[i];
            Av[idx] += m__x__N * acc_p[i];
          }
        }
      }
    }
  }
  /* VecPointwiseDivide :695-696.  PETSc (third-party, not vendored; src/vec/vec/impls/seq/bvec2.c,
   * VecPointwiseDivide_Seq) writes 0 where the denominator is 0: an active node no particle lists (narrow LME
   * kernels, gamma_LME >= ~5: the 1-ring activation reaches further than the cut-off radius) has zero lumped mass. */
  for (int i = 0; i < nactive * ndim; i++) {
    V[i] = Mv[i] != 0.0 ? V[i] / Mv[i] : 0.0;
    Av[i] = Mv[i] != 0.0 ? Av[i] / Mv[i] : 0.0;
  }
  return 0;
}

/* update_*_Deformation_Gradient*, Particles/compute-Strains.c:20-105,176-207 */
static void update_increment_DF(double *DF_p, const double *DeltaU, const double *gradient_p, int nn,
                                int ndim, double diag) {
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) DF_p[i * ndim + j] = diag * (i == j);
  for (int A = 0; A < nn; A++)
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) DF_p[i * ndim + j] += DeltaU[A * ndim + i] * gradient_p[A * ndim + j];
}

static void update_F_n1(double *F_n1, const double *F_n, const double *f_n1, int ndim) {
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) {
      double aux = 0;
      for (int k = 0; k < ndim; k++) aux += f_n1[i * ndim + k] * F_n[k * ndim + j];
      F_n1[i * ndim + j] = aux;
    }
}

static void update_rate_F_n1(double *dt_F_n1, const double *dt_f_n1, const double *F_n,
                             const double *f_n1, const double *dt_F_n, int ndim) {
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) {
      double aux = 0.0;
      for (int k = 0; k < ndim; k++)
        aux += dt_f_n1[i * ndim + k] * F_n[k * ndim + j] + f_n1[i * ndim + k] * dt_F_n[k * ndim + j];
      dt_F_n1[i * ndim + j] = aux;
    }
}

/* __local_compatibility_conditions, U-Newmark-beta.c:1064-1160 (F-bar off).  dU_dt may be NULL
 * (the rate tensors are consumed only by the Newtonian-fluid law, Constitutive.c:84-108). */
int orc_compatibility(const double *dU, const double *dU_dt, orc_particles *P, const orc_mesh *M,
                      const int *nodes2mask) {
  int ndim = M->ndim, T = P->T;
  int STATUS = 0;
#pragma omp parallel for schedule(static) reduction(| : STATUS)
  for (int p = 0; p < P->np; p++) {
    double dN[ORC_MAXNB * 3], dUa[ORC_MAXNB * 3], dVa[ORC_MAXNB * 3];
    int nn = orc_compute_dN(dN, P, M, p);
    if (nn < 0) {
      STATUS |= 1;
      continue;
    }
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    for (int A = 0; A < nn; A++) { /* get_set_field__MeshTools__, Nodes-Tools.c:249-275 */
      int A_mask = nodes2mask[conn[A]];
      for (int i = 0; i < ndim; i++) {
        dUa[A * ndim + i] = dU[A_mask * ndim + i];
        if (dU_dt) dVa[A * ndim + i] = dU_dt[A_mask * ndim + i];
      }
    }
    double *F_n_p = &P->F_n[p * T], *F_n1_p = &P->F_n1[p * T], *DF_p = &P->DF[p * T];
    update_increment_DF(DF_p, dUa, dN, nn, ndim, 1.0);
    update_F_n1(F_n1_p, F_n_p, DF_p, ndim);
    if (dU_dt && P->dt_DF) {
      update_increment_DF(&P->dt_DF[p * T], dVa, dN, nn, ndim, 0.0);
      update_rate_F_n1(&P->dt_F_n1[p * T], &P->dt_DF[p * T], F_n_p, DF_p, &P->dt_F_n[p * T], ndim);
    }
    P->J_n1[p] = I3(F_n1_p, ndim);
    if (P->J_n1[p] <= 0.0) { /* :1137-1142: message + clamp */
      P->J_n1[p] = 0.0;
      if (P->status) P->status[p] |= 4;
    }
  }
  return STATUS;
}

/* left_Cauchy_Green__Particles__, compute-Strains.c:365-384 */
static void left_cauchy_green(double *b, const double *F, int ndim) {
  if (ndim == 2) {
    b[0] = F[0] * F[0] + F[1] * F[1];
    b[1] = F[0] * F[2] + F[1] * F[3];
    b[2] = b[1];
    b[3] = F[2] * F[2] + F[3] * F[3];
  } else {
    b[0] = F[0] * F[0] + F[1] * F[1] + F[2] * F[2];
    b[1] = F[0] * F[3] + F[1] * F[4] + F[2] * F[5];
    b[2] = F[0] * F[6] + F[1] * F[7] + F[2] * F[8];
    b[3] = b[1];
    b[4] = F[3] * F[3] + F[4] * F[4] + F[5] * F[5];
    b[5] = F[3] * F[6] + F[4] * F[7] + F[5] * F[8];
    b[6] = b[2];
    b[7] = b[5];
    b[8] = F[6] * F[6] + F[7] * F[7] + F[8] * F[8];
  }
}

/* compute_Kirchhoff_Stress_Neo_Hookean__Constitutive__, Hyperelastic/Neo-Hookean.c:38-85
 * (+ energy :18-34; I1 = trace of the d x d block, TensorLib.c:113-125 with the 3-D typo read as
 * the obvious trace). */
static int stress_neo_hookean(int ndim, const orc_material *mat, const double *F_n1, double J,
                              double *T, double *W) {
  double G = mat->E / (2 * (1 + mat->nu));
  double lambda = mat->nu * mat->E / ((1 - mat->nu * 2) * (1 + mat->nu));
  double J2 = J * J;
  double c0 = lambda * 0.5 * (J2 - 1.0);
  double b[9];
  left_cauchy_green(b, F_n1, ndim);
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) {
      double Id = (i == j) ? 1.0 : 0.0;
      T[i * ndim + j] = c0 * Id + G * (b[i * ndim + j] - Id);
    }
  if (ndim == 2) T[4] = c0;
  double I1_b = (ndim == 2) ? b[0] + b[3] : b[0] + b[4] + b[8];
  double f_J = 0.25 * lambda * (J * J - 1) - 0.5 * lambda * log(J) - G * log(J);
  *W = f_J + 0.5 * G * (I1_b - ndim);
  return 0;
}

/* rotate principal values to xyz with eigenvector A = COLUMN A (Hencky.c:248-265,
 * Drucker-Prager.c:755-776, :663-710) */
static void ppal_to_xyz(double *T_xyz, const double *T_ppal, const double *eigvec, int ndim) {
  double T_aux[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int A = 0; A < ndim; A++)
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++)
        T_aux[i * ndim + j] += T_ppal[A] * eigvec[A + i * ndim] * eigvec[A + j * ndim];
  for (int i = 0; i < ndim * ndim; i++) T_xyz[i] = T_aux[i];
  if (ndim == 2) T_xyz[4] = T_ppal[2];
}

/* compute_Kirchhoff_Stress_Hencky__Constitutive__, Hyperelastic/Hencky.c:40-94,233-285 */
static int stress_hencky(int ndim, const orc_material *mat, const double *F_n1, double *T, double *W) {
  double b[9], eigvec[9] = {0}, eigval[3] = {0.0, 0.0, 1.0}; /* 2-D: third eigenvalue fixed 1.0, :48 */
  double E = mat->E, nu = mat->nu;
  double Lame = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu));
  double G = E / (2.0 * (1.0 + nu));
  double AA[9] = {Lame + 2 * G, Lame, Lame, Lame, Lame + 2 * G, Lame, Lame, Lame, Lame + 2 * G};
  left_cauchy_green(b, F_n1, ndim);
  if (orc_sym_eigen(eigval, eigvec, b, ndim)) return 1;
  double Eh[3], Tp[3];
  Eh[0] = 0.5 * log(eigval[0]);
  Eh[1] = 0.5 * log(eigval[1]);
  Eh[2] = 0.5 * log(eigval[2]);
  Tp[0] = AA[0] * Eh[0] + AA[1] * Eh[1] + AA[2] * Eh[2];
  Tp[1] = AA[3] * Eh[0] + AA[4] * Eh[1] + AA[5] * Eh[2];
  Tp[2] = AA[6] * Eh[0] + AA[7] * Eh[1] + AA[8] * Eh[2];
  ppal_to_xyz(T, Tp, eigvec, ndim);
  *W = 0.5 * (Tp[0] * Eh[0] + Tp[1] * Eh[1] + Tp[2] * Eh[2]);
  return 0;
}

/* compute_Kirchhoff_Stress_Drucker_Prager__Constitutive__, Plasticity/Drucker-Prager.c:319-613 and
 * its helpers :617-1084.  The plastic branches index eigenvectors row-wise (:957,1059) while the
 * elastic branch and the b_e corrector index column-wise (:770,699); the two agree in the only
 * buildable (2-D) reference because LAPACK's 2x2 eigenvector matrix is symmetric.  Restated with
 * the consistent COLUMN convention (SURVEY.md §7 hard part 3).  C_ep (:1088-1198, implicit tangent
 * only) is not produced. */
/* __compute_trial_b_e, Plasticity/Drucker-Prager.c:617-633: b_tr = d_phi b_e,n d_phi^T on the d x d block
 * (the reference's tests/Constitutive/test.py checks this index convention on a fixed pair of matrices) */
void orc_trial_b_e(double *btr, const double *d_phi, const double *b_e_n, int ndim) {
  for (int i = 0; i < ndim * ndim; i++) btr[i] = 0.0;
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++)
      for (int k = 0; k < ndim; k++)
        for (int l = 0; l < ndim; l++)
          btr[i * ndim + j] += d_phi[i * ndim + k] * b_e_n[k * ndim + l] * d_phi[j * ndim + l];
}

static int stress_drucker_prager(int ndim, const orc_material *mat, const orc_params *prm,
                                 const double *d_phi, const double *b_e_n, double kappa_n,
                                 double eps_n_in, double *T, double *W, double *b_e, double *kappa_out,
                                 double *eps_out, double *C_ep) {
  double eigval[3] = {0, 0, 0}, eigvec[9] = {0}, btr[9] = {0};
  orc_trial_b_e(btr, d_phi, b_e_n, ndim);
  if (orc_sym_eigen(eigval, eigvec, btr, ndim)) return 1;
  if (ndim == 2) eigval[2] = b_e_n[4];

  double Etr[3] = {0.5 * log(eigval[0]), 0.5 * log(eigval[1]), 0.5 * log(eigval[2])};

  double K = mat->E / (3.0 * (1.0 - 2.0 * mat->nu));
  double G = mat->E / (2.0 * (1.0 + mat->nu));
  double p_ref = mat->p_ref;
  double rad_friction_angle = (PI_MATRIXLIB / 180.0) * mat->phi_deg;
  double rad_dilatancy_angle = (PI_MATRIXLIB / 180.0) * mat->psi_deg;
  double exp_param = mat->exponent_ortiz;
  double kappa_0 = mat->kappa_0;
  double eps_0 = mat->eps_0;
  double alpha_F, alpha_Q, beta;
  if (ndim == 2) { /* :362-368 plane-strain match */
    alpha_F = sqrt(2. / 3.) * tan(rad_friction_angle) / sqrt(3. + 4. * dsqr(tan(rad_friction_angle)));
    alpha_Q = sqrt(2. / 3.) * tan(rad_dilatancy_angle) / sqrt(3. + 4. * dsqr(tan(rad_dilatancy_angle)));
    beta = sqrt(2. / 3.) * 3. / sqrt(3. + 4. * dsqr(tan(rad_friction_angle)));
  } else { /* :370-375 */
    alpha_F = sqrt(2 / 3.) * 2 * sin(rad_friction_angle) / (3 - sin(rad_friction_angle));
    alpha_Q = sqrt(2 / 3.) * 2 * sin(rad_dilatancy_angle) / (3 - sin(rad_dilatancy_angle));
    beta = sqrt(2 / 3.) * 6 * cos(rad_friction_angle) / (3 - sin(rad_friction_angle));
  }

  double n[3] = {0, 0, 0}, dEp[3] = {0, 0, 0};
  double PHI, PHI_0, d_PHI;
  double d_gamma_k = 0;
  double eps_n = eps_n_in, eps_k = eps_n, kappa_k = kappa_n, d_kappa_k = 0.0;
  double TOL = prm->tol_radial_returning;
  int MaxIter = prm->max_iter_radial_returning, Iter = 0;
  double Tp[3];

  /* plastic laws start from the n-state, Constitutive.c:160-168 */
  *kappa_out = kappa_n;
  *eps_out = eps_n_in;

  /* __trial_elastic :713-738 */
  double tr = Etr[0] + Etr[1] + Etr[2];
  double Evol = (1.0 / 3.0) * tr;
  double Tvol[3], Tdev[3];
  for (int a = 0; a < 3; a++) {
    Tvol[a] = -p_ref - K * Evol;
    Tdev[a] = 2 * G * (Etr[a] - Evol);
  }
  double pressure = (Tvol[0] + Tvol[1] + Tvol[2]) / 3.0;
  double J2 = sqrt(Tdev[0] * Tdev[0] + Tdev[1] * Tdev[1] + Tdev[2] * Tdev[2]);

#define YIELD_CLASSICAL(dg, kap) \
  (J2 - 2.0 * G * (dg) - 3.0 * alpha_F * (pressure - 3.0 * K * alpha_Q * (dg)) - beta * (kap))

  PHI = PHI_0 = YIELD_CLASSICAL(d_gamma_k, kappa_k); /* :886-896 */

  if (PHI_0 <= TOL_NR) { /* elastic :410-432 */
    for (int a = 0; a < 3; a++) Tp[a] = -Tvol[a] + Tdev[a];
    ppal_to_xyz(T, Tp, eigvec, ndim);
    if (C_ep) /* __tangent_moduli_elastic :1088-1113 */
      for (int i = 0; i < ndim; i++)
        for (int j = 0; j < ndim; j++)
          C_ep[i * ndim + j] = (1.0 / 3.0) * K * 1.0 * 1.0 + 2.0 * G * ((i == j ? 1.0 : 0.0) - (1.0 / 3.0) * 1.0 * 1.0);
  } else {
    if (J2 > TOL_NR) { /* __compute_plastic_flow_direction :798-812 */
      n[0] = Tdev[0] / J2;
      n[1] = Tdev[1] / J2;
      n[2] = Tdev[2] / J2;
    }
    /* __d_kappa :849-863 */
    {
      double base = 1.0 + eps_n / eps_0;
      if (base < 0.0) return 1;
      d_kappa_k = (kappa_0 / (exp_param * eps_0)) * pow(base, 1.0 / exp_param - 1.0);
    }
    /* __compute_pressure_limit :867-882 */
    double ads = sqrt(1.0 + 3.0 * alpha_Q * alpha_Q);
    if (alpha_F == 0.0) return 1;
    double pressure_limit = 3.0 * alpha_Q * K / (2.0 * G) * J2 +
                            beta / (3.0 * alpha_F) * ((J2 / (2.0 * G)) * d_kappa_k * ads + kappa_k);

    if (-pressure < pressure_limit) { /* classical return :457-530 */
      while (fabs(PHI / PHI_0) >= TOL) {
        Iter++;
        if (Iter == MaxIter) break;
        d_PHI = 9.0 * K * alpha_F * alpha_Q - 2.0 * G - beta * d_kappa_k * ads; /* :900-909 */
        if (fabs(d_PHI) < TOL) return 1;
        d_gamma_k += -PHI / d_PHI;
        if (d_gamma_k < 0.0) return 1;
        eps_k = eps_n + d_gamma_k * sqrt(3.0 * alpha_Q * alpha_Q + 1.0); /* __eps :816-829 */
        if (eps_k < 0.0) return 1;
        {
          double base = 1.0 + eps_k / eps_0; /* __kappa :833-847 */
          if (base < 0.0) return 1;
          kappa_k = kappa_0 * pow(base, 1.0 / exp_param);
          if (kappa_k < 0.0) return 1;
          d_kappa_k = (kappa_0 / (exp_param * eps_0)) * pow(base, 1.0 / exp_param - 1.0);
        }
        PHI = YIELD_CLASSICAL(d_gamma_k, kappa_k);
      }
      for (int a = 0; a < 3; a++) /* :913-927 */
        Tp[a] = -Tvol[a] + Tdev[a] + d_gamma_k * (3 * K * alpha_Q - 2 * G * n[a]);
      *eps_out = eps_k; /* :931-947 */
      *kappa_out = kappa_k;
      for (int a = 0; a < 3; a++) dEp[a] = d_gamma_k * (alpha_Q + n[a]);
      ppal_to_xyz(T, Tp, eigvec, ndim);
      if (C_ep) { /* __tangent_moduli_classical :1117-1159 */
        double c0 = 9 * alpha_F * alpha_Q * K + 2 * G + beta * d_kappa_k * sqrt(2. / 3. * (1 + 3 * alpha_Q * alpha_Q));
        double c1 = 1.0 - 9.0 * alpha_F * alpha_Q * K / c0;
        double c2 = 0.0;
        if (J2 > TOL_NR) c2 = d_gamma_k / J2;
        for (int i = 0; i < ndim; i++)
          for (int j = 0; j < ndim; j++)
            C_ep[i * ndim + j] = c1 * K * 1.0 * 1.0 +
                                 2 * G * ((i == j ? 1.0 : 0.0) - (1. / 3.) * (1.0 - 2.0 * G * c2) * 1.0 * 1.0) -
                                 (6.0 * alpha_Q * K * G / c0) * 1.0 * n[j] - (6.0 * alpha_Q * K * G / c0) * n[i] * 1.0 -
                                 4 * G * G * (1.0 / c0 - c2) * n[i] * n[j];
      }
    } else { /* apex return :532-590 */
      double d_gamma_1 = J2 / (2.0 * G);
      double d_gamma_2_k = 0.0;
      d_gamma_k = d_gamma_1 + d_gamma_2_k;
      while (fabs(PHI / PHI_0) >= TOL) {
        Iter++;
        if (Iter == MaxIter) break;
        d_PHI = 3.0 * alpha_Q * K + /* __d_yield_function_apex :1004-1018 */
                3.0 * d_kappa_k * beta * (alpha_Q * alpha_Q) * d_gamma_k /
                    (3.0 * alpha_F *
                     sqrt((d_gamma_1 * d_gamma_1) + 3.0 * (alpha_Q * alpha_Q) * (d_gamma_k * d_gamma_k)));
        if (fabs(d_PHI) < TOL) break;
        d_gamma_2_k += -PHI / d_PHI;
        if (d_gamma_2_k < 0.0) {
          d_gamma_k = 0.0;
          d_gamma_2_k = 0.0;
          break;
        } else {
          d_gamma_k = d_gamma_1 + d_gamma_2_k;
        }
        PHI = (beta / (3.0 * alpha_F) * /* __yield_function_apex :987-1000 */
                   (kappa_k + d_kappa_k * sqrt((d_gamma_1 * d_gamma_1) +
                                               3.0 * (alpha_Q * alpha_Q) * (d_gamma_k * d_gamma_k))) -
               pressure + 3.0 * K * alpha_Q * d_gamma_k);
      }
      eps_k = eps_n + d_gamma_k * sqrt(3.0 * alpha_Q * alpha_Q + 1.0);
      if (eps_k < 0.0) return 1;
      for (int a = 0; a < 3; a++) Tp[a] = -Tvol[a] + d_gamma_k * 3 * K * alpha_Q; /* :1022-1032 */
      *eps_out = eps_k; /* :1036-1051 */
      *kappa_out = kappa_k;
      for (int a = 0; a < 3; a++) dEp[a] = d_gamma_k * alpha_Q + d_gamma_1 * n[a];
      ppal_to_xyz(T, Tp, eigvec, ndim);
      if (C_ep) { /* __tangent_moduli_apex :1163-1198 */
        double c0 = 0.0, c1 = 0.0;
        if (d_gamma_k > 0.0) {
          c0 = (alpha_Q * beta * sqrt(2. / 3.) * d_kappa_k * d_gamma_k) /
               (3.0 * alpha_F * K * sqrt(d_gamma_1 * d_gamma_1 + 3.0 * alpha_Q * alpha_Q * d_gamma_k * d_gamma_k) +
                alpha_Q * beta * sqrt(2. / 3.) * d_kappa_k * d_gamma_k);
          c1 = c0 * K / (2.0 * alpha_Q * G * d_gamma_k);
        }
        for (int i = 0; i < ndim; i++)
          for (int j = 0; j < ndim; j++) C_ep[i * ndim + j] = c0 * K * 1.0 * 1.0 + c1 * 1.0 * n[j];
      }
    }
  }
#undef YIELD_CLASSICAL

  Etr[0] -= dEp[0]; /* :593-597 */
  Etr[1] -= dEp[1];
  Etr[2] -= dEp[2];
  *W = 0.5 * (Tp[0] * Etr[0] + Tp[1] * Etr[1] + Tp[2] * Etr[2]);

  /* __corrector_b_e :663-710 */
  double ev[3] = {exp(2 * Etr[0]), exp(2 * Etr[1]), exp(2 * Etr[2])};
  ppal_to_xyz(b_e, ev, eigvec, ndim); /* same column-wise rotation; 2-D: b_e[4] = ev[2] */
  return 0;
}

/* Stress_integration__Constitutive__, Constitutive/Constitutive.c:18-258 (the three laws on the path) */
/* compute_Kirchhoff_Stress_Von_Mises__Constitutive__, Plasticity/Von-Mises.c:212-392 (helpers :396-757): J2
 * plasticity in principal Hencky strains with combined isotropic (linear + Voce) / kinematic hardening.  The back
 * stress (3 principal values) is updated in place like upstream.  Eigenvectors by column everywhere (upstream
 * indexes them by row in __update_internal_variables_plastic, :673-676; the two agree for dsyev's 2x2 output and
 * 3-D has no compilable reference). */
static int stress_von_mises(int ndim, const orc_material *mat, const orc_params *prm, const double *d_phi,
                            const double *b_e_n, double eps_n_in, double *back, double *T, double *W, double *b_e,
                            double *eps_out, double *C_ep) {
  double eigval[3] = {0, 0, 0}, eigvec[9] = {0}, btr[9] = {0};
  orc_trial_b_e(btr, d_phi, b_e_n, ndim); /* __compute_trial_b_e :396-441 */
  if (orc_sym_eigen(eigval, eigvec, btr, ndim)) return 1;
  if (ndim == 2) eigval[2] = b_e_n[4];
  double Etr[3] = {0.5 * log(eigval[0]), 0.5 * log(eigval[1]), 0.5 * log(eigval[2])};
  double K = mat->E / (3.0 * (1.0 - 2.0 * mat->nu));
  double G = mat->E / (2.0 * (1.0 + mat->nu));
  double sigma_y = mat->kappa_0, H = mat->hardening_modulus, theta = mat->theta_voce;
  double K_0 = mat->K0_voce, K_inf = mat->Kinf_voce, delta = mat->delta_voce;
  double n[3] = {0, 0, 0}, dEp[3] = {0, 0, 0}, T_back[3] = {back[0], back[1], back[2]};
  double kappa_n[2], kappa_k[2], d_kappa_k[2];
  double PHI, PHI_0, d_PHI, J2, eps_n = eps_n_in, eps_k = eps_n, d_gamma_k = 0.0;
  double TOL = prm->tol_radial_returning;
  int MaxIter = prm->max_iter_radial_returning, Iter = 0;
  double Tp[3], Tvol[3], Tdev[3];
  *eps_out = eps_n_in;
  /* __trial_elastic :495-519 */
  double Evol = (1.0 / 3.0) * (Etr[0] + Etr[1] + Etr[2]);
  for (int a = 0; a < 3; a++) {
    Tvol[a] = K * Evol;
    Tdev[a] = 2 * G * (Etr[a] - Evol) - T_back[a];
  }
  J2 = sqrt(Tdev[0] * Tdev[0] + Tdev[1] * Tdev[1] + Tdev[2] * Tdev[2]);
#define VM_KAPPA(k, e)                                                                          \
  do {                                                                                          \
    if ((e) < 0.0) return 1;                                                                    \
    (k)[0] = sigma_y + theta * H * (e) + (K_inf - K_0) * (1 - exp(-delta * (e))); /* :591-604 */ \
    (k)[1] = (1 - theta) * H * (e);                                                             \
  } while (0)
#define VM_YIELD(kk, dg) (J2 - sqrt(2. / 3.) * ((kk)[0] + (kk)[1] - kappa_n[1]) - 2.0 * G * (dg)) /* :624-633 */
  VM_KAPPA(kappa_n, eps_n);
  PHI_0 = VM_YIELD(kappa_n, d_gamma_k);
  kappa_k[0] = kappa_n[0];
  kappa_k[1] = kappa_n[1];
  if (PHI_0 <= 0.0) { /* elastic :283-297 */
    for (int a = 0; a < 3; a++) Tp[a] = Tvol[a] + Tdev[a];
    ppal_to_xyz(T, Tp, eigvec, ndim);
  } else {
    for (int a = 0; a < 3; a++) n[a] = Tdev[a] / J2; /* :580-587 */
    PHI = PHI_0;
    while (fabs(PHI / PHI_0) >= TOL) { /* :312-340 */
      Iter++;
      if (Iter == MaxIter) break;
      if (eps_k < 0.0) return 1;
      d_kappa_k[0] = theta * H + delta * (K_inf - K_0) * exp(-delta * eps_k); /* :608-620 */
      d_kappa_k[1] = (1 - theta) * H;
      d_PHI = -2.0 * G * (1.0 + (d_kappa_k[0] + d_kappa_k[1]) / (3 * G)); /* :637-642 */
      d_gamma_k += -PHI / d_PHI;
      eps_k = eps_n + sqrt(2. / 3.) * d_gamma_k;
      VM_KAPPA(kappa_k, eps_k);
      PHI = VM_YIELD(kappa_k, d_gamma_k);
    }
    double d_K_kin = kappa_k[1] - kappa_n[1];
    for (int a = 0; a < 3; a++) Tp[a] = Tvol[a] + Tdev[a] + T_back[a] - d_gamma_k * 2 * G * n[a]; /* :646-654 */
    *eps_out = eps_k; /* __update_internal_variables_plastic :658-713 */
    for (int a = 0; a < 3; a++) dEp[a] = d_gamma_k * n[a];
    ppal_to_xyz(T, Tp, eigvec, ndim);
    for (int a = 0; a < 3; a++) back[a] += sqrt(2. / 3.) * d_K_kin * n[a];
  }
#undef VM_KAPPA
#undef VM_YIELD
  Etr[0] -= dEp[0];
  Etr[1] -= dEp[1];
  Etr[2] -= dEp[2];
  { /* __corrector_b_e :444-491 */
    double lam[3] = {exp(2 * Etr[0]), exp(2 * Etr[1]), exp(2 * Etr[2])};
    int Tn = ndim == 2 ? 5 : 9;
    for (int i = 0; i < Tn; i++) b_e[i] = 0.0;
    for (int A = 0; A < ndim; A++)
      for (int i = 0; i < ndim; i++)
        for (int j = 0; j < ndim; j++) b_e[i * ndim + j] += lam[A] * eigvec[A + i * ndim] * eigvec[A + j * ndim];
    if (ndim == 2) b_e[4] = lam[2];
  }
  if (C_ep) { /* __tangent_moduli :717-757 */
    double th = 0.0;
    if (J2 > TOL_NR) th = 1.0 - 2.0 * G * d_gamma_k / J2;
    double theta_bar = 1.0 / (1.0 + (kappa_k[0] + kappa_k[1]) / (3.0 * G)) - (1.0 - th);
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++)
        C_ep[i * ndim + j] = K * 1.0 * 1.0 + 2.0 * G * th * ((i == j ? 1.0 : 0.0) - (1.0 / 3.0) * 1.0 * 1.0) -
                             2.0 * G * theta_bar * n[i] * n[j];
  }
  *W = 0.5 * (Tp[0] * Etr[0] + Tp[1] * Etr[1] + Tp[2] * Etr[2]);
  return 0;
}

/* ======================================================================================
 * Matsuoka-Nakai and Lade-Duncan: the monolithic three-invariant return mapping with line search of
 * Plasticity/Matsuoka-Nakai.c:300-700 and Plasticity/Lade-Duncan.c:290-692 (Borja et al. 2003).  The two files share
 * every statement but the yield / potential surfaces (Matsuoka-Nakai.c:961-1053, Lade-Duncan.c:959-1035) and what the
 * plastic branch does with the trial strain (:432-434 against Lade-Duncan.c:430-432), so one routine restates both.
 * Unknowns of the local Newton iteration: principal Kirchhoff stresses (shifted by c cot(phi)), kappa_phi and the
 * plastic multiplier; the 5 x 5 system goes through LAPACKE_dgetrf / dgetrs upstream (:1120-1163), partial pivoting
 * here.  Kept as written: the residual added to the diagonal of the tangent (:505-510), the line search that steps
 * along the NEW residual (:583-587), the elastic branch that leaves E_hencky_k1 at zero so the corrector writes
 * b_e = 1 (:410-424, :694), C_ep written in 2-D only after a plastic step (:1285-1290).  Eigenvectors by column
 * everywhere, as for the other laws (upstream indexes them by row in __update_internal_variables_plastic, :1190).
 * ====================================================================================== */
static int lu_solve5(double A[25], double b[5]) { /* A x = b in place, row-major, partial pivoting (first maximum) */
  for (int k = 0; k < 5; k++) {
    int piv = k;
    double big = fabs(A[k * 5 + k]);
    for (int r = k + 1; r < 5; r++)
      if (fabs(A[r * 5 + k]) > big) {
        big = fabs(A[r * 5 + k]);
        piv = r;
      }
    if (A[piv * 5 + k] == 0.0) return 1; /* dgetrf INFO > 0 (:1130-1147) */
    if (piv != k) {
      for (int c = 0; c < 5; c++) {
        double t = A[k * 5 + c];
        A[k * 5 + c] = A[piv * 5 + c];
        A[piv * 5 + c] = t;
      }
      double t = b[k];
      b[k] = b[piv];
      b[piv] = t;
    }
    for (int r = k + 1; r < 5; r++) {
      double f = A[r * 5 + k] / A[k * 5 + k];
      for (int c = k + 1; c < 5; c++) A[r * 5 + c] -= f * A[k * 5 + c];
      b[r] -= f * b[k];
    }
  }
  for (int k = 4; k >= 0; k--) {
    double t = b[k];
    for (int c = k + 1; c < 5; c++) t -= A[k * 5 + c] * b[c];
    b[k] = t / A[k * 5 + k];
  }
  return 0;
}

static int inverse3_pivot(double A[9]) { /* in place, partial pivoting (dgetrf_ + dgetri_, :1241-1283) */
  double M[3][6];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      M[i][j] = A[i * 3 + j];
      M[i][3 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int k = 0; k < 3; k++) {
    int piv = k;
    for (int r = k + 1; r < 3; r++)
      if (fabs(M[r][k]) > fabs(M[piv][k])) piv = r;
    if (M[piv][k] == 0.0) return 1;
    if (piv != k)
      for (int c = 0; c < 6; c++) {
        double t = M[k][c];
        M[k][c] = M[piv][c];
        M[piv][c] = t;
      }
    double d = M[k][k];
    for (int c = 0; c < 6; c++) M[k][c] /= d;
    for (int r = 0; r < 3; r++) {
      if (r == k) continue;
      double f = M[r][k];
      for (int c = 0; c < 6; c++) M[r][c] -= f * M[k][c];
    }
  }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) A[i * 3 + j] = M[i][3 + j];
  return 0;
}

typedef struct {
  int lade_duncan;
  double I1, I2, I3;
} fric_inv;

static void fric_invariants(fric_inv *v, const double *T) { /* :399-401 */
  v->I1 = T[0] + T[1] + T[2];
  v->I2 = T[0] * T[1] + T[1] * T[2] + T[0] * T[2];
  v->I3 = T[0] * T[1] * T[2];
}
/* __F: Matsuoka-Nakai.c:961-965, Lade-Duncan.c:959-964 */
static double fric_F(const fric_inv *v, double kappa_phi) {
  if (v->lade_duncan) return cbrt((27.0 + kappa_phi) * v->I3) - v->I1;
  return cbrt((9.0 + kappa_phi) * v->I3) - cbrt(v->I1 * v->I2);
}
/* the part the two gradients share in Matsuoka-Nakai (:974-976): d cbrt(I1 I2) / d tau_A */
static double fric_grad_g(const fric_inv *v, const double *T, int A) {
  return (v->I1 * (v->I1 - T[A]) + v->I2) / (3.0 * pow(cbrt(v->I1 * v->I2), 2.0));
}
/* __d_F_d_stress / __d_G_d_stress: Matsuoka-Nakai.c:969-1007, Lade-Duncan.c:968-1001 (same form, kappa_phi | kappa_psi) */
static void fric_dsurf(double *d, const fric_inv *v, const double *T, double kap) {
  for (int A = 0; A < 3; A++) {
    if (v->lade_duncan) d[A] = cbrt((27.0 + kap) * v->I3) / (3.0 * T[A]) - 1.0;
    else d[A] = cbrt((9.0 + kap) * v->I3) / (3.0 * T[A]) - fric_grad_g(v, T, A);
  }
}
/* __d_F_d_kappa_phi: :986-991 | Lade-Duncan.c:979-985 */
static double fric_dF_dkappa(const fric_inv *v, double kappa_phi) {
  double K1 = (v->lade_duncan ? 27.0 : 9.0) + kappa_phi;
  return (1.0 / 3.0) * pow(cbrt(K1), -2.0) * cbrt(v->I3);
}
/* __dd_G_dd_stress: :1011-1040 | Lade-Duncan.c:1005-1017 */
static void fric_ddG(double *dd, const fric_inv *v, const double *T, double kappa_psi) {
  double K2 = (v->lade_duncan ? 27.0 : 9.0) + kappa_psi;
  for (int A = 0; A < 3; A++)
    for (int B = 0; B < 3; B++) {
      double first = (1.0 / 3.0) * cbrt(K2 * v->I3) * (1.0 / (3.0 * T[A] * T[B]) - 1.0 * (A == B) / pow(T[A], 2.0));
      if (v->lade_duncan) dd[A * 3 + B] = first;
      else {
        double ddg = pow(cbrt(v->I1 * v->I2), -2.0) / 3.0 * (3.0 * v->I1 - T[A] - T[B] - v->I1 * (A == B)) -
                     (2.0 / cbrt(v->I1 * v->I2)) * fric_grad_g(v, T, A) * fric_grad_g(v, T, B);
        dd[A * 3 + B] = first - ddg;
      }
    }
}
/* __dd_G_d_stress_d_kappa_psi: :1042-1053 | Lade-Duncan.c:1020-1031 (which drops the 3 cbrt(K2)^2 divisor) */
static void fric_ddG_dkappa(double *d, const fric_inv *v, const double *T, double kappa_psi) {
  for (int A = 0; A < 3; A++) {
    d[A] = cbrt(v->I3) / (3.0 * T[A]);
    if (!v->lade_duncan) d[A] = d[A] / (3.0 * pow(cbrt(9.0 + kappa_psi), 2));
  }
}
/* __residual :1057-1082 */
static double fric_residual(double *R, const double *E_tr, const double *E_k, const double *dG, double kappa_phi,
                            double kappa_hat, double F_k, double dl) {
  R[0] = E_k[0] - E_tr[0] + dl * dG[0];
  R[1] = E_k[1] - E_tr[1] + dl * dG[1];
  R[2] = E_k[2] - E_tr[2] + dl * dG[2];
  R[3] = kappa_phi - kappa_hat;
  R[4] = F_k;
  double n2 = 0.0;
  for (int A = 0; A < 5; A++) n2 += R[A] * R[A];
  return pow(n2, 0.5);
}

static int stress_frictional(int ndim, const orc_material *mat, const orc_params *prm, const double *d_phi,
                             const double *b_e_n, double kappa_in, double eps_in, double *T, double *W, double *b_e,
                             double *kappa_out, double *eps_out, double *C_ep) {
  double eigval[3] = {0, 0, 0}, eigvec[9] = {0}, btr[9] = {0};
  orc_trial_b_e(btr, d_phi, b_e_n, ndim); /* __compute_trial_b_e :705-747 */
  if (orc_sym_eigen(eigval, eigvec, btr, ndim)) return 1;
  if (ndim == 2) eigval[2] = b_e_n[4];
  double E_tr[3] = {0.5 * log(eigval[0]), 0.5 * log(eigval[1]), 0.5 * log(eigval[2])};
  double E_k1[3] = {0, 0, 0}, E_k2[3] = {0, 0, 0};
  fric_inv v;
  v.lade_duncan = mat->type == ORC_MAT_LADE_DUNCAN;

  const double E = mat->E, nu = mat->nu;
  const double Lame = E * nu / ((1.0 + nu) * (1.0 - 2.0 * nu)), G = E / (2.0 * (1.0 + nu));
  const double rad = (PI_MATRIXLIB / 180.0) * mat->phi_deg;
  const double c_cotphi = rad > 0.0 ? mat->cohesion / tan(rad) : 0.0;
  const double alpha = mat->alpha_borja;
  const double *a = mat->a_borja;
  /* __elastic_tangent :799-824: compliance CC and stiffness AA in principal space */
  const double CC[9] = {1.0 / E, -nu / E, -nu / E, -nu / E, 1.0 / E, -nu / E, -nu / E, -nu / E, 1.0 / E};
  const double AA[9] = {Lame + 2 * G, Lame, Lame, Lame, Lame + 2 * G, Lame, Lame, Lame, Lame + 2 * G};
#define FRIC_E_HENCKY(Eo, Tk) /* __E_hencky :841-849 */                                            \
  for (int r_ = 0; r_ < 3; r_++)                                                                   \
    (Eo)[r_] = CC[3 * r_] * ((Tk)[0] + c_cotphi) + CC[3 * r_ + 1] * ((Tk)[1] + c_cotphi) + CC[3 * r_ + 2] * ((Tk)[2] + c_cotphi)
#define FRIC_KAPPA_HAT(Lam) (a[0] * (Lam)*exp(a[1] * v.I1) * exp(-a[2] * (Lam))) /* __kappa :933-937 */

  const double Lambda_n = eps_in;
  const double kappa_n[2] = {kappa_in, alpha * kappa_in};
  const double TOL = prm->tol_radial_returning, TOL_apex = 0.1;
  const int MaxIter_k1 = prm->max_iter_radial_returning, MaxIter_k2 = 10 * prm->max_iter_radial_returning;
  double T_tr[3], T_k1[3], T_k2[3], kappa_k1[2], kappa_k2[2];
  double dG[3] = {0, 0, 0}, ddG[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, ddG_dk[3], dkappa_ds, dkappa_dl, dF[3], dF_dk;
  double R1[5] = {0, 0, 0, 0, 0}, R2[5] = {0, 0, 0, 0, 0}, TM[25];
  double F_0, F_k1, F_k2 = 0.0, N0, N1, N2, Lambda_k1, Lambda_k2, dl1 = 0.0, dl2, delta;

  *kappa_out = kappa_in; /* the n+1 state starts from n, Constitutive.c:196-203 */
  *eps_out = eps_in;
  for (int r = 0; r < 3; r++) /* __trial_elastic :828-837 */
    T_tr[r] = AA[3 * r] * E_tr[0] + AA[3 * r + 1] * E_tr[1] + AA[3 * r + 2] * E_tr[2] - c_cotphi;
  fric_invariants(&v, T_tr);
  F_0 = fric_F(&v, kappa_n[0]);
  T_k1[0] = T_tr[0];
  T_k1[1] = T_tr[1];
  T_k1[2] = T_tr[2];

  if (F_0 <= TOL_NR) { /* elastic :410-424 */
    double Tp[3] = {T_k1[0] + c_cotphi, T_k1[1] + c_cotphi, T_k1[2] + c_cotphi};
    ppal_to_xyz(T, Tp, eigvec, ndim); /* __update_internal_variables_elastic :853-908 */
    if (C_ep) /* __elastic_tangent_moduli :912-929 */
      for (int i = 0; i < ndim; i++)
        for (int j = 0; j < ndim; j++) C_ep[i * ndim + j] = AA[i * 3 + j];
  } else { /* plastic: monolithic Newton with line search :426-690 */
    FRIC_E_HENCKY(E_k1, T_k1);
    for (int r = 0; r < 3; r++) {
      if (v.lade_duncan) E_k1[r] = E_tr[r]; /* Lade-Duncan.c:430-432 */
      else E_tr[r] = E_k1[r];               /* Matsuoka-Nakai.c:432-434 */
    }
    double kappa_hat = FRIC_KAPPA_HAT(Lambda_n);
    fric_dsurf(dG, &v, T_tr, kappa_n[1]);
    N0 = fric_residual(R1, E_tr, E_k1, dG, kappa_n[0], kappa_hat, F_0, 0.0);
    kappa_k1[0] = kappa_n[0];
    kappa_k1[1] = kappa_n[1];
    F_k1 = F_0;
    dl1 = 0.0;
    Lambda_k1 = Lambda_n;
    N1 = N0;
    int Iter_k1 = 0, Iter_k2;
#define FRIC_APEX(Tk) (fabs(((Tk)[0] + (Tk)[1] + (Tk)[2]) / 3.0) < TOL_apex)
#define FRIC_EVAL_K2() /* :550-573 */                                                   \
  do {                                                                                  \
    fric_invariants(&v, T_k2);                                                          \
    FRIC_E_HENCKY(E_k2, T_k2);                                                          \
    kappa_hat = FRIC_KAPPA_HAT(Lambda_k2);                                              \
    fric_dsurf(dG, &v, T_k2, kappa_k2[1]);                                              \
    F_k2 = fric_F(&v, kappa_k2[0]);                                                     \
    N2 = fric_residual(R2, E_tr, E_k2, dG, kappa_k2[0], kappa_hat, F_k2, dl2);          \
  } while (0)
    while ((fabs(N1 / N0) >= TOL) && (fabs(F_k1 / F_0) >= TOL)) {
      delta = 1.0;
      /* hardening derivatives :941-957 (I1 of the last evaluated stress) */
      dkappa_ds = a[0] * a[1] * Lambda_k1 * exp(a[1] * v.I1) * exp(-a[2] * Lambda_k1);
      dkappa_dl = (1 - a[2] * Lambda_k1) * a[0] * exp(a[1] * v.I1) * exp(-a[2] * Lambda_k1);
      fric_dsurf(dF, &v, T_k1, kappa_k1[0]);
      dF_dk = fric_dF_dkappa(&v, kappa_k1[0]);
      fric_ddG(ddG, &v, T_k1, kappa_k1[1]);
      fric_ddG_dkappa(ddG_dk, &v, T_k1, kappa_k1[1]);
      for (int r = 0; r < 3; r++) { /* tangent :472-503 */
        for (int c = 0; c < 3; c++) TM[r * 5 + c] = CC[r * 3 + c] + dl1 * ddG[r * 3 + c];
        TM[r * 5 + 3] = alpha * dl1 * ddG_dk[r];
        TM[r * 5 + 4] = dG[r];
        TM[3 * 5 + r] = -dkappa_ds;
        TM[4 * 5 + r] = dF[r];
      }
      TM[3 * 5 + 3] = 1.0;
      TM[3 * 5 + 4] = -dkappa_dl;
      TM[4 * 5 + 3] = dF_dk;
      TM[4 * 5 + 4] = 0.0;
      for (int r = 0; r < 5; r++) TM[r * 5 + r] += R1[r]; /* "preconditioner" :505-510 */
      if (lu_solve5(TM, R1)) return 1;
      dl2 = dl1 - delta * R1[4];
      if (Lambda_n + dl2 < 0.0) break;
      Lambda_k2 = Lambda_n + dl2;
      T_k2[0] = T_k1[0] - delta * R1[0];
      T_k2[1] = T_k1[1] - delta * R1[1];
      T_k2[2] = T_k1[2] - delta * R1[2];
      kappa_k2[0] = kappa_k1[0] - delta * R1[3];
      kappa_k2[1] = alpha * kappa_k2[0];
      Iter_k2 = 0;
      if (FRIC_APEX(T_k2)) break; /* :536-544: the k2 values it resets are not used again */
      FRIC_EVAL_K2();
      while ((fabs(N2 - N1) > TOL) && (fabs(F_k2 / F_0) >= TOL)) { /* line search :575-629 */
        delta = pow(delta, 2.0) * 0.5 * N1 / (N2 - delta * N1 + N1);
        if ((delta > 1.0) || (delta < 0.0)) break;
        dl2 = dl1 - delta * R2[4];
        if (Lambda_n + dl2 < 0.0) break;
        Lambda_k2 = Lambda_n + dl2;
        T_k2[0] = T_k1[0] - delta * R2[0];
        T_k2[1] = T_k1[1] - delta * R2[1];
        T_k2[2] = T_k1[2] - delta * R2[2];
        kappa_k2[0] = kappa_k1[0] - delta * R2[3];
        kappa_k2[1] = alpha * kappa_k2[0];
        if (FRIC_APEX(T_k2)) {
          Lambda_k2 = Lambda_n;
          kappa_k2[0] = kappa_n[0];
          kappa_k2[1] = alpha * kappa_k2[0];
          T_k2[0] = T_k2[1] = T_k2[2] = 0.0;
          break;
        }
        FRIC_EVAL_K2();
        Iter_k2++;
        if (Iter_k2 == MaxIter_k2) break;
      }
      for (int r = 0; r < 3; r++) { /* :632-649 */
        T_k1[r] = T_k2[r];
        E_k1[r] = E_k2[r];
      }
      kappa_k1[0] = kappa_k2[0];
      kappa_k1[1] = kappa_k2[1];
      Lambda_k1 = Lambda_k2;
      F_k1 = F_k2;
      dl1 = dl2;
      for (int r = 0; r < 5; r++) R1[r] = R2[r];
      N1 = N2;
      Iter_k1++;
      if (FRIC_APEX(T_k1)) { /* :651-659 */
        Lambda_k1 = Lambda_n;
        kappa_k1[0] = kappa_n[0];
        kappa_k1[1] = alpha * kappa_k1[0];
        T_k1[0] = T_k1[1] = T_k1[2] = 0.0;
        break;
      }
      if (Iter_k1 == MaxIter_k1) break; /* :661-679 only prints the condition number */
    }
#undef FRIC_APEX
#undef FRIC_EVAL_K2
    *eps_out = Lambda_k1; /* __update_internal_variables_plastic :1167-1215 */
    *kappa_out = kappa_k1[0];
    double Tp[3] = {T_k1[0] + c_cotphi, T_k1[1] + c_cotphi, T_k1[2] + c_cotphi};
    ppal_to_xyz(T, Tp, eigvec, ndim);
    if (C_ep) { /* __elastoplastic_tangent_moduli :1219-1291: inverse of CC + dlambda ddG; stored in 2-D only */
      double aux[9];
      for (int i = 0; i < 9; i++) aux[i] = CC[i] + dl1 * ddG[i];
      if (inverse3_pivot(aux)) return 1;
      if (ndim == 2) {
        C_ep[0] = aux[0];
        C_ep[1] = aux[1];
        C_ep[2] = aux[3];
        C_ep[3] = aux[4];
      }
    }
  }
#undef FRIC_E_HENCKY
#undef FRIC_KAPPA_HAT
  { /* __corrector_b_e :751-795 with E_hencky_k1 (zero after an elastic step) */
    double lam[3] = {exp(2 * E_k1[0]), exp(2 * E_k1[1]), exp(2 * E_k1[2])};
    ppal_to_xyz(b_e, lam, eigvec, ndim);
  }
  *W = 0.5 * ((T_k1[0] + c_cotphi) * E_tr[0] + (T_k1[1] + c_cotphi) * E_tr[1] + (T_k1[2] + c_cotphi) * E_tr[2]); /* :696-699 */
  return 0;
}

int orc_stress_one(int ndim, const orc_material *mat, const orc_params *prm, const double *F_n1,
                   const double *DF, double J, const double *b_e_n, double kappa_n, double eps_n,
                   double *stress, double *W, double *b_e_n1, double *kappa_n1, double *eps_n1, double *C_ep,
                   double *back_stress) {
  switch (mat->type) {
  case ORC_MAT_NEO_HOOKEAN:
    return stress_neo_hookean(ndim, mat, F_n1, J, stress, W);
  case ORC_MAT_HENCKY:
    return stress_hencky(ndim, mat, F_n1, stress, W);
  case ORC_MAT_DRUCKER_PRAGER:
    return stress_drucker_prager(ndim, mat, prm, DF, b_e_n, kappa_n, eps_n, stress, W, b_e_n1,
                                 kappa_n1, eps_n1, C_ep);
  case ORC_MAT_VON_MISES: { /* Constitutive.c:110-143 */
    double zero_back[3] = {0, 0, 0};
    *kappa_n1 = kappa_n;
    return stress_von_mises(ndim, mat, prm, DF, b_e_n, eps_n, back_stress ? back_stress : zero_back, stress, W,
                            b_e_n1, eps_n1, C_ep);
  }
  case ORC_MAT_MATSUOKA_NAKAI: /* Constitutive.c:180-213 */
  case ORC_MAT_LADE_DUNCAN:    /* Constitutive.c:215-248 */
    return stress_frictional(ndim, mat, prm, DF, b_e_n, kappa_n, eps_n, stress, W, b_e_n1, kappa_n1, eps_n1, C_ep);
  default:
    return 1; /* Constitutive.c:251-256 exit()s */
  }
}

/* __constitutive_update, U-Newmark-beta.c:1208-1242 */
int orc_constitutive(orc_particles *P, const orc_material *mats, const orc_params *prm) {
  int STATUS = 0;
  int T = P->T;
#pragma omp parallel for schedule(static) reduction(| : STATUS)
  for (int p = 0; p < P->np; p++) {
    const orc_material *mat = &mats[P->matidx[p]];
    double dummyb[9], dk, de;
    int st = orc_stress_one(P->ndim, mat, prm, &P->F_n1[p * T], &P->DF[p * T], P->J_n1[p],
                            P->b_e_n ? &P->b_e_n[p * T] : dummyb, P->kappa_n ? P->kappa_n[p] : 0.0,
                            P->eps_n ? P->eps_n[p] : 0.0, &P->stress[p * T], &P->W[p],
                            P->b_e_n1 ? &P->b_e_n1[p * T] : dummyb, P->kappa_n1 ? &P->kappa_n1[p] : &dk,
                            P->eps_n1 ? &P->eps_n1[p] : &de, P->C_ep ? &P->C_ep[p * P->ndim * P->ndim] : NULL,
                            P->back_stress ? &P->back_stress[p * 3] : NULL);
    if (st) {
      if (P->status) P->status[p] |= 8;
      STATUS |= 1;
    }
  }
  return STATUS;
}

static const double *g_tangent_damage = NULL;
void orc_set_tangent_damage(const double *damage_n1) { g_tangent_damage = damage_n1; }

/* ======================================================================================
 * Eigenerosion
 * ====================================================================================== */
/* compute_Beps__Constitutive__, Constitutive/Fracture/Beps.c:16-80.  List_Particles_Node[A] holds the particles whose
 * closest node is A, pushed for p = 0 .. Np-1 (LME.c:126-129, 962-964; push prepends, ChainOp.c:163-182), i.e. in
 * DESCENDING p; the walk goes over NodalLocality_0[I0_p] in chain order and that list, and every hit is pushed
 * (prepended) onto Beps[p]: the array below is Beps[p] in chain order = the reverse of the walk.  A particle whose
 * total displacement is <= 1e-6 keeps its list unless `initialize`. */
int orc_compute_beps(int *beps_n, int *beps, int stride, const orc_particles *P, const orc_mesh *M,
                     const orc_material *mats, int initialize) {
  int np = P->np, ndim = M->ndim, STATUS = 0;
  /* List_Particles_Node as CSR (descending p inside a node) */
  int *cnt = (int *)calloc((size_t)M->nnodes + 1, sizeof(int));
  int *lst = (int *)malloc(sizeof(int) * (size_t)(np > 0 ? np : 1));
  for (int p = 0; p < np; p++) cnt[P->I0[p] + 1]++;
  for (int A = 0; A < M->nnodes; A++) cnt[A + 1] += cnt[A];
  int *fill = (int *)calloc((size_t)M->nnodes, sizeof(int));
  for (int p = np - 1; p >= 0; p--) lst[cnt[P->I0[p]] + fill[P->I0[p]]++] = p;
  for (int p = 0; p < np; p++) {
    const double *dis_p = &P->dis[p * ndim];
    double nd2 = 0.0;
    for (int i = 0; i < ndim; i++) nd2 += dis_p[i] * dis_p[i];
    if (!(sqrt(nd2) > 0.000001 || initialize)) continue; /* :30-36 */
    const double eps_distance_p = mats[P->matidx[p]].Ceps * M->h; /* :39-42 */
    const double *xp = &P->x[p * ndim];
    int tmp[4096], nt = 0;
    const int I0_p = P->I0[p];
    for (int a = M->r1_ptr[I0_p]; a < M->r1_ptr[I0_p + 1]; a++) { /* :49-52 */
      const int A = M->r1[a];
      for (int s = cnt[A]; s < cnt[A + 1]; s++) { /* :56-70 */
        const int q = lst[s];
        const double *xq = &P->x[q * ndim];
        double d2 = 0.0;
        for (int i = 0; i < ndim; i++) d2 += (xp[i] - xq[i]) * (xp[i] - xq[i]);
        if (sqrt(d2) <= eps_distance_p && nt < 4096) tmp[nt++] = q;
      }
    }
    if (nt > stride) {
      STATUS = 1;
      nt = stride;
    }
    beps_n[p] = nt;
    for (int a = 0; a < nt; a++) beps[(size_t)p * stride + a] = tmp[nt - 1 - a];
  }
  free(cnt);
  free(lst);
  free(fill);
  return STATUS;
}

/* __constitutive_update with the damage drivers on, U-Newmark-beta.c:1208-1242: a failed particle (Damage_n == 1)
 * gets W = 0 and keeps its stress */
int orc_constitutive_eroded(orc_particles *P, const orc_material *mats, const orc_params *prm, const double *damage_n) {
  int STATUS = 0;
  int T = P->T;
  for (int p = 0; p < P->np; p++) {
    if (damage_n[p] == 1.0) {
      P->W[p] = 0.0;
      continue;
    }
    const orc_material *mat = &mats[P->matidx[p]];
    double dummyb[9], dk, de;
    int st = orc_stress_one(P->ndim, mat, prm, &P->F_n1[p * T], &P->DF[p * T], P->J_n1[p],
                            P->b_e_n ? &P->b_e_n[p * T] : dummyb, P->kappa_n ? P->kappa_n[p] : 0.0,
                            P->eps_n ? P->eps_n[p] : 0.0, &P->stress[p * T], &P->W[p],
                            P->b_e_n1 ? &P->b_e_n1[p * T] : dummyb, P->kappa_n1 ? &P->kappa_n1[p] : &dk,
                            P->eps_n1 ? &P->eps_n1[p] : &de, P->C_ep ? &P->C_ep[p * P->ndim * P->ndim] : NULL,
                            P->back_stress ? &P->back_stress[p * 3] : NULL);
    if (st) STATUS |= 1;
  }
  return STATUS;
}

/* The damage part of __nodal_internal_forces (U-Newmark-beta.c:1313-1331): for every particle
 * compute_damage__Constitutive__ (Constitutive.c:385-435) -> Eigenerosion__Constitutive__ (EigenErosion.c:29-117),
 * then the Kirchhoff stress of the particle is scaled IN PLACE by (1 - Damage_n1[p]).  The neighbours enter through
 * J_n1, Vol_0, W and Damage_n only, none of which the loop modifies, so the particle order does not matter. */
int orc_eigenerosion_hook(double *damage_n1, const double *damage_n, orc_particles *P, const orc_material *mats,
                          const int *beps_n, const int *beps, int stride, double DeltaX) {
  const int ndim = P->ndim, T = P->T;
  int STATUS = 0;
  for (int p = 0; p < P->np; p++) {
    const orc_material *mat = &mats[P->matidx[p]];
    double *Stress_p = &P->stress[p * T];
    double eigval[3] = {0.0, 0.0, 0.0}, eigvec[9], blk[9];
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) blk[i * ndim + j] = Stress_p[i * ndim + j];
    if (orc_sym_eigen(eigval, eigvec, blk, ndim)) STATUS = 1; /* ascending, like dsyev (TensorLib.c:172-228) */
    if (ndim == 2) eigval[2] = Stress_p[4];
    if ((damage_n[p] < 1.0) && (eigval[0] > 0.0)) {
      const double Ceps_p = mat->Ceps, Gf_p = mat->Gf;
      const double V_p = P->vol0[p] * P->J_n1[p];
      double sum_V = V_p, sum_V_x_W = V_p * P->W[p];
      for (int a = 0; a < beps_n[p]; a++) {
        const int q = beps[(size_t)p * stride + a];
        const double V_q = P->vol0[q] * P->J_n1[q];
        sum_V += V_q;
        if (damage_n[q] < 1.0) sum_V_x_W += V_q * P->W[q];
      }
      const double G_p = (Ceps_p * DeltaX / sum_V) * sum_V_x_W;
      if (G_p > Gf_p) damage_n1[p] = 1.0;
    }
    for (int i = 0; i < T; i++) Stress_p[i] *= (1.0 - damage_n1[p]);
  }
  return STATUS;
}

/* The same hook with Driver_EigenSoftening: compute_damage__Constitutive__ (Constitutive.c:412-432) ->
 * eulerian_almansi__Particles__ (compute-Strains.c:388-429: e = (I - b^-1) / 2, b = F_n1 F_n1^T) ->
 * Eigensoftening__Constitutive__ (EigenSoftening.c:27-163), then the in-place scaling of U-Newmark-beta.c:1321-1330.
 * Restated AS WRITTEN, one particle after the other like the reference at one thread:
 *   - "first principal" = eigval[0] of the ascending dsyev order, i.e. the SMALLEST principal value (:60, :106);
 *   - the neighbour loop ASSIGNS sum_m_x_T_ppal = m_q T_q (no +=, :118): what survives is the term of the LAST list
 *     entry with Damage_n < 1 (chain order = reverse of the walk of compute_Beps, so: the first such particle of the
 *     walk), or the particle's own term when there is none;
 *   - Stress[q] is read while this very loop scales the stresses in place: a neighbour that came earlier in the loop
 *     enters with its Kirchhoff stress already multiplied by (1 - Damage_n1[q]), a later one unscaled;
 *   - StrainF_n and StrainF_n1 are the same array (Constitutive.c:418-419).
 * Beps: with this driver the lists are never initialised (U-Newmark-beta.c:182 tests Driver_EigenErosion only) and
 * compute_Beps (:213-215) only fills the list of a particle that has moved by more than 1e-6: the caller passes
 * orc_compute_beps(..., initialize = 0) over lists that start empty. */
int orc_eigensoftening_hook(double *damage_n1, const double *damage_n, double *strain_f_n1, orc_particles *P,
                            const orc_material *mats, const int *beps_n, const int *beps, int stride) {
  const int ndim = P->ndim, T = P->T;
  int STATUS = 0;
  for (int p = 0; p < P->np; p++) {
    const orc_material *mat = &mats[P->matidx[p]];
    double *Stress_p = &P->stress[p * T];
    double eigval_stress_p[3] = {0.0, 0.0, 0.0}, eigval_stress_q[3], eigval_strain_p[3] = {0.0, 0.0, 0.0}, eigvec[9], blk[9];
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) blk[i * ndim + j] = Stress_p[i * ndim + j];
    if (orc_sym_eigen(eigval_stress_p, eigvec, blk, ndim)) STATUS = 1;
    if (ndim == 2) eigval_stress_p[2] = Stress_p[4];
    /* eulerian_almansi__Particles__ on the d x d block of F_n1 */
    double b[9], b_m1[9], Strain_p[9];
    const double *F = &P->F_n1[p * T];
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) {
        double v = 0.0;
        for (int k = 0; k < ndim; k++) v += F[i * ndim + k] * F[j * ndim + k]; /* left_Cauchy_Green, compute-Strains.c:365-384 */
        b[i * ndim + j] = v;
      }
    if (orc_inverse(b_m1, b, ndim)) STATUS = 1;
    for (int i = 0; i < ndim; i++)
      for (int j = 0; j < ndim; j++) Strain_p[i * ndim + j] = 0.5 * ((i == j ? 1.0 : 0.0) - b_m1[i * ndim + j]);
    const double ft_p = mat->ft, heps_p = mat->heps, wcrit_p = mat->wcrit;
    if ((damage_n[p] == 0.0) && (eigval_stress_p[0] > 0.0)) {
      const double m_p = P->mass[p];
      double sum_m = m_p, sum_m_x_T_ppal = m_p * eigval_stress_p[0];
      for (int a = 0; a < beps_n[p]; a++) {
        const int q = beps[(size_t)p * stride + a];
        const double m_q = P->mass[q];
        sum_m += m_q;
        if (damage_n[q] < 1.0) {
          const double *Stress_q = &P->stress[q * T];
          for (int i = 0; i < ndim; i++)
            for (int j = 0; j < ndim; j++) blk[i * ndim + j] = Stress_q[i * ndim + j];
          if (orc_sym_eigen(eigval_stress_q, eigvec, blk, ndim)) STATUS = 1;
          sum_m_x_T_ppal = m_q * eigval_stress_q[0];
        }
      }
      const double Teps_p = sum_m_x_T_ppal / sum_m;
      if (Teps_p > ft_p) {
        if (orc_sym_eigen(eigval_strain_p, eigvec, Strain_p, ndim)) STATUS = 1;
        strain_f_n1[p] = eigval_strain_p[0];
      }
    } else if ((damage_n[p] != 1.0) && (strain_f_n1[p] > 0)) {
      if (orc_sym_eigen(eigval_strain_p, eigvec, Strain_p, ndim)) STATUS = 1;
      const double aux = (eigval_strain_p[0] - strain_f_n1[p]) * heps_p / wcrit_p;
      const double mx = aux > damage_n[p] ? aux : damage_n[p];
      damage_n1[p] = 1.0 < mx ? 1.0 : mx; /* DMIN(1.0, DMAX(aux, Damage_n[p])) */
    }
    for (int i = 0; i < T; i++) Stress_p[i] *= (1.0 - damage_n1[p]);
  }
  return STATUS;
}

/* __nodal_internal_forces, U-Newmark-beta.c:1257-1374 + push_forward_dN__MeshTools__,
 * Shape-Functions.c:405-448 */
int orc_internal_forces(double *R, const orc_particles *P, const orc_mesh *M, const int *nodes2mask,
                        const int *dofs2mask) {
  int ndim = M->ndim, T = P->T;
  int STATUS = 0;
#pragma omp parallel for schedule(static) reduction(| : STATUS)
  for (int p = 0; p < P->np; p++) {
    double dN[ORC_MAXNB * 3], dN1[ORC_MAXNB * 3], d_phi_mT[9];
    int nn = orc_compute_dN(dN, P, M, p);
    if (nn < 0 || adjunt(d_phi_mT, &P->DF[p * T], ndim)) {
      STATUS |= 1;
      continue;
    }
    for (int A = 0; A < nn; A++)
      for (int i = 0; i < ndim; i++) {
        double s = 0.0;
        for (int j = 0; j < ndim; j++) s += d_phi_mT[i * ndim + j] * dN[A * ndim + j];
        dN1[A * ndim + i] = s;
      }
    double V0_p = P->vol0[p];
    const double *kirchhoff_p = &P->stress[p * T];
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    for (int A = 0; A < nn; A++) {
      int Mask_node_A = nodes2mask[conn[A]];
      double f[3];
      for (int i = 0; i < ndim; i++) {
        f[i] = 0.0;
        for (int j = 0; j < ndim; j++) f[i] += kirchhoff_p[i * ndim + j] * dN1[A * ndim + j];
      }
#pragma omp critical
      {
        for (int i = 0; i < ndim; i++) {
          int idx = Mask_node_A * ndim + i;
          if (dofs2mask[idx] != -1) R[idx] += f[i] * V0_p;
        }
      }
    }
  }
  return STATUS;
}

/* __nodal_traction_forces, U-Newmark-beta.c:1376-1500: over the Neumann contours and their particles, serial.
 * ids = the particle lists one after the other (load_n[l] each), dir / val [nloads][ndim] of the current step.
 * The traction vector T keeps a component from the previous contour where Dir is 0 (:1457-1461 never reset it);
 * A0_p = Vol_0 / Thickness_Plain_Stress in 2-D (:1440), Phi.Area_0 in 3-D (:1442). */
int orc_nodal_traction_forces(double *R, const orc_particles *P, const orc_mesh *M, const int *nodes2mask,
                              const int *dofs2mask, int nloads, const int *load_n, const int *ids, const int *dir,
                              const double *val, double thickness, const double *area0) {
  int ndim = M->ndim, at = 0;
  double T[3] = {0.0, 0.0, 0.0};
  for (int l = 0; l < nloads; l++)
    for (int q = 0; q < load_n[l]; q++) {
      int p = ids[at++];
      double A0_p = (ndim == 2) ? P->vol0[p] / thickness : area0[p];
      double N[ORC_MAXNB];
      int nn = orc_compute_N(N, P, M, p);
      if (nn < 0) return 1;
      const int *conn = &P->list[(size_t)p * ORC_MAXNB];
      for (int i = 0; i < ndim; i++)
        if (dir[l * ndim + i] == 1) T[i] = val[l * ndim + i];
      for (int A = 0; A < nn; A++) {
        int Mask_node_A = nodes2mask[conn[A]];
        for (int i = 0; i < ndim; i++) {
          int idx = Mask_node_A * ndim + i;
          if (dofs2mask[idx] != -1) R[idx] += -N[A] * T[i] * A0_p;
        }
      }
    }
  return 0;
}

/* compute_stiffness_density_Neo_Hookean, Hyperelastic/Neo-Hookean.c:89-141 (the only law whose tangent is
 * restated: the spectral tangents of Hencky and the elastoplastic laws divide by eigenvalue differences down to
 * 1e-14 (Hencky.c:205-214, Elastoplastic-Tangent-Matrix.c:137-147), which no second eigen-solver reproduces) */
static void stiffness_density_neo_hookean(double *Kd, int ndim, const double *dNa_n1, const double *dNb_n1,
                                          const double *dNa_n, const double *dNb_n, const double *F_n,
                                          double J, const orc_material *mat) {
  double G = mat->E / (2 * (1 + mat->nu));
  double lambda = mat->nu * mat->E / ((1 - mat->nu * 2) * (1 + mat->nu));
  double sqr_J = J * J;
  double c0 = lambda * sqr_J;
  double c1 = G - 0.5 * lambda * (sqr_J - 1);
  double b_n[9];
  left_cauchy_green(b_n, F_n, ndim);
  double lenght_0 = 0.0, lenght_0_aux = 0.0;
  for (int i = 0; i < ndim; i++) {
    for (int j = 0; j < ndim; j++) lenght_0_aux += b_n[i * ndim + j] * dNa_n[j];
    lenght_0 += dNb_n[i] * lenght_0_aux;
    lenght_0_aux = 0.0;
  }
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++)
      Kd[i * ndim + j] = c0 * dNa_n1[i] * dNb_n1[j] + G * lenght_0 * (i == j) + c1 * dNa_n1[j] * dNb_n1[i];
}

/* Spectral stiffness density shared by compute_stiffness_density_Hencky__Constitutive__ (Hyperelastic/Hencky.c:98-229:
 * b = F_n1 F_n1^T, moduli AA = [lambda + 2G on the diagonal, lambda elsewhere]) and
 * compute_stiffness_elastoplastic__Constitutive__ (Plasticity/Elastoplastic-Tangent-Matrix.c:42-163: b = b_e,n+1,
 * moduli C_ep).  Eigenpairs of b and eigenvalues of the d x d Kirchhoff block by dsyev (ascending; paired by
 * index, :137-147); eigenvector A = column A.  The quotient (tau_B - tau_A)/(lambda_B - lambda_A) is skipped when
 * |lambda_B - lambda_A| <= 1e-14 exactly like the reference. */
static int stiffness_density_spectral(double *Kd, int ndim, const double *dN_alpha_n1, const double *dN_beta_n1,
                                      const double *b, const double *Cmod, const double *Stress) {
  double eigval_b[3] = {0, 0, 0}, eigvec_b[9] = {0}, eigval_T[3] = {0, 0, 0}, eigvec_T[9] = {0}, Tblk[9];
  for (int i = 0; i < ndim * ndim; i++) {
    Kd[i] = 0.0;
    Tblk[i] = Stress[i];
  }
  if (orc_sym_eigen(eigval_b, eigvec_b, b, ndim)) return 1;
  if (orc_sym_eigen(eigval_T, eigvec_T, Tblk, ndim)) return 1;
  double u__o__v[3][3];
  for (int i = 0; i < ndim; i++)
    for (int j = 0; j < ndim; j++) u__o__v[i][j] = dN_beta_n1[i] * dN_alpha_n1[j];
  for (int A = 0; A < ndim; A++) {
    double u_A = 0.0, v_A = 0.0;
    for (int i = 0; i < ndim; i++) {
      u_A += dN_alpha_n1[i] * eigvec_b[A + i * ndim];
      v_A += dN_beta_n1[i] * eigvec_b[A + i * ndim];
    }
    for (int B = 0; B < ndim; B++) {
      double u_B = 0.0, v_B = 0.0;
      for (int i = 0; i < ndim; i++) {
        u_B += dN_alpha_n1[i] * eigvec_b[B + i * ndim];
        v_B += dN_beta_n1[i] * eigvec_b[B + i * ndim];
      }
      double C_ep_AB = Cmod[A * ndim + B];
      double v_A__dot__u_B = u_B * v_A, u_A__dot__v_B = u_A * v_B, u_B__dot__v_B = u_B * v_B;
      for (int i = 0; i < ndim; i++)
        for (int j = 0; j < ndim; j++) {
          Kd[i * ndim + j] += C_ep_AB * u_A__dot__v_B * eigvec_b[A + i * ndim] * eigvec_b[B + j * ndim];
          if (A != B && fabs(eigval_b[B] - eigval_b[A]) > 1E-14)
            Kd[i * ndim + j] += 0.5 * ((eigval_T[B] - eigval_T[A]) / (eigval_b[B] - eigval_b[A])) *
                                (eigval_b[B] * u_B__dot__v_B * (eigvec_b[A + i * ndim] * eigvec_b[A + j * ndim]) +
                                 eigval_b[A] * v_A__dot__u_B * (eigvec_b[A + i * ndim] * eigvec_b[B + j * ndim]));
        }
    }
  }
  for (int i = 0; i < ndim; i++) /* geometric part */
    for (int j = 0; j < ndim; j++)
      for (int k = 0; k < ndim; k++) Kd[i * ndim + j] += -Stress[i * ndim + k] * u__o__v[k][j];
  return 0;
}

/* exported for tests/test_oracle.py: the reference's own numpy check of this density
 * (tests/Constitutive/Elastoplastic-Tangent-Matrix.py) is held as tests/golden/ref_etm2d.npz */
int orc_stiffness_density_spectral(double *Kd, int ndim, const double *dN_alpha_n1, const double *dN_beta_n1,
                                   const double *b, const double *Cmod, const double *Stress) {
  return stiffness_density_spectral(Kd, ndim, dN_alpha_n1, dN_beta_n1, b, Cmod, Stress);
}

/* __jacobian_evaluation, U-Newmark-beta.c:1646-1830, as a dense matrix K[ntot][ntot] (row-major, masked dof
 * numbering, ntot = nactive*ndim): particle loop with stiffness_density__Constitutive__ (Constitutive.c:262-283)
 * times V0 (:1768-1774), alpha_1 * lumped mass on the diagonal (:1797-1807) and, when dofs2mask is given,
 * MatZeroRowsColumnsIS on the Dirichlet dofs with 1.0 on their diagonal (:1822).  pattern[ntot] (optional) is
 * __create_sparsity_pattern, :1568-1632: the number of structurally visited columns of every row.
 * Laws: Neo-Hookean, Hencky, Drucker-Prager (stiffness_density__Constitutive__, Constitutive.c:262-381); the
 * Drucker-Prager branch reads the C_ep the constitutive update left in P->C_ep. */
int orc_tangent_matrix(double *K, int *pattern, double alpha_1, const double *lumped_mass, const orc_particles *P,
                       const orc_mesh *M, const orc_material *mats, const int *nodes2mask, const int *dofs2mask,
                       int nactive) {
  int ndim = M->ndim, T = P->T;
  size_t ntot = (size_t)nactive * ndim;
  int STATUS = 0;
  unsigned char *visited = pattern ? (unsigned char *)calloc(ntot * ntot, 1) : NULL;
  memset(K, 0, ntot * ntot * sizeof(double));
  for (int p = 0; p < P->np; p++) {
    const orc_material *mat = &mats[P->matidx[p]];
    if (mat->type == ORC_MAT_DRUCKER_PRAGER && (!P->C_ep || !P->b_e_n1)) {
      STATUS = 1;
      break;
    }
    double dN[ORC_MAXNB * 3], dN1[ORC_MAXNB * 3], d_phi_mT[9];
    int nn = orc_compute_dN(dN, P, M, p);
    if (nn < 0 || adjunt(d_phi_mT, &P->DF[p * T], ndim)) { /* push_forward_dN__MeshTools__, Shape-Functions.c:405-448 */
      STATUS = 1;
      continue;
    }
    for (int A = 0; A < nn; A++)
      for (int i = 0; i < ndim; i++) {
        double s = 0.0;
        for (int j = 0; j < ndim; j++) s += d_phi_mT[i * ndim + j] * dN[A * ndim + j];
        dN1[A * ndim + i] = s;
      }
    double V0_p = P->vol0[p];
    /* U-Newmark-beta.c:1757-1764: Stiffness_density_p *= (1 - Damage_n1[p]); the density enters linearly with V0 */
    if (g_tangent_damage) V0_p *= (1.0 - g_tangent_damage[p]);
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    for (int A = 0; A < nn; A++) {
      int Mask_node_A = nodes2mask[conn[A]];
      for (int B = 0; B < nn; B++) {
        int Mask_node_B = nodes2mask[conn[B]];
        double Kd[9];
        if (mat->type == ORC_MAT_NEO_HOOKEAN) {
          stiffness_density_neo_hookean(Kd, ndim, &dN1[A * ndim], &dN1[B * ndim], &dN[A * ndim], &dN[B * ndim],
                                        &P->F_n[p * T], P->J_n1[p], mat);
        } else if (mat->type == ORC_MAT_HENCKY) { /* Hencky.c:98-229 */
          double Lame = mat->E * mat->nu / ((1.0 + mat->nu) * (1.0 - 2.0 * mat->nu));
          double G = mat->E / (2.0 * (1.0 + mat->nu));
          double AA[9], b[9];
          for (int i = 0; i < ndim; i++)
            for (int j = 0; j < ndim; j++) AA[i * ndim + j] = Lame + (i == j ? 2 * G : 0.0);
          left_cauchy_green(b, &P->F_n1[p * T], ndim);
          STATUS |= stiffness_density_spectral(Kd, ndim, &dN1[A * ndim], &dN1[B * ndim], b, AA, &P->stress[p * T]);
        } else { /* Elastoplastic-Tangent-Matrix.c:42-163 */
          STATUS |= stiffness_density_spectral(Kd, ndim, &dN1[A * ndim], &dN1[B * ndim], &P->b_e_n1[p * T],
                                               &P->C_ep[p * ndim * ndim], &P->stress[p * T]);
        }
        for (int i = 0; i < ndim; i++)
          for (int j = 0; j < ndim; j++) {
            size_t at = ((size_t)Mask_node_A * ndim + i) * ntot + (size_t)Mask_node_B * ndim + j;
            K[at] += Kd[i * ndim + j] * V0_p;
            if (visited) visited[at] = 1;
          }
      }
    }
  }
  if (lumped_mass)
    for (size_t d = 0; d < ntot; d++) K[d * ntot + d] += alpha_1 * lumped_mass[d];
  if (dofs2mask)
    for (size_t d = 0; d < ntot; d++)
      if (dofs2mask[d] == -1) {
        for (size_t e = 0; e < ntot; e++) K[d * ntot + e] = K[e * ntot + d] = 0.0;
        K[d * ntot + d] = 1.0;
      }
  if (pattern) {
    for (size_t a = 0; a < ntot; a++) {
      int c = 0;
      for (size_t b = 0; b < ntot; b++) c += visited[a * ntot + b];
      pattern[a] = c;
    }
    free(visited);
  }
  return STATUS;
}

/* __update_particles_internal_variables, U-Newmark-beta.c:1917-1978 */
void orc_roll_state(orc_particles *P) {
  int T = P->T;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < P->np; p++) {
    P->J_n[p] = P->J_n1[p];
    P->rho[p] = P->mass[p] / (P->vol0[p] * P->J_n[p]);
    if (P->kappa_n) P->kappa_n[p] = P->kappa_n1[p];
    if (P->eps_n) P->eps_n[p] = P->eps_n1[p];
    for (int i = 0; i < T; i++) {
      if (P->b_e_n) P->b_e_n[p * T + i] = P->b_e_n1[p * T + i];
      P->F_n[p * T + i] = P->F_n1[p * T + i];
      if (P->dt_F_n) P->dt_F_n[p * T + i] = P->dt_F_n1[p * T + i];
    }
  }
}

/* __update_particles_kinetics_FLIP_PIC, U-Newmark-beta.c:1993-2072 */
int orc_update_kinetics(double alpha_blend, const double *dU, const double *Un_dt, const double *dU_dt,
                        const double *dU_dt2, orc_particles *P, const orc_mesh *M,
                        const int *nodes2mask) {
  int ndim = M->ndim;
  double beta_blend = 1 - alpha_blend;
#pragma omp parallel for schedule(static)
  for (int p = 0; p < P->np; p++) {
    double N[ORC_MAXNB];
    int nn = orc_compute_N(N, P, M, p);
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    for (int i = 0; i < ndim; i++) P->vel[p * ndim + i] = alpha_blend * P->vel[p * ndim + i];
    for (int A = 0; A < nn; A++) {
      int A_mask = nodes2mask[conn[A]];
      for (int i = 0; i < ndim; i++) {
        double DU_pI = N[A] * dU[A_mask * ndim + i];
        double D_V_pI = N[A] * dU_dt[A_mask * ndim + i];
        double V_n_pI = N[A] * Un_dt[A_mask * ndim + i];
        double D_A_pI = N[A] * dU_dt2[A_mask * ndim + i];
        P->acc[p * ndim + i] += D_A_pI;
        P->vel[p * ndim + i] += D_V_pI + beta_blend * V_n_pI;
        P->dis[p * ndim + i] += DU_pI;
        P->x[p * ndim + i] += DU_pI;
      }
    }
  }
  return 0;
}

/* ======================================================================================
 * Explicit predictor-corrector step.  The explicit drivers are stubs in the reference (SURVEY.md
 * fact 2); this composes the maintained stage functions in the order and with the formulas of
 * U-Verlet.c: predictor :229-253, nodal dU projection :301-367, Dirichlet :455-527, local state
 * :530-676 (density :630-632), nodal equilibrium :919-957 (a = g + F/M, reactions on fixed dofs),
 * G2P of acceleration and dU :962-1010, corrector :1024-1084; internal force in the Kirchhoff form of
 * U-Newmark-beta.c:1257-1374 with the sign of U-Verlet.c:783 (Forces -= f_int).
 * ====================================================================================== */
int orc_explicit_step(orc_particles *P, orc_mesh *M, const orc_material *mats, const orc_params *prm,
                      const orc_bcc *bcc, int nbcc, int step, int nsteps, double dt, double gamma,
                      const double *gravity, orc_step_out *out) {
  int ndim = M->ndim, T = P->T, np = P->np;
  int STATUS = 0;

  STATUS |= orc_local_search(P, M, prm);
  if (STATUS) return STATUS;
  int nactive = orc_active_nodes(out->nodes2mask, M);
  out->nactive = nactive;
  int nd = nactive * ndim;
  orc_active_dofs(out->dofs2mask, out->nodes2mask, nactive, ndim, bcc, nbcc, step, nsteps);
  memset(out->mass, 0, sizeof(double) * nd);
  memset(out->dU, 0, sizeof(double) * nd);
  memset(out->force, 0, sizeof(double) * nd);
  memset(out->accel, 0, sizeof(double) * nd);
  memset(out->reaction, 0, sizeof(double) * nd);

  /* __mass_NODES :160-223 == __compute_nodal_lumped_mass */
  orc_lumped_mass(out->mass, P, M, out->nodes2mask);

  /* __predictor_PARTICLES :229-253 */
  for (int p = 0; p < np; p++)
    for (int i = 0; i < ndim; i++) {
      int idx = p * ndim + i;
      P->d_dis[idx] = dt * P->vel[idx] + 0.5 * dsqr(dt) * P->acc[idx];
      P->vel[idx] += (1 - gamma) * dt * P->acc[idx];
    }

  /* __d_displacement_NODES :301-367 */
#pragma omp parallel for schedule(static)
  for (int p = 0; p < np; p++) {
    double N[ORC_MAXNB];
    int nn = orc_compute_N(N, P, M, p);
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    double m_p = P->mass[p];
    for (int A = 0; A < nn; A++) {
      int A_mask = out->nodes2mask[conn[A]];
#pragma omp critical
      {
        for (int i = 0; i < ndim; i++) out->dU[A_mask * ndim + i] += m_p * N[A] * P->d_dis[p * ndim + i];
      }
    }
  }
  /* U-Verlet.c:357-362 divides as written (0/0 at an active node no particle lists); the composition keeps the
   * convention of the maintained driver's VecPointwiseDivide instead (0 where the lumped mass is 0): such a node
   * is in no particle's list, so no particle result depends on the choice. */
  for (int i = 0; i < nd; i++) out->dU[i] = out->mass[i] != 0.0 ? out->dU[i] / out->mass[i] : 0.0;

  /* impose_Dirichlet_Boundary_Conditions :455-527 */
  for (int b = 0; b < nbcc; b++)
    for (int j = 0; j < bcc[b].nnodes; j++) {
      int m = out->nodes2mask[bcc[b].nodes[j]];
      if (m == -1) continue;
      for (int k = 0; k < bcc[b].dim; k++)
        if (bcc[b].dir[k * nsteps + step] == 1) out->dU[m * ndim + k] = bcc[b].value[k * nsteps + step];
    }

  /* __update_Local_State :530-676: kinematics (J <= 0 is fatal here, :608-613) ... */
  STATUS |= orc_compatibility(out->dU, NULL, P, M, out->nodes2mask);
  for (int p = 0; p < np; p++) {
    if (P->J_n1[p] <= 0.0) STATUS |= 1;
    double Delta_J_p = I3(&P->DF[p * T], ndim); /* :630-632 */
    P->rho[p] = P->rho[p] / Delta_J_p;
  }
  if (STATUS) return STATUS;
  /* ... and stress */
  STATUS |= orc_constitutive(P, mats, prm);
  if (STATUS) return STATUS;

  /* nodal forces: Forces -= f_int (U-Verlet.c:783) on every dof (fixed dofs keep it as reaction) */
  {
    int *alldofs = (int *)calloc((size_t)nd, sizeof(int)); /* all free for the accumulation */
    double *fint = (double *)calloc((size_t)nd, sizeof(double));
    STATUS |= orc_internal_forces(fint, P, M, out->nodes2mask, alldofs);
    for (int i = 0; i < nd; i++) out->force[i] = -fint[i];
    free(alldofs);
    free(fint);
  }

  /* solve_Nodal_Equilibrium :919-957 */
  for (int A = 0; A < nactive; A++)
    for (int i = 0; i < ndim; i++) {
      int idx = A * ndim + i;
      if (out->dofs2mask[idx] != -1) {
        out->accel[idx] = out->mass[idx] != 0.0 ? (gravity ? gravity[i] : 0.0) + out->force[idx] / out->mass[idx]
                                                : 0.0; /* massless active node: see the dU division above */
      } else {
        out->accel[idx] = 0.0;
        out->reaction[idx] = out->force[idx];
      }
    }

  /* G2P :962-1010 */
#pragma omp parallel for schedule(static)
  for (int p = 0; p < np; p++) {
    double N[ORC_MAXNB];
    int nn = orc_compute_N(N, P, M, p);
    const int *conn = &P->list[(size_t)p * ORC_MAXNB];
    for (int i = 0; i < ndim; i++) {
      P->acc[p * ndim + i] = 0.0;
      P->d_dis[p * ndim + i] = 0.0;
    }
    for (int A = 0; A < nn; A++) {
      int A_mask = out->nodes2mask[conn[A]];
      for (int i = 0; i < ndim; i++) {
        P->acc[p * ndim + i] += N[A] * out->accel[A_mask * ndim + i];
        P->d_dis[p * ndim + i] += N[A] * out->dU[A_mask * ndim + i];
      }
    }
  }

  /* compute_Explicit_Newmark_Corrector :1024-1084 */
  for (int p = 0; p < np; p++) {
    P->J_n[p] = P->J_n1[p];
    if (P->kappa_n) P->kappa_n[p] = P->kappa_n1[p];
    if (P->eps_n) P->eps_n[p] = P->eps_n1[p];
    if (P->b_e_n)
      for (int i = 0; i < T; i++) P->b_e_n[p * T + i] = P->b_e_n1[p * T + i];
    for (int i = 0; i < ndim; i++) {
      P->vel[p * ndim + i] += gamma * dt * P->acc[p * ndim + i];
      P->x[p * ndim + i] += P->d_dis[p * ndim + i];
      P->dis[p * ndim + i] += P->d_dis[p * ndim + i];
    }
    for (int i = 0; i < T; i++) P->F_n[p * T + i] = P->F_n1[p * T + i];
  }
  return STATUS;
}

/* ======================================================================================
 * uGIMP (BASELINE config 1, plumbing only — uGIMP is not on the GPU path and is unusable in the
 * reference: local_search exits, Shape-Functions.c:70-71, and dN__GIMP__ mixes node and dimension
 * indices, GIMP.c:314-320).  Sip / dSip / N follow Nodes/GIMP.c:235-295; the gradient is the
 * correct tensor product dS(x_j) * prod_{k != j} S(x_k).
 * ====================================================================================== */
double orc_sip_gimp(double L, double lp, double Delta_xp) { /* GIMP.c:235-253 */
  if ((-lp < Delta_xp) && (Delta_xp <= lp)) {
    return 1 - 0.5 * (dsqr(Delta_xp) + lp * lp) * (double)1 / (L * lp);
  } else if (((-L - lp) < Delta_xp) && (Delta_xp <= (-L + lp))) {
    return (double)(0.25 / (L * lp)) * dsqr(L + lp + Delta_xp);
  } else if (((L - lp) < Delta_xp) && (Delta_xp <= (L + lp))) {
    return (double)(0.25 / (L * lp)) * dsqr(L + lp - Delta_xp);
  } else if (((-L + lp) < Delta_xp) && (Delta_xp <= -lp)) {
    return 1 + (double)Delta_xp / L;
  } else if ((lp < Delta_xp) && (Delta_xp <= (L - lp))) {
    return 1 - (double)Delta_xp / L;
  }
  return (double)0.0;
}

double orc_dsip_gimp(double L, double lp, double Delta_xp) { /* GIMP.c:257-273 */
  if (((-L - lp) < Delta_xp) && (Delta_xp <= (-L + lp))) {
    return (double)(0.5 / (L * lp)) * (L + lp + Delta_xp);
  } else if (((-L + lp) < Delta_xp) && (Delta_xp <= -lp)) {
    return (double)1 / L;
  } else if ((-lp < Delta_xp) && (Delta_xp <= lp)) {
    return -(double)Delta_xp / (L * lp);
  } else if ((lp < Delta_xp) && (Delta_xp <= (L - lp))) {
    return -(double)1 / L;
  } else if (((L - lp) < Delta_xp) && (Delta_xp <= (L + lp))) {
    return -(double)(0.5 / (L * lp)) * (L + lp - Delta_xp);
  }
  return (double)0.0;
}

/* N__GIMP__, GIMP.c:277-295; Delta_Xp[nn][ndim] = x_p - x_I, lp[ndim] half particle size */
void orc_N_gimp(double *S, const double *Delta_Xp, int nn, int ndim, const double *lp, double L) {
  for (int i = 0; i < nn; i++) {
    S[i] = 1.0;
    for (int j = 0; j < ndim; j++) S[i] *= orc_sip_gimp(L, lp[j], Delta_Xp[i * ndim + j]);
  }
}

void orc_dN_gimp(double *dS, const double *Delta_Xp, int nn, int ndim, const double *lp, double L) {
  for (int i = 0; i < nn; i++)
    for (int j = 0; j < ndim; j++) {
      double v = orc_dsip_gimp(L, lp[j], Delta_Xp[i * ndim + j]);
      for (int k = 0; k < ndim; k++)
        if (k != j) v *= orc_sip_gimp(L, lp[k], Delta_Xp[i * ndim + k]);
      dS[i * ndim + j] = v;
    }
}
