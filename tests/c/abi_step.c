/* abi_step.c -- a plain C99 caller of the C-ABI (include/nlps_gpu.h), linked against libnlps_gpu.so like the reference's
 * driver would be (no Python, no ctypes): create -> initialize__LME__ -> four fused explicit steps -> download, compared
 * with the committed end state of the "nh3d" golden case (tests/golden/nh3d_abi.bin, written by
 * tests/golden/make_abi_fixture.py from the oracle's vectors).
 *   gcc -std=c99 -Iinclude tests/c/abi_step.c -o abi_step -Lnl-partsol_amd/csrc -lnlps_gpu -Wl,-rpath,$PWD/nl-partsol_amd/csrc -lm
 *   ./abi_step tests/golden/nh3d_abi.bin          (exit code 0 = parity, needs a GPU)                                  */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nlps_gpu.h"

static void *xread(FILE *f, size_t bytes) {
  void *p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "abi_step: short read\n");
    exit(2);
  }
  return p;
}

static double relerr(const double *a, const double *b, size_t n) {
  double s = 1e-300, e = 0.0;
  for (size_t i = 0; i < n; i++) {
    if (fabs(b[i]) > s) s = fabs(b[i]);
    if (fabs(a[i] - b[i]) > e) e = fabs(a[i] - b[i]);
  }
  return e / s;
}

#define CHECK(call)                                                            \
  do {                                                                         \
    if ((call) != 0) {                                                         \
      fprintf(stderr, "abi_step: %s failed: %s\n", #call, nlps_gpu_last_error(h)); \
      return 1;                                                                \
    }                                                                          \
  } while (0)

int main(int argc, char **argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: abi_step nh3d_abi.bin\n");
    return 2;
  }
  FILE *f = fopen(argv[1], "rb");
  if (!f) {
    perror(argv[1]);
    return 2;
  }
  int *hd = (int *)xread(f, 8 * sizeof(int));
  if (hd[0] != 0x4E4C5053 || hd[1] != 3) {
    fprintf(stderr, "abi_step: not a 3-D fixture\n");
    return 2;
  }
  const int d = 3, T = 9, np = hd[5], nsteps = hd[6], nbc = hd[7];
  double *sc = (double *)xread(f, 7 * sizeof(double));
  double *x = (double *)xread(f, sizeof(double) * np * d), *vel = (double *)xread(f, sizeof(double) * np * d);
  double *mass = (double *)xread(f, sizeof(double) * np), *vol0 = (double *)xread(f, sizeof(double) * np);
  double *rho = (double *)xread(f, sizeof(double) * np);
  int *bcn = (int *)xread(f, sizeof(int) * nbc);
  int *I0_ref = (int *)xread(f, sizeof(int) * np);
  double *x_ref = (double *)xread(f, sizeof(double) * np * d), *v_ref = (double *)xread(f, sizeof(double) * np * d);
  double *F_ref = (double *)xread(f, sizeof(double) * np * T), *s_ref = (double *)xread(f, sizeof(double) * np * T);
  fclose(f);

  /* the arrays the reference's Particle struct owns (Types.h:548-623), initialised like U-Analisys.c:33-43 */
  double *dis = calloc((size_t)np * d, sizeof(double)), *acc = calloc((size_t)np * d, sizeof(double));
  double *F_n = calloc((size_t)np * T, sizeof(double)), *J_n = malloc(sizeof(double) * np);
  double *stress = calloc((size_t)np * T, sizeof(double));
  int *matidx = calloc((size_t)np, sizeof(int)), *I0 = calloc((size_t)np, sizeof(int));
  for (int p = 0; p < np; p++) {
    F_n[p * T + 0] = F_n[p * T + 4] = F_n[p * T + 8] = 1.0;
    J_n[p] = 1.0;
  }
  nlps_grid grid;
  memset(&grid, 0, sizeof grid);
  grid.ndim = d;
  for (int a = 0; a < 3; a++) grid.n[a] = hd[2 + a];
  grid.h = sc[0];
  nlps_params prm = {3.0, 1e-6, 1e-10, 10, 1e-14, 10, 0, 0}; /* Read_GramsShapeFun.c:100-104; no damage driver */
  nlps_material mat;
  memset(&mat, 0, sizeof mat);
  mat.type = NLPS_MAT_NEO_HOOKEAN;
  mat.E = sc[2];
  mat.nu = sc[3];
  nlps_particles P;
  memset(&P, 0, sizeof P);
  P.np = np;
  P.x_GC = x;
  P.dis = dis;
  P.vel = vel;
  P.acc = acc;
  P.F_n = F_n;
  P.J_n = J_n;
  P.rho = rho;
  P.mass = mass;
  P.Vol_0 = vol0;
  P.MatIdx = matidx;
  /* Dirichlet plane, all directions on at every step: Dir[k*NumTimeStep+t], Value[k].Fx[t] (Types.h:296-351) */
  int *dir = malloc(sizeof(int) * d * nsteps);
  double *val = calloc((size_t)d * nsteps, sizeof(double));
  for (int i = 0; i < d * nsteps; i++) dir[i] = 1;
  nlps_bcc bc = {nbc, bcn, d, dir, val};

  nlps_gpu *h = NULL;
  if (nlps_gpu_create(&h, &grid, &prm, &mat, 1, &P, nsteps, NULL) != 0) {
    fprintf(stderr, "abi_step: nlps_gpu_create failed: %s\n", h ? nlps_gpu_last_error(h) : "no handle");
    return 1;
  }
  CHECK(nlps_gpu_initialize_lme(h));
  for (int t = 0; t < nsteps; t++) CHECK(nlps_gpu_explicit_step(h, &bc, 1, t, sc[1], 0.5, &sc[4]));
  int flags = -1;
  CHECK(nlps_gpu_status_flags(h, &flags));
  nlps_particles out;
  memset(&out, 0, sizeof out);
  out.np = np;
  out.x_GC = x;
  out.vel = vel;
  out.F_n = F_n;
  out.Stress = stress;
  out.I0 = I0;
  CHECK(nlps_gpu_download_state(h, &out));
  CHECK(nlps_gpu_destroy(h));

  int bad = flags != 0;
  for (int p = 0; p < np; p++) bad |= I0[p] != I0_ref[p];
  const double ex = relerr(x, x_ref, (size_t)np * d), ev = relerr(vel, v_ref, (size_t)np * d);
  const double eF = relerr(F_n, F_ref, (size_t)np * T), es = relerr(stress, s_ref, (size_t)np * T);
  printf("abi_step: %d particles, %d steps, flags %d, closest nodes %s, rel. err x %.2e vel %.2e F %.2e stress %.2e\n", np,
         nsteps, flags, bad ? "DIFFER" : "identical", ex, ev, eF, es);
  if (bad || ex > 1e-9 || ev > 1e-9 || eF > 1e-9 || es > 1e-9) {
    printf("abi_step: FAIL\n");
    return 1;
  }
  printf("abi_step: PASS\n");
  return 0;
}
