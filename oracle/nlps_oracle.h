/*
 * nlps_oracle.h — CPU ORACLE for the NL-PartSol particle<->grid + stress-update hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / CPU baseline.  The product path (nl-partsol_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no golden vectors or asserting tests
 * for this path, and its sources on the path include <lapacke.h> and link LAPACK
 * (nl-partsol/src/Matlib/MatrixOp.c:13, Matlib/TensorLib.c:13, Particles/compute-Strains.c:13,
 * Constitutive/Plasticity/Drucker-Prager.h:25), neither of which exists in the build image, so the
 * reference cannot be compiled here without writing stand-ins.  This file is therefore a careful
 * plain-C restatement of the reference's algorithm, function by function, each citing the reference
 * file:line it follows; the third-party arithmetic it replaces (LAPACK dsyev / dgetrf+dgetri, the
 * pinned dependency is "whatever LAPACK the host links", not vendored) is restated in closed form /
 * cyclic Jacobi and is cross-checked in tests against scipy's LAPACK (same routines).
 *
 * All paths are relative to /root/reference/nl-partsol/src unless noted.
 */
#ifndef NLPS_ORACLE_H
#define NLPS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXNB 128 /* >= 5^3 two-ring candidates */

enum { ORC_MAT_NEO_HOOKEAN = 0, ORC_MAT_HENCKY = 1, ORC_MAT_DRUCKER_PRAGER = 2, ORC_MAT_VON_MISES = 3,
       ORC_MAT_MATSUOKA_NAKAI = 4, ORC_MAT_LADE_DUNCAN = 5 };

/* Background mesh: Types.h:631-760 (only the members the path reads). */
typedef struct {
  int ndim;      /* NumberDimensions, Macros.h:33-37 */
  int n[3];      /* nodes per axis of the structured Q4/H8 grid (x fastest) */
  double origin[3];
  double h;
  int nnodes;
  double *coords;        /* Coordinates.nM  [nnodes][ndim] */
  int *r1_ptr, *r1;      /* NodalLocality_0 (1-ring incl. self), CSR, chain order */
  int *r2_ptr, *r2;      /* NodalLocality   (2-ring incl. self), CSR, chain order */
  double *h_avg;         /* Read_GramsBox.c:460-507 */
  unsigned char *active; /* ActiveNode[] */
} orc_mesh;

/* Particle set: Types.h:548-623 / Fields Types.h:184-283 (AoS rows, row-major, as in the reference). */
typedef struct {
  int np, ndim, T; /* T = 5 (2-D: xx,xy,yx,yy,zz) | 9 */
  double *x, *dis, *vel, *acc, *d_dis;                  /* [np][ndim] */
  double *F_n, *F_n1, *DF, *stress, *b_e_n, *b_e_n1;   /* [np][T] */
  double *dt_F_n, *dt_F_n1, *dt_DF;                     /* [np][T] */
  double *J_n, *J_n1, *rho, *mass, *vol0, *W;           /* [np] */
  double *kappa_n, *kappa_n1, *eps_n, *eps_n1;          /* [np] */
  int *matidx;                                          /* [np] */
  int *I0;                                              /* [np] */
  double *lambda;                                       /* [np][ndim] */
  double *beta;                                         /* [np] */
  int *nn;                                              /* NumberNodes[np] */
  int *list;                                            /* ListNodes as array [np][ORC_MAXNB], chain order */
  int *status;                                          /* per-particle failure flags (build's addition) */
  double *C_ep;                                         /* [np][ndim*ndim] elastoplastic tangent moduli */
  double *back_stress;                                  /* [np][3] principal back stress (Von-Mises), in/out */
} orc_particles;

/* Material: Types.h:359-458 (members the three laws read). */
typedef struct {
  int type;
  double E, nu;
  double phi_deg, psi_deg;      /* phi_Frictional, psi_Frictional */
  double kappa_0;               /* kappa_0 */
  double exponent_ortiz;        /* Exponent_Hardening_Ortiz */
  double eps_0;                 /* Plastic_Strain_0 */
  double p_ref;                 /* ReferencePressure */
  /* Von-Mises (Plasticity/Von-Mises.c:246-253): sigma_y = kappa_0 */
  double hardening_modulus;     /* Hardening_modulus */
  double theta_voce, K0_voce, Kinf_voce, delta_voce; /* *_Hardening_Voce */
  double Ceps, Gf;              /* eigenerosion: normalising constant and critical energy release rate */
  /* Matsuoka-Nakai / Lade-Duncan (Plasticity/Matsuoka-Nakai.c:330-341): with phi_deg, kappa_0 (initial Kappa) and
   * eps_0 (initial EPS of Matsuoka-Nakai, Generate-One-Phase-Analysis.c:624-626) */
  double cohesion;              /* Cohesion */
  double alpha_borja;           /* alpha_Hardening_Borja */
  double a_borja[3];            /* a_Hardening_Borja */
  double ft, heps, wcrit;       /* eigensoftening (Types.h:386-390): tensile strength, band width, critical opening */
} orc_material;

/* Globals snapshot: Globals.h:21,33-58; defaults InOutFun/Read_GramsShapeFun.c:100-104 */
typedef struct {
  double gamma_lme, tol_zero_lme, tol_wrapper_lme;
  int max_iter_lme;
  double tol_radial_returning;
  int max_iter_radial_returning;
} orc_params;

/* Dirichlet boundary (Load, Types.h:296-334): Dir[k*nsteps+t], Value[k].Fx[t] flattened [dim][nsteps] */
typedef struct {
  int nnodes;
  const int *nodes;
  int dim;
  const int *dir;
  const double *value;
} orc_bcc;

/* ---- mesh ---- */
orc_mesh *orc_mesh_build(int ndim, const int n[3], const double origin[3], double h);
void orc_mesh_free(orc_mesh *m);

/* ---- LME (Nodes/LME.c) ---- */
void orc_p_lme(double *p, const double *l, int na, int ndim, const double *lambda, double beta);
int orc_dp_lme(double *dp, const double *l, const double *p, int na, int ndim);
int orc_lambda_newton(const double *l, int na, int ndim, double *lambda, double beta,
                      const orc_params *prm, int *iters);
double orc_rcond_ref(const double *A, int n);
int orc_inverse(double *Am1, const double *A, int n);
int orc_sym_eigen(double *eigval, double *eigvec, const double *A, int n);

int orc_initialize_lme(orc_particles *P, orc_mesh *M, const orc_params *prm);
int orc_local_search(orc_particles *P, orc_mesh *M, const orc_params *prm);
int orc_search_phase1(orc_particles *P, orc_mesh *M);
int orc_search_phase2(orc_particles *P, orc_mesh *M, const orc_params *prm);
int orc_compute_N(double *N, const orc_particles *P, const orc_mesh *M, int p);
int orc_compute_dN(double *dN, const orc_particles *P, const orc_mesh *M, int p);

/* ---- masks (Nodes/Nodes-Tools.c:46-156) ---- */
int orc_active_nodes(int *nodes2mask, const orc_mesh *M);
int orc_active_dofs(int *dofs2mask, const int *nodes2mask, int nactive, int ndof,
                    const orc_bcc *bcc, int nbcc, int step, int nsteps);

/* ---- stage functions (Formulations/Displacements/U-Newmark-beta.c statics) ---- */
int orc_lumped_mass(double *Mv, const orc_particles *P, const orc_mesh *M, const int *nodes2mask);
int orc_nodal_field_n(double *V, double *A, const double *Mv, const orc_particles *P,
                      const orc_mesh *M, const int *nodes2mask, const int *dofs2mask, int nactive);
int orc_compatibility(const double *dU, const double *dU_dt, orc_particles *P, const orc_mesh *M,
                      const int *nodes2mask);
int orc_constitutive(orc_particles *P, const orc_material *mats, const orc_params *prm);
int orc_stress_one(int ndim, const orc_material *mat, const orc_params *prm, const double *F_n1,
                   const double *DF, double J, const double *b_e_n, double kappa_n, double eps_n,
                   double *stress, double *W, double *b_e_n1, double *kappa_n1, double *eps_n1, double *C_ep,
                   double *back_stress);
int orc_internal_forces(double *R, const orc_particles *P, const orc_mesh *M, const int *nodes2mask,
                        const int *dofs2mask);
/* __compute_trial_b_e, Drucker-Prager.c:617-633 */
void orc_trial_b_e(double *btr, const double *d_phi, const double *b_e_n, int ndim);
/* __jacobian_evaluation + __create_sparsity_pattern (U-Newmark-beta.c:1568-1830), Neo-Hookean, dense */
int orc_nodal_traction_forces(double *R, const orc_particles *P, const orc_mesh *M, const int *nodes2mask,
                              const int *dofs2mask, int nloads, const int *load_n, const int *ids, const int *dir,
                              const double *val, double thickness, const double *area0);
int orc_stiffness_density_spectral(double *Kd, int ndim, const double *dN_alpha_n1, const double *dN_beta_n1,
                                   const double *b, const double *Cmod, const double *Stress);
int orc_tangent_matrix(double *K, int *pattern, double alpha_1, const double *lumped_mass, const orc_particles *P,
                       const orc_mesh *M, const orc_material *mats, const int *nodes2mask, const int *dofs2mask,
                       int nactive);
void orc_roll_state(orc_particles *P);

/* ---- eigenerosion: Constitutive/Fracture/Beps.c:16-80, EigenErosion.c:29-117 and the hooks of
 * U-Newmark-beta.c:1218-1224 (__constitutive_update skips failed particles), :1313-1331 (damage + stress scaling inside
 * __nodal_internal_forces), :1757-1764 (stiffness density scaled), :1950-1953 (roll of the damage field).
 * beps_n[np], beps[np][stride]: the epsilon-neighbourhoods as arrays in chain order (push prepends). */
int orc_compute_beps(int *beps_n, int *beps, int stride, const orc_particles *P, const orc_mesh *M,
                     const orc_material *mats, int initialize);
int orc_constitutive_eroded(orc_particles *P, const orc_material *mats, const orc_params *prm, const double *damage_n);
int orc_eigenerosion_hook(double *damage_n1, const double *damage_n, orc_particles *P, const orc_material *mats,
                          const int *beps_n, const int *beps, int stride, double DeltaX);
/* Eigensoftening (Constitutive/Fracture/EigenSoftening.c:27-163 behind compute_damage__Constitutive__,
 * Constitutive.c:412-432): the same hook of __nodal_internal_forces with Driver_EigenSoftening.  strain_f_n1[np] is
 * Phi.Strain_f_n1 (read and written: the reference passes it as both StrainF_n and StrainF_n1). */
int orc_eigensoftening_hook(double *damage_n1, const double *damage_n, double *strain_f_n1, orc_particles *P,
                            const orc_material *mats, const int *beps_n, const int *beps, int stride);
void orc_set_tangent_damage(const double *damage_n1); /* NULL = off: orc_tangent_matrix scales by (1 - damage) */
int orc_update_kinetics(double alpha_blend, const double *dU, const double *Un_dt,
                        const double *dU_dt, const double *dU_dt2, orc_particles *P,
                        const orc_mesh *M, const int *nodes2mask);

/* ---- explicit predictor-corrector composition (U-Verlet.c:229-253,301-367,530-676,947-957,1024-1084) ---- */
typedef struct {
  int nactive;
  int *nodes2mask;   /* [nnodes] */
  int *dofs2mask;    /* [nactive*ndim] */
  double *mass;      /* [nactive*ndim] */
  double *dU;        /* [nactive*ndim] */
  double *force;     /* [nactive*ndim] */
  double *accel;     /* [nactive*ndim] */
  double *reaction;  /* [nactive*ndim] */
} orc_step_out;

int orc_explicit_step(orc_particles *P, orc_mesh *M, const orc_material *mats, const orc_params *prm,
                      const orc_bcc *bcc, int nbcc, int step, int nsteps, double dt, double gamma,
                      const double *gravity, orc_step_out *out);

/* ---- uGIMP (config-1 plumbing, CPU only; Nodes/GIMP.c:235-295) ---- */
double orc_sip_gimp(double L, double lp, double Delta_xp);
double orc_dsip_gimp(double L, double lp, double Delta_xp);
void orc_N_gimp(double *S, const double *Delta_Xp, int nn, int ndim, const double *lp, double L);
void orc_dN_gimp(double *dS, const double *Delta_Xp, int nn, int ndim, const double *lp, double L);

int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
