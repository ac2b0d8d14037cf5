/*
 * nlps_gpu.h — C-ABI of the MI355X (gfx950) implementation of NL-PartSol's particle<->grid transfer
 * and per-particle stress-update hot path.
 *
 * Plain C, plain pointers and sizes.  Every entry point names the reference interface it replaces
 * (file:line relative to nl-partsol/src of migmolper/NL-PartSol @ v1).  The reference passes its fat
 * `Particle` / `Mesh` structs by value (Types.h:548-760); here the same data crosses the boundary as
 * the contiguous row-major arrays those structs already own (`Matrix.nV`, MatrixOp.c:128-181), so the
 * glue a maintainer adds in U-Newmark-beta.c / U-Static.c is pointer plumbing only.
 *
 * Conventions
 *  - return value: 0 = EXIT_SUCCESS, 1 = EXIT_FAILURE (reference convention, U-Newmark-beta.c:199-204);
 *    the library never exit()s; nlps_gpu_last_error() returns the message the reference would have
 *    printed in red on stderr.  A missing GPU / failed HIP call is a failure, never a CPU fallback.
 *  - arithmetic is FP64; index maps are 32-bit int and bit-identical to the reference's.
 *  - particle state is device resident between calls (SoA, sorted by background-grid cell); host
 *    arrays are stale until nlps_gpu_download_state().
 *  - nodal vectors ("Vec" arrays of the PETSc driver) use the reference's masked numbering
 *    [Nactivenodes*Ndim], index = Nodes2Mask[node]*Ndim + i.  Each such pointer may be a HOST pointer
 *    (VecGetArray) or a DEVICE pointer (hipMalloc / torch tensor); the library detects which.
 *  - threading: one host thread per handle (the PETSc thread); all work is issued on one HIP stream.
 */
#ifndef NLPS_GPU_H
#define NLPS_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these are its exports */
#endif

typedef struct nlps_gpu nlps_gpu; /* opaque: device buffers, stream, tables */

#define NLPS_MAXNB 128 /* row stride of downloaded neighbour lists (>= 5^3) */

/* Material.Type strings of Constitutive/Constitutive.c:28-258 that are on the path */
enum {
  NLPS_MAT_NEO_HOOKEAN = 0,   /* "Neo-Hookean-Wriggers", Hyperelastic/Neo-Hookean.c:38-85 */
  NLPS_MAT_HENCKY = 1,        /* "Hencky",               Hyperelastic/Hencky.c:40-94       */
  NLPS_MAT_DRUCKER_PRAGER = 2, /* "Drucker-Prager",      Plasticity/Drucker-Prager.c:319-613 */
  NLPS_MAT_VON_MISES = 3,      /* "Von-Mises",           Plasticity/Von-Mises.c:212-392 */
  NLPS_MAT_MATSUOKA_NAKAI = 4, /* "Matsuoka-Nakai",      Plasticity/Matsuoka-Nakai.c:300-700 */
  NLPS_MAT_LADE_DUNCAN = 5     /* "Lade-Duncan",         Plasticity/Lade-Duncan.c:290-692 */
};

/* Structured background grid (GramsBox mesh, InOutFun/Read_GramsBox.c:54): Q4 / H8 lattice, nodes
 * numbered x-fastest, elements x-fastest with GiD connectivity order.  h_avg = FEM_Mesh.h_avg
 * (Read_GramsBox.c:460-507) or NULL to have the library compute it with the same rule. */
typedef struct {
  int ndim;          /* NumberDimensions, Macros.h:33-37 */
  int n[3];          /* nodes per axis */
  double origin[3];
  double h;          /* lattice spacing (FEM_Mesh.DeltaX) */
  const double *h_avg;
} nlps_grid;

/* Snapshot of the globals the path reads implicitly (Globals.h:21,33-58);
 * defaults InOutFun/Read_GramsShapeFun.c:100-104. */
typedef struct {
  double gamma_lme;        /* gamma_LME        (3.0)   */
  double tol_zero_lme;     /* TOL_zero_LME     (1e-6)  */
  double tol_wrapper_lme;  /* TOL_wrapper_LME  (1e-10) */
  int max_iter_lme;        /* max_iter_LME     (10)    */
  double tol_radial_returning;   /* TOL_Radial_Returning            */
  int max_iter_radial_returning; /* Max_Iterations_Radial_Returning */
  int driver_eigenerosion;       /* Driver_EigenErosion (Globals.h): the damage hooks of the level-B stages (0 = off) */
  int driver_eigensoftening;     /* Driver_EigenSoftening: the same hooks with Eigensoftening__Constitutive__ (0 = off;
                                    both on = eigenerosion, like the if / else if of Constitutive.c:395-412) */
} nlps_params;

/* Material, Types.h:359-458 (members read by the three laws) */
typedef struct {
  int type;
  double E, nu;
  double phi_deg, psi_deg; /* phi_Frictional, psi_Frictional */
  double kappa_0, exponent_ortiz, eps_0 /* Plastic_Strain_0 */, p_ref /* ReferencePressure */;
  /* Von-Mises (Von-Mises.c:246-253): sigma_y = kappa_0, Hardening_modulus, theta / K_0 / K_inf / delta _Hardening_Voce */
  double hardening_modulus, theta_voce, K0_voce, Kinf_voce, delta_voce;
  double Ceps, Gf; /* eigenerosion (Constitutive/Fracture/EigenErosion.c:63-64): normalising constant, Griffith energy */
  /* Matsuoka-Nakai / Lade-Duncan (Matsuoka-Nakai.c:330-341): Cohesion, alpha_Hardening_Borja, a_Hardening_Borja[3];
   * they also read phi_deg; Kappa_n starts at kappa_0 and EPS_n of Matsuoka-Nakai at eps_0 (caller's arrays,
   * Generate-One-Phase-Analysis.c:620-626).  Their readers set TOL_Radial_Returning / Max_Iterations_Radial_Returning
   * to 1e-10 / 20 (Matsuoka-Nakai) and 1e-14 / 10 (Lade-Duncan): nlps_params carries them */
  double cohesion, alpha_borja, a_borja[3];
  double ft, heps, wcrit; /* eigensoftening (Types.h:386-390, EigenSoftening.c:67-69): tensile strength, band width of the
                             cohesive fracture, critical opening displacement */
} nlps_material;

/* Particle fields, Types.h:184-283 / 548-623: HOST pointers to the reference's row-major arrays
 * (MPM_Mesh.Phi.<field>.nV).  T = 5 in 2-D (xx,xy,yx,yy,zz), 9 in 3-D.  Optional members may be NULL. */
typedef struct {
  int np;               /* NumGP */
  double *x_GC;         /* [np][ndim] */
  double *dis;          /* [np][ndim] */
  double *vel;          /* [np][ndim] */
  double *acc;          /* [np][ndim] */
  double *F_n;          /* [np][T] */
  double *F_n1;         /* [np][T]  optional on upload (defaults to F_n) */
  double *DF;           /* [np][T]  optional on upload (identity) */
  double *Stress;       /* [np][T]  optional on upload (0): Kirchhoff stress */
  double *b_e_n;        /* [np][T]  optional (identity) */
  double *b_e_n1;       /* [np][T]  optional */
  double *J_n;          /* [np] */
  double *J_n1;         /* [np]     optional (J_n) */
  double *rho;          /* [np] */
  double *mass;         /* [np] */
  double *Vol_0;        /* [np] */
  double *W;            /* [np]     optional */
  double *Kappa_n, *Kappa_n1, *EPS_n, *EPS_n1; /* [np] optional (0) */
  int *MatIdx;          /* [np] */
  int *I0;              /* [np]      NULL on upload => nlps_gpu_initialize_lme() must be called */
  double *lambda;       /* [np][ndim] (MPM_Mesh.lambda.nV) optional */
  double *Beta;         /* [np]       (MPM_Mesh.Beta.nV)   optional */
  double *dt_F_n;       /* [np][T]  optional (0): rate of F, consumed only by the fluid law upstream */
  double *dt_F_n1;      /* [np][T]  optional */
  double *dt_DF;        /* [np][T]  optional */
  double *C_ep;         /* [np][ndim*ndim] download only: elastoplastic tangent moduli (Drucker-Prager, Von-Mises) */
  double *Back_stress;  /* [np][3] optional (0): principal back stress of Von-Mises (Phi.Back_stress), in/out */
  double *Damage_n;     /* [np] optional (0): Phi.Damage_n  (eigenerosion, driver_eigenerosion != 0) */
  double *Damage_n1;    /* [np] optional (Damage_n): Phi.Damage_n1 */
  double *Strain_f_n;   /* [np] optional (0): Phi.Strain_f_n  (eigensoftening: strain at the onset of fracture) */
  double *Strain_f_n1;  /* [np] optional (Strain_f_n): Phi.Strain_f_n1 -- the one Eigensoftening reads AND writes */
} nlps_particles;

/* Dirichlet boundary = Load of FEM_Mesh.Bounds (Types.h:296-351), flattened:
 * dir[k*nsteps + t] = Dir[k*NumTimeStep+t], value[k*nsteps + t] = Value[k].Fx[t]. */
typedef struct {
  int nnodes;
  const int *nodes;
  int dim;
  const int *dir;
  const double *value;
} nlps_bcc;

/* ------------------------------------------------------------------ lifetime */

/* Uploads the particle set (AoS -> cell-sorted SoA) and builds the grid tables.
 * hip_stream: a hipStream_t to issue all work on (NULL = the library creates its own). */
int nlps_gpu_create(nlps_gpu **h, const nlps_grid *grid, const nlps_params *prm,
                    const nlps_material *mats, int nmats, const nlps_particles *host, int nsteps,
                    void *hip_stream);
int nlps_gpu_destroy(nlps_gpu *h);
const char *nlps_gpu_last_error(const nlps_gpu *h);
int nlps_gpu_synchronize(nlps_gpu *h);

/* Copies the device state back into the caller's arrays in the caller's particle order (before VTK/CSV
 * output, Outputs/WriteVtk.c:95-266, or host tangent assembly).  NULL members are skipped. */
int nlps_gpu_download_state(nlps_gpu *h, nlps_particles *host);
/* MPM_Mesh.NumberNodes[p] and ListNodes[p] as arrays (chain order = nodal_set__Particles__ order,
 * Particles/Particles-Tools.c:71-82): nn[np], list[np][NLPS_MAXNB]. */
int nlps_gpu_download_lists(nlps_gpu *h, int *nn, int *list);
/* The shape functions and their gradients of particles first .. first + count - 1 (caller's order) in the order of
 * ListNodes[p]: what compute_N__ShapeFun__ / compute_dN__ShapeFun__ return for the LME family (Shape-Functions.c:145-262;
 * p__LME__ LME.c:716-762, dp__LME__ LME.c:836-891, with the particle's current lambda and beta).  N[count][NLPS_MAXNB],
 * dN[count][NLPS_MAXNB][ndim], entries behind NumberNodes[p] are 0; either may be NULL.  A level-A entry for callers that
 * interpolate fields of their own (outputs, contact, diagnostics); the step kernels never store these values. */
int nlps_gpu_shape_functions(nlps_gpu *h, int first, int count, double *N, double *dN);
/* FEM_Mesh.ActiveNode[nnodes] as bytes */
int nlps_gpu_download_active(nlps_gpu *h, unsigned char *active);
/* OR of the per-particle failure flags (1 Newton, 2 connectivity, 4 J<=0, 8 constitutive, 16 halo) */
int nlps_gpu_status_flags(nlps_gpu *h, int *flags);

/* ------------------------------------------------------------------ level-B stage calls */

/* initialise_shapefun__MeshTools__ -> initialize__LME__, Nodes/LME.c:45-173 */
int nlps_gpu_initialize_lme(nlps_gpu *h);

/* local_search__MeshTools__ (Nodes/Shape-Functions.c:31-90) -> local_search__LME__ (Nodes/LME.c:895-1015):
 * I0 update, 1-ring activation, neighbour lists, beta, lambda Newton. */
int nlps_gpu_local_search(nlps_gpu *h);

/* get_active_nodes__MeshTools__ + get_active_dofs__MeshTools__ (Nodes/Nodes-Tools.c:46-156).
 * nodes2mask[nnodes] / dofs2mask[nactive*ndim] may be NULL (kept on the device only). */
/* Node order of the masked numbering.  get_active_nodes__MeshTools__ hands out its running index in the order of the
 * mesh file's nodes (Nodes/Nodes-Tools.c:46-66); the library works in lattice numbering (x fastest), which is the same
 * thing only for a file numbered that way.  lattice_of_file[A] = lattice node of file node A (the map
 * nlps_host_lattice_from_nodes returns; nnodes entries, copied): from now on Nodes2Mask -- and with it every masked
 * vector, the dof mask and the COO rows / columns of the tangent -- counts the active nodes in FILE order, and the
 * nodes2mask array of nlps_gpu_active_masks is indexed by file node.  Everything else of this interface that names a
 * node (I0, lists, Dirichlet node sets, h_avg, active flags) stays in lattice numbering.  NULL = lattice order. */
int nlps_gpu_set_node_numbering(nlps_gpu *h, const int *lattice_of_file);
int nlps_gpu_active_masks(nlps_gpu *h, const nlps_bcc *bcc, int nbcc, int step, int *nactive,
                          int *nfree_dofs, int *nodes2mask, int *dofs2mask);

/* __compute_nodal_lumped_mass, U-Newmark-beta.c:528-597.  M[nactive*ndim] is overwritten. */
int nlps_gpu_lumped_mass(nlps_gpu *h, double *M);

/* __get_nodal_field_n, U-Newmark-beta.c:615-696.  V, A overwritten; needs M from the call above. */
int nlps_gpu_nodal_field_n(nlps_gpu *h, double *V, double *A, const double *M);

/* __local_compatibility_conditions, U-Newmark-beta.c:1064-1160: DF, F_n1, J_n1 and, when dU_dt is given,
 * the rate tensors dt_DF, dt_F_n1 (compute-Strains.c:48-72,176-207; they feed only the Newtonian-fluid law,
 * Constitutive.c:84-108, so dU_dt may be NULL). */
int nlps_gpu_compatibility(nlps_gpu *h, const double *dU, const double *dU_dt);

/* __constitutive_update -> Stress_integration__Constitutive__, U-Newmark-beta.c:1208-1242,
 * Constitutive/Constitutive.c:18-258 */
int nlps_gpu_constitutive(nlps_gpu *h);

/* __nodal_internal_forces, U-Newmark-beta.c:1257-1374 (+ push_forward_dN__MeshTools__,
 * Shape-Functions.c:405-448).  R[nactive*ndim] is ACCUMULATED into; Dirichlet dofs are skipped. */
int nlps_gpu_internal_forces(nlps_gpu *h, double *R);
/* __nodal_traction_forces (U-Newmark-beta.c:1376-1500): R_A -= N_pA T A0_p over the particles of the Neumann contours
 * (the reference runs it serially on the host between the internal and the inertial forces).  loads[l]
 * is a Load like the Dirichlet ones with nodes = PARTICLE indices in the caller's order; the traction of the current
 * step is value where dir is 1 and, as upstream, the previous contour's value where it is 0.  A0_p is Vol_0 / thickness
 * in 2-D (Thickness_Plain_Stress) and area0[p] (Phi.Area_0, caller's order) in 3-D.  R: masked, host or device. */
int nlps_gpu_nodal_traction_forces(nlps_gpu *h, double *R, const nlps_bcc *loads, int nloads, int step,
                                   double thickness, const double *area0);

/* __update_particles_internal_variables, U-Newmark-beta.c:1917-1978 */
int nlps_gpu_roll_state(nlps_gpu *h);

/* __update_particles_kinetics_FLIP_PIC, U-Newmark-beta.c:1993-2072 */
int nlps_gpu_update_kinetics(nlps_gpu *h, double alpha_blend, const double *dU, const double *Un_dt,
                             const double *dU_dt, const double *dU_dt2);

/* Fused explicit predictor-corrector step (stage order and formulas of U-Verlet.c:229-253, 301-367,
 * 455-527, 530-676, 919-1010, 1024-1084; internal force in the Kirchhoff form of
 * U-Newmark-beta.c:1257-1374).  gravity[ndim] may be NULL.  One P2G+stress+G2P particle step. */
int nlps_gpu_explicit_step(nlps_gpu *h, const nlps_bcc *bcc, int nbcc, int step, double dt,
                           double gamma, const double *gravity);
/* Number of active nodes after the last search (computes Nodes2Mask on the device). */
int nlps_gpu_num_active(nlps_gpu *h, int *nactive);
/* Nodal results of the last explicit step in masked numbering (any pointer may be NULL).  On one GPU without a ghost
 * exchange the step itself keeps only the nodal sums (mass, momentum, force): dU, accelerations and reactions are made
 * from them by this call (or before a level-B stage overwrites the sums), two small kernels. */
int nlps_gpu_explicit_nodal(nlps_gpu *h, double *mass, double *dU, double *force, double *accel,
                            double *reaction);

/* Maintenance: physically re-sorts the device-resident particle arrays by (tile of the closest node,
 * corner type, closest node) so that memory order keeps matching the tile binning as particles move.
 * Results are unaffected (downloads always use the caller's particle order).  explicit_step calls it
 * every `every_n_steps` steps (default 50, 0 = never). */
int nlps_gpu_resort(nlps_gpu *h);
int nlps_gpu_set_resort_interval(nlps_gpu *h, int every_n_steps);
/* Adaptive re-sort of the fused explicit step (ON by default with budget 0.8, min_steps 4, whenever the interval above
 * is not 0; off in deterministic mode: the trigger reads a count the device publishes asynchronously, so the step that
 * re-sorts -- never a result beyond rounding -- can differ from run to run): the search stage counts the particles that are no
 * longer in the tile their memory slot was sorted into; every step adds that share of the cloud to a debt, and when
 * the debt since the last re-sort exceeds `budget` (and at least min_steps steps have passed) the step re-sorts ahead
 * of the interval above.  budget = 0 switches it off; an interval of 0 switches every re-sort off.  A budget of 0.6
 * is about the cost of one re-sort in units of the slowdown displaced particles cause.  No reference
 * counterpart (the CPU path has no memory order to keep). */
int nlps_gpu_set_adaptive_resort(nlps_gpu *gpu, double budget, int min_steps);

/* ------------------------------------------------------------------ multi-GPU: ghost-layer exchange over RCCL, owned by
 * the library.  One process per GPU; the particles are range-partitioned into slabs along the slowest grid
 * axis, rank r touches node layers [layer_lo[r], layer_hi[r]] (inclusive, every rank passes the ranges of ALL ranks).
 * After every nodal scatter of a stage function or of the fused explicit step the layers shared with rank r-1 / r+1
 * are summed (doubles) or OR-ed (activation flags): ncclSend / ncclRecv of the two contiguous slices on a
 * library-owned stream, behind the tiles that do not touch a shared layer (mode 0), or one ncclAllReduce of the whole
 * array (mode 1, BASELINE.json's wording).  librccl.so.1 is dlopen()ed at attach time.
 * nlps_gpu_rccl_unique_id: rank 0 makes the 128-byte ncclUniqueId, the host driver hands it to the other ranks
 * (MPI_Bcast, a file, ...).  _attach creates the communicator (ncclCommInitRank, collective), restricts the per-step
 * nodal work to the rank's layers (nlps_gpu_set_node_window) and declares the shared layers (nlps_gpu_set_ghost_bands);
 * _attach_comm takes an existing ncclComm_t.  Call before nlps_gpu_initialize_lme.
 * nlps_gpu_rccl_reduce: implicit driver on several ranks -- a masked vector (device pointer, n doubles: residual,
 * lumped mass) summed onto `root`, the rank that runs the PETSc solve, or onto all ranks (root < 0). */
int nlps_gpu_rccl_unique_id(void *id128);
int nlps_gpu_rccl_attach(nlps_gpu *h, const void *id128, int rank, int world, const int *layer_lo,
                         const int *layer_hi, int mode);
int nlps_gpu_rccl_attach_comm(nlps_gpu *h, void *nccl_comm, int rank, int world, const int *layer_lo,
                              const int *layer_hi, int mode);
int nlps_gpu_rccl_detach(nlps_gpu *h);
int nlps_gpu_rccl_reduce(nlps_gpu *h, double *vec, size_t n, int root);
/* What the attached communicator reports (ncclCommCount, ncclCommUserRank) and the overlap choreography in force
 * (0 blocking exchanges, 1 split launches, 2 one launch per stage); any pointer may be NULL. */
int nlps_gpu_rccl_info(nlps_gpu *h, int *nranks, int *rank, int *overlap_mode);
/* world-size-1 self-test of the exchange (one-GPU boxes): the rank is its own two neighbours -- the lowest three
 * layers of its range are exchanged with the highest three by ncclSend / ncclRecv to itself; overlap = the two-phase
 * (side-stream) form. */
int nlps_gpu_rccl_selftest_exchange(nlps_gpu *h, void *dptr, int nfield, int elem_bytes, int kind, int overlap);

/* Clouds whose materials follow several laws (Constitutive.c:28-258 dispatches per particle on MatProp.Type): the
 * fused stress stage runs either as one launch per law of the kernel compiled for that law (mode 1: right when the
 * materials sit in blocks, nearly every tile of closest nodes then holds one law) or as one kernel that dispatches on
 * the law at run time (mode 2: right when the laws are interleaved particle by particle).  nlps_gpu_create picks the
 * mode from the share of tiles that hold more than one law; results are identical. */
int nlps_gpu_set_law_launch_mode(nlps_gpu *h, int mode);

/* Run-to-run bit-reproducible results ("race detection"): with on != 0 every nodal sum of the fused
 * explicit step is accumulated in a FIXED order -- the per-tile particle lists are sorted, every wave accumulates into
 * a window of its own, the windows and the tiles are combined in index order (no floating-point atomics between
 * workgroups).  Slower than the default path (atomic accumulation in arrival order); same results to rounding. */
int nlps_gpu_set_deterministic(nlps_gpu *h, int on);

/* ------------------------------------------------------------------ per-dof updates of the implicit driver (a21)
 * Vectors of N_A*d doubles in masked numbering, host (VecGetArray) or device pointers.  alpha = the six Newmark
 * parameters alpha_1..alpha_6 of __compute_Newmark_parameters (U-Newmark-beta.c:497-514). */
/* __form_initial_guess, :879-957: optional explicit trial dU = dt*Un_dt + dt^2/2*Un_dt2, then the Dirichlet
 * values of the active boundary dofs at `step` */
int nlps_gpu_form_initial_guess(nlps_gpu *h, double *dU, const double *Un_dt, const double *Un_dt2, double dt,
                                int use_explicit_trial, const nlps_bcc *bcc, int nbcc, int step);
/* __compute_nodal_velocity_increments / __compute_nodal_kinetic_increments, :1834-1906 (either output may be NULL) */
int nlps_gpu_nodal_kinetic_increments(nlps_gpu *h, double *dU_dt, double *dU_dt2, const double *dU,
                                      const double *Un_dt, const double *Un_dt2, const double *alpha);
/* __nodal_inertial_forces, :1519-1557: R += M (alpha_1 dU - alpha_2 Un_dt - alpha_3 Un_dt2 - b) on the free dofs */
int nlps_gpu_nodal_inertial_forces(nlps_gpu *h, double *R, const double *M, const double *dU, const double *Un_dt,
                                   const double *Un_dt2, const double *alpha, const double *gravity);

/* __lagrangian_evaluation, U-Newmark-beta.c:970-1058 -- the callback SNES runs at every Newton iterate and every
 * line-search trial of the maintained driver -- as ONE call: VecZeroEntries(R), then
 *   __compute_nodal_velocity_increments (:1018), __local_compatibility_conditions (:1021-1022),
 *   __constitutive_update (:1026), __nodal_internal_forces (:1028-1029), __nodal_traction_forces (:1031-1032),
 *   __nodal_inertial_forces (:1034-1036).
 * On the device: the caller's dU is gathered, DF / F_n1 / J_n1 (J <= 0 clamped to 0 like :1137-1142), the stress update
 * and the scatter of the internal force run as ONE pass over the particles with DF and tau handed over in registers
 * (what nlps_gpu_compatibility + _constitutive + _internal_forces do in three passes with DF, F_n1, tau through HBM in
 * between), then one nodal kernel adds the traction and inertial terms and leaves 0 on the Dirichlet dofs.
 * The particle state afterwards is what the three separate stages leave: DF, F_n1, J_n1, Stress, W, b_e_n1, Kappa_n1,
 * EPS_n1, C_ep (the n state is not touched: the residual is evaluated many times from it), so
 * nlps_gpu_tangent_assemble / nlps_gpu_roll_state / nlps_gpu_update_kinetics follow it exactly as they follow the stages.
 * R, dU, Un_dt, Un_dt2, M: masked [N_A*d], host (VecGetArray) or device pointers; R is OVERWRITTEN.  alpha[6] as for
 * the per-dof updates above, gravity[ndim] or NULL, loads / nloads / step / thickness / area0 as for
 * nlps_gpu_nodal_traction_forces (nloads = 0: none).  Needs nlps_gpu_local_search + nlps_gpu_active_masks first.
 * flags: NLPS_LAGR_RATES   also the rate tensors dt_DF, dt_F_n1 (compute-Strains.c:48-72,176-207; only the Newtonian-fluid
 *                          law, which is not on this path, reads them: off by default) -- runs the separate stages;
 *        NLPS_LAGR_SEPARATE the composition of the separate stage calls (same results; what the fused form is tested and
 *                          timed against).  The separate stages also run when the damage hooks are on
 *                          (driver_eigenerosion / _eigensoftening: every stress before any force).  A cloud of several laws
 *                          runs one fused launch per law present.
 *        NLPS_LAGR_SAME_STEP host vectors only: Un_dt, Un_dt2 and M are the ones of the previous evaluation (they do not
 *                          change inside one SNES solve, U-Newmark-beta.c:241-300): their device copies are reused, an
 *                          evaluation then moves dU in and R out and nothing else (2 of 5 transfers).  An error without an
 *                          earlier evaluation since nlps_gpu_active_masks. */
enum { NLPS_LAGR_RATES = 1, NLPS_LAGR_SEPARATE = 2, NLPS_LAGR_SAME_STEP = 4 };
int nlps_gpu_lagrangian_evaluation(nlps_gpu *h, double *R, const double *dU, const double *Un_dt, const double *Un_dt2,
                                   const double *M, const double *alpha, const double *gravity, const nlps_bcc *loads,
                                   int nloads, int step, double thickness, const double *area0, int flags);

/* ------------------------------------------------------------------ tangent assembly */

/* __jacobian_evaluation (U-Newmark-beta.c:1646-1830) with stiffness_density__Constitutive__
 * (Constitutive.c:262-381): Neo-Hookean (Hyperelastic/Neo-Hookean.c:89-141), Hencky (Hencky.c:98-229) and
 * Drucker-Prager (Plasticity/Elastoplastic-Tangent-Matrix.c:42-163, with the C_ep of the last constitutive
 * update).  The d x d blocks V0 * stiffness_density(A, B) of every node pair a particle connects are summed on
 * the device, from the state the compatibility + constitutive stages left (DF, F_n, F_n1, J_n1, tau, b_e,n+1,
 * C_ep) and the current lists / lambda.  *nnz = number of COO entries = d*d * (number of structurally visited node
 * pairs, the reference's sparsity pattern).  Needs nlps_gpu_active_masks() first. */
int nlps_gpu_tangent_assemble(nlps_gpu *h, long long *nnz);
/* grouped != 0 (default): one workgroup per closest node sums the blocks of the particles sharing it before the
 * atomics; 0: one wave per particle (kept for comparison, same result up to summation order). */
int nlps_gpu_tangent_set_grouped(nlps_gpu *h, int grouped);
/* The assembled matrix as COO triplets (masked dof numbering, every visited pair present even when its value is
 * zero, like MatSetValues ... ADD_VALUES): rows[nnz], cols[nnz], vals[nnz], host or device pointers.
 * lumped_mass (masked [N_A*d], may be NULL): alpha_1 * M is added on the diagonal (:1797-1807).
 * apply_dirichlet != 0: rows and columns of the dofs fixed at the step given to nlps_gpu_active_masks() become
 * identity rows (MatZeroRowsColumnsIS, :1822). */
int nlps_gpu_tangent_coo(nlps_gpu *h, double alpha_1, const double *lumped_mass, int apply_dirichlet, int *rows,
                         int *cols, double *vals);
/* __create_sparsity_pattern (U-Newmark-beta.c:1568-1632): visited columns per dof row, nnz_per_row[N_A*d] */
int nlps_gpu_sparsity_pattern(nlps_gpu *h, int *nnz_per_row);

/* ------------------------------------------------------------------ multi-GPU hooks */

/* Halo exchange callback, invoked by explicit_step / the P2G stages after a nodal scatter, on the
 * handle's stream order.  dptr = device array [nnodes_grid][nfield] of `elem_bytes`-sized elements in
 * GRID numbering (x fastest, slab axis slowest => a slab halo is one contiguous byte range);
 * kind: 0 = sum doubles, 1 = OR bytes.  The callee sums/ORs the halo layers with the neighbouring
 * ranks (RCCL send/recv or all-reduce).  NULL = single GPU.
 * phase: 0 = exchange now, in the order of the handle's stream; 1 = start the exchange (the callee may run it on
 * a stream of its own, after making that stream wait for the work queued on the handle's stream so far);
 * 2 = make the handle's stream wait for the exchange started by the matching phase-1 call.  Phases 1/2 are only
 * used by nlps_gpu_explicit_step when nlps_gpu_set_ghost_bands(..., overlap = 1) was called. */
typedef int (*nlps_halo_fn)(void *ctx, void *dptr, int nfield, int elem_bytes, int kind, int phase);
int nlps_gpu_set_halo_exchange(nlps_gpu *h, nlps_halo_fn fn, void *ctx);
/* Promise that the 5^d stencils of this rank's particles stay inside node layers [layer_lo, layer_hi] of the
 * slab (slowest) axis: the per-step nodal work of explicit_step / local_search (resets, nodal kernels, tile
 * launches) is limited to that window instead of the whole grid, so a rank's cost does not grow with the number
 * of ranks.  A particle that leaves the window raises status flag 16.  Default: the whole grid. */
int nlps_gpu_set_node_window(nlps_gpu *h, int layer_lo, int layer_hi);
/* Ghost bands of this rank: node layers <= band_lo and >= band_hi (slab axis) are shared with a neighbouring
 * rank (pass band_lo < 0 / band_hi >= n_layers for "none").  overlap: 0 = every exchange blocks in place; 1 =
 * explicit_step launches the tiles that touch a band and the others separately and every exchange runs behind the
 * interior tiles of the next stage (two-phase callback above, or the library's RCCL path); 2 = ONE launch per stage
 * with the boundary tiles first: the last of them releases the library's exchange stream through a device flag and
 * the exchange runs beside the interior tiles of the same launch (library RCCL path only; what nlps_gpu_rccl_attach
 * selects for world > 1). */
int nlps_gpu_set_ghost_bands(nlps_gpu *h, int band_lo, int band_hi, int overlap);
/* Range of node layers along the slab axis this rank's particles may touch (5^d stencil reach). */
int nlps_gpu_touched_layers(nlps_gpu *h, int *lo, int *hi);

/* ---- particle migration between slab ranks.  Ownership follows the closest node: a particle whose
 * I0 lies below node layer keep_lo moves to the rank below, above keep_hi to the rank above.  The caller moves the
 * packed rows between ranks (RCCL send/recv, see nl-partsol_amd/halo.py::SlabHalo.migrate) between the two calls.
 * Call it every few steps, before a particle can leave the node window (status flag 16). */
/* step 1: select and pack.  n_down / n_up rows of row_words 8-byte words each wait in two device buffers owned by
 * the handle (valid until the next select). */
int nlps_gpu_migration_select(nlps_gpu *h, int keep_lo, int keep_hi, int *n_down, int *n_up, int *row_words,
                              void **down_rows, void **up_rows);
/* step 2: the selected particles leave, n_a + n_b immigrants (packed rows received from the two neighbours, host
 * or device pointers, may be NULL/0) join; the arrays are re-sorted.  Capacity is fixed at create:
 * np + max(np/4, 1024) particles. */
int nlps_gpu_migration_commit(nlps_gpu *h, const void *rows_a, int n_a, const void *rows_b, int n_b);
/* Both steps with the transport in between done by the library over the communicator of nlps_gpu_rccl_attach (a C driver
 * needs nothing else): select and pack, the row counts and then the rows exchanged with rank - 1 / rank + 1 by
 * ncclSend / ncclRecv on the handle's stream, commit.  Collective over the communicator: every rank calls it at the
 * same step.  An edge rank's keep range is clamped to the grid on its outer side.  Any output may be NULL. */
int nlps_gpu_rccl_migrate(nlps_gpu *h, int keep_lo, int keep_hi, int *sent_down, int *sent_up, int *received);
/* world-size-1 self-test of that transport (one-GPU boxes): the rank is its own two neighbours, the particles that
 * leave the keep range come straight back as immigrants (no clamping). */
int nlps_gpu_rccl_selftest_migrate(nlps_gpu *h, int keep_lo, int keep_hi, int *sent_down, int *sent_up, int *received);
/* current number of particles of this handle */
int nlps_gpu_num_particles(nlps_gpu *h, int *np);
/* global particle ids (default: the index in the arrays given to nlps_gpu_create).  After the first migration
 * nlps_gpu_download_state / _lists / _ids return their rows in ascending id (arrays of nlps_gpu_num_particles rows). */
int nlps_gpu_set_particle_ids(nlps_gpu *h, const int *ids);
int nlps_gpu_download_ids(nlps_gpu *h, int *ids);

/* ------------------------------------------------------------------ host-only helpers (no GPU needed) */

/* Stencil-order tables the library derives for the canonical grid numbering (see csrc/nlps_tables.hpp):
 * rank1[27][27]  chain position of each 3^d offset of NodalLocality_0 per 1-ring boundary class,
 * order2[125][125] / count2[125]  walk order of NodalLocality per 2-ring boundary class,
 * h_avg1[27]     mean 1-ring distance per class for h = 1 (Read_GramsBox.c:460-507). */
int nlps_host_stencil_tables(int ndim, unsigned char *rank1, unsigned char *order2, unsigned char *count2,
                             double *h_avg1);

/* ---- input formats: what stands on the input side of the path in the reference.
 * GiD ASCII meshes exactly as Nodes/Read-GID-Mesh.c:225-430 reads them: line 0 "MESH dimension d ElemType T Nnode n",
 * a Coordinates ... End Coordinates block of 4-word lines "id x y z" (also in 2-D), an Elements ... End Elements
 * block of "id n1 .. nN" lines, 1-based.  All four return 0 or 1; nlps_host_io_last_error() holds the reason. */
typedef struct nlps_gid_info {
  int ndim, nnodes, nelem, nodes_per_elem;
  char elem_type[32]; /* Triangle | Quadrilateral | Tetrahedra | Hexahedra */
} nlps_gid_info;
const char *nlps_host_io_last_error(void);
/* Read_Mesh_Information, Read-GID-Mesh.c:225-300 (stricter: a malformed line inside a block is an error, where the
 * reference would count it and leave the node unread) */
int nlps_host_gid_mesh_info(const char *path, nlps_gid_info *info);
/* Fill_Coordinates :304-352 and Fill_Linear_Conectivity :356-408.  coords[nnodes][ndim]; conn[nelem][nodes_per_elem]
 * 0-based and in the order the reference's chains hold the nodes, which is the REVERSE of the file order
 * (push__SetLib__ prepends, ChainOp.c:163-182). */
int nlps_host_gid_mesh_read(const char *path, const nlps_gid_info *info, double *coords, int *conn);
/* The structured lattice behind a background mesh (the GramsBox of configs 1-5): spacing, nodes per axis, origin,
 * and canon[file node] = lattice id (x fastest), the numbering every node-indexed array of this library uses.
 * Replaces the O(N_nodes x N_elem) neighbour construction of Read_GramsBox.c:293-456 and the element search of
 * LME.c:63-115 by the closed-form tables of nlps_host_stencil_tables.  Fails if the nodes are not such a lattice. */
int nlps_host_lattice_from_nodes(int ndim, int nnodes, const double *coords, double *h, int n[3], double origin[3],
                                 int *canon);
/* Particles of a body mesh: initial_position__Particles__ (Particles-Tools.c:8-28; Q4.c:342-452 with 1, 4, 5 or 9,
 * H8.c:389-575 with 1, 8 or 27, T3.c:337-440 with 1, 3, 4 or 9 and T4.c:322-420 with 1, 4 or 10 particles per
 * element) and the volumes of initialise_particles
 * (Generate-One-Phase-Analysis.c:569-625: element volume by 2^d-point quadrature / particles per element; thickness
 * is Thickness_Plain_Stress of the 2-D build).  x[nelem * gp][ndim], vol0[nelem * gp], particle p = e * gp + j. */
int nlps_host_particles_from_mesh(const nlps_gid_info *info, const double *coords, const int *conn, int gp_per_elem,
                                  double thickness, double *x, double *vol0);

/* ---- command file (.nlp), the subset the path needs before its first step: NLPS-Solver (Read_GramsTime.c:44-378, with
 * its defaults and check_Solver), GramsShapeFun (Read_GramsShapeFun.c:20-200), the mesh of GramsBox
 * (Read_GramsBox.c:235-262) and the body mesh / particles per element of One-Phase-Analysis
 * (Generate-One-Phase-Analysis.c:386-445); file names come back joined to the directory of the command file
 * (generate_route).  Materials, initial values, boundary conditions and outputs are not read.  Where the reference
 * prints and exit()s, this returns 1 with the same wording in nlps_host_io_last_error(). */
typedef struct nlps_deck {
  char box_mesh[512], body_mesh[512];
  int gp_per_elem;
  char scheme[64]; /* NLPS-Solver Type */
  double CFL, Cel;
  int i0, N;
  double epsilon_mass_matrix, beta_newmark, gamma_newmark, tol_newmark, rb_generalized_alpha, tol_generalized_alpha;
  int max_iter, explicit_trial;
  char shape_fun[16]; /* GramsShapeFun Type */
  double gamma_lme, tol_zero_lme, tol_wrapper_lme;
  int max_iter_lme;
  char wrapper_lme[32];
} nlps_deck;
int nlps_host_read_deck(const char *path, nlps_deck *deck);
/* The Define-Material(idx=i,Model=m) { property = value ... } blocks of the same file for the four laws of this path
 * (Read_GramsMaterials2.c:51-175; property names, defaults and completeness checks of Material/Hyperelastic/
 * Neo-Hookean.c, Hencky.c, Material/Plasticity/Drucker-Prager.c incl. its default reference plastic strain, and
 * Von-Mises.c).  mats / rho / idx (may be NULL) have room for max_materials entries, *nmats comes back.  Any other
 * model, or Fbar = true, is an error: this path does not cover them. */
int nlps_host_read_materials(const char *path, int max_materials, nlps_material *mats, double *rho, int *idx,
                             int *nmats);
/* The Dirichlet boundaries of the same file, GramsBoundary (File=nodes.txt) { BcDirichlet V.x curve.txt | NULL ... }
 * (NLPS-Read-u-Dirichlet-Boundary-Conditions.c:46-300, File2Chain.c, ReadCurve.c with its six curve kinds), in the
 * layout of nlps_bcc: boundary b has nnodes[b] nodes (concatenated in nodes[], in the reference's reversed file
 * order and in FILE numbering: map them through canon[] of nlps_host_lattice_from_nodes), dir and value
 * [b][ndim][nsteps].  With nodes / dir / value NULL only the counts come back (*nbounds, and nnodes[] if given). */
int nlps_host_read_boundaries(const char *path, int ndim, int nsteps, int max_bounds, int node_cap, int *nbounds,
                              int *nnodes, int *nodes, int *dir, double *value);
/* The Neumann contours of the same file, Define-Neumann-Boundary(File=elements.txt) { T.x curve.txt | NULL ... }
 * (NLPS-Read-u-Neumann-Boundary-Conditions.c:150-352): as above, but the list names elements of the body mesh and every
 * element stands for its particles e * gp_per_elem + j; the result is what nlps_gpu_nodal_traction_forces takes. */
int nlps_host_read_neumann(const char *path, int ndim, int nsteps, int gp_per_elem, int max_bounds, int node_cap,
                           int *nbounds, int *nnodes, int *nodes, int *dir, double *value);
/* Assign-material-to-particles (MatIdx=i, Particles=elements.txt) (Generate-One-Phase-Analysis.c:458-566):
 * matidx[nparticles] is updated in place for the particles of the listed body elements. */
int nlps_host_read_material_assignment(const char *path, int gp_per_elem, int nmaterials, int nparticles, int *matidx);
/* The initial velocities of the same file, GramsInitials (Nodes=list.txt) { Value=[vx,vy,vz] }
 * (Read_GramsInitials.c:7-186): the list names ELEMENTS of the body mesh (0-based), all gp_per_elem particles of a
 * listed element get the value.  vel[nparticles][ndim] is updated in place. */
int nlps_host_read_initials(const char *path, int ndim, int gp_per_elem, int nparticles, double *vel);
/* The gravity field of the same file, generate-gravity-field-constant { g.x = .. } or generate-gravity-field-curve
 * { g = file.csv } (Read_Generate_Gravity_Field.c:170-371): g[nsteps][ndim], row t is the gravity argument of
 * nlps_gpu_explicit_step at step t; *found (may be NULL) tells whether the file holds such a block. */
int nlps_host_read_gravity(const char *path, int ndim, int nsteps, double *g, int *found);
/* The output block of the same file, GramsOutputs (i=int) { DIR=dir Particles-file=name Nodes-file=name Out-...=true|false }
 * (Outputs/Read_GramsOutputs.c:25-345): interval, directory (joined to the command file's; it has to exist), file
 * names, and the Out_* switches of the blocks nlps_host_write_particles_vtk writes; switches of other blocks are
 * accepted and counted in `unsupported` when on.  found = 0 if the file has no such block. */
typedef struct nlps_outputs {
  int found, results_time_step;
  char dir[512], particles_file[128], nodes_file[128];
  int global_coordinates, mass, density, nodal_idx, material_idx, velocity, acceleration, displacement, stress,
      volumetric_stress, deformation_gradient, energy, eps;
  int unsupported;
} nlps_outputs;
int nlps_host_read_outputs(const char *path, nlps_outputs *out);

/* ---- output format: the particle file of particle_results_vtk__InOutFun__ (InOutFun/Outputs/WriteVtk.c:95-266):
 * legacy ASCII VTK, one vertex cell per particle, numbers as %.20g, blocks in the reference's order.  Arrays are in
 * the layout of nlps_gpu_download_state ([np][ndim], tensors [np][5 | 9]); a NULL array leaves its block out, like
 * the reference's Out_* switches.  flags add the blocks that derive from shared arrays. */
typedef struct nlps_vtk_fields {
  const double *x;            /* POINTS (required); X_GC with NLPS_VTK_X_GC */
  const double *mass, *rho;   /* MASS, DENSITY */
  const int *I0, *matidx;     /* ELEM_i, MatIdx */
  const double *vel, *acc, *dis; /* VELOCITY, ACCELERATION, DISPLACEMENT */
  const double *stress;       /* STRESS (2-D: zz from slot 4); P with NLPS_VTK_P */
  const double *F_n;          /* DEFORMATION-GRADIENT */
  const double *W;            /* Energy-Potential + Energy-Kinetic with NLPS_VTK_ENERGY (needs vel, mass) */
  const double *eps;          /* EPS */
} nlps_vtk_fields;
enum { NLPS_VTK_X_GC = 1, NLPS_VTK_P = 2, NLPS_VTK_ENERGY = 4 };
int nlps_host_write_particles_vtk(const char *path, int results_time_step, int ndim, int np,
                                  const nlps_vtk_fields *fields, int flags);
/* The nodal file of nodal_results_vtk__InOutFun__ (WriteVtk.c:269-405): the background mesh in file numbering (info,
 * coords, conn as read by nlps_host_gid_mesh_read), the active-node mask and the reactions.  active[nnodes] and
 * reactions[nnodes][ndim] are the library's lattice-numbered arrays (nlps_gpu_download_active,
 * nlps_gpu_explicit_nodal); canon[file node] = lattice node, NULL = identity. */
int nlps_host_write_nodes_vtk(const char *path, const nlps_gid_info *info, const double *coords, const int *conn,
                              const int *canon, const unsigned char *active, const double *reactions);

/* ------------------------------------------------------------------ measurement */

/* Time (ms, HIP events on the handle's stream) of the kernels of the last explicit step:
 * [0] search+activate, [1] lists+Newton+P2G mass/momentum, [2] G2P grad+F+stress+P2G force,
 * [3] G2P kinematics+roll, [4] nodal/mask kernels, [5] a calibration bracket around a kernel of K3's grid that
 * does nothing (the part of a one-kernel bracket that is not kernel time).  Enabled with nlps_gpu_set_timing(h,1). */
int nlps_gpu_set_timing(nlps_gpu *h, int on);
int nlps_gpu_get_timing(nlps_gpu *h, float ms[8]);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
