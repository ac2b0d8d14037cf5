/* nlps_glue.c — the reference-side binding of include/nlps_gpu.h.
 *
 * This file is meant to be dropped into nl-partsol/src/Formulations/Displacements/ and compiled WITH the
 * reference (it includes the reference's own headers; nothing of them is copied here).  It marshals the
 * reference's Particle / Mesh / Material / Boundaries structures (Types.h:14-797) into the plain-pointer
 * structures of the C-ABI, so that U-Newmark-beta.c / U-Static.c can replace their static stage functions by
 * the nlps_gpu_* calls listed in INTEGRATION.md.  It cannot be linked into the reference here (the reference needs
 * PETSc and LAPACK); tests/test_abi.py::test_glue_compiles_against_the_reference_headers compiles it to an object
 * against the real Types.h in both dimensions whenever the reference tree is present and checks the nlps_gpu_* symbols
 * it leaves undefined against the library's exports.
 *
 * Node numbering: the library indexes every nodal array in LATTICE numbering (x fastest).  A GiD box mesh may number
 * its nodes differently, so the glue keeps canon[file node] = lattice node (nlps_host_lattice_from_nodes) and maps
 * I0, the Dirichlet node lists, h_avg and Nodes2Mask through it, both ways.
 *
 *   cc -std=gnu99 [-DUSE_PLAINSTRAIN] -I<nl-partsol>/src -I<this repo>/include -c nlps_glue.c
 */
#include <math.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "Macros.h"
#include "Types.h"
#include "Globals.h"

#include "nlps_gpu.h"

/* law ids of the C-ABI from Material.Type, the strings of Constitutive.c:28-258 */
static int nlps_glue_law(const Material *M) {
  if (strcmp(M->Type, "Neo-Hookean-Wriggers") == 0) return NLPS_MAT_NEO_HOOKEAN;
  if (strcmp(M->Type, "Hencky") == 0) return NLPS_MAT_HENCKY;
  if (strcmp(M->Type, "Drucker-Prager") == 0) return NLPS_MAT_DRUCKER_PRAGER;
  if (strcmp(M->Type, "Von-Mises") == 0) return NLPS_MAT_VON_MISES;
  if (strcmp(M->Type, "Matsuoka-Nakai") == 0) return NLPS_MAT_MATSUOKA_NAKAI;
  if (strcmp(M->Type, "Lade-Duncan") == 0) return NLPS_MAT_LADE_DUNCAN;
  return -1;
}

/* what the binding keeps between calls */
typedef struct {
  nlps_gpu *gpu;
  int N;          /* FEM_Mesh.NumNodesMesh */
  int identity;   /* the mesh file already numbers its nodes x fastest */
  int *canon;     /* [N] file node -> lattice node */
  int *file_of;   /* [N] lattice node -> file node */
  int *I0_l;      /* [NumGP] closest nodes in lattice numbering (upload / download staging) */
  double *h_avg_l; /* [N] FEM_Mesh.h_avg in lattice numbering */
} nlps_glue;

void nlps_glue_free(nlps_glue *G) {
  if (G == NULL) return;
  if (G->gpu) nlps_gpu_destroy(G->gpu);
  free(G->canon);
  free(G->file_of);
  free(G->I0_l);
  free(G->h_avg_l);
  memset(G, 0, sizeof *G);
}

/* The GramsBox lattice behind FEM_Mesh.Coordinates: spacing, nodes per axis, origin and the node map.  Fails loudly
 * when the background mesh is not a structured lattice (such meshes stay on the CPU path). */
static int nlps_glue_lattice(nlps_glue *G, const Mesh *FEM_Mesh, nlps_grid *g) {
  const int Ndim = NumberDimensions;
  const int N = FEM_Mesh->NumNodesMesh;
  G->N = N;
  G->canon = (int *)malloc((size_t)N * sizeof(int));
  G->file_of = (int *)malloc((size_t)N * sizeof(int));
  G->h_avg_l = (double *)malloc((size_t)N * sizeof(double));
  if (G->canon == NULL || G->file_of == NULL || G->h_avg_l == NULL) return EXIT_FAILURE;
  memset(g, 0, sizeof *g);
  g->ndim = Ndim;
  /* Coordinates.nV is the contiguous [N][Ndim] storage of the Matrix (MatrixOp.c:128-181) */
  if (nlps_host_lattice_from_nodes(Ndim, N, FEM_Mesh->Coordinates.nV, &g->h, g->n, g->origin, G->canon) != 0) {
    fprintf(stderr, "" RED "nlps_glue: the background mesh is not a structured lattice: %s" RESET "\n",
            nlps_host_io_last_error());
    return EXIT_FAILURE;
  }
  G->identity = 1;
  for (int A = 0; A < N; A++) {
    if (G->canon[A] < 0 || G->canon[A] >= N) return EXIT_FAILURE;
    G->file_of[G->canon[A]] = A;
    G->identity &= (G->canon[A] == A);
    G->h_avg_l[G->canon[A]] = FEM_Mesh->h_avg[A];
  }
  g->h_avg = G->h_avg_l;
  return EXIT_SUCCESS;
}

/* The particle arrays the reference already owns (Matrix.nV is the contiguous row-major storage,
 * Matlib/MatrixOp.c:128-181). */
static nlps_particles nlps_glue_particles(Particle MPM_Mesh) {
  nlps_particles p;
  memset(&p, 0, sizeof p);
  p.np = MPM_Mesh.NumGP;
  p.x_GC = MPM_Mesh.Phi.x_GC.nV;
  p.dis = MPM_Mesh.Phi.dis.nV;
  p.vel = MPM_Mesh.Phi.vel.nV;
  p.acc = MPM_Mesh.Phi.acc.nV;
  p.F_n = MPM_Mesh.Phi.F_n.nV;
  p.F_n1 = MPM_Mesh.Phi.F_n1.nV;
  p.DF = MPM_Mesh.Phi.DF.nV;
  p.Stress = MPM_Mesh.Phi.Stress.nV;
  p.b_e_n = MPM_Mesh.Phi.b_e_n.nV;
  p.b_e_n1 = MPM_Mesh.Phi.b_e_n1.nV;
  p.J_n = MPM_Mesh.Phi.J_n.nV;
  p.J_n1 = MPM_Mesh.Phi.J_n1.nV;
  p.rho = MPM_Mesh.Phi.rho.nV;
  p.mass = MPM_Mesh.Phi.mass.nV;
  p.Vol_0 = MPM_Mesh.Phi.Vol_0.nV;
  p.W = MPM_Mesh.Phi.W;
  p.Kappa_n = MPM_Mesh.Phi.Kappa_n;
  p.Kappa_n1 = MPM_Mesh.Phi.Kappa_n1;
  p.EPS_n = MPM_Mesh.Phi.EPS_n;
  p.EPS_n1 = MPM_Mesh.Phi.EPS_n1;
  p.MatIdx = MPM_Mesh.MatIdx;
  p.I0 = MPM_Mesh.I0;
  p.lambda = MPM_Mesh.lambda.nV;
  p.Beta = MPM_Mesh.Beta.nV;
  p.dt_F_n = MPM_Mesh.Phi.dt_F_n.nV;
  p.dt_F_n1 = MPM_Mesh.Phi.dt_F_n1.nV;
  p.dt_DF = MPM_Mesh.Phi.dt_DF.nV;
  p.C_ep = MPM_Mesh.Phi.C_ep.nV;
  p.Back_stress = MPM_Mesh.Phi.Back_stress.nV;
  p.Damage_n = MPM_Mesh.Phi.Damage_n;
  p.Damage_n1 = MPM_Mesh.Phi.Damage_n1;
  p.Strain_f_n = MPM_Mesh.Phi.Strain_f_n;
  p.Strain_f_n1 = MPM_Mesh.Phi.Strain_f_n1;
  return p;
}

/* after initialise_shapefun__MeshTools__ (driver-nl-partsol.c:344): upload everything once */
int nlps_glue_create(nlps_glue *G, Mesh FEM_Mesh, Particle MPM_Mesh, Time_Int_Params Parameters_Solver) {
  nlps_grid g;
  memset(G, 0, sizeof *G);
  if (strcmp(wrapper_LME, "Newton-Raphson") != 0) { /* LME.c:97-101: Nelder-Mead also changes the initial lambda */
    fprintf(stderr, "" RED "nlps_glue: wrapper_LME = %s stays on the CPU path (the GPU path restates Newton-Raphson)" RESET "\n",
            wrapper_LME);
    return EXIT_FAILURE;
  }
  if (nlps_glue_lattice(G, &FEM_Mesh, &g) == EXIT_FAILURE) return EXIT_FAILURE;
  /* snapshot of the globals the level-A functions read implicitly (Globals.h:33-58) */
  nlps_params prm = {gamma_LME, TOL_zero_LME, TOL_wrapper_LME, max_iter_LME, TOL_Radial_Returning,
                     Max_Iterations_Radial_Returning, Driver_EigenErosion ? 1 : 0, Driver_EigenSoftening ? 1 : 0};
  const int Nmat = MPM_Mesh.NumberMaterials;
  nlps_material *mats = (nlps_material *)calloc((size_t)Nmat, sizeof(nlps_material));
  if (mats == NULL) return EXIT_FAILURE;
  for (int m = 0; m < Nmat; m++) {
    const Material *M = &MPM_Mesh.Mat[m];
    const int law = nlps_glue_law(M);
    if (law < 0) {
      fprintf(stderr, "" RED "nlps_glue: material %s stays on the CPU path" RESET "\n", M->Type);
      free(mats);
      return EXIT_FAILURE;
    }
    mats[m].type = law;
    mats[m].E = M->E;
    mats[m].nu = M->nu;
    mats[m].phi_deg = M->phi_Frictional;
    mats[m].psi_deg = M->psi_Frictional;
    mats[m].kappa_0 = M->kappa_0;
    mats[m].exponent_ortiz = M->Exponent_Hardening_Ortiz;
    mats[m].eps_0 = M->Plastic_Strain_0;
    mats[m].p_ref = M->ReferencePressure;
    mats[m].hardening_modulus = M->Hardening_modulus;
    mats[m].theta_voce = M->theta_Hardening_Voce;
    mats[m].K0_voce = M->K_0_Hardening_Voce;
    mats[m].Kinf_voce = M->K_inf_Hardening_Voce;
    mats[m].delta_voce = M->delta_Hardening_Voce;
    mats[m].Ceps = M->Ceps;
    mats[m].Gf = M->Gf;
    mats[m].ft = M->ft;
    mats[m].heps = M->heps;
    mats[m].wcrit = M->wcrit;
    mats[m].cohesion = M->Cohesion;
    mats[m].alpha_borja = M->alpha_Hardening_Borja;
    for (int k = 0; k < 3; k++) mats[m].a_borja[k] = M->a_Hardening_Borja[k];
  }
  nlps_particles p = nlps_glue_particles(MPM_Mesh);
  /* closest nodes in lattice numbering */
  G->I0_l = (int *)malloc((size_t)(p.np > 0 ? p.np : 1) * sizeof(int));
  if (G->I0_l == NULL) {
    free(mats);
    return EXIT_FAILURE;
  }
  for (int q = 0; q < p.np; q++) G->I0_l[q] = G->canon[MPM_Mesh.I0[q]];
  p.I0 = G->I0_l;
  const int STATUS = nlps_gpu_create(&G->gpu, &g, &prm, mats, Nmat, &p, Parameters_Solver.NumTimeStep, NULL);
  free(mats);
  if (STATUS != EXIT_SUCCESS) {
    fprintf(stderr, "" RED "%s" RESET "\n", G->gpu ? nlps_gpu_last_error(G->gpu) : "nlps_gpu_create");
    return STATUS;
  }
  /* a mesh file that is not numbered x-fastest: the masked numbering follows the FILE's node order (Nodes-Tools.c:46-66) */
  if (!G->identity && nlps_gpu_set_node_numbering(G->gpu, G->canon) != EXIT_SUCCESS) {
    fprintf(stderr, "" RED "%s" RESET "\n", nlps_gpu_last_error(G->gpu));
    return EXIT_FAILURE;
  }
  return STATUS;
}

/* FEM_Mesh.Bounds (Types.h:296-351) flattened to the value[k*NumTimeStep + t] layout of nlps_bcc, node lists in
 * lattice numbering.  The caller frees bcc[i].nodes, bcc[i].value and bcc. */
nlps_bcc *nlps_glue_boundaries(const nlps_glue *G, Mesh FEM_Mesh, int NumTimeStep, int *nbcc) {
  const int NumBounds = FEM_Mesh.Bounds.NumBounds;
  nlps_bcc *bcc = (nlps_bcc *)calloc((size_t)(NumBounds > 0 ? NumBounds : 1), sizeof(nlps_bcc));
  if (bcc == NULL) return NULL;
  for (int i = 0; i < NumBounds; i++) {
    const Load *L = &FEM_Mesh.Bounds.BCC_i[i];
    double *value = (double *)calloc((size_t)L->Dim * NumTimeStep, sizeof(double));
    if (value == NULL) return NULL;
    for (int k = 0; k < L->Dim; k++)
      for (int t = 0; t < NumTimeStep && t < L->Value[k].Num; t++) value[(size_t)k * NumTimeStep + t] = L->Value[k].Fx[t];
    int *nodes = (int *)malloc((size_t)(L->NumNodes > 0 ? L->NumNodes : 1) * sizeof(int));
    if (nodes == NULL) return NULL;
    for (int q = 0; q < L->NumNodes; q++) nodes[q] = G->canon[L->Nodes[q]];
    bcc[i].nnodes = L->NumNodes;
    bcc[i].nodes = nodes;
    bcc[i].dim = L->Dim;
    bcc[i].dir = L->Dir; /* Dir[k*NumTimeStep + t], Nodes-Tools.c:130 */
    bcc[i].value = value;
  }
  *nbcc = NumBounds;
  return bcc;
}

/* get_active_nodes__MeshTools__ + get_active_dofs__MeshTools__ (U-Newmark-beta.c:205-209): the Mask structures
 * the host still needs for the PETSc sizes.  Nodes2Mask arrays are malloc'd like the reference's (caller frees). */
int nlps_glue_masks(const nlps_glue *G, Mesh FEM_Mesh, const nlps_bcc *bcc, int nbcc, int TimeStep, Mask *ActiveNodes,
                    Mask *ActiveDOFs) {
  const int Ndim = NumberDimensions;
  nlps_gpu *GPU = G->gpu;
  int Nactivenodes = 0, Nfree = 0;
  ActiveNodes->Nodes2Mask = (int *)malloc((size_t)FEM_Mesh.NumNodesMesh * sizeof(int));
  ActiveDOFs->Nodes2Mask = (int *)malloc((size_t)FEM_Mesh.NumNodesMesh * Ndim * sizeof(int));
  if (ActiveNodes->Nodes2Mask == NULL || ActiveDOFs->Nodes2Mask == NULL) return EXIT_FAILURE;
  if (nlps_gpu_active_masks(GPU, bcc, nbcc, TimeStep, &Nactivenodes, &Nfree, ActiveNodes->Nodes2Mask,
                            ActiveDOFs->Nodes2Mask) != EXIT_SUCCESS) {
    fprintf(stderr, "" RED "%s" RESET "\n", nlps_gpu_last_error(GPU));
    return EXIT_FAILURE;
  }
  /* (for a mesh file that is not numbered x-fastest nlps_glue_create has handed the file's node order to the library,
   * nlps_gpu_set_node_numbering: Nodes2Mask arrives indexed by file node with the running index in file order, exactly
   * the array of get_active_nodes__MeshTools__, and the dof mask -- indexed by MASKED node, Nodes-Tools.c:96-135 -- with it) */
  ActiveNodes->Nactivenodes = Nactivenodes;
  ActiveDOFs->Nactivenodes = Nfree;
  return EXIT_SUCCESS;
}

/* before particle_results_vtk__InOutFun__ (U-Newmark-beta.c:409) or any host-side use of MPM_Mesh.Phi */
int nlps_glue_download(const nlps_glue *G, Particle MPM_Mesh) {
  nlps_particles p = nlps_glue_particles(MPM_Mesh);
  p.I0 = G->I0_l;
  const int STATUS = nlps_gpu_download_state(G->gpu, &p);
  if (STATUS != EXIT_SUCCESS) fprintf(stderr, "" RED "%s" RESET "\n", nlps_gpu_last_error(G->gpu));
  for (int q = 0; q < p.np && STATUS == EXIT_SUCCESS; q++) MPM_Mesh.I0[q] = G->file_of[G->I0_l[q]];
  return STATUS;
}

/* U-Static.c:1380-1470, __update_Particles(dU, MPM_Mesh, FEM_Mesh, ActiveNodes): the roll of the internal variables
 * and x_GC, dis += sum_A N_pA dU_A; the quasi-static driver has no velocities to update (NULL rate vectors). */
int nlps_glue_update_particles_static(const nlps_glue *G, const double *dU /* VecGetArrayRead(dU) */) {
  if (nlps_gpu_roll_state(G->gpu) != EXIT_SUCCESS || nlps_gpu_update_kinetics(G->gpu, 1.0, dU, NULL, NULL, NULL) != EXIT_SUCCESS) {
    fprintf(stderr, "" RED "%s" RESET "\n", nlps_gpu_last_error(G->gpu));
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}

/* U-Newmark-beta.c:970-1058, the body of __lagrangian_evaluation(snes, dU, Lagrangian, ctx) between its VecGetArray and
 * VecRestoreArray calls: the reference's
 *     VecZeroEntries(Lagrangian); dU_dt = __compute_nodal_velocity_increments(...);
 *     __local_compatibility_conditions(...); __constitutive_update(...); __nodal_internal_forces(...);
 *     __nodal_traction_forces(...); __nodal_inertial_forces(...);
 * becomes ONE library call.  The pointers are the driver's own (VecGetArray / VecGetArrayRead of Lagrangian, dU and of
 * ctx->Lumped_Mass, U_n_dt, U_n_dt2); alpha[6] = {alpha_1 .. alpha_6} of ctx->Time_Integration_Params (:497-514);
 * neumann[] = MPM_Mesh.Neumann_Contours flattened like the Dirichlet boundaries (nodes = particle indices);
 * first_evaluation_of_the_step != 0 for the first residual of a time step (the three vectors that do not change during
 * the SNES solve are then copied to the device, afterwards reused).  Returns EXIT_SUCCESS / EXIT_FAILURE like the static. */
int nlps_glue_lagrangian_evaluation(const nlps_glue *G, Particle MPM_Mesh, double *Lagrangian_ptr, const double *dU_ptr,
                                    const double *Un_dt_ptr, const double *Un_dt2_ptr, const double *Lumped_Mass_ptr,
                                    const double alpha[6], const double *gravity /* b of :1519-1557, or NULL */,
                                    const nlps_bcc *neumann, int nneumann, int TimeStep, int first_evaluation_of_the_step) {
#if NumberDimensions == 2
  const double thickness = Thickness_Plain_Stress;
  const double *area0 = NULL;
#else
  const double thickness = 1.0;
  const double *area0 = MPM_Mesh.Phi.Area_0.nV; /* :1440-1444 */
#endif
  const int flags = first_evaluation_of_the_step ? 0 : NLPS_LAGR_SAME_STEP;
  const int STATUS = nlps_gpu_lagrangian_evaluation(G->gpu, Lagrangian_ptr, dU_ptr, Un_dt_ptr, Un_dt2_ptr, Lumped_Mass_ptr, alpha,
                                                    gravity, neumann, nneumann, TimeStep, thickness, area0, flags);
  if (STATUS != EXIT_SUCCESS) fprintf(stderr, "" RED "%s" RESET "\n", nlps_gpu_last_error(G->gpu));
  return STATUS;
}

/* U-Static.c:492-551, the same callback of the quasi-static driver: no rate vectors, the inertial term is - M b
 * (:1005-1033).  The dynamic call with alpha = 0 gives exactly that (0 * dU - 0 * v - 0 * a - b); dU stands in for the
 * two rate vectors, which are multiplied by zero. */
int nlps_glue_lagrangian_evaluation_static(const nlps_glue *G, Particle MPM_Mesh, double *Lagrangian_ptr, const double *dU_ptr,
                                           const double *Lumped_Mass_ptr, const double *b /* gravity or NULL */,
                                           const nlps_bcc *neumann, int nneumann, int TimeStep) {
  static const double alpha0[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  return nlps_glue_lagrangian_evaluation(G, MPM_Mesh, Lagrangian_ptr, dU_ptr, dU_ptr, dU_ptr, Lumped_Mass_ptr, alpha0, b, neumann,
                                         nneumann, TimeStep, 1);
}
