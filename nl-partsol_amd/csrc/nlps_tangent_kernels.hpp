// tangent (Jacobian) assembly of the implicit driver, Neo-Hookean particles.
//
// Restates __jacobian_evaluation (Formulations/Displacements/U-Newmark-beta.c:1646-1830) with
// stiffness_density__Constitutive__ -> compute_stiffness_density_Neo_Hookean (Constitutive.c:262-283,
// Hyperelastic/Neo-Hookean.c:89-141) and __create_sparsity_pattern (:1568-1632).
//
// Layout.  On the GramsBox lattice two nodes can only share a particle when they are at most 4 nodes apart per
// axis, so the block-sparse matrix is a dense "stencil" array K[node A][offset B-A in [-4,4]^d][d][d] in GRID
// numbering plus one byte per (A, offset) that records the structural visits of the reference's
// MatSetValues/sparsity pattern.  One wave assembles one particle: the lanes first build the particle's member
// table in LDS (grad N^n, its push-forward DF^-T grad N^n and b_n grad N^n per member), then share the
// N_n x N_n node pairs and add their d x d blocks with global f64 atomics (the pair space is far too large
// for LDS staging: 81 / 729 blocks per window node).  k_tangent_emit then walks the stencil array and writes
// COO triplets in the masked dof numbering PETSc uses, with alpha_1 * M on the diagonal (:1797-1807) and the
// Dirichlet rows/columns reduced to the identity (MatZeroRowsColumnsIS, :1822).
#pragma once

// Vol_0 of the particle; with the damage drivers on the stiffness density is scaled by (1 - Damage_n1)
// (U-Newmark-beta.c:1757-1764) and it enters every block linearly with Vol_0
__device__ __forceinline__ double tangent_vol(const PView& P, int p) {
  const double V0 = PF(P, F_VOL0, p);
  return P.erosion ? V0 * (1.0 - PF(P, F_DMG1, p)) : V0;
}

template <int ND>
struct TanCfg {
  static constexpr int S = (ND == 3) ? 729 : 81;    // stencil offsets per row node
  static constexpr int MAXM = (ND == 3) ? 125 : 25;  // members of one particle
};

template <int ND>
__device__ __forceinline__ int tangent_offset_index(int sA, int sB) {
  const int dx = (sB % 5) - (sA % 5), dy = ((sB / 5) % 5) - ((sA / 5) % 5), dz = (ND == 3) ? (sB / 25) - (sA / 25) : 0;
  return (dx + 4) + 9 * ((dy + 4) + (ND == 3 ? 9 * (dz + 4) : 0));
}

template <int ND>
__global__ __launch_bounds__(64) void k_tangent_nh(PView P, GridD g, const MatD* __restrict__ mats,
                                                   double* __restrict__ Kst, unsigned char* __restrict__ touched,
                                                   int* __restrict__ gstatus) {
  constexpr int S = TanCfg<ND>::S, MAXM = TanCfg<ND>::MAXM, KN = Lme<ND>::KN;
  __shared__ double tab[6][5];  // ex, ey, ez, lx, ly, lz of this particle
  __shared__ double gn[MAXM][ND], g1[MAXM][ND], ub[MAXM][ND];
  __shared__ int mnode[MAXM], mcode[MAXM];
  const int p = blockIdx.x, lane = threadIdx.x;
  if (p >= P.np) return;
  Lme<ND> c;
  double lam[ND], beta;
  if (!load_lme<ND>(P, g, p, c, lam, beta)) return;  // no neighbourhood: flagged by the search already
  const MatD m = mats[P.mat[p]];
  if (m.type != NLPS_MAT_NEO_HOOKEAN) {  // only this law's tangent is restated
    if (lane == 0) {
      atomicOr(&P.status[p], ST_CONSTITUTIVE);
      atomicOr(gstatus, ST_CONSTITUTIVE);
    }
    return;
  }
  double Zinv, r[ND], J[ND * ND], Jm1[ND * ND], DF[ND * ND], DFm1[ND * ND], Fn[ND * ND], bn[ND * ND], zz;
  lme_moments_h<ND>(c, Zinv, r, J);
  load_block<ND>(P, F_DF, p, DF, zz);
  load_block<ND>(P, fFN(P), p, Fn, zz);
  if (!inverse<ND>(Jm1, J) || !inverse<ND>(DFm1, DF)) {  // dp__LME__ (LME.c:836-891), push_forward_dN (Shape-Functions.c:405-448)
    if (lane == 0) {
      atomicOr(&P.status[p], ST_NEWTON);
      atomicOr(gstatus, ST_NEWTON);
    }
    return;
  }
  left_cauchy_green<ND>(bn, Fn);
#pragma unroll
  for (int i = 0; i < 5; i++)  // every lane holds the same factors; static register indices only
    if (lane == i) {
      tab[0][i] = c.ex[i];
      tab[1][i] = c.ey[i];
      tab[2][i] = (ND == 3) ? c.ez[i % KN] : 1.0;
      tab[3][i] = c.lx[i];
      tab[4][i] = c.ly[i];
      tab[5][i] = (ND == 3) ? c.lz[i % KN] : 0.0;
    }
  __syncthreads();
  int nn = 0;
  for (int s0 = 0; s0 < MAXM; s0 += 64) {
    const int s = s0 + lane;
    const bool mem = s < MAXM && c.on(s);
    const u64 bal = __ballot(mem);
    if (mem) {
      const int pos = nn + (int)__popcll(bal & ((1ull << lane) - 1ull));
      const int i = s % 5, j = (s / 5) % 5, k = s / 25;
      const double l[3] = {tab[3][i], tab[4][j], tab[5][k]};
      const double pa = tab[0][i] * tab[1][j] * tab[2][k] * Zinv;
      double ga[ND];
#pragma unroll
      for (int a = 0; a < ND; a++) {
        double v = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < ND; b2++) v = fma(Jm1[a * ND + b2], l[b2], v);
        ga[a] = -pa * v;
      }
#pragma unroll
      for (int a = 0; a < ND; a++) {
        double v1 = 0.0, vb = 0.0;
#pragma unroll
        for (int b2 = 0; b2 < ND; b2++) {
          v1 = fma(DFm1[b2 * ND + a], ga[b2], v1);  // DF^-T
          vb = fma(bn[a * ND + b2], ga[b2], vb);
        }
        gn[pos][a] = ga[a];
        g1[pos][a] = v1;
        ub[pos][a] = vb;
      }
      mnode[pos] = c.I0 + c.node_offset(g, i, j, k);
      mcode[pos] = s;
    }
    nn += (int)__popcll(bal);
  }
  __syncthreads();
  const double Jp = PF(P, F_JN1, p), sqrJ = Jp * Jp, V0 = tangent_vol(P, p);
  const double c0 = m.lame * sqrJ, c1 = m.G - 0.5 * m.lame * (sqrJ - 1);  // Neo-Hookean.c:107-110
  for (int q = lane; q < nn * nn; q += 64) {
    const int A = q / nn, B = q - A * nn;
    const size_t blk = (size_t)mnode[A] * S + tangent_offset_index<ND>(mcode[A], mcode[B]);
    double len0 = 0.0;
#pragma unroll
    for (int a = 0; a < ND; a++) len0 = fma(gn[B][a], ub[A][a], len0);  // dN_beta_n . (b_n dN_alpha_n)
    double* out = Kst + blk * (ND * ND);
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) {
        const double v = c0 * g1[A][i] * g1[B][j] + (i == j ? m.G * len0 : 0.0) + c1 * g1[A][j] * g1[B][i];
        atomic_add_f64(out + i * ND + j, v * V0);
      }
    touched[blk] = 1;
  }
}

// ---- grouped form: one workgroup per closest node I0.  The particles that share an I0 share their stencil, so
// the d x d blocks of a node pair are summed over the group in registers and reach memory with one atomic per
// entry instead of one per particle (the pair space of one particle, 625 / 15 625 blocks, is what made the
// per-particle form atomic-bound).  The group's member tables live in LDS, indexed by the stencil code
// s = i + 5 j + 25 k (zeros for non-members); TAN_GROUP particles per pass.
static constexpr int TAN_GROUP = 8;
// Threads of the grouped kernel: 8 waves, one per particle of a chunk in phase A.  Phase B stages the d x d blocks of
// a wave's 64 node pairs in LDS and hands them to memory ELEMENT by element: lanes then follow consecutive doubles of
// the stencil array -- pairs (sA, sB) with consecutive sB along x are consecutive blocks of row node A, 5 blocks = 45
// doubles = 360 contiguous bytes in 3-D -- instead of every lane adding into its own block, 72 bytes from its
// neighbour's.  The float atomics of this chip run at their full rate for wave instructions that cover 256 contiguous
// bytes or two 128-byte segments and an order of magnitude below it for 64 scattered words (MI355X guide, global float
// atomics): the blocks are this kernel's whole traffic, 140 GB of added bytes per million particles in 3-D.
static constexpr int TAN_NT = 512, TAN_NW = TAN_NT / 64;

// group heads of the I0-sorted particle list: head[g] = first position of group g, ngroups = count
__global__ void k_tangent_groups(int np, const unsigned long long* __restrict__ keys, int* __restrict__ head,
                                 int* __restrict__ ngroups) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= np) return;
  if (s == 0 || keys[s] != keys[s - 1]) head[atomicAdd(ngroups, 1)] = s;
}

__global__ void k_tangent_keys(PView P, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  keys[p] = (unsigned long long)P.I0[p];
  vals[p] = p;
}

template <int ND>
__global__ __launch_bounds__(TAN_NT) void k_tangent_nh_grouped(PView P, GridD g, const MatD* __restrict__ mats, int np,
                                                            const unsigned long long* __restrict__ keys,
                                                            const int* __restrict__ sorted, const int* __restrict__ head,
                                                            const int* __restrict__ ngroups, double* __restrict__ Kst,
                                                            unsigned char* __restrict__ touched,
                                                            int* __restrict__ gstatus, int sym) {
  constexpr int S = TanCfg<ND>::S, MAXM = TanCfg<ND>::MAXM, KN = Lme<ND>::KN;
  __shared__ double gn[TAN_GROUP][MAXM][ND], g1[TAN_GROUP][MAXM][ND], ub[TAN_GROUP][MAXM][ND];
  __shared__ double coef[TAN_GROUP][3];  // V0*c0, V0*c1, V0*G of each particle (Neo-Hookean)
  // spectral laws (Hencky.c:98-229, Elastoplastic-Tangent-Matrix.c:42-163): eigenvectors (column A) and eigenvalues
  // of b, eigenvalues of the Kirchhoff block, moduli (AA or C_ep), the Kirchhoff block itself, V0
  __shared__ double sp_n[TAN_GROUP][ND * ND], sp_lam[TAN_GROUP][ND], sp_tauv[TAN_GROUP][ND], sp_C[TAN_GROUP][ND * ND],
      sp_tau[TAN_GROUP][ND * ND], sp_V0[TAN_GROUP];
  // spectral laws: the block of a node pair is bilinear in the pushed-forward gradients of its two nodes,
  // K_ij = sum_kl g1_A[k] g1_B[l] D[k][l][i][j] with ONE fourth-order tensor per particle (V0 folded in, built in phase A
  // from the arrays above): phase B then costs d^2 products + d^4 FMAs per pair and particle instead of re-deriving the
  // principal-axes moduli (three divisions among them) for each of the particle's 15 625 pairs
  __shared__ __attribute__((aligned(16))) double spD[TAN_GROUP][(ND * ND * ND * ND + 1) & ~1];  // (rows of an even number of doubles: read as double2)
  __shared__ int law_of[TAN_GROUP];
  __shared__ u64 mem[TAN_GROUP][2];
  __shared__ double tab[TAN_NW][6][5];  // per wave: ex, ey, ez, lx, ly, lz of the particle it is building
  __shared__ double stage_v[TAN_NW][64 * ND * ND];  // phase B: the blocks of the wave's 64 pairs, pair-major
  __shared__ long long stage_b[TAN_NW][64];         // their block index in the stencil array, -1 = nothing to add
  if ((int)blockIdx.x >= *ngroups) return;
  const int first = head[blockIdx.x];
  const unsigned long long key = keys[first];
  int last = first;
  while (last + 1 < np && keys[last + 1] == key) last++;  // groups are small (particles per node)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int I0 = (int)key;
  const int i0 = I0 % g.n[0], j0 = (I0 / g.n[0]) % g.n[1], k0 = I0 / (g.n[0] * g.n[1]);
  for (int chunk = first; chunk <= last; chunk += TAN_GROUP) {
    const int nb = min(TAN_GROUP, last + 1 - chunk);
    __syncthreads();
    // phase A: wave w builds the tables of particle w of the chunk
    for (int jj = wave; jj < nb; jj += TAN_NW) {
      const int p = sorted[chunk + jj];
      Lme<ND> c;
      double lam[ND], beta;
      const bool ok_lists = load_lme<ND>(P, g, p, c, lam, beta);
      const MatD m = mats[P.mat[p]];
      double Zinv = 0.0, r[ND], J[ND * ND], Jm1[ND * ND], DF[ND * ND], DFm1[ND * ND], Fn[ND * ND], bn[ND * ND], zz;
      bool ok = ok_lists;
      const bool spectral = m.type != NLPS_MAT_NEO_HOOKEAN;
      double nv[ND * ND];  // eigenvectors of b (spectral laws)
      if (ok) {
        lme_moments_h<ND>(c, Zinv, r, J);
        load_block<ND>(P, F_DF, p, DF, zz);
        load_block<ND>(P, fFN(P), p, Fn, zz);
        if (!inverse<ND>(Jm1, J) || !inverse<ND>(DFm1, DF)) {
          if (lane == 0) {
            atomicOr(&P.status[p], ST_NEWTON);
            atomicOr(gstatus, ST_NEWTON);
          }
          ok = false;
        }
      }
      if (ok && !spectral) left_cauchy_green<ND>(bn, Fn);
      if (ok && spectral) {
        double bmat[ND * ND], tau[ND * ND], lamb[3] = {0, 0, 0}, tauv[3] = {0, 0, 0}, tv[ND * ND], Cm[ND * ND];
        load_block<ND>(P, F_TAU, p, tau, zz);
        if (m.type == NLPS_MAT_HENCKY) {
          double F1[ND * ND];
          load_block<ND>(P, fFN1(P), p, F1, zz);
          left_cauchy_green<ND>(bmat, F1);
#pragma unroll
          for (int i = 0; i < ND; i++)
#pragma unroll
            for (int j = 0; j < ND; j++) Cm[i * ND + j] = m.lame + (i == j ? 2 * m.G : 0.0);
        } else {
          load_block<ND>(P, fBEN1(P), p, bmat, zz);
#pragma unroll
          for (int q = 0; q < ND * ND; q++) Cm[q] = PF(P, F_CEP + q, p);
        }
        sym_eigen<ND>(lamb, nv, bmat);
        sym_eigen<ND>(tauv, tv, tau);
        if (lane == 0) {
#pragma unroll
          for (int q = 0; q < ND * ND; q++) {
            sp_n[jj][q] = nv[q];
            sp_C[jj][q] = Cm[q];
            sp_tau[jj][q] = tau[q];
          }
#pragma unroll
          for (int a = 0; a < ND; a++) {
            sp_lam[jj][a] = lamb[a];
            sp_tauv[jj][a] = tauv[a];
          }
        }
      }
      if (lane == 0) {
        law_of[jj] = m.type;
        sp_V0[jj] = ok ? tangent_vol(P, p) : 0.0;
      }
#pragma unroll
      for (int i = 0; i < 5; i++)
        if (lane == i) {
          tab[wave][0][i] = c.ex[i];
          tab[wave][1][i] = c.ey[i];
          tab[wave][2][i] = (ND == 3) ? c.ez[i % KN] : 1.0;
          tab[wave][3][i] = c.lx[i];
          tab[wave][4][i] = c.ly[i];
          tab[wave][5][i] = (ND == 3) ? c.lz[i % KN] : 0.0;
        }
      if (lane == 0) {
        const double Jp = PF(P, F_JN1, p), sqrJ = Jp * Jp, V0 = ok ? tangent_vol(P, p) : 0.0;
        coef[jj][0] = V0 * (m.lame * sqrJ);                       // Neo-Hookean.c:107-110
        coef[jj][1] = V0 * (m.G - 0.5 * m.lame * (sqrJ - 1));
        coef[jj][2] = V0 * m.G;
        mem[jj][0] = ok ? c.mlo : 0ull;
        mem[jj][1] = ok ? c.mhi : 0ull;
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      if (spectral) {
        // D[k][l][i][j] = V0 ( -delta_kj tau_il + sum_AB C_AB N_kA N_lB N_iA N_jB
        //                      + sum_{A != B, |lam_B - lam_A| > 1e-14} hq_AB (lam_A N_lA N_kB N_iA N_jB + lam_B N_kB N_lB N_iA N_jA) ),
        // hq_AB = (tau_B - tau_A) / (2 (lam_B - lam_A)), N_iA = component i of eigenvector A of b: the expansion of
        // W_AB + delta_AB D_A of the comment in phase B with a = N^T g1_A, b = N^T g1_B, minus (tau g1_B) (x) g1_A
        constexpr int E4 = ND * ND * ND * ND;
        for (int e4 = lane; e4 < E4; e4 += 64) {
          const int j = e4 % ND, i = (e4 / ND) % ND, l = (e4 / (ND * ND)) % ND, k = e4 / (ND * ND * ND);
          double v = (k == j) ? -sp_tau[jj][i * ND + l] : 0.0;
#pragma unroll
          for (int A = 0; A < ND; A++) {
            const double NiA = sp_n[jj][A + i * ND], NkA = sp_n[jj][A + k * ND], NlA = sp_n[jj][A + l * ND], NjA = sp_n[jj][A + j * ND];
#pragma unroll
            for (int B = 0; B < ND; B++) {
              const double NjB = sp_n[jj][B + j * ND], NlB = sp_n[jj][B + l * ND], NkB = sp_n[jj][B + k * ND];
              v = fma(sp_C[jj][A * ND + B] * NkA * NlB, NiA * NjB, v);
              if (A != B) {
                const double dl = sp_lam[jj][B] - sp_lam[jj][A];
                if (fabs(dl) > 1E-14) {
                  const double hq = 0.5 * ((sp_tauv[jj][B] - sp_tauv[jj][A]) / dl);
                  v = fma(hq * sp_lam[jj][A] * NlA * NkB, NiA * NjB, v);
                  v = fma(hq * sp_lam[jj][B] * NkB * NlB, NiA * NjA, v);
                }
              }
            }
          }
          spD[jj][e4] = ok ? sp_V0[jj] * v : 0.0;
        }
      }
      for (int s = lane; s < MAXM; s += 64) {
        const bool on = ok && c.on(s);
        const int i = s % 5, j = (s / 5) % 5, k = s / 25;
        const double l[3] = {tab[wave][3][i], tab[wave][4][j], tab[wave][5][k]};
        const double pa = on ? tab[wave][0][i] * tab[wave][1][j] * tab[wave][2][k] * Zinv : 0.0;
        double ga[ND];
#pragma unroll
        for (int a = 0; a < ND; a++) {
          double v = 0.0;
#pragma unroll
          for (int b2 = 0; b2 < ND; b2++) v = fma(on ? Jm1[a * ND + b2] : 0.0, l[b2], v);
          ga[a] = -pa * v;
        }
#pragma unroll
        for (int a = 0; a < ND; a++) {
          double v1 = 0.0, vb = 0.0;
#pragma unroll
          for (int b2 = 0; b2 < ND; b2++) {
            v1 = fma(on ? DFm1[b2 * ND + a] : 0.0, ga[b2], v1);
            vb = fma(on ? bn[a * ND + b2] : 0.0, ga[b2], vb);
          }
          gn[jj][s][a] = on ? ga[a] : 0.0;
          g1[jj][s][a] = on ? v1 : 0.0;
          ub[jj][s][a] = on ? vb : 0.0;
        }
        if (spectral) {  // gn <- projections of the pushed-forward gradient on the eigenvectors of b
          double pr[ND];
#pragma unroll
          for (int A = 0; A < ND; A++) {
            double v = 0.0;
#pragma unroll
            for (int i2 = 0; i2 < ND; i2++) v = fma(g1[jj][s][i2], ok ? nv[A + i2 * ND] : 0.0, v);
            pr[A] = v;
          }
#pragma unroll
          for (int A = 0; A < ND; A++) gn[jj][s][A] = on ? pr[A] : 0.0;
        }
      }
    }
    __syncthreads();
    // phase B: the threads share the (sA, sB) pairs of the stencil; blocks are summed over the chunk
    // (every lane of a wave makes the same trips: the staging below is wave-cooperative)
    // sym (a cloud of Neo-Hookean particles only: block (B, A) is the transpose of block (A, B), Neo-Hookean.c:89-141):
    // only the pairs with sB >= sA are assembled -- in the stencil code s = i + 5 j + 25 k that is exactly "offset B - A
    // lexicographically >= 0", the upper half of every row of the stencil array -- and k_tangent_emit reads the lower
    // half from the mirrored block.  Half the added bytes, which are what this kernel's time is.  The triangle is
    // walked as H = MAXM / 2 double rows (row r with its MAXM - r pairs + row MAXM - 1 - r with its r + 1) of MAXM + 1
    // pairs and the middle row, so that consecutive lanes still follow consecutive sB.
    constexpr int H = MAXM / 2, NPAIR_SYM = H * (MAXM + 1) + H + 1;
    const int npair = sym ? NPAIR_SYM : MAXM * MAXM;
    for (int q0 = wave * 64; q0 < npair; q0 += TAN_NT) {
      const int q = q0 + lane;
      const bool valid = q < npair;
      int sA = 0, sB = 0;
      if (valid && !sym) {
        sA = q / MAXM;
        sB = q - sA * MAXM;
      } else if (valid) {
        if (q < H * (MAXM + 1)) {
          const int r = q / (MAXM + 1), c = q - r * (MAXM + 1);
          if (c < MAXM - r) {
            sA = r;
            sB = r + c;
          } else {
            sA = MAXM - 1 - r;
            sB = sA + (c - (MAXM - r));
          }
        } else {
          sA = H;
          sB = H + (q - H * (MAXM + 1));
        }
      }
      double acc[ND * ND];
#pragma unroll
      for (int e = 0; e < ND * ND; e++) acc[e] = 0.0;
      bool any = false;
      for (int jj = 0; valid && jj < nb; jj++) {
        const bool inA = sA < 64 ? (mem[jj][0] >> sA) & 1ull : (mem[jj][1] >> (sA - 64)) & 1ull;
        const bool inB = sB < 64 ? (mem[jj][0] >> sB) & 1ull : (mem[jj][1] >> (sB - 64)) & 1ull;
        if (!(inA && inB)) continue;
        any = true;
        if (law_of[jj] != NLPS_MAT_NEO_HOOKEAN) {
          // K = V0 [ sum_AB (W_AB + delta_AB D_A) n_A (x) n_B - (tau g1_B) (x) g1_A ],  a = proj(sA), b = proj(sB):
          // W_AB = C_AB a_A b_B + [A != B] 1/2 q_AB lam_A b_A a_B ,  D_A = sum_{B != A} 1/2 q_AB lam_B a_B b_B ,
          // q_AB = (tau_B - tau_A) / (lam_B - lam_A) where |lam_B - lam_A| > 1e-14
          // -- as the contraction of the particle's tensor spD (phase A) with g1_A (x) g1_B
          // (the tensor is the same for every lane: read as double2, ds_read_b128 broadcasts -- pairs of ds_read2_b64, which
          // the back end forms from scalar reads, take twice the LDS cycles)
          constexpr int E = ND * ND, E4 = E * E;
          double o[E];
#pragma unroll
          for (int k = 0; k < ND; k++)
#pragma unroll
            for (int l = 0; l < ND; l++) o[k * ND + l] = g1[jj][sA][k] * g1[jj][sB][l];
          const double2* D2 = reinterpret_cast<const double2*>(&spD[jj][0]);
          // all reads first, then the arithmetic (the scheduler otherwise alternates them with a wait each: at two waves
          // per SIMD the LDS latency of ~40 such waits per pair and particle was most of this path's time)
          double2 d[(E4 + 1) / 2];
#pragma unroll
          for (int t = 0; t < (E4 + 1) / 2; t++) d[t] = D2[t];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int t = 0; t < E4 / 2; t++) {
            acc[(2 * t) % E] = fma(o[(2 * t) / E], d[t].x, acc[(2 * t) % E]);
            acc[(2 * t + 1) % E] = fma(o[(2 * t + 1) / E], d[t].y, acc[(2 * t + 1) % E]);
          }
          if (E4 & 1) acc[(E4 - 1) % E] = fma(o[(E4 - 1) / E], d[E4 / 2].x, acc[(E4 - 1) % E]);
          continue;
        }
        double len0 = 0.0;
#pragma unroll
        for (int a = 0; a < ND; a++) len0 = fma(gn[jj][sB][a], ub[jj][sA][a], len0);
        const double k0c = coef[jj][0], k1c = coef[jj][1], kG = coef[jj][2] * len0;
#pragma unroll
        for (int i = 0; i < ND; i++)
#pragma unroll
          for (int j = 0; j < ND; j++)
            acc[i * ND + j] += k0c * g1[jj][sA][i] * g1[jj][sB][j] + (i == j ? kG : 0.0) + k1c * g1[jj][sA][j] * g1[jj][sB][i];
      }
      const int ia = sA % 5, ja = (sA / 5) % 5, ka = sA / 25;
      const int nodeA = (i0 + ia - 2) + g.n[0] * ((j0 + ja - 2) + (ND == 3 ? g.n[1] * (k0 + ka - 2) : 0));
      const int oAB = tangent_offset_index<ND>(sA, sB);
      const long long blk = any ? (long long)nodeA * S + oAB : -1ll;
      if (any) {
        touched[blk] = 1;
        if (sym && sB != sA) {  // the structural visit of the mirrored pair (B, A), whose values the emit kernel derives
          const int ib = sB % 5, jb = (sB / 5) % 5, kb = sB / 25;
          const int nodeB = (i0 + ib - 2) + g.n[0] * ((j0 + jb - 2) + (ND == 3 ? g.n[1] * (k0 + kb - 2) : 0));
          touched[(size_t)nodeB * S + (S - 1 - oAB)] = 1;
        }
      }
      constexpr int E = ND * ND;
#pragma unroll
      for (int e = 0; e < E; e++) stage_v[wave][lane * E + e] = acc[e];
      stage_b[wave][lane] = blk;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int t = 0; t < E; t++) {
        const int m = t * 64 + lane;  // element m of the wave's 64 x E doubles: block m / E, entry m % E
        const long long b = stage_b[wave][m / E];
        const double v = stage_v[wave][m];
        if (b >= 0) atomic_add_f64(Kst + (size_t)b * E + (m % E), v);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();  // (the staging rows are rewritten by the next trip)
    }
  }
}

// number of structurally visited blocks of every row node (both ends active by construction); one wave per row node
template <int ND>
__global__ __launch_bounds__(256) void k_tangent_count(int nnodes, const unsigned char* __restrict__ touched, int* __restrict__ cnt) {
  constexpr int S = TanCfg<ND>::S;
  const int lane = threadIdx.x & 63;
  const int A = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (A >= nnodes) return;
  const unsigned char* t = touched + (size_t)A * S;
  int c = 0;
  for (int s0 = 0; s0 < S; s0 += 64) c += (int)__popcll(__ballot(s0 + lane < S && t[s0 + lane]));
  if (lane == 0) cnt[A] = c;
}

// __create_sparsity_pattern, U-Newmark-beta.c:1568-1632: visited columns per dof row (masked numbering)
template <int ND>
__global__ void k_tangent_pattern(int nnodes, const int* __restrict__ cnt, const int* __restrict__ n2m,
                                  int* __restrict__ pattern) {
  const int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= nnodes || n2m[A] < 0) return;
#pragma unroll
  for (int i = 0; i < ND; i++) pattern[n2m[A] * ND + i] = ND * cnt[A];
}

// COO triplets in masked dof numbering; entry order: row node (grid order), stencil offset, i, j.  One wave per row
// node: it lists the node's visited offsets in LDS (ballots keep them ascending), then its lanes follow the node's
// triplets element by element -- reads of the stencil array and writes of rows / cols / vals are contiguous (one thread
// per row node walking its 729 offsets wrote 36 / 72 bytes apart from its neighbours: 60 GB/s).
template <int ND>
__global__ __launch_bounds__(256) void k_tangent_emit(int nnodes, GridD g, const unsigned char* __restrict__ touched,
                                                      const double* __restrict__ Kst, const int* __restrict__ offs,
                                                      const int* __restrict__ n2m, const int* __restrict__ d2m, double alpha_1,
                                                      const double* __restrict__ mass, int* __restrict__ rows,
                                                      int* __restrict__ cols, double* __restrict__ vals, int sym) {
  constexpr int S = TanCfg<ND>::S, E = ND * ND;
  __shared__ unsigned short list[4][S];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int A = blockIdx.x * 4 + wave;
  if (A >= nnodes) return;  // (whole waves: the kernel has no workgroup barrier)
  const unsigned char* t = touched + (size_t)A * S;
  int cnt = 0;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool on = s < S && t[s];
    const u64 bal = __ballot(on);
    if (on) list[wave][cnt + (int)__popcll(bal & ((1ull << lane) - 1ull))] = (unsigned short)s;
    cnt += (int)__popcll(bal);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const int mA = n2m[A];
  const size_t e0 = (size_t)offs[A] * E;
  for (int m = lane; m < cnt * E; m += 64) {
    const int bp = m / E, e = m - bp * E, s = list[wave][bp];
    const int dx = s % 9 - 4, dy = (s / 9) % 9 - 4, dz = (ND == 3) ? s / 81 - 4 : 0;
    const int B = A + dx + g.n[0] * (dy + g.n[1] * dz);
    const int mB = n2m[B];
    const int i = e / ND, j = e - i * ND;
    const int ra = mA * ND + i, cb = mB * ND + j;
    // sym: the lower half of a row (offset before the centre) was not assembled: block (A, B) = block (B, A)^T
    double v = (sym && s < S / 2) ? Kst[((size_t)B * S + (S - 1 - s)) * E + (j * ND + i)] : Kst[((size_t)A * S + s) * E + e];
    if (ra == cb && mass) v += alpha_1 * mass[ra];  // :1797-1807
    if (d2m && (d2m[ra] == -1 || d2m[cb] == -1)) v = (ra == cb) ? 1.0 : 0.0;  // MatZeroRowsColumnsIS, :1822
    rows[e0 + m] = ra;
    cols[e0 + m] = cb;
    vals[e0 + m] = v;
  }
}
