// nlps_gpu.hip — MI355X (gfx950) kernels and C-ABI host code for NL-PartSol's particle<->grid +
// stress-update hot path.  See include/nlps_gpu.h for the boundary and DESIGN.md for the layout.
//
// Layout in HBM
//   particles : SoA, one contiguous run of `npad` doubles per scalar component (enum below), physically
//               sorted (tile of the closest node I0, corner type, I0) at upload and by the periodic device
//               re-sort; lane p of a wave reads consecutive doubles of every component => coalesced
//               512-B wave loads.
//   nodes     : AoS per node in GRID numbering (x fastest, slab axis slowest): nm[node][1+d] =
//               {mass, momentum}, dU[node][d], force[node][d], accel[node][d]; a slab halo is one
//               contiguous byte range.  active[node], seed[node], fixed[node][d] are bytes.
// One lane = one particle.  All LME quantities (15 separable exp factors, Z, r, J, J^-1, DF, tau)
// live in that lane's registers; no MFMA (<=3x3 contractions).  The particle<->grid kernels work per tile of
// closest nodes with the tile's node window in LDS (nlps_tile_kernels.hpp); this file holds the search, the
// nodal kernels, the per-particle level-B kernels and the C-ABI host code.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <thread>
#include <hipcub/hipcub.hpp>

#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is dlopen()ed when a communicator is attached

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <map>
#include <string>
#include <vector>

#include "../../include/nlps_gpu.h"
#include "nlps_device.hpp"
#include "nlps_tables.hpp"

#define LAUNCH_ND(kern2, kern3, grid, ...)                                                     \
  do {                                                                                         \
    if (h->nd == 2) hipLaunchKernelGGL(kern2, dim3(grid), dim3(BLK), 0, h->stream, __VA_ARGS__); \
    else hipLaunchKernelGGL(kern3, dim3(grid), dim3(BLK), 0, h->stream, __VA_ARGS__);          \
  } while (0)

using namespace nlps;

// ------------------------------------------------------------------------------------------------
// particle component table
// ------------------------------------------------------------------------------------------------
enum {
  F_X = 0, F_DIS = 3, F_VEL = 6, F_ACC = 9, F_DDIS = 12,
  F_FN = 15, F_FN1 = 24, F_DF = 33, F_TAU = 42, F_BEN = 51, F_BEN1 = 60,
  F_JN = 69, F_JN1 = 70, F_RHO = 71, F_MASS = 72, F_VOL0 = 73, F_W = 74,
  F_KN = 75, F_KN1 = 76, F_EN = 77, F_EN1 = 78, F_LAM = 79, F_BETA = 82,
  F_LAMP = 83,  // lambda of the previous step (Newton start extrapolation)
  F_BACK = 86,  // principal back stress (Von-Mises), 3 components
  // rho * J, the invariant of the explicit density update rho <- rho / det DF (U-Verlet.c:630-632: det DF = J_n+1 / J_n):
  // the fused step leaves rho alone and k_copy_n_to_n1 returns rho = (rho J) / J_n when somebody asks (k_init_rhoj)
  F_RHOJ = 89,
  F_CEP = 90,   // C_ep[ndim*ndim] (Drucker-Prager / Von-Mises tangent moduli, implicit driver only)
  F_DTFN = 99, F_DTFN1 = 108, F_DTDF = 117,  // rate tensors (level-B compatibility with dU_dt)
  F_DMG = 126, F_DMG1 = 127,                 // Damage_n, Damage_n1 (eigenerosion / eigensoftening, level B only)
  F_STRF = 128, F_STRF1 = 129,               // Strain_f_n, Strain_f_n1 (eigensoftening)
  // eigenerosion: position and closest node of every particle when the epsilon-neighbourhoods were initialised
  // (compute_Beps__Constitutive__(.., true), U-Newmark-beta.c:182-183): what the FROZEN list of a particle that has
  // not moved since is made of (Beps.c:30-36).  The closest node travels as a double like any other field.
  F_X0 = 130, F_I00 = 133,
  NFD = 134
};

struct PView {
  int np;
  size_t npad;
  double* d;  // [NFD][npad]
  int* I0;
  int* I0n;  // closest node for the position the last explicit step wrote (k5_tile's search ahead); = I0 otherwise.  I0
             // itself stays the node of the last search until the next one adopts I0n (downloads, migration see I0)
  int* mat;
  int* nn;
  int* status;
  u64* mlo;
  u64* mhi;
  int* tile;  // tile of I0 (per-step binning)
  int* rank;  // arrival rank inside the tile
  int flip;   // 1: the n / n+1 slots of F and b_e are swapped (the explicit step rolls them by renaming)
  int erosion;  // Driver_EigenErosion or Driver_EigenSoftening: the damage hooks of the level-B stages are on
  int softening;  // ... with Eigensoftening__Constitutive__ (Driver_EigenSoftening and not Driver_EigenErosion)
};
#define PF(P, f, p) ((P).d[(size_t)(f) * (P).npad + (size_t)(p)])
// first component of F_n, F_n+1, b_e,n, b_e,n+1 under the current renaming
__host__ __device__ __forceinline__ int fFN(const PView& P) { return P.flip ? F_FN1 : F_FN; }
__host__ __device__ __forceinline__ int fFN1(const PView& P) { return P.flip ? F_FN : F_FN1; }
__host__ __device__ __forceinline__ int fBEN(const PView& P) { return P.flip ? F_BEN1 : F_BEN; }
__host__ __device__ __forceinline__ int fBEN1(const PView& P) { return P.flip ? F_BEN : F_BEN1; }

struct NView {
  unsigned char* active;  // [nnodes]
  unsigned char* seed;    // [nnodes] 1 where some particle has this node as I0 (dilate_node turns it into `active`)
  const double* h_avg;    // [nnodes]
  // [nnodes] {beta, T2, Ra, -}: beta__LME__ of a particle whose closest node this is (gamma / h_avg^2, LME.c:177-185),
  // the cut-off radius Ra of the NEXT list built with that beta (LME.c:1052) and its exact squared form T2 =
  // max{t : fl(sqrt(t)) <= Ra}; functions of the node alone, made once at create by the device functions K2 would
  // otherwise call per particle and step (two divisions and three or more square roots)
  const double4* beta_t2;
  double* nm;             // [nnodes][1+ND]
  double* dU;             // [nnodes][ND]
  double* force;          // [nnodes][ND]
  double* accel;          // [nnodes][ND]
  double* reaction;       // [nnodes][ND]
  unsigned char* fixed;   // [nnodes][ND]
};

#define ST_NEWTON 1
#define ST_CONNECT 2
#define ST_JACOBIAN 4
#define ST_CONSTITUTIVE 8
#define ST_HALO 16

static constexpr int BLK = 256;

struct TileCnt {
  int nt[3];
  int* count;       // [ntiles] particles per tile, nullptr = no binning
  int tile0, ntw;   // tiles of the node window (nlps_gpu_set_node_window); default: all
  int win_lo, win_hi, plane;  // node layers (slowest axis) of the window, nodes per layer
  int* gstatus;
  // adaptive re-sort (nlps_gpu_set_adaptive_resort), nullptr = off: home[slot] = tile of the particle in that memory
  // slot at the first search after the last re-sort (rehome != 0: this search records it); foreign = 64 counters,
  // 128 B apart, of the particles found in another tile than their slot's home
  int* home;
  int rehome;
  int* foreign;
  // canonical tile lists without a per-tile sort (k_fill_order): node_cnt[node] counts the particles of every closest
  // node, nrank[p] = the rank of p inside its node (arrival order of the atomic); nullptr = off
  int* node_cnt;
  int* nrank;
  // defer != 0 (the search that rides on k5_tile, canonical lists on): only COUNT here -- the tile and node counters by
  // atomics whose result nobody waits for -- and leave the ranks (positions inside the tile list and inside the node) to
  // k_fill_orders of the next step, which takes them from cursors at full occupancy.  K5 runs three waves per SIMD: a
  // returning atomic there is a stall nothing covers (+20 us of K5's 110 at 1 M particles).
  int defer;
};
template <int ND>
struct TileCfg;
template <int ND>
__device__ __forceinline__ int tile_of_node(const GridD& g, const int* nt, int I0);

__device__ __forceinline__ void atomic_add_f64(double* addr, double v) { unsafeAtomicAdd(addr, v); }

// Bins a particle to the tile of its I0.  Wave-aggregated: one returning atomic per (wave, tile); the
// lanes of a wave that share a tile get CONSECUTIVE slots in lane order, so order[] keeps runs of 64
// memory-consecutive particles together (coalesced loads in the tile kernels).  Every lane of the wave
// must call this (invalid lanes pass valid = false).
template <int ND>
__device__ __forceinline__ void bin_particle(const PView& P, const GridD& g, const TileCnt& tc, int p, int I0,
                                             bool valid) {
  if (!tc.count) return;
  int t = valid ? tile_of_node<ND>(g, tc.nt, I0) : -1;
  if (valid) {  // the 5^d stencil of I0 must stay inside this rank's node window
    const int layer = I0 / tc.plane, nl = g.n[ND - 1];
    const int s_lo = layer - 2 < 0 ? 0 : layer - 2, s_hi = layer + 2 > nl - 1 ? nl - 1 : layer + 2;
    if (s_lo < tc.win_lo || s_hi > tc.win_hi) {
      // flagged AND left out of the lists: its stencil holds nodes nothing of this rank resets or exchanges (the per-node
      // counters of the canonical lists among them), so no kernel may take it; it keeps its state, the re-sort keeps it
      // (k_append_unlisted), a migration hands it on
      atomicOr(&P.status[p], ST_HALO);
      atomicOr(tc.gstatus, ST_HALO);
      t = -1;
      valid = false;
    }
    if (valid && (t < tc.tile0 || t >= tc.tile0 + tc.ntw)) {  // not binned: its tile is never launched
      t = -1;
      valid = false;
    }
  }
  const int lane = threadIdx.x & 63;
  if (tc.defer) {  // count only (see TileCnt::defer)
    bool todo_l = valid;
    while (true) {
      const u64 todo = __ballot(todo_l);
      if (!todo) break;
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const int t0 = __shfl(t, leader);
      const bool mine = todo_l && (t == t0);
      const u64 same = __ballot(mine);
      if (lane == leader) atomicAdd(&tc.count[t0], (int)__popcll(same));  // (no result wanted: no wait)
      if (mine) todo_l = false;
    }
    if (p < P.np) {
      P.tile[p] = t;
      if (tc.node_cnt && valid) atomicAdd(&tc.node_cnt[I0], 1);
    }
  } else {
  // groups of lanes that share a tile: leader lane, group size and the lane's place in its group, by ballots alone ...
  int rank = 0;
  {
    int my_leader = lane, my_off = 0, my_cnt = 0;
    bool todo_l = valid;
    while (true) {
      const u64 todo = __ballot(todo_l);
      if (!todo) break;
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const int t0 = __shfl(t, leader);
      const bool mine = todo_l && (t == t0);
      const u64 same = __ballot(mine);
      if (mine) {
        my_leader = leader;
        my_off = (int)__popcll(same & ((1ull << lane) - 1ull));
        my_cnt = (int)__popcll(same);
        todo_l = false;
      }
    }
    // ... then ONE returning atomic per group, all groups of the wave in the same instruction: one round trip to the
    // counters however many tiles the wave's particles are in (a loop with the atomic inside paid one per tile: K5 of a
    // stirred cloud 0.226 ms, of which 0.07 waiting here)
    int base = 0;
    if (valid && lane == my_leader) base = atomicAdd(&tc.count[t], my_cnt);
    base = __shfl(base, my_leader);
    if (valid) rank = base + my_off;
  }
  if (p < P.np) {
    P.tile[p] = t;  // -1: not binned (failed element search or outside the node window)
    P.rank[p] = rank;
    // (the lanes of a wave mostly hold distinct closest nodes -- the memory order is the canonical one -- so these
    // atomics spread over the node array)
    if (tc.node_cnt) tc.nrank[p] = valid ? atomicAdd(&tc.node_cnt[I0], 1) : 0;
  }
  }
  if (tc.home) {
    bool away = false;
    if (p < P.np) {
      if (tc.rehome) tc.home[p] = t;
      else away = t != tc.home[p];
    }
    const u64 m = __ballot(away);
    if (lane == 0 && m) atomicAdd(&tc.foreign[32 * ((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & 63)], (int)__popcll(m));
  }
}

template <int ND>
__device__ __forceinline__ int class3_of(const GridD& g, const int* ijk) {
  int c = 0, mul = 1;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    int ca = (a < ND) ? (ijk[a] == 0 ? 0 : (ijk[a] == g.n[a] - 1 ? 2 : 1)) : 1;
    c += ca * mul;
    mul *= 3;
  }
  return c;
}

// ------------------------------------------------------------------------------------------------
// S1a: closest-node update (LME.c:913-945, Nodes-Tools.c:476-538) + 1-ring activation (LME.c:949-960)
// ------------------------------------------------------------------------------------------------
// One launch instead of five hipMemsetAsync (each of which costs one or two 5-us fill kernels): resets the
// search seeds and tile counters and, for the fused explicit step, the nodal accumulators of the node window.
template <int ND>
__global__ void k_step_clear(int n0, int nnodes, NView N, int* __restrict__ tile_count, int ntiles, int nodal,
                             int* __restrict__ node_cnt, int bins) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (bins && t < ntiles) tile_count[t] = 0;
  if (t >= nnodes) return;
  const size_t A = (size_t)n0 + t;
  if (bins) {  // (not when the last step's k5_tile has done this step's search: its seeds and counters are the input)
    N.seed[A] = 0;  // Shape-Functions.c:38-46
    if (node_cnt) node_cnt[A] = 0;
  }
  if (nodal) {
#pragma unroll
    for (int a = 0; a < 1 + ND; a++) N.nm[A * (1 + ND) + a] = 0.0;
#pragma unroll
    for (int a = 0; a < ND; a++) {
      N.force[A * ND + a] = 0.0;
      N.fixed[A * ND + a] = 0;
    }
  }
}

// 1-ring activation (LME.c:949-960): the particles only mark their I0 (one byte store each instead of 3^d
// scattered ones); node n is active iff some I0 lies in its 1-ring.
template <int ND>
__device__ __forceinline__ void dilate_node(int A, const GridD& g, const NView& N) {
  const int i0 = A % g.n[0], j0 = (A / g.n[0]) % g.n[1], k0 = A / (g.n[0] * g.n[1]);
  unsigned any = 0u;
#pragma unroll
  for (int dk = (ND == 3 ? -1 : 0); dk <= (ND == 3 ? 1 : 0); dk++)
#pragma unroll
    for (int dj = -1; dj <= 1; dj++)
#pragma unroll
      for (int di = -1; di <= 1; di++) {
        const int i = i0 + di, j = j0 + dj, k = k0 + dk;
        const bool ok = i >= 0 && i < g.n[0] && j >= 0 && j < g.n[1] && (ND == 2 || (k >= 0 && k < g.n[2]));
        if (ok) any |= N.seed[i + g.n[0] * (j + g.n[1] * k)];
      }
  N.active[A] = any ? 1 : 0;
}

// The closest-node update of local_search__LME__ (LME.c:924-930): argmin of the distance over the 1-ring of the
// PREVIOUS closest node, ties by chain position (get_closest_node__MeshTools__, Nodes-Tools.c:476-538).
template <int ND>
__device__ __forceinline__ int closest_node_update(const GridD& g, const uint8_t* __restrict__ rank1, const double* x, int I0) {
  int ijk[3] = {I0 % g.n[0], (I0 / g.n[0]) % g.n[1], I0 / (g.n[0] * g.n[1])};
  const uint8_t* rk = rank1 + 27 * class3_of<ND>(g, ijk);
  // Of the 3^d candidates only those that can be the minimum are evaluated.  Along an axis where the particle is
  // clearly closer to the old node's plane than to either neighbouring plane (|d| <= 0.49 h) a step to a neighbouring
  // plane adds at least 0.02 h^2 to the squared distance -- orders of magnitude beyond rounding -- and along any axis
  // the step AWAY from the particle adds more than h^2: such candidates lose in the reference's loop as well.  What is
  // left is the old node and, per axis with |d| > 0.49 h, its neighbour on the particle's side: at most 2^d nodes,
  // usually one.  Their squared distances are summed like point_distance__MeshTools__ (Nodes-Tools.c:397-420:
  // DIST += pow(d,2) in axis order), the winner is the reference's: lexicographic minimum of (sqrt(D), chain position)
  // = first strict minimum of the chain walk (Nodes-Tools.c:476-538).
  int sgn[3] = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < ND; a++) {
    const double d = x[a] - (g.o[a] + g.h * (double)ijk[a]);
    if (fabs(d) > 0.49 * g.h) sgn[a] = d > 0.0 ? 1 : -1;
  }
  double best = 0.0;
  int bestrank = 256, bi = ijk[0], bj = ijk[1], bk = ijk[2];
#pragma unroll
  for (int q = 0; q < (1 << ND); q++) {
    const int di = (q & 1) ? sgn[0] : 0, dj = (q & 2) ? sgn[1] : 0, dk = (ND == 3 && (q & 4)) ? sgn[2] : 0;
    // a combination that asks for a step along an axis that has none repeats another one
    const bool dup = ((q & 1) && sgn[0] == 0) || ((q & 2) && sgn[1] == 0) || (ND == 3 && (q & 4) && sgn[2] == 0);
    const int i = ijk[0] + di, j = ijk[1] + dj, k = ijk[2] + dk;
    const bool ok = !dup && i >= 0 && i < g.n[0] && j >= 0 && j < g.n[1] && (ND == 2 || (k >= 0 && k < g.n[2]));
    if (ok) {
      double Dc = 0.0, t;
      t = x[0] - (g.o[0] + g.h * (double)i);
      Dc += t * t;
      t = x[1] - (g.o[1] + g.h * (double)j);
      Dc += t * t;
      if (ND == 3) {
        t = x[ND - 1] - (g.o[2] + g.h * (double)k);
        Dc += t * t;
      }
      const double d = sqrt(Dc);
      const int rank = rk[(di + 1) + 3 * (dj + 1) + 9 * (dk + 1)];
      if (bestrank == 256 || d < best || (d == best && rank < bestrank)) {
        best = d;
        bestrank = rank;
        bi = i;
        bj = j;
        bk = k;
      }
    }
  }
  return bi + g.n[0] * (bj + g.n[1] * bk);
}

// update != 0: the search proper.  update == 0: seeds and bins only, with the closest nodes as they are -- the last
// kernel of the fused explicit step (k5_tile) has already updated them for the positions it wrote, and a second update
// would start from the new node's 1-ring instead of the old one's (LME.c:927-929).
template <int ND>
__global__ __launch_bounds__(BLK) void k_search(PView P, GridD g, NView N, const uint8_t* __restrict__ rank1, TileCnt tc,
                                                int update) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  const bool valid = p < P.np;
  int I0 = 0;
  if (valid) {
    I0 = P.I0[p];
    if (update) {
      double x[ND], aux = 0.0;
#pragma unroll
      for (int a = 0; a < ND; a++) {
        x[a] = PF(P, F_X + a, p);
        aux += dsqr(PF(P, F_DIS + a, p));
      }
      if (sqrt(aux) > 0.0) {  // norm__MatrixLib__(dis_p,2) > 0, LME.c:924
        I0 = closest_node_update<ND>(g, rank1, x, I0);
        P.I0[p] = I0;
      }
      P.I0n[p] = I0;
    } else {  // adopt the update the last explicit step made for the position it wrote
      const int In = P.I0n[p];
      if (In != I0) P.I0[p] = In;
      I0 = In;
    }
  }
  bin_particle<ND>(P, g, tc, p, I0, valid);
  // (only a particle that is in a tile list activates nodes: one left out -- outside the node window -- takes no part)
  if (valid && (!tc.count || P.tile[p] >= 0)) N.seed[I0] = 1;
}

// the closest nodes k5_tile found ahead become THE closest nodes (paths whose list kernels do not do it on their way)
__global__ void k_commit_I0(int np, const int* __restrict__ I0n, int* __restrict__ I0) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < np) I0[p] = I0n[p];
}

// initialize__LME__ first loop (LME.c:63-115): element search + closest element node
template <int ND>
__global__ __launch_bounds__(BLK) void k_init_I0(PView P, GridD g, NView N, TileCnt tc) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  bool valid = p < P.np;
  int bestnode = 0;
  if (valid) {
    double x[ND];
    int c[3] = {0, 0, 0};
    bool found = true;
#pragma unroll
    for (int a = 0; a < ND; a++) {
      x[a] = PF(P, F_X + a, p);
      int nc = g.n[a] - 1;
      int ci = (int)floor((x[a] - g.o[a]) / g.h);
      ci = ci < 0 ? 0 : (ci > nc - 1 ? nc - 1 : ci);
      while (ci > 0 && x[a] <= g.o[a] + g.h * (double)ci) ci--;
      while (ci < nc - 1 && x[a] > g.o[a] + g.h * (double)(ci + 1)) ci++;
      if (x[a] < g.o[a] + g.h * (double)ci || x[a] > g.o[a] + g.h * (double)(ci + 1)) found = false;
      c[a] = ci;
    }
    if (!found) {
      atomicOr(&P.status[p], ST_CONNECT);
      atomicOr(tc.gstatus, ST_CONNECT);  // the particle is not binned, so no later kernel would report it
      valid = false;
    } else {
      // connectivity chain = reverse GiD file order (Read-GID-Mesh.c:411-413): rank of corner (a,b,t)
      const int fileQ[4][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}};
      double best = 0.0;
      int bestrank = 256;
#pragma unroll
      for (int t = 0; t < (ND == 3 ? 2 : 1); t++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          int filepos = 4 * t + q;
          int nnod = (ND == 3) ? 8 : 4;
          int rank = nnod - 1 - filepos;
          int i = c[0] + fileQ[q][0], j = c[1] + fileQ[q][1], k = c[2] + t;
          double D = 0.0, tt;
          tt = x[0] - (g.o[0] + g.h * (double)i);
          D += tt * tt;
          tt = x[1] - (g.o[1] + g.h * (double)j);
          D += tt * tt;
          if (ND == 3) {
            tt = x[ND - 1] - (g.o[2] + g.h * (double)k);
            D += tt * tt;
          }
          D = sqrt(D);
          if (bestrank == 256 || D < best || (D == best && rank < bestrank)) {
            best = D;
            bestrank = rank;
            bestnode = i + g.n[0] * (j + g.n[1] * (ND == 3 ? k : 0));
          }
        }
      P.I0[p] = bestnode;
      P.I0n[p] = bestnode;
      N.seed[bestnode] = 1;
    }
  }
  bin_particle<ND>(P, g, tc, p, bestnode, valid);
}

template <int ND>
__device__ __forceinline__ bool load_lme(const PView& P, const GridD& g, int p, Lme<ND>& c, double* lam, double& beta) {
  double x[ND];
#pragma unroll
  for (int a = 0; a < ND; a++) {
    x[a] = PF(P, F_X + a, p);
    lam[a] = PF(P, F_LAM + a, p);
  }
  beta = PF(P, F_BETA, p);
  c.geom(g, x, P.I0[p]);
  c.mlo = P.mlo[p];
  c.mhi = P.mhi[p];
  c.factors(lam, beta, g.h);
  return (c.mlo | c.mhi) != 0ull;
}

// ------------------------------------------------------------------------------------------------
// S3 (+S4): G2P of grad(dU) -> DF, F_n1, J (U-Newmark-beta.c:1064-1160 / U-Verlet.c:530-632), stress
// (Constitutive.c:18-258), P2G of the internal force (U-Newmark-beta.c:1257-1374)
// ------------------------------------------------------------------------------------------------
template <int ND>
__device__ __forceinline__ void load_block(const PView& P, int f0, int p, double* t, double& zz) {
#pragma unroll
  for (int s = 0; s < ND * ND; s++) t[s] = PF(P, f0 + s, p);
  zz = (ND == 2) ? PF(P, f0 + 4, p) : 0.0;
}
template <int ND>
__device__ __forceinline__ void store_block(const PView& P, int f0, int p, const double* t, double zz, bool with_zz) {
#pragma unroll
  for (int s = 0; s < ND * ND; s++) PF(P, f0 + s, p) = t[s];
  if (ND == 2 && with_zz) PF(P, f0 + 4, p) = zz;
}

// LAW = -1: dispatch on the particle's material at run time (mixed clouds); LAW = 0/1/2: the whole
// cloud uses that one law, so only its code (and register footprint) is compiled into the kernel.
// CEP: also keep the Drucker-Prager tangent moduli (only the implicit driver's Jacobian reads them).
// FRIC: the Matsuoka-Nakai / Lade-Duncan code is compiled in (its 5 x 5 Newton system would otherwise set the register
// budget of every other law).  A kernel without it is never launched on a cloud that holds the law: the level-B
// stress kernel exists in both forms, the fused step runs one launch per law for such a cloud (nlps_gpu_create forces
// k3_per_law, nlps_gpu_set_law_launch_mode refuses the dispatch kernel).
// LAZY (the fused explicit step): the hyperelastic laws do not store tau and W -- they are functions of F_n+1 (and J)
// alone, nothing in the step reads them back, and k_copy_n_to_n1 recomputes them with the same code from the same
// stored inputs when a level-B stage or a download asks (80 B per particle less to store in K3).
template <int ND, int LAW = -1, bool CEP = false, bool FRIC = (LAW == NLPS_KLAW_FRICTIONAL), bool LAZY = false>
__device__ __forceinline__ int stress_update(const PView& P, int p, const MatD* __restrict__ mats, const ParamsD& prm,
                                             const double* Fn1, const double* DF, double J, double* tau, int mat_idx = -1) {
  MatD m = mats[mat_idx >= 0 ? mat_idx : P.mat[p]];  // (mat_idx: the caller has read the particle's material index already)
  StressIO<ND> o;
  o.fail = 0;
  o.kappa = 0.0;
  o.eps = 0.0;
  o.cep_keep = false;
  const int law = (LAW >= 0) ? LAW : m.type;
  if (law == NLPS_MAT_NEO_HOOKEAN) {
    law_neo_hookean<ND>(m, Fn1, J, o);
  } else if (law == NLPS_MAT_HENCKY) {
    law_hencky<ND>(m, Fn1, o);
  } else {
    double be[ND * ND], bzz;
    load_block<ND>(P, fBEN(P), p, be, bzz);
    if (law == NLPS_MAT_VON_MISES) {
      o.kappa = PF(P, F_KN, p);
#pragma unroll
      for (int a = 0; a < 3; a++) o.back[a] = PF(P, F_BACK + a, p);
      law_von_mises<ND>(m, prm, DF, be, bzz, PF(P, F_EN, p), o);
#pragma unroll
      for (int a = 0; a < 3; a++) PF(P, F_BACK + a, p) = o.back[a];  // in place, like upstream (Constitutive.c:116)
    } else if (FRIC && law == NLPS_KLAW_FRICTIONAL) {
      law_frictional<ND>(m, prm, DF, be, bzz, PF(P, F_KN, p), PF(P, F_EN, p), o);
    } else {
      law_drucker_prager<ND>(m, prm, DF, be, bzz, PF(P, F_KN, p), PF(P, F_EN, p), o);
    }
    store_block<ND>(P, fBEN1(P), p, o.be, o.be_zz, true);
    PF(P, LAZY ? F_KN : F_KN1, p) = o.kappa;  // fused step: straight to the n slots (read above, by this thread)
    PF(P, LAZY ? F_EN : F_EN1, p) = o.eps;
    if (CEP && !(FRIC && o.cep_keep)) {
#pragma unroll
      for (int q = 0; q < ND * ND; q++) PF(P, F_CEP + q, p) = o.cep[q];
    }
  }
#pragma unroll
  for (int s = 0; s < ND * ND; s++) tau[s] = o.tau[s];
  if (!(LAZY && (law == NLPS_MAT_NEO_HOOKEAN || law == NLPS_MAT_HENCKY))) {
    store_block<ND>(P, F_TAU, p, o.tau, o.tau_zz, true);
    PF(P, F_W, p) = o.W;
  }
  return o.fail ? ST_CONSTITUTIVE : 0;
}

// B[i][m] = sign * V0 * sum_jq tau[i][j] DF^-T[j][q] J^-1[q][m];  f_A = p_A * B l_A
template <int ND>
__device__ __forceinline__ bool force_operator(double* B, const double* tau, const double* DF, const double* Jm1,
                                               double V0, double sign) {
  NLPS_FP_CONTRACT
  double DFt[ND * ND], DFmT[ND * ND], M1[ND * ND];
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int j = 0; j < ND; j++) DFt[i * ND + j] = DF[j * ND + i];
  if (!inverse<ND>(DFmT, DFt)) return false;  // compute_adjunt__TensorLib__, TensorLib.c:829-905
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int q = 0; q < ND; q++) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < ND; j++) s += tau[i * ND + j] * DFmT[j * ND + q];
      M1[i * ND + q] = s;
    }
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int m = 0; m < ND; m++) {
      double s = 0.0;
#pragma unroll
      for (int q = 0; q < ND; q++) s += M1[i * ND + q] * Jm1[q * ND + m];
      B[i * ND + m] = -sign * V0 * s;  // grad p_a = -p_a J^-1 l_a
    }
  return true;
}

// ------------------------------------------------------------------------------------------------
// Eigenerosion: the damage part of __nodal_internal_forces, U-Newmark-beta.c:1313-1331 ->
// compute_damage__Constitutive__ (Constitutive.c:385-435) -> Eigenerosion__Constitutive__ (EigenErosion.c:29-117).
// The epsilon-neighbourhood Beps[p] (Beps.c:16-80: particles q whose closest node lies in the 1-ring of I0_p, within
// Ceps * DeltaX of p) is not stored: the particles are sorted by closest node once per call (first/last = the run of
// every node in `sorted`) and every particle walks the <= 3^d runs around its own closest node.  x_GC does not change
// between the search and the force stage, so this is the list the reference built after its search.  A particle that
// has not moved by more than 1e-6 since the start keeps the list of the initialisation (Beps.c:30-36): it walks the
// same runs of the SNAPSHOT taken before the first search (positions and closest nodes of Initialize_Beps = true,
// fields F_X0 / F_I00, tables first0 / last0 / sorted0) -- the same members, wherever they have moved since.
// ------------------------------------------------------------------------------------------------
__global__ void k_node_ranges(int np, const unsigned long long* __restrict__ keys, int* __restrict__ first,
                              int* __restrict__ last) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= np) return;
  const unsigned long long k = keys[i];
  if (i == 0 || keys[i - 1] != k) first[k] = i;
  if (i == np - 1 || keys[i + 1] != k) last[k] = i + 1;
}

__global__ void k_beps_snapshot(PView P, int ND) {  // Initialize_Beps = true: the configuration the frozen lists belong to
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  for (int a = 0; a < ND; a++) PF(P, F_X0 + a, p) = PF(P, F_X + a, p);
  PF(P, F_I00, p) = (double)P.I0[p];
}
__global__ void k_beps_keys0(PView P, unsigned long long* __restrict__ keys, int* __restrict__ vals) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  keys[p] = (unsigned long long)(long long)PF(P, F_I00, p);
  vals[p] = p;
}

// first0 / last0 / sorted0: the same tables for the closest nodes of the snapshot (nullptr: no frozen lists)
template <int ND>
__global__ __launch_bounds__(BLK) void k_damage(PView P, GridD g, const MatD* __restrict__ mats, const int* __restrict__ first,
                                                const int* __restrict__ last, const int* __restrict__ sorted,
                                                const int* __restrict__ first0, const int* __restrict__ last0,
                                                const int* __restrict__ sorted0, double DeltaX) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= P.np) return;
  constexpr int T = (ND == 2) ? 5 : 9;
  const double Dn = PF(P, F_DMG, p);
  double Dn1 = PF(P, F_DMG1, p);
  double tau[ND * ND], z, w[3] = {0.0, 0.0, 0.0}, v[ND * ND];
  load_block<ND>(P, F_TAU, p, tau, z);
  sym_eigen<ND>(w, v, tau);  // ascending like dsyev: w[0] is the smallest principal Kirchhoff stress
  if (Dn < 1.0 && w[0] > 0.0) {
    const MatD m = mats[P.mat[p]];
    const double eps = m.Ceps * DeltaX;
    // a particle whose total displacement is <= 1e-6 keeps the list it had (Beps.c:30-36): the one of the snapshot,
    // i.e. the particles around its closest node THEN that were within eps of it THEN -- wherever they are now
    double xp[ND], d2p = 0.0;
#pragma unroll
    for (int a = 0; a < ND; a++) d2p += dsqr(PF(P, F_DIS + a, p));
    const bool frozen = first0 && !(sqrt(d2p) > 0.000001);
    const int fx = frozen ? F_X0 : F_X;
    if (frozen) {
      first = first0;
      last = last0;
      sorted = sorted0;
    }
#pragma unroll
    for (int a = 0; a < ND; a++) xp[a] = PF(P, fx + a, p);
    const double V_p = PF(P, F_VOL0, p) * PF(P, F_JN1, p);
    double sum_V = V_p, sum_VW = V_p * PF(P, F_W, p);
    const int I0 = frozen ? (int)PF(P, F_I00, p) : P.I0[p];
    const int i0 = I0 % g.n[0], j0 = (I0 / g.n[0]) % g.n[1], k0 = I0 / (g.n[0] * g.n[1]);
    for (int dk = (ND == 3 ? -1 : 0); dk <= (ND == 3 ? 1 : 0); dk++)
      for (int dj = -1; dj <= 1; dj++)
        for (int di = -1; di <= 1; di++) {
          const int i = i0 + di, j = j0 + dj, k = k0 + dk;
          if (i < 0 || i >= g.n[0] || j < 0 || j >= g.n[1] || (ND == 3 && (k < 0 || k >= g.n[2]))) continue;
          const int A = i + g.n[0] * (j + g.n[1] * k);
          for (int s = first[A]; s < last[A]; s++) {
            const int q = sorted[s];
            double d2 = 0.0;
#pragma unroll
            for (int a = 0; a < ND; a++) {
              const double d = xp[a] - PF(P, fx + a, q);
              d2 += d * d;
            }
            if (sqrt(d2) <= eps) {  // q = p included, like the reference's list
              const double V_q = PF(P, F_VOL0, q) * PF(P, F_JN1, q);
              sum_V += V_q;
              if (PF(P, F_DMG, q) < 1.0) sum_VW += V_q * PF(P, F_W, q);
            }
          }
        }
    const double G_p = (m.Ceps * DeltaX / sum_V) * sum_VW;
    if (G_p > m.Gf) {
      Dn1 = 1.0;
      PF(P, F_DMG1, p) = 1.0;
    }
  }
  const double sc = 1.0 - Dn1;  // kirchhoff_p[i] *= (1 - Damage_n1[p]), in place (:1321-1330)
#pragma unroll
  for (int s = 0; s < T; s++) PF(P, F_TAU + s, p) = PF(P, F_TAU + s, p) * sc;
}

// ------------------------------------------------------------------------------------------------
// Eigensoftening: the same hook with Driver_EigenSoftening -> Eigensoftening__Constitutive__
// (EigenSoftening.c:27-163), restated AS WRITTEN: "first principal" value = eigval[0] of the ascending dsyev order (the
// smallest, :60, :106); the neighbour loop ASSIGNS its term (no +=, :118); StrainF_n and StrainF_n1 are one array
// (Constitutive.c:418-419).
// The reference's loop runs one particle after the other and scales every Kirchhoff stress in place while later
// particles still read it; two passes give the same numbers in parallel:
//   pass 1 (per particle, no neighbour): the smallest principal Kirchhoff stress T0 of the UNSCALED stress, and the
//           damage update of the second branch (:146-160), which reads the particle's own strain and history only;
//   pass 2 (first branch, :71-144): mass sum over the epsilon-neighbourhood and the ONE neighbour term that survives the
//           assignment of :118 -- the first particle of compute_Beps's walk (chain of NodalLocality_0[I0_p], inside a
//           node descending particle index, Beps.c:49-70) with Damage_n < 1 -- whose stress counts scaled by
//           (1 - Damage_n1) iff its index in the CALLER's order is below p's (it came earlier in the reference's loop);
//           then every stress is scaled in place (U-Newmark-beta.c:1321-1330).
// Lists: with this driver Beps is never initialised (U-Newmark-beta.c:182 tests Driver_EigenErosion only), a particle
// whose total displacement is <= 1e-6 has an empty list (Beps.c:30-36), any other the list of its current position.
// ------------------------------------------------------------------------------------------------
template <int ND>
__device__ __forceinline__ double almansi_min_principal(const double* F) {  // eulerian_almansi__Particles__ + eigval[0]
  double b[ND * ND], bm1[ND * ND], e[ND * ND], w[3] = {0.0, 0.0, 0.0}, v[ND * ND];
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int j = 0; j < ND; j++) {
      double a = 0.0;
#pragma unroll
      for (int k = 0; k < ND; k++) a += F[i * ND + k] * F[j * ND + k];
      b[i * ND + j] = a;
    }
  inverse<ND>(bm1, b);
#pragma unroll
  for (int i = 0; i < ND; i++)
#pragma unroll
    for (int j = 0; j < ND; j++) e[i * ND + j] = 0.5 * ((i == j ? 1.0 : 0.0) - bm1[i * ND + j]);
  sym_eigen<ND>(w, v, e);
  return w[0];
}
template <int ND>
__global__ __launch_bounds__(BLK) void k_soften_pass1(PView P, const MatD* __restrict__ mats, double* __restrict__ T0) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= P.np) return;
  double tau[ND * ND], z, w[3] = {0.0, 0.0, 0.0}, v[ND * ND];
  load_block<ND>(P, F_TAU, p, tau, z);
  sym_eigen<ND>(w, v, tau);
  T0[p] = w[0];
  const double Dn = PF(P, F_DMG, p), sf = PF(P, F_STRF1, p);
  if (Dn == 0.0 && w[0] > 0.0) return;  // first branch: pass 2
  if (Dn != 1.0 && sf > 0.0) {          // :146-160
    double F[ND * ND], fz;
    load_block<ND>(P, fFN1(P), p, F, fz);
    const MatD m = mats[P.mat[p]];
    const double aux = (almansi_min_principal<ND>(F) - sf) * m.heps / m.wcrit;
    const double mx = aux > Dn ? aux : Dn;
    PF(P, F_DMG1, p) = 1.0 < mx ? 1.0 : mx;
  }
}
template <int ND>
__global__ __launch_bounds__(BLK) void k_soften_pass2(PView P, GridD g, const MatD* __restrict__ mats, const int* __restrict__ first,
                                                      const int* __restrict__ last, const int* __restrict__ sorted,
                                                      const uint8_t* __restrict__ rank1, const int* __restrict__ orig,
                                                      const double* __restrict__ T0, double DeltaX) {
  const int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= P.np) return;
  constexpr int T = (ND == 2) ? 5 : 9;
  const double Dn = PF(P, F_DMG, p), T0p = T0[p];
  if (Dn == 0.0 && T0p > 0.0) {
    const MatD m = mats[P.mat[p]];
    const double m_p = PF(P, F_MASS, p);
    double sum_m = m_p, term = m_p * T0p;
    double xp[ND], d2p = 0.0;
#pragma unroll
    for (int a = 0; a < ND; a++) {
      xp[a] = PF(P, F_X + a, p);
      d2p += dsqr(PF(P, F_DIS + a, p));
    }
    if (sqrt(d2p) > 0.000001) {  // Beps.c:30-36 (lists are never initialised with this driver: empty otherwise)
      const double eps = m.Ceps * DeltaX;
      const int I0 = P.I0[p], op = orig[p];
      int ijk[3] = {I0 % g.n[0], (I0 / g.n[0]) % g.n[1], I0 / (g.n[0] * g.n[1])};
      const uint8_t* rk = rank1 + 27 * class3_of<ND>(g, ijk);
      long long bestkey = 0x7fffffffffffffffll;
      int bestq = -1;
      for (int dk = (ND == 3 ? -1 : 0); dk <= (ND == 3 ? 1 : 0); dk++)
        for (int dj = -1; dj <= 1; dj++)
          for (int di = -1; di <= 1; di++) {
            const int i = ijk[0] + di, j = ijk[1] + dj, k = ijk[2] + dk;
            if (i < 0 || i >= g.n[0] || j < 0 || j >= g.n[1] || (ND == 3 && (k < 0 || k >= g.n[2]))) continue;
            const int A = i + g.n[0] * (j + g.n[1] * k);
            const long long nodekey = (long long)rk[(di + 1) + 3 * (dj + 1) + 9 * (dk + 1)] << 32;
            for (int s = first[A]; s < last[A]; s++) {
              const int q = sorted[s];
              double d2 = 0.0;
#pragma unroll
              for (int a = 0; a < ND; a++) {
                const double d = xp[a] - PF(P, F_X + a, q);
                d2 += d * d;
              }
              if (sqrt(d2) <= eps) {  // q = p included, like the reference's list
                sum_m += PF(P, F_MASS, q);
                if (PF(P, F_DMG, q) < 1.0) {
                  // the walk visits the nodes in chain order and a node's particles in descending caller index; the
                  // reference's assignment keeps the FIRST such particle of the walk
                  const long long key = nodekey + (long long)(0x7fffffff - orig[q]);
                  if (key < bestkey) {
                    bestkey = key;
                    bestq = q;
                  }
                }
              }
            }
          }
      if (bestq >= 0) {
        const double sc = (orig[bestq] < op) ? (1.0 - PF(P, F_DMG1, bestq)) : 1.0;  // scaled in place already, or not yet
        term = PF(P, F_MASS, bestq) * (T0[bestq] * sc);
      }
    }
    const double Teps = term / sum_m;
    if (Teps > m.ft) {
      double F[ND * ND], fz;
      load_block<ND>(P, fFN1(P), p, F, fz);
      PF(P, F_STRF1, p) = almansi_min_principal<ND>(F);
    }
  }
  const double sc = 1.0 - PF(P, F_DMG1, p);  // kirchhoff_p[i] *= (1 - Damage_n1[p]), in place (:1321-1330)
#pragma unroll
  for (int s = 0; s < T; s++) PF(P, F_TAU + s, p) = PF(P, F_TAU + s, p) * sc;
}

// __constitutive_update (U-Newmark-beta.c:1208-1242)
template <int ND, bool FRIC>
__global__ __launch_bounds__(BLK) void k_stress(PView P, const MatD* __restrict__ mats, ParamsD prm,
                                                int* __restrict__ gstatus) {
  int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= P.np) return;
  if (P.erosion && PF(P, F_DMG, p) == 1.0) {  // failed particle: U-Newmark-beta.c:1218-1224
    PF(P, F_W, p) = 0.0;
    return;
  }
  double Fn1[ND * ND], DF[ND * ND], tau[ND * ND], z;
  load_block<ND>(P, fFN1(P), p, Fn1, z);
  load_block<ND>(P, F_DF, p, DF, z);
  int st = stress_update<ND, -1, true, FRIC>(P, p, mats, prm, Fn1, DF, PF(P, F_JN1, p), tau);
  if (st) {
    atomicOr(&P.status[p], st);
    atomicOr(gstatus, st);
  }
}

// __update_particles_internal_variables (U-Newmark-beta.c:1917-1978)
// After explicit steps the n+1 slots of F and b_e hold the previous step's values (rolled by renaming);
// the reference's copy semantics (n+1 == n after the roll) are restored on demand, before any level-B stage
// or download looks at them.
template <int ND>
__global__ __launch_bounds__(BLK) void k_copy_n_to_n1(PView P, const MatD* __restrict__ mats) {
  int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= P.np) return;
  constexpr int T = (ND == 2) ? 5 : 9;
  {
    // DF of the last explicit step, which K3 did not store: the n slot holds F_n+1 of that step, the n+1 slot still
    // the F_n it started from (the roll is a renaming), so DF = F_n+1 F_n^-1 on the d x d block (the 2-D zz slot of DF
    // stays 1, compute-Strains.c:20-44 never touches it)
    double Fnew[ND * ND], Fold[ND * ND], inv[ND * ND], z;
    load_block<ND>(P, fFN(P), p, Fnew, z);
    load_block<ND>(P, fFN1(P), p, Fold, z);
    if (inverse<ND>(inv, Fold)) {
#pragma unroll
      for (int i = 0; i < ND; i++)
#pragma unroll
        for (int j = 0; j < ND; j++) {
          double a2 = 0.0;
#pragma unroll
          for (int k2 = 0; k2 < ND; k2++) a2 += Fnew[i * ND + k2] * inv[k2 * ND + j];
          PF(P, F_DF + i * ND + j, p) = a2;
        }
    }
    // Kirchhoff stress and energy of the hyperelastic laws, which the fused step did not store (stress_update LAZY):
    // the same functions of the same stored F_n+1 and J
    const MatD m = mats[P.mat[p]];
    if (m.type == NLPS_MAT_NEO_HOOKEAN || m.type == NLPS_MAT_HENCKY) {
      StressIO<ND> o;
      o.fail = 0;
      if (m.type == NLPS_MAT_NEO_HOOKEAN) law_neo_hookean<ND>(m, Fnew, PF(P, F_JN, p), o);
      else law_hencky<ND>(m, Fnew, o);
      store_block<ND>(P, F_TAU, p, o.tau, o.tau_zz, true);
      PF(P, F_W, p) = o.W;
    }
  }
#pragma unroll
  for (int s = 0; s < T; s++) {
    PF(P, fFN1(P) + s, p) = PF(P, fFN(P) + s, p);
    PF(P, fBEN1(P) + s, p) = PF(P, fBEN(P) + s, p);
  }
  PF(P, F_JN1, p) = PF(P, F_JN, p);
  PF(P, F_KN1, p) = PF(P, F_KN, p);
  PF(P, F_EN1, p) = PF(P, F_EN, p);
  PF(P, F_RHO, p) = PF(P, F_RHOJ, p) / PF(P, F_JN, p);
}

// first explicit step after the upload or after level-B stages: the invariant of the density update (F_RHOJ)
__global__ __launch_bounds__(BLK) void k_init_rhoj(PView P) {
  int p = blockIdx.x * BLK + threadIdx.x;
  if (p < P.np) PF(P, F_RHOJ, p) = PF(P, F_RHO, p) * PF(P, F_JN, p);
}

template <int ND>
__global__ __launch_bounds__(BLK) void k_roll(PView P) {
  int p = blockIdx.x * BLK + threadIdx.x;
  if (p >= P.np) return;
  double J = PF(P, F_JN1, p);
  PF(P, F_JN, p) = J;
  PF(P, F_RHO, p) = PF(P, F_MASS, p) / (PF(P, F_VOL0, p) * J);
  PF(P, F_KN, p) = PF(P, F_KN1, p);
  PF(P, F_EN, p) = PF(P, F_EN1, p);
  if (P.erosion) PF(P, F_DMG, p) = PF(P, F_DMG1, p);  // U-Newmark-beta.c:1950-1953
  if (P.softening) PF(P, F_STRF, p) = PF(P, F_STRF1, p);  // :1954-1956
  constexpr int T = (ND == 2) ? 5 : 9;
#pragma unroll
  for (int s = 0; s < T; s++) {
    PF(P, fBEN(P) + s, p) = PF(P, fBEN1(P) + s, p);
    PF(P, fFN(P) + s, p) = PF(P, fFN1(P) + s, p);
    PF(P, F_DTFN + s, p) = PF(P, F_DTFN1 + s, p);
  }
}

// ------------------------------------------------------------------------------------------------
// nodal kernels (grid numbering)
// ------------------------------------------------------------------------------------------------
// Dirichlet sets of one step for the nodal kernel: set i fixes the directions `bits[i]` of its nodes at v[i]; a node
// finds its sets in the bit mask bcmask[A] (built once per set of lists, ensure_bcs), later sets override earlier ones
// like the sequence of k_bc launches they replace (one launch per set and step: 5 us each).
#define NLPS_MAX_BC_INLINE 8
struct BcStep {
  int n;
  int dim[NLPS_MAX_BC_INLINE], bits[NLPS_MAX_BC_INLINE];
  double v[NLPS_MAX_BC_INLINE][3];
};
__global__ void k_bc_mark(const int* __restrict__ nodes, int n, unsigned bit, unsigned* __restrict__ mask) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < n) atomicOr(&mask[nodes[q]], bit);
}

template <int ND>
__global__ void k_nodal_dU(int n0, int nnodes, int n0b, int nnodesb, NView N, const unsigned* __restrict__ bcmask,
                           BcStep bc) {  // U-Verlet.c:357-362 + impose_Dirichlet_Boundary_Conditions :455-527
  int A = blockIdx.x * blockDim.x + threadIdx.x;  // two node ranges: [n0, n0+nnodes) and [n0b, n0b+nnodesb)
  if (A >= nnodes + nnodesb) return;
  A = A < nnodes ? n0 + A : n0b + (A - nnodes);
  double M = N.nm[(size_t)A * (1 + ND)];
  // an active node no particle lists (narrow LME kernels: the 1-ring activation reaches further than the cut-off
  // radius) has M = 0: its value is 0 like VecPointwiseDivide's in the maintained driver, never 0/0 (the gather
  // kernels read every window slot, a NaN there would poison the zero-weighted non-members)
  const bool active = N.active[A];
  bool act = active && M != 0.0;
  double val[ND];
  bool fix[ND];
#pragma unroll
  for (int a = 0; a < ND; a++) {
    val[a] = act ? N.nm[(size_t)A * (1 + ND) + 1 + a] / M : 0.0;
    fix[a] = false;
  }
  const unsigned bm = (bcmask && active) ? bcmask[A] : 0u;
  if (bm) {
    for (int i = 0; i < bc.n; i++) {
      if (!((bm >> i) & 1u)) continue;
#pragma unroll
      for (int k = 0; k < ND; k++)
        if (k < bc.dim[i] && ((bc.bits[i] >> k) & 1)) {
          val[k] = bc.v[i][k];
          fix[k] = true;
        }
    }
  }
#pragma unroll
  for (int a = 0; a < ND; a++) {
    N.dU[(size_t)A * ND + a] = val[a];
    if (fix[a]) N.fixed[(size_t)A * ND + a] = 1;
  }
}

template <int ND>
__global__ void k_bc(const int* __restrict__ nodes, int n, int dim, int dirbits, double v0, double v1, double v2,
                     NView N, int r0, int r1, int inside) {  // impose_Dirichlet_Boundary_Conditions, U-Verlet.c:455-527
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int A = nodes[q];
  if (((A >= r0 && A < r1) ? 1 : 0) != inside) return;  // node-range filter (interior / ghost-band passes)
  if (!N.active[A]) return;
  double v[3] = {v0, v1, v2};
#pragma unroll
  for (int k = 0; k < ND; k++)
    if (k < dim && ((dirbits >> k) & 1)) {
      N.dU[(size_t)A * ND + k] = v[k];
      N.fixed[(size_t)A * ND + k] = 1;
    }
}

// clr_* (search fused into k5_tile): this kernel runs in front of it and resets what that search accumulates into -- the
// seeds and per-node counters of the nodes it visits, and (first launch of the stage only) the tile counters
template <int ND>
__global__ void k_nodal_accel(int n0, int nnodes, int n0b, int nnodesb, NView N, double g0, double g1, double g2,
                              int clr_seed, int* __restrict__ clr_node_cnt, int* __restrict__ clr_tile_count, int ntiles) {  // U-Verlet.c:947-957
  int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (clr_tile_count)
    for (int t = A; t < ntiles; t += gridDim.x * blockDim.x) clr_tile_count[t] = 0;
  if (A >= nnodes + nnodesb) return;
  A = A < nnodes ? n0 + A : n0b + (A - nnodes);
  if (clr_seed) {
    N.seed[A] = 0;
    if (clr_node_cnt) clr_node_cnt[A] = 0;
  }
  double M = N.nm[(size_t)A * (1 + ND)];
  bool act = N.active[A] && M != 0.0;  // massless active node: see k_nodal_dU
  double gv[3] = {g0, g1, g2};
#pragma unroll
  for (int a = 0; a < ND; a++) {
    size_t o = (size_t)A * ND + a;
    double f = N.force[o];
    bool fx = N.fixed[o];
    N.accel[o] = (act && !fx) ? gv[a] + f / M : 0.0;
    N.reaction[o] = (act && fx) ? f : 0.0;
  }
}

// masked <-> grid numbering
__global__ void k_expand(double* __restrict__ grid, const double* __restrict__ masked, const int* __restrict__ n2m,
                         int nnodes, int nf) {
  int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= nnodes) return;
  int m = n2m[A];
  for (int f = 0; f < nf; f++) grid[(size_t)A * nf + f] = (m >= 0) ? masked[(size_t)m * nf + f] : 0.0;
}

// k_expand of the residual call, which also resets the force accumulator of the node window (one launch for the two)
__global__ void k_expand_reset(double* __restrict__ grid, const double* __restrict__ masked, const int* __restrict__ n2m,
                               int nnodes, int nf, double* __restrict__ zero, int n0, int nwn) {
  int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= nnodes) return;
  int m = n2m[A];
  for (int f = 0; f < nf; f++) grid[(size_t)A * nf + f] = (m >= 0) ? masked[(size_t)m * nf + f] : 0.0;
  if (A >= n0 && A < n0 + nwn)
    for (int f = 0; f < nf; f++) zero[(size_t)A * nf + f] = 0.0;
}

// mode 0: out = grid[A*gstride + goff + (bcast?0:f)]
// mode 1: out += ... skipping fixed dofs (d2m == -1)
// mode 2: out = (fixed ? 0 : grid) / div[idx]
__global__ void k_compact(double* __restrict__ out, const double* __restrict__ grid, const int* __restrict__ n2m,
                          const int* __restrict__ d2m, const double* __restrict__ div, int nnodes, int nf, int gstride,
                          int goff, int bcast, int mode) {
  int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= nnodes) return;
  int m = n2m[A];
  if (m < 0) return;
  for (int f = 0; f < nf; f++) {
    size_t idx = (size_t)m * nf + f;
    double v = grid[(size_t)A * gstride + goff + (bcast ? 0 : f)];
    if (mode == 0) out[idx] = v;
    else if (mode == 1) {
      if (d2m[idx] != -1) out[idx] += v;
    } else {
      const double dv = div[idx];  // VecPointwiseDivide (U-Newmark-beta.c:695-696): 0 where the lumped mass is 0
      out[idx] = dv != 0.0 ? ((d2m[idx] == -1) ? 0.0 : v) / dv : 0.0;
    }
  }
}

// ---- exclusive scan of byte flags -> (flag ? running index : -1); 1024 items per block
__global__ void k_scan_count(const unsigned char* __restrict__ flags, int n, int invert, int* __restrict__ bsum) {
  __shared__ int sh[256];
  int base = blockIdx.x * 1024 + threadIdx.x * 4, c = 0;
  for (int q = 0; q < 4; q++)
    if (base + q < n) c += ((flags[base + q] != 0) != (invert != 0));
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0];
}
__global__ void k_scan_top(int* __restrict__ bsum, int nb, int* __restrict__ total) {
  // one 1024-thread block; each thread owns a contiguous chunk
  __shared__ int sh[1024];
  int chunk = (nb + 1023) / 1024;
  int lo = threadIdx.x * chunk, hi = min(nb, lo + chunk), c = 0;
  for (int q = lo; q < hi; q++) c += bsum[q];
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    int v = ((int)threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  int run = sh[threadIdx.x] - c;
  for (int q = lo; q < hi; q++) {
    int v = bsum[q];
    bsum[q] = run;
    run += v;
  }
  if (threadIdx.x == 1023) *total = sh[1023];
}
__global__ void k_scan_write(const unsigned char* __restrict__ flags, int n, int invert, const int* __restrict__ bsum,
                             int* __restrict__ out) {
  __shared__ int sh[256];
  int base = blockIdx.x * 1024 + threadIdx.x * 4;
  int f[4], c = 0;
  for (int q = 0; q < 4; q++) {
    f[q] = (base + q < n) ? ((flags[base + q] != 0) != (invert != 0)) : 0;
    c += f[q];
  }
  sh[threadIdx.x] = c;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    int v = ((int)threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += v;
    __syncthreads();
  }
  int run = bsum[blockIdx.x] + sh[threadIdx.x] - c;
  for (int q = 0; q < 4; q++)
    if (base + q < n) {
      out[base + q] = f[q] ? run : -1;
      run += f[q];
    }
}
// get_active_dofs__MeshTools__ marking loop (Nodes-Tools.c:96-135) in masked numbering
__global__ void k_mark_fixed_masked(const int* __restrict__ nodes, int n, int dim, int dirbits, int ndof,
                                    const int* __restrict__ n2m, unsigned char* __restrict__ fixedm) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  int m = n2m[nodes[q]];
  if (m < 0) return;
  for (int k = 0; k < dim; k++)
    if ((dirbits >> k) & 1) fixedm[(size_t)m * ndof + k] = 1;
}

#include "nlps_tile_kernels.hpp"

// The two kernels between the search and the tile lists in one launch (they do not depend on each other and both are
// too small to fill the chip: 8 us + 10 us -> 10 us): workgroup 0 = exclusive scan of the tile counts + work lists
// (tile_scan_block), the others = 1-ring activation of 1024 nodes each.
// Layer tables of the canonical tile lists: layer r of a tile = the r-th particle of every node that has one, nodes in
// lattice order.  The nodes of a tile form NNW words of 64 (3-D: 4^3 nodes = 1 word, 2-D: 16^2 = 4 words);
// mask[tile][r][w] = the nodes of word w that reach layer r, base[tile][r][w] = list offset of the first of them.  A
// particle of node l with rank r inside its node (bin_particle) sits at
// start[tile] + base[r][l / 64] + popcount(mask[r][l / 64] & below(l % 64)).  base[tile][0][0] < 0: some node is
// deeper than LMAX, the tile keeps the order of the binning.  Replaces the per-tile counting sort (k_tile_order,
// 13.6 us) outside deterministic mode.
struct TileTab {
  static constexpr int LMAX = 32;
  int nt[3];
  const int* node_cnt;
  unsigned long long* mask;
  int* base;
};
template <int ND>
struct TileTabCfg {
  static constexpr int NN = (ND == 3) ? TileCfg<3>::TB * TileCfg<3>::TB * TileCfg<3>::TB : TileCfg<2>::TB * TileCfg<2>::TB;
  static constexpr int NNW = NN / 64;
};

template <int ND>
__global__ __launch_bounds__(1024) void k_dilate_scan(int n0, int nnodes, GridD g, NView N, TileScanArgs ts,
                                                      int* __restrict__ foreign, int* __restrict__ foreign_host, TileTab tab,
                                                      int clear_nodal) {
  if (blockIdx.x == 0) {
    if (foreign && threadIdx.x < 64) {  // this step's count of displaced particles (bin_particle) -> pinned host word
      int v = foreign[32 * threadIdx.x];
      foreign[32 * threadIdx.x] = 0;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
      if (threadIdx.x == 0) *foreign_host = v;
    }
    tile_scan_block(ts);
    return;
  }
  const int nbd = (nnodes + 1023) >> 10;  // workgroups of the activation
  if ((int)blockIdx.x <= nbd) {
    const int A = ((int)blockIdx.x - 1) * 1024 + (int)threadIdx.x;
    if (A < nnodes) {
      dilate_node<ND>(n0 + A, g, N);
      if (clear_nodal) {  // (k_step_clear's nodal part, when that kernel has nothing else to do: the search ran ahead)
        const size_t B = (size_t)n0 + A;
#pragma unroll
        for (int a = 0; a < 1 + ND; a++) N.nm[B * (1 + ND) + a] = 0.0;
#pragma unroll
        for (int a = 0; a < ND; a++) {
          N.force[B * ND + a] = 0.0;
          N.fixed[B * ND + a] = 0;
        }
      }
    }
    return;
  }
  // layer tables of the canonical tile lists (TileTab), one wave per tile
  constexpr int TB = TileCfg<ND>::TB, NNW = TileTabCfg<ND>::NNW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = ((int)blockIdx.x - 1 - nbd) * 16 + wave;
  if (q >= ts.n) return;
  const int tile = ts.tile0 + q;
  const int tx = tile % tab.nt[0], ty = (tile / tab.nt[0]) % tab.nt[1], tz = tile / (tab.nt[0] * tab.nt[1]);
  int c[NNW], maxc = 0;
#pragma unroll
  for (int w = 0; w < NNW; w++) {
    const int l = w * 64 + lane;
    const int bx = l % TB, by = (l / TB) % TB, bz = l / (TB * TB);
    const int i = tx * TB + bx, j = ty * TB + by, k = (ND == 3) ? tz * TB + bz : 0;
    const bool in = i < g.n[0] && j < g.n[1] && (ND == 2 || k < g.n[2]);
    c[w] = in ? tab.node_cnt[i + g.n[0] * (j + g.n[1] * k)] : 0;
    maxc = max(maxc, c[w]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) maxc = max(maxc, __shfl_xor(maxc, off));
  unsigned long long* tm = tab.mask + (size_t)tile * TileTab::LMAX * NNW;
  int* tb = tab.base + (size_t)tile * TileTab::LMAX * NNW;
  if (maxc > TileTab::LMAX) {  // deeper than the table: this tile keeps the order of the binning
    if (lane == 0) tb[0] = -1;
    return;
  }
  int run = 0;
  for (int r = 0; r < maxc; r++) {
#pragma unroll
    for (int w = 0; w < NNW; w++) {
      const unsigned long long m = __ballot(c[w] > r);
      if (lane == 0) {
        tm[r * NNW + w] = m;
        tb[r * NNW + w] = run;
      }
      run += (int)__popcll(m);
    }
  }
  if (maxc == 0 && lane == 0) tb[0] = 0;
}
// Both tile lists in one pass over the particles: order = as binned (position = rank of the wave-aggregated tile atomic:
// runs of memory-consecutive particles), order2 = canonical (TileTab).
// cursor != nullptr: the binning only counted (TileCnt::defer) -- the position inside the tile list comes from a cursor
// that starts at the list's first slot (tile_scan_block), one wave-aggregated atomic per (wave, tile), lanes in memory
// order; the rank inside the closest node counts the node's counter DOWN (every value c - 1 .. 0 once; the counter is
// back at zero for the next binning, the layer tables of this step were made from it before this kernel).
template <int ND>
__global__ __launch_bounds__(BLK) void k_fill_orders(int np, const int* __restrict__ tile, const int* __restrict__ rank,
                                                     const int* __restrict__ nrank, int* __restrict__ I0a,
                                                     const int* __restrict__ I0n,
                                                     const int* __restrict__ start, GridD g, TileTab tab,
                                                     int* __restrict__ order, int* __restrict__ order2,
                                                     int* __restrict__ cursor, int* __restrict__ node_cnt) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int t = p < np ? tile[p] : -1;
  int pos = -1, I0 = 0, r = 0;
  if (t >= 0) {  // the search was done ahead by k5_tile: its closest node becomes THE closest node (a particle that is in
    I0 = I0a[p];  // no list was not visited: its I0n equals I0)
    if (I0n) {
      const int In = I0n[p];
      if (In != I0) I0a[p] = I0 = In;
    }
    // (asked for first: its round trip runs under the one of the list cursor below)
    if (cursor) r = atomicSub(&node_cnt[I0], 1) - 1;
  }
  if (cursor) {  // (every lane of the wave takes part)
    const int lane = threadIdx.x & 63;
    int my_leader = lane, my_off = 0, my_cnt = 0;
    bool todo_l = t >= 0;
    while (true) {  // (groups by ballots, then one atomic per group in ONE instruction: see bin_particle)
      const u64 todo = __ballot(todo_l);
      if (!todo) break;
      const int leader = __ffsll((unsigned long long)todo) - 1;
      const int t0 = __shfl(t, leader);
      const bool mine = todo_l && (t == t0);
      const u64 same = __ballot(mine);
      if (mine) {
        my_leader = leader;
        my_off = (int)__popcll(same & ((1ull << lane) - 1ull));
        my_cnt = (int)__popcll(same);
        todo_l = false;
      }
    }
    int base = 0;
    if (t >= 0 && lane == my_leader) base = atomicAdd(&cursor[t], my_cnt);
    base = __shfl(base, my_leader);
    if (t >= 0) pos = base + my_off;
  }
  if (t < 0) return;
  constexpr int TB = TileCfg<ND>::TB;
  const int s0 = start[t];
  if (!cursor) pos = s0 + rank[p];
  order[pos] = p;
  constexpr int NNW = TileTabCfg<ND>::NNW;
  const int* tb = tab.base + (size_t)t * TileTab::LMAX * NNW;
  if (tb[0] < 0) {
    order2[pos] = p;
    return;
  }
  if (!cursor) r = nrank[p];
  const int bx = (I0 % g.n[0]) % TB, by = ((I0 / g.n[0]) % g.n[1]) % TB, bz = (ND == 3) ? (I0 / (g.n[0] * g.n[1])) % TB : 0;
  const int l = bx + TB * (by + TB * bz), w = l >> 6;
  const unsigned long long m = tab.mask[((size_t)t * TileTab::LMAX + r) * NNW + w];
  order2[s0 + tb[r * NNW + w] + (int)__popcll(m & ((1ull << (l & 63)) - 1ull))] = p;
}
#include "nlps_tangent_kernels.hpp"

// ------------------------------------------------------------------------------------------------
// physical re-sort of the particle SoA (maintenance, every few dozen steps): restores the
// (tile of I0, corner type, I0 in tile) memory order that makes waves hit distinct window slots
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ void k_sort_keys(PView P, GridD g, TileCnt tc, unsigned long long* __restrict__ keys, int* __restrict__ vals,
                            const unsigned char* __restrict__ leaving, const MatD* __restrict__ mats) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  if (leaving && leaving[p]) {  // migration: the emigrants sort behind everything that stays
    keys[p] = ~0ull;
    vals[p] = p;
    return;
  }
  constexpr int TB = TileCfg<ND>::TB;
  const int I0 = P.I0[p];
  int ijk[3] = {I0 % g.n[0], (I0 / g.n[0]) % g.n[1], I0 / (g.n[0] * g.n[1])};
  unsigned long long kt = 0, kn = 0, kc = 0, mt = 1, mn = 1, mc = 1;
#pragma unroll
  for (int a = 0; a < ND; a++) {
    const double xi = (PF(P, F_X + a, p) - g.o[a]) / g.h;
    int c = (int)floor(xi);
    c = c < 0 ? 0 : (c > g.n[a] - 2 ? g.n[a] - 2 : c);
    kt += mt * (unsigned long long)(ijk[a] / TB);
    mt *= (unsigned long long)tc.nt[a];
    kn += mn * (unsigned long long)(ijk[a] % TB);
    mn *= TB;
    kc += mc * (unsigned long long)(ijk[a] > c ? 1 : 0);
    mc *= 2;
  }
  (void)mats;  // (tile, law, corner type, node) was tried for clouds with several laws: K3's per-law launches load
               // coalesced, but K2's waves lose their distinct closest nodes (0.25 -> 0.35 ms) -- net loss
  keys[p] = (kt * mc + kc) * mn + kn;
  vals[p] = p;
}
// ---- particle migration between slab ranks: packed rows of MIG_WORDS 8-byte words:
// the NFD fields in canonical order (F_n / b_e,n in their n slots whatever the current renaming), then
// I0, material, NumberNodes, status, caller index, global id (as integers in words), then the two mask words.
static constexpr int MIG_WORDS = NFD + 8;
__device__ __forceinline__ int mig_field(const PView& P, int f) {  // canonical field -> current physical field
  if (!P.flip) return f;
  if (f >= F_FN && f < F_FN + 9) return f + (F_FN1 - F_FN);
  if (f >= F_FN1 && f < F_FN1 + 9) return f - (F_FN1 - F_FN);
  if (f >= F_BEN && f < F_BEN + 9) return f + (F_BEN1 - F_BEN);
  if (f >= F_BEN1 && f < F_BEN1 + 9) return f - (F_BEN1 - F_BEN);
  return f;
}
template <int ND>
__global__ void k_mig_flag(PView P, GridD g, int keep_lo, int keep_hi, unsigned char* __restrict__ leaving,
                           int* __restrict__ slot, int* __restrict__ counts) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np) return;
  const int layer = P.I0[p] / (g.nnodes / g.n[ND - 1]);
  const int dir = layer < keep_lo ? 1 : (layer > keep_hi ? 2 : 0);
  leaving[p] = (unsigned char)dir;
  if (dir) slot[p] = atomicAdd(&counts[dir - 1], 1);
}
__global__ void k_mig_pack(PView P, const unsigned char* __restrict__ leaving, const int* __restrict__ slot,
                           const int* __restrict__ perm, const int* __restrict__ gid, double* __restrict__ down,
                           double* __restrict__ up) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P.np || !leaving[p]) return;
  double* row = (leaving[p] == 1 ? down : up) + (size_t)slot[p] * MIG_WORDS;
  for (int f = 0; f < NFD; f++) row[f] = PF(P, mig_field(P, f), p);
  long long* w = reinterpret_cast<long long*>(row + NFD);
  w[0] = P.I0[p];
  w[1] = P.mat[p];
  w[2] = P.nn[p];
  w[3] = P.status[p];
  w[4] = perm[p];
  w[5] = gid[p];
  w[6] = (long long)P.mlo[p];
  w[7] = (long long)P.mhi[p];
}
__global__ void k_mig_unpack(PView P, int first, int n, const double* __restrict__ rows, int* __restrict__ perm,
                             int* __restrict__ gid, unsigned char* __restrict__ leaving) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  const int p = first + j;
  const double* row = rows + (size_t)j * MIG_WORDS;
  for (int f = 0; f < NFD; f++) PF(P, mig_field(P, f), p) = row[f];
  const long long* w = reinterpret_cast<const long long*>(row + NFD);
  P.I0[p] = (int)w[0];
  P.I0n[p] = (int)w[0];
  P.mat[p] = (int)w[1];
  P.nn[p] = (int)w[2];
  P.status[p] = (int)w[3];
  perm[p] = (int)w[4];
  gid[p] = (int)w[5];
  P.mlo[p] = (u64)w[6];
  P.mhi[p] = (u64)w[7];
  leaving[p] = 0;
}

template <class T>
__global__ void k_copy(T* __restrict__ out, const T* __restrict__ in, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}

// all particle fields of the re-sort in one launch: thread = new slot, blockIdx.y = group of GATHER_FIELDS fields
static constexpr int GATHER_FIELDS = 8;
__global__ void k_gather_fields(double* __restrict__ out, const double* __restrict__ in, const int* __restrict__ idx,
                                int n, size_t npad, int nf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = (size_t)idx[i];
  const int f0 = blockIdx.y * GATHER_FIELDS, f1 = min(nf, f0 + GATHER_FIELDS);
  for (int f = f0; f < f1; f++) out[(size_t)f * npad + i] = in[(size_t)f * npad + j];
}

// the live components of the re-sort as ONE launch: blockIdx.y = group of GATHER_FIELDS entries of a list of components
struct FieldList {
  int n;
  unsigned char f[128];
};
__global__ void k_gather_field_list(double* __restrict__ out, const double* __restrict__ in, const int* __restrict__ idx,
                                    int n, size_t npad, FieldList fl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const size_t j = (size_t)idx[i];
  const int q0 = blockIdx.y * GATHER_FIELDS, q1 = min(fl.n, q0 + GATHER_FIELDS);
  for (int q = q0; q < q1; q++) {
    const size_t f = fl.f[q];
    out[f * npad + i] = in[f * npad + j];
  }
}

// the integer arrays of the particles (closest nodes, material, list length, status, permutation, ids; the two mask
// words) through one scratch block: one gather launch and one copy-back launch instead of two per array
struct SmallArrays {
  int* ia[7];
  unsigned long long* ua[2];
};
__global__ void k_gather_small(SmallArrays A, const int* __restrict__ idx, int n, size_t npad, int* __restrict__ si,
                               unsigned long long* __restrict__ su) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = idx[i];
#pragma unroll
  for (int a = 0; a < 7; a++) si[(size_t)a * npad + i] = A.ia[a][j];
#pragma unroll
  for (int a = 0; a < 2; a++) su[(size_t)a * npad + i] = A.ua[a][j];
}
__global__ void k_copy_small(SmallArrays A, int n, size_t npad, const int* __restrict__ si,
                             const unsigned long long* __restrict__ su) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
#pragma unroll
  for (int a = 0; a < 7; a++) A.ia[a][i] = si[(size_t)a * npad + i];
#pragma unroll
  for (int a = 0; a < 2; a++) A.ua[a][i] = su[(size_t)a * npad + i];
}

template <class T>
__global__ void k_gather(T* __restrict__ out, const T* __restrict__ in, const int* __restrict__ idx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[idx[i]];
}

template <class T>
__global__ void k_scatter(T* __restrict__ out, const T* __restrict__ in, const int* __restrict__ idx, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[idx[i]] = in[i];
}

// Periodic re-sort with the tile lists as the permutation: the particles that are in NO list of the step that built the
// lists (flagged by its search instead of binned: failed element search, stencil outside the node window) go behind them.
// Membership is read off the lists themselves -- P.tile[] may already belong to the NEXT step (the search k5_tile runs
// ahead), so "tile < 0" would miss a particle that was listed and is flagged now, and count one twice.
__global__ void k_mark_listed(int np, const int* __restrict__ last_start, const int* __restrict__ last_count,
                              const int* __restrict__ order, unsigned char* __restrict__ listed) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = min(np, *last_start + *last_count);
  if (q < total) listed[order[q]] = 1;
}
__global__ void k_append_unlisted(int np, const unsigned char* __restrict__ listed, const int* __restrict__ last_start,
                                  const int* __restrict__ last_count, int* __restrict__ counter, int* __restrict__ order) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np || listed[p]) return;
  const int pos = *last_start + *last_count + atomicAdd(counter, 1);
  if (pos < np) order[pos] = p;
}

__global__ void k_node_tables(int nn, const double* __restrict__ h_avg, double gamma_lme, double neg_log_tol_zero,
                              double4* __restrict__ out) {
  const int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= nn) return;
  const double hv = h_avg[A];
  const double beta = gamma_lme / (hv * hv);
  const double Ra = sqrt(neg_log_tol_zero / beta);
  out[A] = make_double4(beta, sqrt_threshold(Ra), Ra, 0.0);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
struct BcDev {
  const int* host_nodes;
  int n;
  int* dnodes;
};


struct nlps_gpu {
  int nd, T;
  GridD g;
  ParamsD prm;
  nlps_params hprm;
  int nsteps;
  hipStream_t stream;
  bool own_stream;
  std::string err;

  PView P;
  std::vector<int> perm;  // sorted slot -> caller's particle index
  NView N;
  double* h_avg_d;
  double4* beta_t2_d = nullptr;  // NView::beta_t2
  MatD* mats_d;
  int nmats;
  int uniform_law;  // material law shared by every particle, or -1
  int law_present;  // bit l set: some material follows law l
  bool k3_per_law = true;  // several laws: one K3 launch per law (few mixed tiles) or one run-time-dispatch kernel
  uint8_t* rank1_d;
  nlps_host::StencilTables tab;

  // masks
  int* n2m_d;
  int* d2m_d;
  // nlps_gpu_set_node_numbering: lattice node of file node A (canon_d), and the scratch of the file-order mask scan
  int* canon_d = nullptr;
  unsigned char* mask_flags_d = nullptr;
  int* mask_idx_d = nullptr;
  unsigned char* fixedm_d;
  int* bsum_d;
  int* total_d;
  int* gstatus_d;
  int nactive, nfree;
  bool masks_valid;
  bool binned;  // order[] / tile tables describe the current I0s
  // The last kernel of the fused explicit step (k5_tile<., ., true>) does the search of the NEXT step for the particles
  // it moves.  searched: the closest nodes already belong to the current positions (a search that follows must not
  // update them a second time, LME.c:927-929); ahead: seeds, tile counters, per-particle tile / rank and per-node
  // counters of that search are in place, the next explicit step starts at the activation kernel.
  bool searched = false, ahead = false;
  int fuse_search = 1;  // developer switch NLPS_FUSE_SEARCH
  int lazy_nodal = 1;  // the folded explicit step (k3_tile_lazy / k5_tile_lazy): 1 below 2 M particles, 2 always, 0 never (NLPS_LAZY_NODAL)
  int ncu = 256;
  bool nodal_stale = false;
  BcStep last_bc;
  const unsigned* last_bm = nullptr;
  double last_gv[3] = {0, 0, 0};
  // migration
  int* gid_d = nullptr;               // global particle id (default: the caller's index)
  unsigned char* leaving_d = nullptr;  // 0 stay, 1 leaves downwards, 2 upwards (between select and commit)
  int* mig_slot_d = nullptr;
  int* mig_cnt_d = nullptr;
  double *mig_down_d = nullptr, *mig_up_d = nullptr;
  int mig_n[2] = {0, 0};
  bool mig_selected = false;
  bool migrated = false;             // downloads are ordered by ascending global id from now on
  bool rolled = false;  // explicit steps renamed the n/n+1 tensor slots since the last materialise_roll()
  bool level_b_fields = false;  // C_ep / rate tensors hold data (a level-B constitutive or rate call was made)

  // scratch nodal arrays
  double* gridA;  // [nnodes][2*ND] general purpose
  double* gridB;  // [nnodes][ND] x4 for kinetics
  double* maskedA;
  size_t maskedA_cap;

  std::vector<BcDev> bcs;
  unsigned* bcmask_d = nullptr;  // per node: bit i = member of Dirichlet set i (<= NLPS_MAX_BC_INLINE sets; k_nodal_dU)

  // periodic physical re-sort
  int* perm_d;            // sorted slot -> caller's particle index (device copy of perm)
  bool perm_dirty;        // device perm newer than the host copy
  int resort_every, steps_since_sort;
  unsigned long long *skey_d, *skey2_d;
  int *sval_d, *sval2_d;
  void* cub_tmp;
  size_t cub_tmp_bytes;
  double* gather_tmp;     // [npad] scratch for the gather of the integer arrays
  double* Pd_alt = nullptr;  // twin of P.d, target of the re-sort (allocated at the first one)

  // per-step tile binning
  int nt[3], ntiles;
  int* tile_count_d;
  int* tile_count2_d = nullptr;  // the counters the search ahead (k5_tile) fills while tile_count_d still sizes the lists in use
  int* tile_start_d;
  int2 *work1_d = nullptr, *work2_d = nullptr;  // compacted (tile, part) work lists, see TileD
  int resort_from_lists = 1;     // developer switch NLPS_RESORT_FROM_LISTS (resort)
  // canonical lists from per-node counters (TileTab): node_cnt[nnodes], nrank[npad], layer tables [ntiles][LMAX]
  int *node_cnt_d = nullptr, *nrank_d = nullptr, *tabo_d = nullptr;
  int* tile_cursor_d = nullptr;  // [ntiles + 1] list cursors of the deferred ranks (TileCnt::defer, k_fill_orders)
  int defer_ranks = 1;           // the search riding on K5 only counts; ranks come from k_fill_orders (debug option defer_ranks)
  bool ranks_deferred = false;   // ... and did so in the step before: the lists of this step take their ranks from cursors
  unsigned long long* tabm_d = nullptr;
  int node_lists_on = 1;         // developer switch NLPS_NODE_LISTS (0: the per-tile counting sort k_tile_order)
  // adaptive re-sort (nlps_gpu_set_adaptive_resort): see TileCnt::home.  The count of displaced particles of a step
  // reaches the pinned host word at the end of its search stage; explicit_step adds count / NumGP to `debt` every
  // step and re-sorts ahead of the interval when the debt since the last re-sort exceeds `adaptive_resort`
  int* home_d = nullptr;
  int* foreign_d = nullptr;
  int* foreign_h = nullptr;
  int* status_h = nullptr;  // pinned landing word of check_status (one asynchronous copy + one synchronise per check)
  int* status_hd = nullptr; // the same word as the device sees it (hipHostGetDevicePointer): a kernel at the end of a call can leave the status there itself
  bool rehome = true;
  double adaptive_resort = 0.8, debt = 0.0;  // default budget: about one re-sort's cost
  int adaptive_min_steps = 4;
  int* nwork_d = nullptr;   // ranges[3 classes][2 splits][begin,end] of the work lists (tile_scan_block)
  int *dmg_first_d = nullptr, *dmg_last_d = nullptr;  // eigenerosion: run of every node in the I0-sorted particle list
  int *dmg_first0_d = nullptr, *dmg_last0_d = nullptr, *dmg_sorted0_d = nullptr;  // the same for the snapshot's closest nodes
  bool beps_snapshot = false;  // F_X0 / F_I00 hold the configuration of Initialize_Beps = true
  double* slab_d = nullptr; // P2G window slabs [ntiles][K2_SPLIT][1+ND][NW] (TileD::slab), deterministic mode only
  bool deterministic = false;
  // canonical (layer, closest node) order of every tile list each step (k_tile_order) for the LDS-atomic-bound K2 and
  // K3; the memory-bound K5 keeps the lists as binned.  13 us per step at 1 M particles.  A freshly sorted cloud has its
  // lists in that order already (0.685 vs 0.677 ms/step), but once particles have changed closest node -- 45 steps into
  // the bench cloud's fall -- K2 runs 0.231 instead of 0.300 ms and K3 0.260 instead of 0.304, and the stirred cloud of
  // DESIGN.md 0.84 instead of 0.95 ms/step
  int tile_ordering = 1;
  int band_lo = -(1 << 30), band_hi = 1 << 30;  // ghost bands: layers <= band_lo and >= band_hi are shared with neighbours
  int overlap = 0;          // halo exchanges: 0 blocking in place; 1 behind the interior tiles of the NEXT stage (split
                            // launches, two-phase callback); 2 behind the interior tiles of the SAME launch (library RCCL only)
  unsigned long long* phase_d = nullptr;
  double* vec_d = nullptr;  // scratch pool for host vectors of the a21 per-dof updates
  size_t vec_cap = 0;
  // nlps_gpu_lagrangian_evaluation with host vectors: device copies of Un_dt, Un_dt2, M, which do not change between the
  // evaluations of one SNES solve (NLPS_LAGR_SAME_STEP reuses them: two transfers per evaluation instead of five)
  double* lagr_d = nullptr;
  size_t lagr_cap = 0;
  bool lagr_valid = false;
  // tangent assembly, allocated on first use
  double* kst_d = nullptr;           // [nnodes][S][d*d]
  unsigned char* ktouched_d = nullptr;  // [nnodes][S]
  int *kcnt_d = nullptr, *koffs_d = nullptr;  // visited blocks per row node, exclusive scan
  int *khead_d = nullptr, *kng_d = nullptr;   // group heads of the I0-sorted particle list, group count
  void* kscan_tmp = nullptr;
  size_t kscan_bytes = 0;
  long long knnz_blocks = -1;
  bool tangent_grouped = true;  // one workgroup per closest node (false: one wave per particle, kept for comparison)
  bool tangent_symmetric = true;  // Neo-Hookean clouds: only the upper half of every row is assembled (nlps_gpu_debug_option "tangent_symmetric")
  bool ktan_sym = false;          // how the last nlps_gpu_tangent_assemble filled the stencil array (nlps_gpu_tangent_coo mirrors the rest)
  int* order_d;
  int* order2_d = nullptr;  // canonical tile lists (k_tile_order), allocated on first use

  nlps_halo_fn halo;
  void* halo_ctx;
  struct RcclHalo* rccl = nullptr;  // ghost-layer exchange over RCCL owned by the library (nlps_gpu_rccl_attach)
  bool rccl_wait_value = false;     // the attached exchange can be released through signal memory (overlap mode 2)

  bool timing;
  hipEvent_t ev[8];
  hipEvent_t evw[12] = {};  // brackets of the exchanges a step waits for (timing only)
  int nwait = 0;
  float ms[8];
  int slab_lo, slab_hi;
  int win_lo, win_hi;  // node window (layers of the slowest axis) the per-step nodal work is limited to
  int n0, nwn;         // first node / node count of the window
  int tile0, ntw;      // first tile / tile count of the window
};

#define HIPCHK(call)                                                                         \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess) {                                                                  \
      h->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      fprintf(stderr, "\033[1;31mError in nlps_gpu: %s\033[0m\n", h->err.c_str());           \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

static inline int nblk(int n, int b = BLK) { return n > 0 ? (n + b - 1) / b : 1; }
static int materialise_nodal(nlps_gpu* h);  // the nodal arrays the folded explicit step left unmade (defined with the step)

static bool is_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  hipError_t e = hipPointerGetAttributes(&a, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

static MatD make_mat(const nlps_material& m, int nd) {
  MatD d;
  memset(&d, 0, sizeof(d));
  d.type = m.type;
  d.E = m.E;
  d.nu = m.nu;
  d.G = m.E / (2.0 * (1.0 + m.nu));
  d.lame = m.nu * m.E / ((1 - m.nu * 2) * (1 + m.nu));
  d.K = m.E / (3.0 * (1.0 - 2.0 * m.nu));
  const double PI = 3.14159265358979323846;
  double rf = (PI / 180.0) * m.phi_deg, rd = (PI / 180.0) * m.psi_deg;
  auto sq = [](double a) { return a == 0.0 ? 0.0 : a * a; };
  if (nd == 2) {  // Drucker-Prager.c:362-368
    d.alpha_F = sqrt(2. / 3.) * tan(rf) / sqrt(3. + 4. * sq(tan(rf)));
    d.alpha_Q = sqrt(2. / 3.) * tan(rd) / sqrt(3. + 4. * sq(tan(rd)));
    d.beta_dp = sqrt(2. / 3.) * 3. / sqrt(3. + 4. * sq(tan(rf)));
  } else {  // :370-375
    d.alpha_F = sqrt(2 / 3.) * 2 * sin(rf) / (3 - sin(rf));
    d.alpha_Q = sqrt(2 / 3.) * 2 * sin(rd) / (3 - sin(rd));
    d.beta_dp = sqrt(2 / 3.) * 6 * cos(rf) / (3 - sin(rf));
  }
  d.kappa_0 = m.kappa_0;
  d.exp_param = m.exponent_ortiz;
  d.eps_0 = m.eps_0;
  d.p_ref = m.p_ref;
  d.H = m.hardening_modulus;
  d.theta = m.theta_voce;
  d.K_0 = m.K0_voce;
  d.K_inf = m.Kinf_voce;
  d.delta = m.delta_voce;
  d.Ceps = m.Ceps;
  d.ft = m.ft;
  d.heps = m.heps;
  d.wcrit = m.wcrit;
  d.Gf = m.Gf;
  if (m.type == NLPS_MAT_MATSUOKA_NAKAI || m.type == NLPS_MAT_LADE_DUNCAN) {  // one kernel law, two surfaces
    d.type = NLPS_KLAW_FRICTIONAL;
    d.surface = m.type == NLPS_MAT_LADE_DUNCAN;
    d.c_cotphi = rf > 0.0 ? m.cohesion / tan(rf) : 0.0;  // Matsuoka-Nakai.c:334-336
    d.alpha_b = m.alpha_borja;
    for (int i = 0; i < 3; i++) d.a_b[i] = m.a_borja[i];
  }
  return d;
}
static inline int klaw_of(int type) { return type >= NLPS_KLAW_FRICTIONAL ? NLPS_KLAW_FRICTIONAL : type; }

// h_avg exactly as compute_nodal_distance_local does it (Read_GramsBox.c:460-507): chain order sum
static void host_h_avg(const nlps_grid& G, const nlps_host::StencilTables& tab, std::vector<double>& out) {
  int nd = G.ndim;
  int n[3] = {G.n[0], G.n[1], nd == 3 ? G.n[2] : 1};
  size_t nn = (size_t)n[0] * n[1] * n[2];
  out.resize(nn);
  // per class: offsets sorted by chain position
  std::vector<std::vector<std::array<int, 3>>> chains(27);
  for (int cls = 0; cls < 27; cls++) {
    std::vector<std::pair<int, int>> v;
    for (int o = 0; o < 27; o++)
      if (tab.rank1[cls][o] != 255) v.push_back({tab.rank1[cls][o], o});
    std::sort(v.begin(), v.end());
    for (auto& pr : v) chains[cls].push_back({pr.second % 3 - 1, (pr.second / 3) % 3 - 1, pr.second / 9 - 1});
  }
  for (size_t I = 0; I < nn; I++) {
    int ijk[3] = {(int)(I % n[0]), (int)((I / n[0]) % n[1]), (int)(I / ((size_t)n[0] * n[1]))};
    int cls = 0, mul = 1;
    for (int a = 0; a < 3; a++) {
      int ca = a < nd ? nlps_host::class3(ijk[a], n[a]) : 1;
      cls += ca * mul;
      mul *= 3;
    }
    double avg = 0.0;
    int cnt = 0;
    for (auto& o : chains[cls]) {
      if (o[0] == 0 && o[1] == 0 && o[2] == 0) continue;
      double aux = 0.0;
      for (int a = 0; a < nd; a++) {
        double xb = G.origin[a] + G.h * (double)(ijk[a] + o[a]);
        double xa = G.origin[a] + G.h * (double)ijk[a];
        double d = xb - xa;
        aux += (d == 0.0 ? 0.0 : d * d);
      }
      avg += pow(aux, 0.5);
      cnt++;
    }
    out[I] = avg / (double)cnt;
  }
}

template <class T>
static int dev_alloc(nlps_gpu* h, T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T)));
  // default stream, like the uploads that may follow: on the handle's stream a caller-provided NON-BLOCKING
  // stream (torch's are) could run this after them and wipe the upload
  HIPCHK(hipMemset(*p, 0, n * sizeof(T)));
  return 0;
}

static int ensure_masked(nlps_gpu* h, size_t n) {
  if (n <= h->maskedA_cap) return 0;
  if (h->maskedA) HIPCHK(hipFree(h->maskedA));
  HIPCHK(hipMalloc((void**)&h->maskedA, n * sizeof(double)));
  h->maskedA_cap = n;
  return 0;
}

extern "C" int nlps_host_stencil_tables(int ndim, unsigned char* rank1, unsigned char* order2, unsigned char* count2,
                                        double* h_avg1) {
  if (ndim != 2 && ndim != 3) return 1;
  nlps_host::StencilTables t = nlps_host::build_tables(ndim);
  if (rank1) memcpy(rank1, t.rank1, sizeof(t.rank1));
  if (order2) memcpy(order2, t.order2, sizeof(t.order2));
  if (count2) memcpy(count2, t.count2, sizeof(t.count2));
  if (h_avg1) memcpy(h_avg1, t.h_avg1, sizeof(t.h_avg1));
  return 0;
}

extern "C" const char* nlps_gpu_last_error(const nlps_gpu* h) { return h ? h->err.c_str() : "null handle"; }

extern "C" int nlps_gpu_synchronize(nlps_gpu* h) {
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int nlps_gpu_set_halo_exchange(nlps_gpu* h, nlps_halo_fn fn, void* ctx) {
  h->halo = fn;
  h->halo_ctx = ctx;
  return 0;
}

static void apply_window(nlps_gpu* h, int lo, int hi) {
  const int nd = h->nd, nl = h->g.n[nd - 1], plane = h->g.nnodes / nl;
  const int TB = nd == 3 ? TileCfg<3>::TB : TileCfg<2>::TB;
  const int tiles_per_layer = h->ntiles / h->nt[nd - 1];
  h->win_lo = lo;
  h->win_hi = hi;
  h->n0 = lo * plane;
  h->nwn = (hi - lo + 1) * plane;
  h->tile0 = (lo / TB) * tiles_per_layer;
  h->ntw = (hi / TB - lo / TB + 1) * tiles_per_layer;
}

extern "C" int nlps_gpu_set_node_window(nlps_gpu* h, int layer_lo, int layer_hi) {
  const int nl = h->g.n[h->nd - 1];
  if (layer_lo < 0 || layer_hi >= nl || layer_lo > layer_hi) {
    h->err = "nlps_gpu_set_node_window: layers outside the grid";
    return 1;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  apply_window(h, layer_lo, layer_hi);
  // clean slate outside the new window (nothing resets those nodes any more); the nodal results of an explicit step
  // before this call are gone with it, made or not
  h->nodal_stale = false;
  const size_t nn = (size_t)h->g.nnodes, ND = (size_t)h->nd;
  HIPCHK(hipMemsetAsync(h->N.active, 0, nn, h->stream));
  HIPCHK(hipMemsetAsync(h->N.seed, 0, nn, h->stream));
  HIPCHK(hipMemsetAsync(h->N.nm, 0, nn * (1 + ND) * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(h->N.dU, 0, nn * ND * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(h->N.force, 0, nn * ND * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(h->N.accel, 0, nn * ND * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(h->N.reaction, 0, nn * ND * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(h->N.fixed, 0, nn * ND, h->stream));
  HIPCHK(hipMemsetAsync(h->tile_count_d, 0, ((size_t)h->ntiles + 1) * sizeof(int), h->stream));
  h->masks_valid = false;
  h->binned = false;
  h->ahead = false;
  return 0;
}

#if NLPS_PHASE_TIMING
extern "C" __attribute__((visibility("default"))) int nlps_gpu_debug_phases(nlps_gpu* h, unsigned long long* out, int reset) {
  HIPCHK(hipStreamSynchronize(h->stream));
  std::vector<unsigned long long> tmp(16 * 1024);
  HIPCHK(hipMemcpy(tmp.data(), h->phase_d, tmp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (int k = 0; k < 16; k++) {
    out[k] = 0;
    for (int b = 0; b < 1024; b++) out[k] += tmp[k + 16 * b];
  }
  if (reset) HIPCHK(hipMemset(h->phase_d, 0, tmp.size() * sizeof(unsigned long long)));
  return 0;
}
#endif

extern "C" int nlps_gpu_set_ghost_bands(nlps_gpu* h, int band_lo, int band_hi, int overlap) {
  if (materialise_nodal(h)) return 1;  // (the node ranges of the last step's nodal kernels are about to change)
  h->band_lo = band_lo;
  h->band_hi = band_hi;
  if (overlap == 2 && !h->rccl_wait_value) {
    h->err = "nlps_gpu_set_ghost_bands: overlap 2 needs the library's RCCL path (nlps_gpu_rccl_attach)";
    return 1;
  }
  h->overlap = overlap;
  return 0;
}

extern "C" int nlps_gpu_touched_layers(nlps_gpu* h, int* lo, int* hi) {
  *lo = h->slab_lo;
  *hi = h->slab_hi;
  return 0;
}

extern "C" int nlps_gpu_set_timing(nlps_gpu* h, int on) {
  h->timing = on != 0;
  return 0;
}
extern "C" int nlps_gpu_get_timing(nlps_gpu* h, float ms[8]) {
  for (int i = 0; i < 8; i++) ms[i] = h->ms[i];
  return 0;
}

static int upload_field(nlps_gpu* h, int f, int ncomp, const double* src, int stride, std::vector<double>& tmp,
                        const double* dflt_diag /*identity rows*/, double dflt) {
  // gathers component c of the caller's AoS rows into the sorted SoA slot
  int np = h->P.np;
  for (int c = 0; c < ncomp; c++) {
    for (int s = 0; s < np; s++) {
      int p = h->perm[s];
      tmp[s] = src ? src[(size_t)p * stride + c] : (dflt_diag ? dflt_diag[c] : dflt);
    }
    for (size_t s = np; s < h->P.npad; s++) tmp[s] = 0.0;
    HIPCHK(hipMemcpy(h->P.d + (size_t)(f + c) * h->P.npad, tmp.data(), h->P.npad * sizeof(double),
                     hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int nlps_gpu_create(nlps_gpu** out, const nlps_grid* grid, const nlps_params* prm,
                               const nlps_material* mats, int nmats, const nlps_particles* host, int nsteps,
                               void* hip_stream) {
  nlps_gpu* h = new nlps_gpu();
  *out = h;
  h->nd = grid->ndim;
  h->T = grid->ndim == 2 ? 5 : 9;
  h->nsteps = nsteps;
  h->halo = nullptr;
  h->halo_ctx = nullptr;
  h->timing = false;
  h->masks_valid = false;
  h->binned = false;
  h->maskedA = nullptr;
  h->maskedA_cap = 0;
  h->perm_d = nullptr;
  h->perm_dirty = false;
  h->resort_every = 50;
  h->steps_since_sort = 0;
  h->skey_d = h->skey2_d = nullptr;
  h->sval_d = h->sval2_d = nullptr;
  h->cub_tmp = nullptr;
  h->cub_tmp_bytes = 0;
  h->gather_tmp = nullptr;
  memset(h->ms, 0, sizeof(h->ms));
  if (grid->ndim != 2 && grid->ndim != 3) {
    h->err = "ndim must be 2 or 3";
    return 1;
  }
  for (int a = 0; a < grid->ndim; a++)
    if (grid->n[a] < 5) {
      h->err = "structured grid needs >= 5 nodes per axis";
      return 1;
    }
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) {
    h->err = "no HIP device: the MI355X path has no CPU fallback";
    return 1;
  }
  if (hip_stream) {
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
  } else {
    HIPCHK(hipStreamCreate(&h->stream));
    h->own_stream = true;
  }
  for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&h->ev[i]));

  GridD& g = h->g;
  g.nd = grid->ndim;
  for (int a = 0; a < 3; a++) {
    g.n[a] = a < g.nd ? grid->n[a] : 1;
    g.o[a] = a < g.nd ? grid->origin[a] : 0.0;
  }
  g.h = grid->h;
  g.nnodes = g.n[0] * g.n[1] * g.n[2];
  {
    int TB = g.nd == 3 ? TileCfg<3>::TB : TileCfg<2>::TB;
    for (int a = 0; a < 3; a++) h->nt[a] = a < g.nd ? (g.n[a] + TB - 1) / TB : 1;
    h->ntiles = h->nt[0] * h->nt[1] * h->nt[2];
  }
  apply_window(h, 0, g.n[g.nd - 1] - 1);
  h->hprm = *prm;
  h->prm.gamma_lme = prm->gamma_lme;
  h->prm.neg_log_tol_zero = -log(prm->tol_zero_lme);
  h->prm.tol_wrapper = prm->tol_wrapper_lme;
  h->prm.max_iter_lme = prm->max_iter_lme;
  h->prm.tol_radial = prm->tol_radial_returning;
  h->prm.max_iter_radial = prm->max_iter_radial_returning;
  h->P.erosion = prm->driver_eigenerosion != 0 || prm->driver_eigensoftening != 0;
  h->P.softening = prm->driver_eigensoftening != 0 && prm->driver_eigenerosion == 0;
#if NLPS_DEV  // developer builds only (tools/build_variant.sh): the shipped library reads no environment variable
  if (const char* e = getenv("NLPS_TILE_ORDERING")) h->tile_ordering = atoi(e);  // see k_tile_order
  if (const char* e = getenv("NLPS_RESORT_FROM_LISTS")) h->resort_from_lists = atoi(e);
  if (const char* e = getenv("NLPS_FUSE_SEARCH")) h->fuse_search = atoi(e);
  if (const char* e = getenv("NLPS_LAZY_NODAL")) h->lazy_nodal = atoi(e);
#endif
  {
    int dev = 0, ncu = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && ncu > 0)
      h->ncu = ncu;
  }

  h->tab = nlps_host::build_tables(g.nd);
  HIPCHK(hipMalloc((void**)&h->rank1_d, 27 * 27));
  HIPCHK(hipMemcpy(h->rank1_d, h->tab.rank1, 27 * 27, hipMemcpyHostToDevice));

  // nodal arrays
  size_t nn = (size_t)g.nnodes;
  int ND = g.nd;
  if (dev_alloc(h, &h->N.active, nn)) return 1;
  if (dev_alloc(h, &h->N.seed, nn)) return 1;
  if (dev_alloc(h, &h->N.nm, nn * (1 + ND))) return 1;
  if (dev_alloc(h, &h->N.dU, nn * ND)) return 1;
  if (dev_alloc(h, &h->N.force, nn * ND)) return 1;
  if (dev_alloc(h, &h->N.accel, nn * ND)) return 1;
  if (dev_alloc(h, &h->N.reaction, nn * ND)) return 1;
  if (dev_alloc(h, &h->N.fixed, nn * ND)) return 1;
  if (dev_alloc(h, &h->h_avg_d, nn)) return 1;
  if (dev_alloc(h, &h->n2m_d, nn)) return 1;
  if (dev_alloc(h, &h->d2m_d, nn * ND)) return 1;
  if (dev_alloc(h, &h->fixedm_d, nn * ND)) return 1;
  if (dev_alloc(h, &h->bsum_d, nn * ND / 1024 + 2)) return 1;
  if (dev_alloc(h, &h->total_d, 4)) return 1;
  if (dev_alloc(h, &h->gstatus_d, 4)) return 1;
  if (dev_alloc(h, &h->gridA, nn * 2 * ND)) return 1;
  if (dev_alloc(h, &h->gridB, nn * 4 * ND)) return 1;
  {
    std::vector<double> hv;
    if (grid->h_avg) hv.assign(grid->h_avg, grid->h_avg + nn);
    else host_h_avg(*grid, h->tab, hv);
    HIPCHK(hipMemcpy(h->h_avg_d, hv.data(), nn * sizeof(double), hipMemcpyHostToDevice));
    h->N.h_avg = h->h_avg_d;
    HIPCHK(hipMalloc((void**)&h->beta_t2_d, nn * sizeof(double4)));
    hipLaunchKernelGGL(k_node_tables, dim3(nblk((int)nn)), dim3(BLK), 0, 0, (int)nn, (const double*)h->h_avg_d,
                       h->prm.gamma_lme, h->prm.neg_log_tol_zero, h->beta_t2_d);
    HIPCHK(hipGetLastError());
    h->N.beta_t2 = h->beta_t2_d;
  }
  // materials
  h->nmats = nmats;
  h->uniform_law = nmats > 0 ? klaw_of(mats[0].type) : -1;
  h->law_present = 0;
  for (int i = 0; i < nmats; i++) {
    if (mats[i].type < 0 || mats[i].type > NLPS_MAT_LADE_DUNCAN) {
      h->err = "material type outside 0..5 (Neo-Hookean, Hencky, Drucker-Prager, Von-Mises, Matsuoka-Nakai, Lade-Duncan)";
      return 1;
    }
    if (klaw_of(mats[i].type) != klaw_of(mats[0].type)) h->uniform_law = -1;
    h->law_present |= 1 << klaw_of(mats[i].type);
  }
  {
    std::vector<MatD> md(nmats);
    for (int i = 0; i < nmats; i++) md[i] = make_mat(mats[i], g.nd);
    HIPCHK(hipMalloc((void**)&h->mats_d, sizeof(MatD) * nmats));
    HIPCHK(hipMemcpy(h->mats_d, md.data(), sizeof(MatD) * nmats, hipMemcpyHostToDevice));
  }

  // particles: sort by background-grid cell
  int np = host->np;
  h->P.np = np;
  // capacity: room for immigrants (nlps_gpu_migration_commit), fixed at create
  h->P.npad = ((size_t)np + std::max<size_t>((size_t)np / 4, 1024) + 255) / 256 * 256;
  h->perm.resize(np);
  std::iota(h->perm.begin(), h->perm.end(), 0);
  {
    std::vector<long long> key(np);
    std::vector<unsigned char> tile_laws(h->uniform_law < 0 ? (size_t)h->ntiles : 0, 0);
    int slab_axis = ND - 1;
    int lo = 1 << 30, hi = -1;
    const int TB = ND == 3 ? TileCfg<3>::TB : TileCfg<2>::TB;
    for (int p = 0; p < np; p++) {
      // Physical order = (tile of the closest node, corner type, closest node inside the tile).
      // "Corner type" = which corner of its cell the closest node is.  Particles that share a closest
      // node come from different cells, i.e. have different corner types, so any 64 consecutive
      // particles (one wave) have 64 DISTINCT closest nodes: their window accesses (LDS atomics of
      // the scatter, LDS reads of the gather) never collide on an address and spread over all banks.
      long long kt = 0, kn = 0, kc = 0, mt = 1, mn = 1, mc = 1;
      for (int a = 0; a < ND; a++) {
        int nc = g.n[a] - 1;
        double xi = (host->x_GC[(size_t)p * ND + a] - g.o[a]) / g.h;
        int c = (int)floor(xi);
        c = c < 0 ? 0 : (c > nc - 1 ? nc - 1 : c);
        int nd = (int)floor(xi + 0.5);
        nd = nd < 0 ? 0 : (nd > g.n[a] - 1 ? g.n[a] - 1 : nd);
        kt += mt * (nd / TB);
        mt *= h->nt[a];
        kn += mn * (nd % TB);
        mn *= TB;
        kc += mc * (nd > c ? 1 : 0);
        mc *= 2;
        if (a == slab_axis) {
          lo = std::min(lo, c);
          hi = std::max(hi, c + 1);
        }
      }
      const int mi = host->MatIdx ? host->MatIdx[p] : 0;
      if (mi < 0 || mi >= nmats) {
        h->err = "MatIdx outside the material table";
        return 1;
      }
      key[p] = (kt * mc + kc) * mn + kn;
      if (h->uniform_law < 0) {  // which laws meet in which tile (decides how K3 treats a cloud with several laws)
        unsigned char& m = tile_laws[(size_t)kt];
        m |= (unsigned char)(1 << klaw_of(mats[mi].type));
      }
    }
    if (h->uniform_law < 0) {
      size_t used = 0, mixed = 0;
      for (unsigned char m : tile_laws) {
        used += m != 0;
        mixed += (m & (m - 1)) != 0;
      }
      // blocks of different materials (a footing on soil): nearly every tile holds one law and K3 runs as one launch of
      // the single-law kernel per law, each taking its tiles (0 scratch, the speed of the uniform cloud).  Laws
      // interleaved particle by particle: every launch would touch every tile and every cache line (measured 0.68 ms
      // against 0.51 ms for the one kernel that dispatches on the law at run time), so that kernel stays for them.
      h->k3_per_law = used > 0 && 4 * mixed <= used;
      if (h->law_present & (1 << NLPS_KLAW_FRICTIONAL)) h->k3_per_law = 1;  // not in the dispatch kernel (stress_update)
    }
    std::stable_sort(h->perm.begin(), h->perm.end(), [&](int a, int b) { return key[a] < key[b]; });
    h->slab_lo = std::max(0, lo - 3);
    h->slab_hi = std::min(g.n[slab_axis] - 1, hi + 3);
  }
  HIPCHK(hipMalloc((void**)&h->P.d, (size_t)NFD * h->P.npad * sizeof(double)));
  HIPCHK(hipMemset(h->P.d, 0, (size_t)NFD * h->P.npad * sizeof(double)));
  if (dev_alloc(h, &h->P.I0, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.I0n, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.mat, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.nn, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.status, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.mlo, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.mhi, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.tile, h->P.npad)) return 1;
  if (dev_alloc(h, &h->P.rank, h->P.npad)) return 1;
  if (dev_alloc(h, &h->order_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->tile_count_d, (size_t)h->ntiles + 1)) return 1;
  if (dev_alloc(h, &h->tile_count2_d, (size_t)h->ntiles + 1)) return 1;
  if (dev_alloc(h, &h->tile_start_d, (size_t)h->ntiles + 1)) return 1;
  if (dev_alloc(h, &h->work1_d, (size_t)h->ntiles)) return 1;
  if (dev_alloc(h, &h->work2_d, (size_t)h->ntiles * 2)) return 1;
  if (dev_alloc(h, &h->nwork_d, 16)) return 1;
#if NLPS_DEV
  if (const char* e = getenv("NLPS_ADAPTIVE_RESORT")) h->adaptive_resort = atof(e);  // (0 = off)
  if (const char* e = getenv("NLPS_NODE_LISTS")) h->node_lists_on = atoi(e);
#endif
  if (dev_alloc(h, &h->node_cnt_d, (size_t)h->g.nnodes)) return 1;
  if (dev_alloc(h, &h->nrank_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->tile_cursor_d, (size_t)h->ntiles + 1)) return 1;
  {
    const size_t nnw = h->g.nd == 3 ? TileTabCfg<3>::NNW : TileTabCfg<2>::NNW;
    if (dev_alloc(h, &h->tabo_d, (size_t)h->ntiles * TileTab::LMAX * nnw)) return 1;
    if (dev_alloc(h, &h->tabm_d, (size_t)h->ntiles * TileTab::LMAX * nnw)) return 1;
  }
  if (dev_alloc(h, &h->home_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->foreign_d, 64 * 32)) return 1;
  HIPCHK(hipHostMalloc((void**)&h->foreign_h, sizeof(int), hipHostMallocDefault));
  *h->foreign_h = 0;
  HIPCHK(hipHostMalloc((void**)&h->status_h, sizeof(int), hipHostMallocDefault));
  *h->status_h = 0;
  if (hipHostGetDevicePointer((void**)&h->status_hd, h->status_h, 0) != hipSuccess) h->status_hd = nullptr;  // (then check_status copies)
#if NLPS_PHASE_TIMING
  if (dev_alloc(h, &h->phase_d, 16 * 1024)) return 1;
#endif
  {
    std::vector<double> tmp(h->P.npad);
    int T = h->T;
    double idrow[9] = {0};
    idrow[0] = 1.0;
    idrow[ND + 1] = 1.0;
    idrow[T - 1] = 1.0;
    if (upload_field(h, F_X, ND, host->x_GC, ND, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_DIS, ND, host->dis, ND, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_VEL, ND, host->vel, ND, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_ACC, ND, host->acc, ND, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_FN, T, host->F_n, T, tmp, idrow, 0.0)) return 1;
    if (upload_field(h, F_FN1, T, host->F_n1 ? host->F_n1 : host->F_n, T, tmp, idrow, 0.0)) return 1;
    if (upload_field(h, F_DF, T, host->DF, T, tmp, idrow, 0.0)) return 1;
    if (upload_field(h, F_TAU, T, host->Stress, T, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_BEN, T, host->b_e_n, T, tmp, idrow, 0.0)) return 1;
    if (upload_field(h, F_BEN1, T, host->b_e_n1 ? host->b_e_n1 : host->b_e_n, T, tmp, idrow, 0.0)) return 1;
    if (upload_field(h, F_JN, 1, host->J_n, 1, tmp, nullptr, 1.0)) return 1;
    if (upload_field(h, F_JN1, 1, host->J_n1 ? host->J_n1 : host->J_n, 1, tmp, nullptr, 1.0)) return 1;
    if (upload_field(h, F_RHO, 1, host->rho, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_MASS, 1, host->mass, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_VOL0, 1, host->Vol_0, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_W, 1, host->W, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_KN, 1, host->Kappa_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_KN1, 1, host->Kappa_n1 ? host->Kappa_n1 : host->Kappa_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_EN, 1, host->EPS_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_EN1, 1, host->EPS_n1 ? host->EPS_n1 : host->EPS_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_LAM, ND, host->lambda, ND, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_BETA, 1, host->Beta, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_LAMP, ND, host->lambda, ND, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_BACK, 3, host->Back_stress, 3, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_DMG, 1, host->Damage_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_DMG1, 1, host->Damage_n1 ? host->Damage_n1 : host->Damage_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_STRF, 1, host->Strain_f_n, 1, tmp, nullptr, 0.0)) return 1;
    if (upload_field(h, F_STRF1, 1, host->Strain_f_n1 ? host->Strain_f_n1 : host->Strain_f_n, 1, tmp, nullptr, 0.0)) return 1;
    if (h->P.erosion) h->level_b_fields = true;  // the damage fields travel with the re-sort
    if (host->dt_F_n || host->dt_F_n1 || host->dt_DF) h->level_b_fields = true;
    if (host->dt_F_n && upload_field(h, F_DTFN, T, host->dt_F_n, T, tmp, nullptr, 0.0)) return 1;
    if (host->dt_F_n1 && upload_field(h, F_DTFN1, T, host->dt_F_n1, T, tmp, nullptr, 0.0)) return 1;
    if (host->dt_DF && upload_field(h, F_DTDF, T, host->dt_DF, T, tmp, nullptr, 0.0)) return 1;
    std::vector<int> it(h->P.npad, 0);
    for (int s = 0; s < np; s++) it[s] = host->MatIdx ? host->MatIdx[h->perm[s]] : 0;
    HIPCHK(hipMemcpy(h->P.mat, it.data(), h->P.npad * sizeof(int), hipMemcpyHostToDevice));
    if (host->I0) {
      for (int s = 0; s < np; s++) it[s] = host->I0[h->perm[s]];
      HIPCHK(hipMemcpy(h->P.I0, it.data(), h->P.npad * sizeof(int), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(h->P.I0n, it.data(), h->P.npad * sizeof(int), hipMemcpyHostToDevice));
    }
  }
  // re-sort buffers
  if (dev_alloc(h, &h->perm_d, h->P.npad)) return 1;
  HIPCHK(hipMemcpy(h->perm_d, h->perm.data(), (size_t)np * sizeof(int), hipMemcpyHostToDevice));
  if (dev_alloc(h, &h->gid_d, h->P.npad)) return 1;
  HIPCHK(hipMemcpy(h->gid_d, h->perm.data(), (size_t)np * sizeof(int), hipMemcpyHostToDevice));
  if (dev_alloc(h, &h->leaving_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->mig_slot_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->mig_cnt_d, 2)) return 1;
  if (dev_alloc(h, &h->skey_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->skey2_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->sval_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->sval2_d, h->P.npad)) return 1;
  if (dev_alloc(h, &h->gather_tmp, h->P.npad)) return 1;
  HIPCHK(hipcub::DeviceRadixSort::SortPairs(nullptr, h->cub_tmp_bytes, h->skey_d, h->skey2_d, h->sval_d, h->sval2_d,
                                            (int)h->P.npad, 0, 64, h->stream));
  HIPCHK(hipMalloc(&h->cub_tmp, h->cub_tmp_bytes + 16));
  HIPCHK(hipStreamSynchronize(h->stream));
  // the uploads above are ordered on the default stream only: a caller-provided non-blocking stream (torch
  // streams are) would not wait for their DMA
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

// Physical re-sort of every particle array by (tile of I0, corner type, I0 in tile).
// live_only: called at the head of an explicit step -- the fields that step rewrites in full before anything reads them
// (d_dis, the n+1 slots of F and b_e, DF, tau, J_n+1, W, kappa_n+1, eps_n+1: 43 of the 89 components) are not moved.
static int resort(nlps_gpu* h, const unsigned char* leaving = nullptr, bool live_only = false) {
  const int np = h->P.np;
  if (np == 0) return 0;
  // The tile lists of the last step in canonical order (k_tile_order: layer r = the r-th particle of every closest node,
  // nodes in lattice order) ARE the memory order the kernels want -- each wave then reads 64 consecutive slots that hit
  // 64 distinct window rows -- so the periodic re-sort of the fused step takes them as its permutation: no keys, no
  // radix sort.  Otherwise (first sort, migration, level-B callers): sort by (tile, corner type, node).
  const bool from_lists = !leaving && live_only && h->binned && h->order2_d && (h->tile_ordering || h->deterministic) &&
                          h->resort_from_lists && h->ntw > 0;
  if (!from_lists) {
    TileCnt tc;
    for (int a = 0; a < 3; a++) tc.nt[a] = h->nt[a];
    tc.count = nullptr;
    tc.home = nullptr;
    tc.node_cnt = nullptr;
    if (h->nd == 2) hipLaunchKernelGGL(k_sort_keys<2>, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->P, h->g, tc, h->skey_d, h->sval_d, leaving, h->mats_d);
    else hipLaunchKernelGGL(k_sort_keys<3>, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->P, h->g, tc, h->skey_d, h->sval_d, leaving, h->mats_d);
    HIPCHK(hipGetLastError());
    size_t bytes = h->cub_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(h->cub_tmp, bytes, h->skey_d, h->skey2_d, h->sval_d, h->sval2_d, np, 0, 64,
                                              h->stream));
  }
  if (from_lists) {  // particles that are in no list go behind the lists (k_mark_listed / k_append_unlisted)
    unsigned char* listed = reinterpret_cast<unsigned char*>(h->gather_tmp);  // [npad] bytes of the gather scratch, idle here
    HIPCHK(hipMemsetAsync(h->mig_cnt_d, 0, sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(listed, 0, (size_t)np, h->stream));
    const int last = h->tile0 + h->ntw - 1;
    hipLaunchKernelGGL(k_mark_listed, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, (const int*)h->tile_start_d + last,
                       (const int*)h->tile_count_d + last, (const int*)h->order2_d, listed);
    hipLaunchKernelGGL(k_append_unlisted, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, (const unsigned char*)listed,
                       (const int*)h->tile_start_d + last, (const int*)h->tile_count_d + last, h->mig_cnt_d, h->order2_d);
  }
  const int* idx = from_lists ? h->order2_d : h->sval2_d;  // new slot -> old slot
  const size_t npad = h->P.npad;
  const int nf = h->level_b_fields ? (int)NFD : (int)F_CEP;  // C_ep and the rate tensors only exist for level B
  // the field block moves into its twin in one launch and the two swap roles (no copy back; the twin costs a second
  // NFD x npad block of HBM, allocated at the first re-sort)
  if (!h->Pd_alt) {
    HIPCHK(hipMalloc((void**)&h->Pd_alt, (size_t)NFD * npad * sizeof(double)));
    HIPCHK(hipMemsetAsync(h->Pd_alt, 0, (size_t)NFD * npad * sizeof(double), h->stream));
  }
  if (live_only && !h->level_b_fields) {
    // contiguous runs of live components (enum at the top of this file; F and b_e by their current n slots)
    const int fn = fFN(h->P), ben = fBEN(h->P);
    const int runs[][2] = {{F_X, 12}, {fn, 9}, {ben, 9}, {F_JN, 1}, {F_RHO, 3}, {F_KN, 1}, {F_EN, 1}, {F_LAM, 11}};
    FieldList fl;
    fl.n = 0;
    for (auto& r : runs)
      for (int q = 0; q < r[1]; q++) fl.f[fl.n++] = (unsigned char)(r[0] + q);
    if (h->nd == 2) {  // the zz slots of F_n+1 and DF are never written by the 2-D kernels (they stay 1 from the upload)
      fl.f[fl.n++] = (unsigned char)(fFN1(h->P) + 4);
      fl.f[fl.n++] = (unsigned char)(F_DF + 4);
    }
    static_assert((int)NFD <= 255, "component indices travel as bytes");
    hipLaunchKernelGGL(k_gather_field_list, dim3(nblk(np), (fl.n + GATHER_FIELDS - 1) / GATHER_FIELDS), dim3(BLK), 0, h->stream,
                       h->Pd_alt, (const double*)h->P.d, idx, np, npad, fl);
  } else {
    hipLaunchKernelGGL(k_gather_fields, dim3(nblk(np), (nf + GATHER_FIELDS - 1) / GATHER_FIELDS), dim3(BLK), 0, h->stream,
                       h->Pd_alt, (const double*)h->P.d, idx, np, npad, nf);
  }
  std::swap(h->P.d, h->Pd_alt);
  {
    // (the twin block now holds the OLD field values, which nothing reads any more: its head is the scratch of the
    // integer arrays -- 2 x npad doubles for the mask words, then 7 x npad ints)
    SmallArrays A;
    int* iarr[7] = {h->P.I0, h->P.I0n, h->P.mat, h->P.nn, h->P.status, h->perm_d, h->gid_d};
    for (int a = 0; a < 7; a++) A.ia[a] = iarr[a];
    A.ua[0] = h->P.mlo;
    A.ua[1] = h->P.mhi;
    static_assert((int)NFD >= 6, "the scratch of the integer arrays needs 2 + 3.5 components of the twin block");
    unsigned long long* su = reinterpret_cast<unsigned long long*>(h->Pd_alt);
    int* si = reinterpret_cast<int*>(h->Pd_alt + 2 * npad);
    hipLaunchKernelGGL(k_gather_small, dim3(nblk(np)), dim3(BLK), 0, h->stream, A, idx, np, npad, si, su);
    hipLaunchKernelGGL(k_copy_small, dim3(nblk(np)), dim3(BLK), 0, h->stream, A, np, npad, (const int*)si,
                       (const unsigned long long*)su);
  }
  HIPCHK(hipGetLastError());
  h->perm_dirty = true;
  h->binned = false;
  h->ahead = false;  // (the per-particle tile / rank of a search done ahead belong to the old slots)
  h->steps_since_sort = 0;
  h->rehome = true;  // the slots have new owners: the next search records their tiles
  h->debt = 0.0;
  return 0;
}

static int refresh_perm(nlps_gpu* h) {
  if (!h->perm_dirty) return 0;
  HIPCHK(hipStreamSynchronize(h->stream));
  const int np = h->P.np;
  h->perm.resize(np);
  if (!h->migrated) {
    HIPCHK(hipMemcpy(h->perm.data(), h->perm_d, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
  } else if (np > 0) {
    // after a migration the caller's original order is gone: rows come back in ascending global id
    std::vector<int> gid(np), idx(np);
    HIPCHK(hipMemcpy(gid.data(), h->gid_d, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return gid[a] < gid[b]; });
    for (int r = 0; r < np; r++) h->perm[idx[r]] = r;
  }
  h->perm_dirty = false;
  return 0;
}

extern "C" int nlps_gpu_num_particles(nlps_gpu* h, int* np) {
  *np = h->P.np;
  return 0;
}

extern "C" int nlps_gpu_set_particle_ids(nlps_gpu* h, const int* ids) {
  if (refresh_perm(h)) return 1;
  const int np = h->P.np;
  std::vector<int> g(np);
  for (int s = 0; s < np; s++) g[s] = ids[h->perm[s]];
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->gid_d, g.data(), (size_t)np * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

extern "C" int nlps_gpu_download_ids(nlps_gpu* h, int* ids) {
  if (refresh_perm(h)) return 1;
  const int np = h->P.np;
  std::vector<int> g(np);
  HIPCHK(hipStreamSynchronize(h->stream));
  if (np) HIPCHK(hipMemcpy(g.data(), h->gid_d, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
  for (int s = 0; s < np; s++) ids[h->perm[s]] = g[s];
  return 0;
}

// Migration, step 1: particles whose closest node lies below layer keep_lo leave "down", above keep_hi "up";
// their packed rows are written to two device buffers owned by the handle.
extern "C" int nlps_gpu_migration_select(nlps_gpu* h, int keep_lo, int keep_hi, int* n_down, int* n_up, int* row_words,
                                         void** down_rows, void** up_rows) {
  const int np = h->P.np;
  HIPCHK(hipMemsetAsync(h->mig_cnt_d, 0, 2 * sizeof(int), h->stream));
  HIPCHK(hipMemsetAsync(h->leaving_d, 0, h->P.npad, h->stream));
  if (np > 0) LAUNCH_ND((k_mig_flag<2>), (k_mig_flag<3>), nblk(np), h->P, h->g, keep_lo, keep_hi, h->leaving_d, h->mig_slot_d, h->mig_cnt_d);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(h->mig_n, h->mig_cnt_d, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->mig_down_d) (void)hipFree(h->mig_down_d);
  if (h->mig_up_d) (void)hipFree(h->mig_up_d);
  h->mig_down_d = h->mig_up_d = nullptr;
  HIPCHK(hipMalloc((void**)&h->mig_down_d, ((size_t)h->mig_n[0] + 1) * MIG_WORDS * sizeof(double)));
  HIPCHK(hipMalloc((void**)&h->mig_up_d, ((size_t)h->mig_n[1] + 1) * MIG_WORDS * sizeof(double)));
  if (h->mig_n[0] + h->mig_n[1] > 0) {
    hipLaunchKernelGGL(k_mig_pack, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->P, h->leaving_d, h->mig_slot_d, h->perm_d,
                       h->gid_d, h->mig_down_d, h->mig_up_d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  *n_down = h->mig_n[0];
  *n_up = h->mig_n[1];
  *row_words = MIG_WORDS;
  if (down_rows) *down_rows = h->mig_down_d;
  if (up_rows) *up_rows = h->mig_up_d;
  h->mig_selected = true;
  return 0;
}

// Migration, step 2: the selected particles leave, the immigrants (packed rows from the neighbours, host or device
// pointers) join, the arrays are re-sorted and the next search re-bins everything.
extern "C" int nlps_gpu_migration_commit(nlps_gpu* h, const void* rows_a, int n_a, const void* rows_b, int n_b) {
  if (!h->mig_selected) {
    h->err = "nlps_gpu_migration_commit: call nlps_gpu_migration_select() first";
    return 1;
  }
  const int np = h->P.np, n_in = n_a + n_b, n_out = h->mig_n[0] + h->mig_n[1];
  if ((size_t)np + n_in > h->P.npad) {
    h->err = "nlps_gpu_migration_commit: more immigrants than the capacity reserved at create (np + max(np/4, 1024))";
    return 1;
  }
  h->mig_selected = false;
  if (n_in == 0 && n_out == 0) return 0;
  const void* src[2] = {rows_a, rows_b};
  const int cnt[2] = {n_a, n_b};
  int first = np;
  for (int k = 0; k < 2; k++) {
    if (cnt[k] <= 0) continue;
    double* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, (size_t)cnt[k] * MIG_WORDS * sizeof(double)));
    HIPCHK(hipMemcpyAsync(tmp, src[k], (size_t)cnt[k] * MIG_WORDS * sizeof(double), hipMemcpyDefault, h->stream));
    hipLaunchKernelGGL(k_mig_unpack, dim3(nblk(cnt[k])), dim3(BLK), 0, h->stream, h->P, first, cnt[k], tmp, h->perm_d,
                       h->gid_d, h->leaving_d);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    (void)hipFree(tmp);
    first += cnt[k];
  }
  h->P.np = np + n_in;
  h->searched = false;  // the immigrants' closest nodes are those of their last search: everyone searches again
  if (resort(h, h->leaving_d)) return 1;  // emigrants sort to the end ...
  h->P.np = np + n_in - n_out;           // ... and fall off
  h->migrated = true;
  h->perm_dirty = true;
  h->masks_valid = false;
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}

extern "C" int nlps_gpu_resort(nlps_gpu* h) { return resort(h); }
extern "C" __attribute__((visibility("default"))) int nlps_gpu_debug_set_tile_ordering(nlps_gpu* h, int on) {
  h->tile_ordering = on;  // developer switch (tools/kbench.py --no-order)
  h->ahead = false;
  return 0;
}
// Developer / test switches as an explicit call (never read from the environment in the shipped build): which of the
// equivalent launch forms a handle uses.  Not part of include/nlps_gpu.h.  Results do not depend on any of them.
extern "C" __attribute__((visibility("default"))) int nlps_gpu_debug_option(nlps_gpu* h, const char* name, double value) {
  const std::string k(name ? name : "");
  if (k == "lazy_nodal") h->lazy_nodal = (int)value;             // folded explicit step: 0 never, 1 below 2 M particles, 2 always
  else if (k == "fuse_search") h->fuse_search = (int)value;      // the next step's search on K5: 0 off, 1 with binning, 2 without
  else if (k == "resort_from_lists") h->resort_from_lists = (int)value;
  else if (k == "node_lists") h->node_lists_on = (int)value;
  else if (k == "tile_ordering") h->tile_ordering = (int)value;
  else if (k == "defer_ranks") h->defer_ranks = (int)value;  // the riding search only counts, k_fill_orders hands out the ranks
  else if (k == "tangent_symmetric") h->tangent_symmetric = value != 0;  // Neo-Hookean clouds: half rows + mirror (1) or every pair (0)
  else {
    h->err = "nlps_gpu_debug_option: unknown option " + k;
    return 1;
  }
  h->ahead = false;  // (lists made under another form are not reused)
  return 0;
}
extern "C" int nlps_gpu_set_law_launch_mode(nlps_gpu* h, int mode) {
  if (mode != 1 && mode != 2) {
    h->err = "nlps_gpu_set_law_launch_mode: 1 = one launch per law, 2 = one kernel dispatching on the law";
    return 1;
  }
  if (mode == 2 && (h->law_present & (1 << NLPS_KLAW_FRICTIONAL))) {
    h->err = "nlps_gpu_set_law_launch_mode: the dispatch kernel does not hold Matsuoka-Nakai / Lade-Duncan";
    return 1;
  }
  h->k3_per_law = mode == 1;
  return 0;
}
extern "C" int nlps_gpu_set_deterministic(nlps_gpu* h, int on) {
  h->deterministic = on != 0;
  h->ahead = false;
  h->rehome = true;
  h->debt = 0.0;
  return 0;
}
extern "C" int nlps_gpu_set_resort_interval(nlps_gpu* h, int every_n_steps) {
  h->resort_every = every_n_steps;
  h->steps_since_sort = 0;  // the interval counts from this call
  return 0;
}
extern "C" int nlps_gpu_set_adaptive_resort(nlps_gpu* h, double budget, int min_steps) {
  if (budget < 0.0 || min_steps < 1) {
    h->err = "nlps_gpu_set_adaptive_resort: budget >= 0 (0 = off), min_steps >= 1";
    return 1;
  }
  h->adaptive_resort = budget;
  h->adaptive_min_steps = min_steps;
  h->ahead = false;
  h->rehome = true;
  h->debt = 0.0;
  return 0;
}
extern "C" __attribute__((visibility("default"))) int nlps_gpu_debug_displaced(nlps_gpu* h, int* count, double* debt) {
  HIPCHK(hipStreamSynchronize(h->stream));  // developer read-out: the count of the last completed search stage
  *count = h->foreign_h ? *(volatile int*)h->foreign_h : 0;
  *debt = h->debt;
  return 0;
}

extern "C" int nlps_gpu_destroy(nlps_gpu* h) {
  if (!h) return 0;
  (void)nlps_gpu_rccl_detach(h);
  (void)hipStreamSynchronize(h->stream);
  void* ptrs[] = {h->P.d, h->Pd_alt, h->P.I0, h->P.I0n, h->P.mat, h->P.nn, h->P.status, h->P.mlo, h->P.mhi, h->N.active, h->N.seed, h->N.nm,
                  h->N.dU, h->N.force, h->N.accel, h->N.reaction, h->N.fixed, h->h_avg_d, h->beta_t2_d, h->n2m_d, h->d2m_d, h->canon_d, h->mask_flags_d, h->mask_idx_d,
                  h->fixedm_d, h->bsum_d, h->total_d, h->gstatus_d, h->gridA, h->gridB, h->maskedA, h->mats_d,
                  h->rank1_d, h->P.tile, h->P.rank, h->order_d, h->order2_d, h->tile_count_d, h->tile_count2_d, h->tile_start_d, h->work1_d, h->work2_d, h->nwork_d, h->slab_d, h->dmg_first_d, h->dmg_last_d, h->dmg_first0_d, h->dmg_last0_d, h->dmg_sorted0_d, h->perm_d, h->skey_d, h->skey2_d, h->sval_d, h->sval2_d,
                  h->gather_tmp, h->cub_tmp, h->gid_d, h->leaving_d, h->mig_slot_d, h->mig_cnt_d, h->mig_down_d, h->mig_up_d, h->kst_d, h->ktouched_d, h->kcnt_d, h->koffs_d, h->kscan_tmp, h->khead_d, h->kng_d, h->vec_d, h->bcmask_d, h->home_d, h->foreign_d, h->node_cnt_d, h->nrank_d, h->tabo_d, h->tabm_d, h->tile_cursor_d};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& b : h->bcs)
    if (b.dnodes) (void)hipFree(b.dnodes);
  if (h->foreign_h) (void)hipHostFree(h->foreign_h);
  if (h->lagr_d) (void)hipFree(h->lagr_d);
  if (h->status_h) (void)hipHostFree(h->status_h);
  for (int i = 0; i < 8; i++) (void)hipEventDestroy(h->ev[i]);
  for (hipEvent_t e : h->evw)
    if (e) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return 0;
}

static int download_field(nlps_gpu* h, int f, int ncomp, double* dst, int stride, std::vector<double>& tmp) {
  if (!dst) return 0;
  int np = h->P.np;
  for (int c = 0; c < ncomp; c++) {
    HIPCHK(hipMemcpy(tmp.data(), h->P.d + (size_t)(f + c) * h->P.npad, (size_t)np * sizeof(double),
                     hipMemcpyDeviceToHost));
    for (int s = 0; s < np; s++) dst[(size_t)h->perm[s] * stride + c] = tmp[s];
  }
  return 0;
}

static int materialise_roll(nlps_gpu* h);

extern "C" int nlps_gpu_download_state(nlps_gpu* h, nlps_particles* o) {
  if (materialise_roll(h)) return 1;
  HIPCHK(hipStreamSynchronize(h->stream));
  if (refresh_perm(h)) return 1;
  int ND = h->nd, T = h->T, np = h->P.np;
  std::vector<double> tmp(h->P.npad);
  if (download_field(h, F_X, ND, o->x_GC, ND, tmp)) return 1;
  if (download_field(h, F_DIS, ND, o->dis, ND, tmp)) return 1;
  if (download_field(h, F_VEL, ND, o->vel, ND, tmp)) return 1;
  if (download_field(h, F_ACC, ND, o->acc, ND, tmp)) return 1;
  if (download_field(h, fFN(h->P), T, o->F_n, T, tmp)) return 1;
  if (download_field(h, fFN1(h->P), T, o->F_n1, T, tmp)) return 1;
  if (download_field(h, F_DF, T, o->DF, T, tmp)) return 1;
  if (download_field(h, F_TAU, T, o->Stress, T, tmp)) return 1;
  if (download_field(h, fBEN(h->P), T, o->b_e_n, T, tmp)) return 1;
  if (download_field(h, fBEN1(h->P), T, o->b_e_n1, T, tmp)) return 1;
  if (download_field(h, F_JN, 1, o->J_n, 1, tmp)) return 1;
  if (download_field(h, F_JN1, 1, o->J_n1, 1, tmp)) return 1;
  if (download_field(h, F_RHO, 1, o->rho, 1, tmp)) return 1;
  if (download_field(h, F_MASS, 1, o->mass, 1, tmp)) return 1;
  if (download_field(h, F_VOL0, 1, o->Vol_0, 1, tmp)) return 1;
  if (download_field(h, F_W, 1, o->W, 1, tmp)) return 1;
  if (download_field(h, F_KN, 1, o->Kappa_n, 1, tmp)) return 1;
  if (download_field(h, F_KN1, 1, o->Kappa_n1, 1, tmp)) return 1;
  if (download_field(h, F_EN, 1, o->EPS_n, 1, tmp)) return 1;
  if (download_field(h, F_EN1, 1, o->EPS_n1, 1, tmp)) return 1;
  if (download_field(h, F_LAM, ND, o->lambda, ND, tmp)) return 1;
  if (download_field(h, F_BETA, 1, o->Beta, 1, tmp)) return 1;
  if (download_field(h, F_DTFN, T, o->dt_F_n, T, tmp)) return 1;
  if (download_field(h, F_DTFN1, T, o->dt_F_n1, T, tmp)) return 1;
  if (download_field(h, F_DTDF, T, o->dt_DF, T, tmp)) return 1;
  if (download_field(h, F_CEP, ND * ND, o->C_ep, ND * ND, tmp)) return 1;
  if (download_field(h, F_BACK, 3, o->Back_stress, 3, tmp)) return 1;
  if (download_field(h, F_DMG, 1, o->Damage_n, 1, tmp)) return 1;
  if (download_field(h, F_DMG1, 1, o->Damage_n1, 1, tmp)) return 1;
  if (download_field(h, F_STRF, 1, o->Strain_f_n, 1, tmp)) return 1;
  if (download_field(h, F_STRF1, 1, o->Strain_f_n1, 1, tmp)) return 1;
  if (o->I0) {
    std::vector<int> it(np);
    HIPCHK(hipMemcpy(it.data(), h->P.I0, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
    for (int s = 0; s < np; s++) o->I0[h->perm[s]] = it[s];
  }
  return 0;
}

extern "C" int nlps_gpu_download_lists(nlps_gpu* h, int* nn_out, int* list) {
  HIPCHK(hipStreamSynchronize(h->stream));
  if (refresh_perm(h)) return 1;
  int np = h->P.np, ND = h->nd;
  std::vector<int> I0(np);
  std::vector<u64> lo(np), hi(np);
  HIPCHK(hipMemcpy(I0.data(), h->P.I0, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(lo.data(), h->P.mlo, (size_t)np * sizeof(u64), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hi.data(), h->P.mhi, (size_t)np * sizeof(u64), hipMemcpyDeviceToHost));
  const GridD& g = h->g;
  for (int s = 0; s < np; s++) {
    int p = h->perm[s];
    int ijk[3] = {I0[s] % g.n[0], (I0[s] / g.n[0]) % g.n[1], I0[s] / (g.n[0] * g.n[1])};
    int cls = 0, mul = 1;
    for (int a = 0; a < 3; a++) {
      cls += (a < ND ? nlps_host::class5(ijk[a], g.n[a]) : 2) * mul;
      mul *= 5;
    }
    // walk NodalLocality[I0] in chain order, keep members, then reverse (prepend-push, LME.c:1077)
    int tmp[NLPS_MAXNB], nt = 0;
    for (int q = 0; q < h->tab.count2[cls]; q++) {
      int b = h->tab.order2[cls][q];
      bool on = b < 64 ? ((lo[s] >> b) & 1ull) : ((hi[s] >> (b - 64)) & 1ull);
      if (!on) continue;
      int i = b % 5, j = (b / 5) % 5, k = b / 25;
      tmp[nt++] = I0[s] + (i - 2) + g.n[0] * ((j - 2) + (ND == 3 ? g.n[1] * (k - 2) : 0));
    }
    nn_out[p] = nt;
    for (int a = 0; a < nt; a++) list[(size_t)p * NLPS_MAXNB + a] = tmp[nt - 1 - a];
    for (int a = nt; a < NLPS_MAXNB; a++) list[(size_t)p * NLPS_MAXNB + a] = -1;
  }
  return 0;
}

static int check_status(nlps_gpu* h, int fatal_mask, const char* where, bool mirrored = false);
// Level A: the shape functions themselves.  One thread per requested particle (device slot): p_a = e_a / Z for the 5^d
// stencil slots (0 for non-members) and dp_a = -p_a J^-1 l_a (LME.c:836-891: r and J from the same p), written per SLOT;
// the host puts them into the particle's list order (the walk of nlps_gpu_download_lists).
template <int ND>
__global__ void k_shape_slots(PView P, GridD g, const int* __restrict__ slots, int n, double* __restrict__ Ns,
                              double* __restrict__ dNs, int* __restrict__ gstatus) {
  constexpr int NS = (ND == 3) ? 125 : 25, KN = Lme<ND>::KN;
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int p = slots[q];
  double* Nq = Ns + (size_t)q * NS;
  double* dq = dNs + (size_t)q * NS * ND;
  for (int b = 0; b < NS; b++) {
    Nq[b] = 0.0;
    for (int a = 0; a < ND; a++) dq[b * ND + a] = 0.0;
  }
  Lme<ND> c;
  double lam[ND], beta;
  if (!load_lme<ND>(P, g, p, c, lam, beta)) return;
  // plain sums in slot order (not the wave-cooperative row form of the step kernels: lanes hold unrelated particles here)
  double Z = 0.0;
  for (int b = 0; b < NS; b++)
    if (c.on(b)) Z += c.ex[b % 5] * c.ey[(b / 5) % 5] * ((ND == 3) ? c.ez[(b / 25) % KN] : 1.0);
  const double Zinv = 1.0 / Z;
  double r[ND], J[ND * ND], Jm1[ND * ND];
  for (int a = 0; a < ND; a++) r[a] = 0.0;
  for (int a = 0; a < ND * ND; a++) J[a] = 0.0;
  for (int b = 0; b < NS; b++) {
    if (!c.on(b)) continue;
    const int i = b % 5, j = (b / 5) % 5, k = b / 25;
    const double pa = c.ex[i] * c.ey[j] * ((ND == 3) ? c.ez[k % KN] : 1.0) * Zinv;
    const double l[3] = {c.lx[i], c.ly[j], (ND == 3) ? c.lz[k % KN] : 0.0};
    for (int a = 0; a < ND; a++) {
      r[a] += pa * l[a];
      for (int m = 0; m < ND; m++) J[a * ND + m] += pa * l[a] * l[m];
    }
  }
  for (int a = 0; a < ND; a++)
    for (int m = 0; m < ND; m++) J[a * ND + m] -= r[a] * r[m];
  if (!inverse<ND>(Jm1, J)) {
    atomicOr(&P.status[p], ST_NEWTON);
    atomicOr(gstatus, ST_NEWTON);
    return;
  }
  for (int b = 0; b < NS; b++) {
    if (!c.on(b)) continue;
    const int i = b % 5, j = (b / 5) % 5, k = b / 25;
    const double pa = c.ex[i] * c.ey[j] * ((ND == 3) ? c.ez[k % KN] : 1.0) * Zinv;
    const double l[3] = {c.lx[i], c.ly[j], (ND == 3) ? c.lz[k % KN] : 0.0};
    Nq[b] = pa;
    for (int a = 0; a < ND; a++) {
      double v = 0.0;
      for (int m = 0; m < ND; m++) v += Jm1[a * ND + m] * l[m];
      dq[b * ND + a] = -pa * v;
    }
  }
}

extern "C" int nlps_gpu_shape_functions(nlps_gpu* h, int first, int count, double* N_out, double* dN_out) {
  const int np = h->P.np, ND = h->nd, NS = ND == 3 ? 125 : 25;
  if (first < 0 || count < 0 || first + count > np) {
    h->err = "nlps_gpu_shape_functions(): particle range outside the cloud";
    return 1;
  }
  if (count == 0) return 0;
  if (refresh_perm(h)) return 1;
  // device slots of the caller's particles first .. first + count - 1
  std::vector<int> slot_of(np), slots(count), I0(np);
  for (int s = 0; s < np; s++) slot_of[h->perm[s]] = s;
  for (int q = 0; q < count; q++) slots[q] = slot_of[first + q];
  int* slots_d = nullptr;
  double *Ns_d = nullptr, *dNs_d = nullptr;
  HIPCHK(hipMalloc((void**)&slots_d, (size_t)count * sizeof(int)));
  HIPCHK(hipMalloc((void**)&Ns_d, (size_t)count * NS * sizeof(double)));
  HIPCHK(hipMalloc((void**)&dNs_d, (size_t)count * NS * ND * sizeof(double)));
  HIPCHK(hipMemcpyAsync(slots_d, slots.data(), (size_t)count * sizeof(int), hipMemcpyHostToDevice, h->stream));
  if (ND == 2) hipLaunchKernelGGL(k_shape_slots<2>, dim3(nblk(count)), dim3(BLK), 0, h->stream, h->P, h->g, slots_d, count, Ns_d, dNs_d, h->gstatus_d);
  else hipLaunchKernelGGL(k_shape_slots<3>, dim3(nblk(count)), dim3(BLK), 0, h->stream, h->P, h->g, slots_d, count, Ns_d, dNs_d, h->gstatus_d);
  HIPCHK(hipGetLastError());
  std::vector<double> Ns((size_t)count * NS), dNs((size_t)count * NS * ND);
  std::vector<u64> lo(np), hi(np);
  HIPCHK(hipMemcpyAsync(Ns.data(), Ns_d, Ns.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipMemcpyAsync(dNs.data(), dNs_d, dNs.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(I0.data(), h->P.I0, (size_t)np * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(lo.data(), h->P.mlo, (size_t)np * sizeof(u64), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hi.data(), h->P.mhi, (size_t)np * sizeof(u64), hipMemcpyDeviceToHost));
  (void)hipFree(slots_d);
  (void)hipFree(Ns_d);
  (void)hipFree(dNs_d);
  const GridD& g = h->g;
  for (int q = 0; q < count; q++) {
    const int s = slots[q];
    int ijk[3] = {I0[s] % g.n[0], (I0[s] / g.n[0]) % g.n[1], I0[s] / (g.n[0] * g.n[1])};
    int cls = 0, mul = 1;
    for (int a = 0; a < 3; a++) {
      cls += (a < ND ? nlps_host::class5(ijk[a], g.n[a]) : 2) * mul;
      mul *= 5;
    }
    int tmp[NLPS_MAXNB], nt = 0;  // the members in chain order, then reversed: the order of ListNodes (nlps_gpu_download_lists)
    for (int w = 0; w < h->tab.count2[cls]; w++) {
      const int b = h->tab.order2[cls][w];
      const bool on = b < 64 ? ((lo[s] >> b) & 1ull) : ((hi[s] >> (b - 64)) & 1ull);
      if (on) tmp[nt++] = b;
    }
    for (int a = 0; a < NLPS_MAXNB; a++) {
      const int b = a < nt ? tmp[nt - 1 - a] : -1;
      if (N_out) N_out[(size_t)q * NLPS_MAXNB + a] = b >= 0 ? Ns[(size_t)q * NS + b] : 0.0;
      if (dN_out)
        for (int d = 0; d < ND; d++)
          dN_out[((size_t)q * NLPS_MAXNB + a) * ND + d] = b >= 0 ? dNs[((size_t)q * NS + b) * ND + d] : 0.0;
    }
  }
  return check_status(h, ST_NEWTON, "nlps_gpu_shape_functions()");
}

extern "C" int nlps_gpu_download_active(nlps_gpu* h, unsigned char* active) {
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(active, h->N.active, (size_t)h->g.nnodes, hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int nlps_gpu_status_flags(nlps_gpu* h, int* flags) {
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(flags, h->gstatus_d, sizeof(int), hipMemcpyDeviceToHost));
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Ghost-layer exchange over RCCL, owned by the library (no Python, no callback).
// Slab partition along the slowest grid axis: rank r may touch node layers [lo[r], hi[r]]; the layers it shares with
// rank r-1 / r+1 are one contiguous slice of every grid-numbered nodal array.  An exchange = ncclSend of the slice
// + ncclRecv of the neighbour's into a receive buffer (one group, two xGMI links busy at once), then slice += buffer
// (sums) or slice = max(slice, buffer) (flags).  Mode 1 keeps the simple all-reduce of the whole array.
// Phases (nlps_halo_fn): 0 = in the order of the handle's stream; 1 = on the library's side stream, behind everything
// queued on the handle's stream so far; 2 = the handle's stream waits for it.
// ------------------------------------------------------------------------------------------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
};
static RcclApi g_rccl;
static const char* rccl_load() {  // nullptr = loaded
  if (g_rccl.lib) return nullptr;
  // the soname every ROCm build of RCCL carries: a copy the host process has loaded already (torch's, the MPI
  // driver's) is reused, otherwise the system one is loaded
  void* L = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!L) L = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!L) return "librccl.so.1 not found (dlopen)";
#define NLPS_SYM(field, name)                                  \
  *(void**)(&g_rccl.field) = dlsym(L, name);                   \
  if (!g_rccl.field) return "RCCL symbol missing: " name;
  NLPS_SYM(GetUniqueId, "ncclGetUniqueId")
  NLPS_SYM(CommInitRank, "ncclCommInitRank")
  NLPS_SYM(CommDestroy, "ncclCommDestroy")
  NLPS_SYM(GroupStart, "ncclGroupStart")
  NLPS_SYM(GroupEnd, "ncclGroupEnd")
  NLPS_SYM(Send, "ncclSend")
  NLPS_SYM(Recv, "ncclRecv")
  NLPS_SYM(AllReduce, "ncclAllReduce")
  NLPS_SYM(Reduce, "ncclReduce")
  NLPS_SYM(GetErrorString, "ncclGetErrorString")
  NLPS_SYM(CommCount, "ncclCommCount")
  NLPS_SYM(CommUserRank, "ncclCommUserRank")
  NLPS_SYM(CommAbort, "ncclCommAbort")
#undef NLPS_SYM
  g_rccl.lib = L;
  return nullptr;
}

struct RcclHalo {
  ncclComm_t comm = nullptr;
  bool own_comm = false;
  int rank = 0, world = 1, mode = 0;  // mode 0: neighbour send/recv, 1: all-reduce of the whole array
  std::vector<int> lo, hi;            // node layers rank r may touch (inclusive)
  hipStream_t side = nullptr;
  void* rbuf[2] = {nullptr, nullptr};  // receive buffers: from rank-1, from rank+1
  size_t rbuf_bytes[2] = {0, 0};
  struct Ev {
    hipEvent_t start = nullptr, done = nullptr;
    bool pending = false;
  };
  std::map<const void*, Ev> ev;  // one pair of events per nodal array, re-recorded every step
  bool self_loop = false;        // world 1 self-test: the rank is its own two neighbours
  int* mig_cnt_d = nullptr;      // nlps_gpu_rccl_migrate: {rows to below, rows to above, rows from below, rows from above}
  // single-launch overlap (TileD::sig_flag): counters in device memory, flags in signal memory, one pair per stage
  unsigned* sig_cnt = nullptr;   // [2]
  unsigned* sig_flag[2] = {nullptr, nullptr};
  unsigned sig_seq = 0;
  bool can_wait_value = false;
};

// Holds the exchange stream until the boundary tiles of the launch in flight on the handle's stream have published
// `seq` (tile_signal).  One lane polls an agent-scope load with s_sleep in between; it occupies one wave slot of one CU
// and depends on nothing but the flag, and it gives up after about one second of the constant 100 MHz clock
// (wall_clock64), so that a launch that never happens cannot hang it (hipStreamWaitValue32 does the same through the
// host: measured 55 us from the store to the next command, against a few us for this kernel).
__global__ void k_wait_flag(const unsigned* __restrict__ flag, unsigned seq, int* __restrict__ gstatus) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 100000000ull) {
    const unsigned v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
    if ((int)(v - seq) >= 0) return;
    __builtin_amdgcn_s_sleep(32);
  }
  atomicOr(gstatus, ST_HALO);  // timed out: reported like a particle outside the node window
}

template <class T>
__global__ void k_halo_add(T* __restrict__ a, const T* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += b[i];
}
__global__ void k_halo_max(unsigned char* __restrict__ a, const unsigned char* __restrict__ b, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = a[i] > b[i] ? a[i] : b[i];
}

#define RCCLCHK(call)                                                                                   \
  do {                                                                                                  \
    ncclResult_t r_ = (call);                                                                           \
    if (r_ != ncclSuccess) {                                                                            \
      h->err = std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?");   \
      fprintf(stderr, "\033[1;31mError in nlps_gpu (RCCL): %s\033[0m\n", h->err.c_str());               \
      return 1;                                                                                         \
    }                                                                                                   \
  } while (0)

// the exchange itself, in the order of stream s
static int rccl_exchange_on(nlps_gpu* h, void* dptr, int nfield, int elem, int kind, hipStream_t s) {
  RcclHalo* R = h->rccl;
  const int nl = h->g.n[h->nd - 1];
  const size_t plane = (size_t)h->g.nnodes / nl;
  const ncclDataType_t dt = elem == 8 ? ncclDouble : ncclUint8;
  if (R->mode == 1) {
    if (R->world > 1)
      RCCLCHK(g_rccl.AllReduce(dptr, dptr, (size_t)h->g.nnodes * nfield, dt, kind == 0 ? ncclSum : ncclMax, R->comm, s));
    return 0;
  }
  struct Part {
    char* sl;
    size_t count;
    int peer, k;
  } parts[2];
  int np = 0;
  for (int k = 0; k < 2; k++) {
    const int nb = R->rank + (k == 0 ? -1 : 1);
    int a, b;
    if (R->self_loop) {  // self-test: the lowest / highest three layers stand for the two shared slices
      a = k == 0 ? R->lo[0] : std::max(R->lo[0], R->hi[0] - 2);
      b = k == 0 ? std::min(R->hi[0], R->lo[0] + 2) : R->hi[0];
    } else {
      if (nb < 0 || nb >= R->world) continue;
      a = std::max(R->lo[R->rank], R->lo[nb]);
      b = std::min(R->hi[R->rank], R->hi[nb]);
      if (a > b) continue;
    }
    const size_t count = (size_t)(b - a + 1) * plane * nfield, bytes = count * elem;
    if (bytes > R->rbuf_bytes[k]) {
      HIPCHK(hipDeviceSynchronize());
      if (R->rbuf[k]) HIPCHK(hipFree(R->rbuf[k]));
      HIPCHK(hipMalloc(&R->rbuf[k], bytes));
      R->rbuf_bytes[k] = bytes;
    }
    parts[np++] = {(char*)dptr + (size_t)a * plane * nfield * elem, count, R->self_loop ? R->rank : nb, k};
  }
  if (np == 0) return 0;
  RCCLCHK(g_rccl.GroupStart());
  for (int q = 0; q < np; q++) {
    // self-test: what goes "down" comes back as what arrives "from above" and vice versa
    RCCLCHK(g_rccl.Send(parts[q].sl, parts[q].count, dt, parts[q].peer, R->comm, s));
    RCCLCHK(g_rccl.Recv(R->rbuf[R->self_loop ? 1 - parts[q].k : parts[q].k], parts[q].count, dt, parts[q].peer, R->comm, s));
  }
  RCCLCHK(g_rccl.GroupEnd());
  // after the sends in stream order, so the neighbour got the un-summed slice
  for (int q = 0; q < np; q++) {
    const size_t n = parts[q].count;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (elem == 8 && kind == 0)
      hipLaunchKernelGGL(k_halo_add<double>, dim3(grid), dim3(256), 0, s, (double*)parts[q].sl, (const double*)R->rbuf[parts[q].k], n);
    else
      hipLaunchKernelGGL(k_halo_max, dim3(grid), dim3(256), 0, s, (unsigned char*)parts[q].sl, (const unsigned char*)R->rbuf[parts[q].k], n);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int rccl_halo(nlps_gpu* h, void* dptr, int nfield, int elem, int kind, int phase) {
  RcclHalo* R = h->rccl;
  // (a rank without neighbours still goes through the stream choreography: that is how a one-GPU box rehearses it)
  if (phase == 0) {
    // the receive buffers are shared with the side stream: an exchange still pending there finishes first
    for (auto& kv : R->ev)
      if (kv.second.pending) {
        HIPCHK(hipStreamWaitEvent(h->stream, kv.second.done, 0));
        kv.second.pending = false;
      }
    return rccl_exchange_on(h, dptr, nfield, elem, kind, h->stream);
  }
  RcclHalo::Ev& e = R->ev[dptr];
  if (!e.start) {
    HIPCHK(hipEventCreateWithFlags(&e.start, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e.done, hipEventDisableTiming));
  }
  if (phase == 1) {
    HIPCHK(hipEventRecord(e.start, h->stream));
    HIPCHK(hipStreamWaitEvent(R->side, e.start, 0));
    if (rccl_exchange_on(h, dptr, nfield, elem, kind, R->side)) return 1;
    HIPCHK(hipEventRecord(e.done, R->side));
    e.pending = true;
    return 0;
  }
  if (phase == 3 || phase == 4) {  // start behind the boundary tiles of the launch that is in flight (stage 0 = K2, 1 = K3)
    const int stage = phase - 3;
    hipLaunchKernelGGL(k_wait_flag, dim3(1), dim3(64), 0, R->side, R->sig_flag[stage], R->sig_seq, h->gstatus_d);
    if (rccl_exchange_on(h, dptr, nfield, elem, kind, R->side)) return 1;
    HIPCHK(hipEventRecord(e.done, R->side));
    e.pending = true;
    return 0;
  }
  if (e.pending) {
    HIPCHK(hipStreamWaitEvent(h->stream, e.done, 0));
    e.pending = false;
  }
  return 0;
}

extern "C" int nlps_gpu_rccl_unique_id(void* id128) {
  if (rccl_load()) return 1;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return 1;
  memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
  return 0;
}

extern "C" int nlps_gpu_rccl_detach(nlps_gpu* h);
static int rccl_attach_common(nlps_gpu* h, ncclComm_t comm, bool own, int rank, int world, const int* layer_lo,
                              const int* layer_hi, int mode) {
  if (h->rccl) {
    h->err = "nlps_gpu_rccl_attach: a communicator is attached already (nlps_gpu_rccl_detach first)";
    return 1;
  }
  if (materialise_nodal(h)) return 1;  // (window and bands change below)
  const int nl = h->g.n[h->nd - 1];
  if (world < 1 || rank < 0 || rank >= world || (mode != 0 && mode != 1)) {
    h->err = "nlps_gpu_rccl_attach: bad rank / world / mode";
    return 1;
  }
  RcclHalo* R = new RcclHalo();
  R->comm = comm;
  R->own_comm = own;
  R->rank = rank;
  R->world = world;
  R->mode = mode;
  for (int r = 0; r < world; r++) {
    const int a = layer_lo ? layer_lo[r] : 0, b = layer_hi ? layer_hi[r] : nl - 1;
    if (a < 0 || b >= nl || a > b) {
      delete R;
      h->err = "nlps_gpu_rccl_attach: layer range outside the grid";
      return 1;
    }
    R->lo.push_back(a);
    R->hi.push_back(b);
  }
  for (int r = 0; r + 2 < world; r++)
    if (R->hi[r] >= R->lo[r + 2]) {
      delete R;
      h->err = "nlps_gpu_rccl_attach: slabs too thin (a rank overlaps its second neighbour)";
      return 1;
    }
  if (hipStreamCreateWithFlags(&R->side, hipStreamNonBlocking) != hipSuccess) {
    delete R;
    h->err = "nlps_gpu_rccl_attach: hipStreamCreateWithFlags failed";
    return 1;
  }
  {
    bool ok = hipMalloc((void**)&R->sig_cnt, 2 * sizeof(unsigned)) == hipSuccess &&
              hipMemset(R->sig_cnt, 0, 2 * sizeof(unsigned)) == hipSuccess;
    for (int k = 0; k < 2 && ok; k++)
      ok = hipMalloc((void**)&R->sig_flag[k], 8) == hipSuccess && hipMemset(R->sig_flag[k], 0, 8) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    R->can_wait_value = ok;
  }
  h->rccl = R;
  h->rccl_wait_value = R->can_wait_value;
  // ghost bands (layers shared with a neighbour) and the node window follow from the layer ranges
  const int band_lo = rank > 0 ? std::min(R->hi[rank], R->hi[rank - 1]) : -1;
  const int band_hi = rank + 1 < world ? std::max(R->lo[rank], R->lo[rank + 1]) : nl;
  h->band_lo = rank > 0 && R->hi[rank - 1] >= R->lo[rank] ? band_lo : -(1 << 30);
  h->band_hi = rank + 1 < world && R->lo[rank + 1] <= R->hi[rank] ? band_hi : (1 << 30);
  // 2: one launch per stage, the exchange released by the boundary tiles through signal memory; 1: split launches
  h->overlap = world > 1 ? (R->can_wait_value ? 2 : 1) : 0;
  if (world > 1 && nlps_gpu_set_node_window(h, R->lo[rank], R->hi[rank])) {
    // the handle must not keep a half-attached exchange: the caller still owns `comm` (it destroys its own, or the
    // wrapper below destroys the one it made), so the detach here must leave it alone
    const std::string err = h->err;
    R->own_comm = false;
    (void)nlps_gpu_rccl_detach(h);
    h->err = err;
    return 1;
  }
  return 0;
}

extern "C" int nlps_gpu_rccl_attach(nlps_gpu* h, const void* id128, int rank, int world, const int* layer_lo,
                                    const int* layer_hi, int mode) {
  if (const char* e = rccl_load()) {
    h->err = e;
    return 1;
  }
  if (h->rccl) {  // before the collective below: a second attach must not block on (or create) a communicator
    h->err = "nlps_gpu_rccl_attach: a communicator is attached already (nlps_gpu_rccl_detach first)";
    return 1;
  }
  ncclUniqueId id;
  memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
  ncclComm_t comm = nullptr;
  RCCLCHK(g_rccl.CommInitRank(&comm, world, id, rank));
  if (rccl_attach_common(h, comm, true, rank, world, layer_lo, layer_hi, mode)) {
    (void)g_rccl.CommDestroy(comm);
    return 1;
  }
  return 0;
}

extern "C" int nlps_gpu_rccl_attach_comm(nlps_gpu* h, void* nccl_comm, int rank, int world, const int* layer_lo,
                                         const int* layer_hi, int mode) {
  if (const char* e = rccl_load()) {
    h->err = e;
    return 1;
  }
  return rccl_attach_common(h, (ncclComm_t)nccl_comm, false, rank, world, layer_lo, layer_hi, mode);
}

extern "C" int nlps_gpu_rccl_detach(nlps_gpu* h) {
  RcclHalo* R = h->rccl;
  if (!R) return 0;
  if (materialise_nodal(h)) return 1;
  (void)hipDeviceSynchronize();
  for (auto& kv : R->ev) {
    if (kv.second.start) (void)hipEventDestroy(kv.second.start);
    if (kv.second.done) (void)hipEventDestroy(kv.second.done);
  }
  for (int k = 0; k < 2; k++) {
    if (R->rbuf[k]) (void)hipFree(R->rbuf[k]);
    if (R->sig_flag[k]) (void)hipFree(R->sig_flag[k]);
  }
  if (R->sig_cnt) (void)hipFree(R->sig_cnt);
  if (R->mig_cnt_d) (void)hipFree(R->mig_cnt_d);
  if (R->side) (void)hipStreamDestroy(R->side);
  if (R->own_comm && R->comm) (void)g_rccl.CommDestroy(R->comm);
  delete R;
  h->rccl = nullptr;
  h->rccl_wait_value = false;
  h->overlap = 0;
  h->band_lo = -(1 << 30);
  h->band_hi = 1 << 30;
  return 0;
}

// Implicit driver on several ranks: the masked residual / lumped-mass vector of every rank summed onto
// the rank that runs the PETSc solve (root >= 0) or onto all ranks (root < 0).  vec: device pointer, n doubles.
extern "C" int nlps_gpu_rccl_reduce(nlps_gpu* h, double* vec, size_t n, int root) {
  RcclHalo* R = h->rccl;
  if (!R) {
    h->err = "nlps_gpu_rccl_reduce: no communicator attached";
    return 1;
  }
  if (R->world == 1 || n == 0) return 0;
  if (root < 0) RCCLCHK(g_rccl.AllReduce(vec, vec, n, ncclDouble, ncclSum, R->comm, h->stream));
  else RCCLCHK(g_rccl.Reduce(vec, vec, n, ncclDouble, ncclSum, root, R->comm, h->stream));
  return 0;
}

// World-size-1 self-test of the exchange machinery (a one-GPU box cannot hold two ranks): the rank acts as its own
// two neighbours, i.e. the lowest three layers of its range are exchanged with the highest three through
// ncclSend / ncclRecv to itself.  After the call array[low slice] += old array[high slice] and vice versa.
extern "C" int nlps_gpu_rccl_selftest_exchange(nlps_gpu* h, void* dptr, int nfield, int elem_bytes, int kind, int overlap) {
  RcclHalo* R = h->rccl;
  if (!R || R->world != 1) {
    h->err = "nlps_gpu_rccl_selftest_exchange: needs an attached communicator of world size 1";
    return 1;
  }
  R->self_loop = true;
  int st = 0;
  if (overlap) st = rccl_halo(h, dptr, nfield, elem_bytes, kind, 1) || rccl_halo(h, dptr, nfield, elem_bytes, kind, 2);
  else st = rccl_halo(h, dptr, nfield, elem_bytes, kind, 0);
  R->self_loop = false;
  return st;
}

static int halo_dispatch(nlps_gpu* h, void* dptr, int nfield, int elem, int kind, int phase);
// What the attached communicator says about itself (ncclCommCount / ncclCommUserRank: the rank count RCCL really
// runs with, not the one the caller asked for) and the overlap choreography in force (0 blocking, 1 split launches,
// 2 one launch per stage).
extern "C" int nlps_gpu_rccl_info(nlps_gpu* h, int* nranks, int* rank, int* overlap_mode) {
  RcclHalo* R = h->rccl;
  if (!R) {
    h->err = "nlps_gpu_rccl_info: no communicator attached";
    return 1;
  }
  int n = 0, r = 0;
  RCCLCHK(g_rccl.CommCount(R->comm, &n));
  RCCLCHK(g_rccl.CommUserRank(R->comm, &r));
  if (nranks) *nranks = n;
  if (rank) *rank = r;
  if (overlap_mode) *overlap_mode = h->overlap;
  return 0;
}

// Particle migration between slab ranks with the transport inside the library (a C driver needs no Python):
// select + pack (nlps_gpu_migration_select), the row counts and then the rows themselves exchanged with the two
// neighbours by ncclSend / ncclRecv on the handle's stream, commit.  self_loop (world 1 only): the rank is its own two
// neighbours, so what leaves comes straight back -- every call on the wire runs on a one-GPU box.
// Collective: every rank makes the same sequence of RCCL calls whatever happens locally.  A rank whose select fails or
// whose capacity would overflow says so in a status word that all ranks agree on (one 4-byte all-reduce) BEFORE any row
// moves: then every rank returns 1 with its cloud unchanged (a select without commit has no effect).  A HIP / RCCL error
// after that point cannot be negotiated: the communicator is aborted so that the peers get an error, never a hang.
static int rccl_migrate(nlps_gpu* h, int keep_lo, int keep_hi, int* sent_down, int* sent_up, int* received, bool self_loop) {
  RcclHalo* R = h->rccl;
  if (!R) {
    h->err = "nlps_gpu_rccl_migrate: no communicator attached (nlps_gpu_rccl_attach)";
    return 1;
  }
  if (self_loop && R->world != 1) {
    h->err = "nlps_gpu_rccl_selftest_migrate: needs an attached communicator of world size 1";
    return 1;
  }
  const int nl = h->g.n[h->nd - 1];
  if (!self_loop) {  // an edge rank has no neighbour on its outer side: nothing may be selected towards it
    if (R->rank == 0) keep_lo = 0;
    if (R->rank == R->world - 1) keep_hi = nl - 1;
  }
  double* rows_in[2] = {nullptr, nullptr};
  auto fatal = [&](const std::string& what) {  // unrecoverable local error in the middle of the collective
    h->err = "nlps_gpu_rccl_migrate: " + what + " (communicator aborted)";
    fprintf(stderr, "\033[1;31mError in nlps_gpu: %s\033[0m\n", h->err.c_str());
    for (int k = 0; k < 2; k++)
      if (rows_in[k]) (void)hipFree(rows_in[k]);
    if (R->comm) (void)g_rccl.CommAbort(R->comm);
    R->comm = nullptr;
    return 1;
  };
#define MIG_HIP(call)                                                                   \
  do {                                                                                  \
    hipError_t e_ = (call);                                                             \
    if (e_ != hipSuccess) return fatal(std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define MIG_RCCL(call)                                                                  \
  do {                                                                                  \
    ncclResult_t r_ = (call);                                                           \
    if (r_ != ncclSuccess) return fatal(std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
  } while (0)
  if (!R->comm) {
    h->err = "nlps_gpu_rccl_migrate: the communicator was aborted by an earlier failure";
    return 1;
  }
  int n_out[2] = {0, 0}, rw = 0;
  void* rows_out[2] = {nullptr, nullptr};
  int bad = nlps_gpu_migration_select(h, keep_lo, keep_hi, &n_out[0], &n_out[1], &rw, &rows_out[0], &rows_out[1]) ? 1 : 0;
  const std::string select_err = bad ? h->err : std::string();
  if (bad) n_out[0] = n_out[1] = 0;
  if (sent_down) *sent_down = n_out[0];
  if (sent_up) *sent_up = n_out[1];
  if (received) *received = 0;
  int peer[2] = {R->rank - 1, R->rank + 1};
  bool has[2] = {peer[0] >= 0, peer[1] < R->world};
  if (self_loop) {
    peer[0] = peer[1] = R->rank;
    has[0] = has[1] = true;
  }
  if (!has[0] && !has[1]) {  // world 1: nobody to agree with
    if (bad) return 1;
    return nlps_gpu_migration_commit(h, nullptr, 0, nullptr, 0);
  }
  if (!R->mig_cnt_d) MIG_HIP(hipMalloc((void**)&R->mig_cnt_d, 8 * sizeof(int)));
  // 1. how many rows come from each neighbour
  int cnt[8] = {n_out[0], n_out[1], 0, 0, 0, 0, 0, 0};
  MIG_HIP(hipMemcpyAsync(R->mig_cnt_d, cnt, 8 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  MIG_RCCL(g_rccl.GroupStart());
  for (int k = 0; k < 2; k++) {
    if (!has[k]) continue;
    // (self-loop: sends and receives between one pair match in the order posted, so what went "down" arrives as
    // "from below" -- which neighbour it stands for does not matter to the commit)
    MIG_RCCL(g_rccl.Send(R->mig_cnt_d + k, 1, ncclInt32, peer[k], R->comm, h->stream));
    MIG_RCCL(g_rccl.Recv(R->mig_cnt_d + 2 + k, 1, ncclInt32, peer[k], R->comm, h->stream));
  }
  MIG_RCCL(g_rccl.GroupEnd());
  MIG_HIP(hipMemcpyAsync(cnt, R->mig_cnt_d, 4 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  MIG_HIP(hipStreamSynchronize(h->stream));
  const int n_in[2] = {cnt[2], cnt[3]};
  const bool overflow = n_in[0] < 0 || n_in[1] < 0 || (size_t)h->P.np + n_in[0] + n_in[1] > h->P.npad;
  // 2. one status word for the whole communicator: 0 go, 1 some rank cannot
  int status[2] = {(bad || overflow) ? 1 : 0, 0};
  MIG_HIP(hipMemcpyAsync(R->mig_cnt_d + 4, status, sizeof(int), hipMemcpyHostToDevice, h->stream));
  MIG_RCCL(g_rccl.AllReduce(R->mig_cnt_d + 4, R->mig_cnt_d + 5, 1, ncclInt32, ncclMax, R->comm, h->stream));
  MIG_HIP(hipMemcpyAsync(status + 1, R->mig_cnt_d + 5, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  MIG_HIP(hipStreamSynchronize(h->stream));
  if (status[1] != 0) {  // every rank takes this branch together; nothing has moved, nothing is committed
    if (bad) h->err = select_err;
    else if (overflow) h->err = "nlps_gpu_rccl_migrate: more immigrants than the capacity reserved at create (np + max(np/4, 1024))";
    else h->err = "nlps_gpu_rccl_migrate: another rank of the communicator could not migrate (its select failed or its capacity is exhausted); no particle moved";
    return 1;
  }
  // 3. the rows
  for (int k = 0; k < 2; k++)
    if (n_in[k] > 0) MIG_HIP(hipMalloc((void**)&rows_in[k], (size_t)n_in[k] * rw * sizeof(double)));
  MIG_RCCL(g_rccl.GroupStart());
  for (int k = 0; k < 2; k++) {
    if (!has[k]) continue;
    if (n_out[k] > 0) MIG_RCCL(g_rccl.Send(rows_out[k], (size_t)n_out[k] * rw, ncclDouble, peer[k], R->comm, h->stream));
    if (n_in[k] > 0) MIG_RCCL(g_rccl.Recv(rows_in[k], (size_t)n_in[k] * rw, ncclDouble, peer[k], R->comm, h->stream));
  }
  MIG_RCCL(g_rccl.GroupEnd());
  MIG_HIP(hipStreamSynchronize(h->stream));
#undef MIG_HIP
#undef MIG_RCCL
  const int st = nlps_gpu_migration_commit(h, rows_in[0], n_in[0], rows_in[1], n_in[1]);
  for (int k = 0; k < 2; k++)
    if (rows_in[k]) (void)hipFree(rows_in[k]);
  if (received) *received = n_in[0] + n_in[1];
  return st;
}
extern "C" int nlps_gpu_rccl_migrate(nlps_gpu* h, int keep_lo, int keep_hi, int* sent_down, int* sent_up, int* received) {
  return rccl_migrate(h, keep_lo, keep_hi, sent_down, sent_up, received, false);
}
extern "C" int nlps_gpu_rccl_selftest_migrate(nlps_gpu* h, int keep_lo, int keep_hi, int* sent_down, int* sent_up,
                                              int* received) {
  return rccl_migrate(h, keep_lo, keep_hi, sent_down, sent_up, received, true);
}

static int halo(nlps_gpu* h, void* dptr, int nfield, int elem, int kind, int phase = 0) {
  // (with timing on, the exchanges the handle's stream has to wait for -- blocking ones, and the pick-up of overlapped
  // ones -- are bracketed: their sum is what a step loses to communication, nlps_gpu_get_timing slot 6)
  const bool bracket = h->timing && (phase == 0 || phase == 2) && (h->rccl || h->halo) && h->nwait < 6;
  if (bracket) {
    if (!h->evw[2 * h->nwait]) {
      HIPCHK(hipEventCreate(&h->evw[2 * h->nwait]));
      HIPCHK(hipEventCreate(&h->evw[2 * h->nwait + 1]));
    }
    HIPCHK(hipEventRecord(h->evw[2 * h->nwait], h->stream));
  }
  const int st_ = halo_dispatch(h, dptr, nfield, elem, kind, phase);
  if (bracket) {
    HIPCHK(hipEventRecord(h->evw[2 * h->nwait + 1], h->stream));
    h->nwait++;
  }
  return st_;
}
static int halo_dispatch(nlps_gpu* h, void* dptr, int nfield, int elem, int kind, int phase) {
  if (h->rccl) return rccl_halo(h, dptr, nfield, elem, kind, phase);
  if (!h->halo) return 0;
  int st = h->halo(h->halo_ctx, dptr, nfield, elem, kind, phase);
  if (st) {
    h->err = "halo exchange callback failed";
    return 1;
  }
  return 0;
}



static int compute_node_mask(nlps_gpu* h) {
  int nn = h->g.nnodes, nb = (nn + 1023) / 1024;
  if (h->canon_d) {
    // the running index of get_active_nodes__MeshTools__ (Nodes-Tools.c:46-66) follows the mesh FILE's node order: scan
    // the flags in that order, then hand every lattice node its index
    hipLaunchKernelGGL(k_gather<unsigned char>, dim3(nblk(nn)), dim3(BLK), 0, h->stream, h->mask_flags_d,
                       (const unsigned char*)h->N.active, (const int*)h->canon_d, nn);
    hipLaunchKernelGGL(k_scan_count, dim3(nb), dim3(256), 0, h->stream, h->mask_flags_d, nn, 0, h->bsum_d);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, h->stream, h->bsum_d, nb, h->total_d);
    hipLaunchKernelGGL(k_scan_write, dim3(nb), dim3(256), 0, h->stream, h->mask_flags_d, nn, 0, h->bsum_d, h->mask_idx_d);
    hipLaunchKernelGGL(k_scatter<int>, dim3(nblk(nn)), dim3(BLK), 0, h->stream, h->n2m_d, (const int*)h->mask_idx_d,
                       (const int*)h->canon_d, nn);
    HIPCHK(hipGetLastError());
    return 0;
  }
  hipLaunchKernelGGL(k_scan_count, dim3(nb), dim3(256), 0, h->stream, h->N.active, nn, 0, h->bsum_d);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, h->stream, h->bsum_d, nb, h->total_d);
  hipLaunchKernelGGL(k_scan_write, dim3(nb), dim3(256), 0, h->stream, h->N.active, nn, 0, h->bsum_d, h->n2m_d);
  HIPCHK(hipGetLastError());
  return 0;
}

// Masked numbering in the node order of the mesh file (see include/nlps_gpu.h)
extern "C" int nlps_gpu_set_node_numbering(nlps_gpu* h, const int* lattice_of_file) {
  const int nn = h->g.nnodes;
  h->masks_valid = false;
  if (!lattice_of_file) {
    if (h->canon_d) {
      HIPCHK(hipStreamSynchronize(h->stream));
      (void)hipFree(h->canon_d);
      h->canon_d = nullptr;
    }
    return 0;
  }
  std::vector<unsigned char> seen(nn, 0);
  for (int A = 0; A < nn; A++) {
    const int l = lattice_of_file[A];
    if (l < 0 || l >= nn || seen[l]) {
      h->err = "nlps_gpu_set_node_numbering: lattice_of_file is not a permutation of the grid's nodes";
      return 1;
    }
    seen[l] = 1;
  }
  if (!h->canon_d) {
    HIPCHK(hipMalloc((void**)&h->canon_d, (size_t)nn * sizeof(int)));
    if (!h->mask_flags_d) HIPCHK(hipMalloc((void**)&h->mask_flags_d, (size_t)nn));
    if (!h->mask_idx_d) HIPCHK(hipMalloc((void**)&h->mask_idx_d, (size_t)nn * sizeof(int)));
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  HIPCHK(hipMemcpy(h->canon_d, lattice_of_file, (size_t)nn * sizeof(int), hipMemcpyHostToDevice));
  return 0;
}

// canonical tile lists from per-node counters (TileTab) instead of the per-tile counting sort: every mode but the
// deterministic one (which needs ranks that do not depend on the arrival order of an atomic)
static bool node_lists(const nlps_gpu* h) { return h->tile_ordering && !h->deterministic && h->node_lists_on; }
static TileCnt tile_cnt(nlps_gpu* h, bool on) {
  TileCnt tc;
  for (int a = 0; a < 3; a++) tc.nt[a] = h->nt[a];
  tc.count = on ? h->tile_count_d : nullptr;
  tc.tile0 = h->tile0;
  tc.ntw = h->ntw;
  tc.win_lo = h->win_lo;
  tc.win_hi = h->win_hi;
  tc.plane = h->g.nnodes / h->g.n[h->nd - 1];
  tc.gstatus = h->gstatus_d;
  tc.home = nullptr;
  tc.rehome = 0;
  tc.foreign = nullptr;
  tc.node_cnt = (on && node_lists(h)) ? h->node_cnt_d : nullptr;
  tc.nrank = h->nrank_d;
  tc.defer = 0;
  if (on && h->adaptive_resort > 0.0 && !h->deterministic) {
    tc.home = h->home_d;
    tc.foreign = h->foreign_d;
    tc.rehome = h->rehome ? 1 : 0;
    h->rehome = false;
  }
  return tc;
}
static TileD tile_view(nlps_gpu* h, int cls = 0) {  // cls: 0 all tiles, 1 boundary, 2 interior
  TileD td;
  for (int a = 0; a < 3; a++) td.nt[a] = h->nt[a];
  td.ntiles = h->ntiles;
  td.tile0 = h->tile0;
  td.ntw = h->ntw;
  td.slab = h->deterministic ? h->slab_d : nullptr;
  td.slab_n = 1;
  td.slab_slot = 0;
  td.sig_cnt = nullptr;
  td.sig_flag = nullptr;
  td.sig_seq = 0;
  td.work[0] = h->work1_d;
  td.work[1] = h->work2_d;
  td.range = h->nwork_d + 4 * cls;
  td.phase = h->phase_d;
  td.start = h->tile_start_d;
  td.count = h->tile_count_d;
  td.order_m = h->order_d;
  td.order = (h->tile_ordering || h->deterministic) ? h->order2_d : h->order_d;
  return td;
}

// S1: closest node + activation (+ binning of the particles to I0-tiles when `tiled`), then lists,
// beta and the Newton iteration (+ predictor and P2G of mass / m*dD when `p2g`, tiled form only)
// node ranges of the per-step nodal kernels: part 0 = the whole node window, 1 = the nodes outside the ghost
// bands, 2 = the ghost bands (two ranges)
struct NodeRanges {
  int a0, an, b0, bn;
};
static NodeRanges node_ranges(nlps_gpu* h, int part) {
  const int plane = h->g.nnodes / h->g.n[h->nd - 1];
  const int lo = h->win_lo, hi = h->win_hi;
  const int il = std::max(lo, std::min(hi + 1, h->band_lo + 1));  // first layer outside the low band
  const int ih = std::min(hi + 1, std::max(il, h->band_hi));      // first layer of the high band
  if (part == 0) return {lo * plane, (hi - lo + 1) * plane, 0, 0};
  if (part == 1) return {il * plane, (ih - il) * plane, 0, 0};
  return {lo * plane, (il - lo) * plane, ih * plane, (hi + 1 - ih) * plane};
}

// The folded explicit step never stored dU, the accelerations and the reactions of its step: made here,
// from the nodal sums of that step, before somebody reads them or overwrites the sums they come from.
static int materialise_nodal(nlps_gpu* h) {
  if (!h->nodal_stale) return 0;
  const NodeRanges r = node_ranges(h, 0);
  LAUNCH_ND((k_nodal_dU<2>), (k_nodal_dU<3>), nblk(r.an + r.bn), r.a0, r.an, r.b0, r.bn, h->N, h->last_bm, h->last_bc);
  LAUNCH_ND((k_nodal_accel<2>), (k_nodal_accel<3>), nblk(r.an + r.bn), r.a0, r.an, r.b0, r.bn, h->N, h->last_gv[0],
            h->last_gv[1], h->last_gv[2], 0, (int*)nullptr, (int*)nullptr, 0);
  HIPCHK(hipGetLastError());
  h->nodal_stale = false;
  return 0;
}

// arms the boundary-done signal of a single-launch stage (overlap mode 2): stage 0 = K2, 1 = K3
static void arm_signal(nlps_gpu* h, TileD& td, int stage) {
  RcclHalo* R = h->rccl;
  td.sig_cnt = R->sig_cnt + stage;
  td.sig_flag = R->sig_flag[stage];
  td.sig_seq = ++R->sig_seq;
}

static void launch_k2(nlps_gpu* h, bool p2g, int cls, double dt, double gamma_nm, bool signal = false) {
  TileD td = tile_view(h, cls);
  if (signal) arm_signal(h, td, 0);
  if (h->deterministic && p2g) {  // one wave per tile, sorted list, slab flush (nlps_gpu_set_deterministic)
    const dim3 grid1(h->ntw), blk1(64);
    if (h->nd == 2) hipLaunchKernelGGL((k2_tile<2, true, 64, 1>), grid1, blk1, 0, h->stream, h->P, h->g, h->N, td, h->prm, dt, gamma_nm, h->gstatus_d);
    else hipLaunchKernelGGL((k2_tile<3, true, 64, 1>), grid1, blk1, 0, h->stream, h->P, h->g, h->N, td, h->prm, dt, gamma_nm, h->gstatus_d);
    return;
  }
  const dim3 blk(BLK);
#define NLPS_K2L(NDv, P2Gv)                                                                              \
  do {                                                                                                   \
    hipLaunchKernelGGL((k2_tile<NDv, P2Gv>), dim3(h->ntw * K2_SPLIT), blk, 0, h->stream, h->P, h->g, h->N, td,         \
                       h->prm, dt, gamma_nm, h->gstatus_d);                                              \
  } while (0)
  if (h->nd == 2) {
    if (p2g) NLPS_K2L(2, true);
    else NLPS_K2L(2, false);
  } else {
    if (p2g) NLPS_K2L(3, true);
    else NLPS_K2L(3, false);
  }
#undef NLPS_K2L
}

// S1: closest-node update + 1-ring activation + binning of the particles to I0-tiles, then lists, beta and the
// Newton iteration (+ predictor and P2G of mass / m*dD when `p2g`).  With `overlap` the exchange of the active
// flags runs behind the tiles that do not touch a ghost band.
static int search_and_lists(nlps_gpu* h, bool init, bool p2g, double dt, double gamma_nm, int overlap = 0,
                            bool launch_lists_kernel = true) {
  int np = h->P.np;
  if (!p2g && materialise_nodal(h)) return 1;  // (a level-B search rewrites the active flags the lazy nodal arrays depend on)
  // ahead: the search of this step was done by the last kernel of the previous one (k5_tile<., ., true>); only the
  // nodal accumulators are reset here
  const bool ahead = h->ahead && !init;
  const bool deferred = ahead && h->ranks_deferred && node_lists(h);  // that search only counted: ranks from cursors (k_fill_orders)
  h->ranks_deferred = false;
  if (ahead) std::swap(h->tile_count_d, h->tile_count2_d);  // the counters that search filled size the lists from here on
  const bool clear_in_dilate = ahead && p2g;  // nothing but the nodal accumulators to reset: k_dilate_scan does it
  if (!clear_in_dilate)
    LAUNCH_ND((k_step_clear<2>), (k_step_clear<3>), nblk(std::max(h->nwn, h->ntw)), h->n0, h->nwn, h->N,
              h->tile_count_d + h->tile0, h->ntw, p2g ? 1 : 0, node_lists(h) ? h->node_cnt_d : nullptr, ahead ? 0 : 1);
  if (!ahead) {
    TileCnt tc = tile_cnt(h, true);
    if (init) LAUNCH_ND((k_init_I0<2>), (k_init_I0<3>), nblk(np), h->P, h->g, h->N, tc);
    else LAUNCH_ND((k_search<2>), (k_search<3>), nblk(np), h->P, h->g, h->N, h->rank1_d, tc, h->searched ? 0 : 1);
  }
  h->searched = h->ahead = false;  // consumed
  {
    const int TB = h->nd == 3 ? TileCfg<3>::TB : TileCfg<2>::TB;
    TileScanArgs ts{h->tile_count_d + h->tile0, h->tile_start_d + h->tile0, h->ntw, h->tile0, h->ntiles / h->nt[h->nd - 1], TB,
                    h->band_lo, h->band_hi, h->work1_d, h->work2_d, h->nwork_d, deferred ? h->tile_cursor_d + h->tile0 : nullptr};
    TileTab tab;
    for (int a = 0; a < 3; a++) tab.nt[a] = h->nt[a];
    tab.node_cnt = h->node_cnt_d;
    tab.mask = h->tabm_d;
    tab.base = h->tabo_d;
    const int nb = 1 + (h->nwn + 1023) / 1024 + (node_lists(h) ? (h->ntw + 15) / 16 : 0);
    int* fo = (h->adaptive_resort > 0.0 && !h->deterministic) ? h->foreign_d : nullptr;
    if (h->nd == 2) hipLaunchKernelGGL(k_dilate_scan<2>, dim3(nb), dim3(1024), 0, h->stream, h->n0, h->nwn, h->g, h->N, ts, fo, h->foreign_h, tab, clear_in_dilate ? 1 : 0);
    else hipLaunchKernelGGL(k_dilate_scan<3>, dim3(nb), dim3(1024), 0, h->stream, h->n0, h->nwn, h->g, h->N, ts, fo, h->foreign_h, tab, clear_in_dilate ? 1 : 0);
  }
  HIPCHK(hipGetLastError());
  if (halo(h, h->N.active, 1, 1, 1, overlap ? 1 : 0)) return 1;
  if ((h->deterministic || h->tile_ordering) && !h->order2_d) {
    HIPCHK(hipMalloc((void**)&h->order2_d, h->P.npad * sizeof(int)));
    HIPCHK(hipMemsetAsync(h->order2_d, 0, h->P.npad * sizeof(int), h->stream));
  }
  if (node_lists(h)) {  // both lists in one pass, the canonical one through the layer tables of this step (TileTab)
    TileTab tab;
    for (int a = 0; a < 3; a++) tab.nt[a] = h->nt[a];
    tab.node_cnt = h->node_cnt_d;
    tab.mask = h->tabm_d;
    tab.base = h->tabo_d;
    const int* adopt = ahead ? h->P.I0n : nullptr;
    if (h->nd == 2) hipLaunchKernelGGL(k_fill_orders<2>, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, h->P.tile, h->P.rank, h->nrank_d, h->P.I0, adopt, h->tile_start_d, h->g, tab, h->order_d, h->order2_d, deferred ? h->tile_cursor_d : (int*)nullptr, h->node_cnt_d);
    else hipLaunchKernelGGL(k_fill_orders<3>, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, h->P.tile, h->P.rank, h->nrank_d, h->P.I0, adopt, h->tile_start_d, h->g, tab, h->order_d, h->order2_d, deferred ? h->tile_cursor_d : (int*)nullptr, h->node_cnt_d);
  } else {
    if (ahead) hipLaunchKernelGGL(k_commit_I0, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, (const int*)h->P.I0n, h->P.I0);
    hipLaunchKernelGGL(k_fill_order, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, h->P.tile, h->P.rank, h->tile_start_d,
                       h->order_d);
    if (h->deterministic) {
      TileD td = tile_view(h, 0);
      if (h->nd == 2) hipLaunchKernelGGL((k_tile_order<2, true>), dim3(h->ntw), dim3(256), 0, h->stream, h->P, h->g, td, (const int*)h->order_d, h->order2_d);
      else hipLaunchKernelGGL((k_tile_order<3, true>), dim3(h->ntw), dim3(256), 0, h->stream, h->P, h->g, td, (const int*)h->order_d, h->order2_d);
    } else if (h->tile_ordering) {
      TileD td = tile_view(h, 0);
      if (h->nd == 2) hipLaunchKernelGGL((k_tile_order<2, false>), dim3(h->ntw), dim3(256), 0, h->stream, h->P, h->g, td, (const int*)h->order_d, h->order2_d);
      else hipLaunchKernelGGL((k_tile_order<3, false>), dim3(h->ntw), dim3(256), 0, h->stream, h->P, h->g, td, (const int*)h->order_d, h->order2_d);
    }
  }
  HIPCHK(hipGetLastError());
  if (h->timing) HIPCHK(hipEventRecord(h->ev[1], h->stream));
  if (overlap == 1) {
    launch_k2(h, p2g, 2, dt, gamma_nm);
    if (halo(h, h->N.active, 1, 1, 1, 2)) return 1;
    launch_k2(h, p2g, 1, dt, gamma_nm);
  } else if (overlap == 2) {  // the flags travelled behind the binning kernels; one launch, boundary tiles first
    if (halo(h, h->N.active, 1, 1, 1, 2)) return 1;
    launch_k2(h, p2g, 0, dt, gamma_nm, true);
  } else if (launch_lists_kernel) {
    launch_k2(h, p2g, 0, dt, gamma_nm);
  }
  HIPCHK(hipGetLastError());
  h->masks_valid = false;
  h->binned = true;
  return 0;
}

static int materialise_roll(nlps_gpu* h) {
  if (!h->rolled) return 0;
  LAUNCH_ND((k_copy_n_to_n1<2>), (k_copy_n_to_n1<3>), nblk(h->P.np), h->P, h->mats_d);
  HIPCHK(hipGetLastError());
  h->rolled = false;
  return 0;
}

// mirrored: the last kernel of the call has already stored the status word into the pinned host word (status_hd)
static int check_status(nlps_gpu* h, int fatal_mask, const char* where, bool mirrored) {
  // the status word travels to a pinned host word in stream order: one wait instead of a synchronise and a blocking copy
  if (!mirrored) HIPCHK(hipMemcpyAsync(h->status_h, h->gstatus_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  const int st = *(volatile int*)h->status_h;
  if (st & fatal_mask) {
    char buf[160];
    snprintf(buf, sizeof buf, "Error in %s: particle failure flags 0x%x (1 Newton, 2 connectivity, 4 J<=0, 8 law, 16 outside node window)",
             where, st);
    h->err = buf;
    fprintf(stderr, "\033[1;31m%s\033[0m\n", buf);
    return 1;
  }
  return 0;
}

extern "C" int nlps_gpu_initialize_lme(nlps_gpu* h) {
  // beta and lambda start at zero (Generate-One-Phase-Analysis.c:190-192) => first list is the
  // whole active 2-ring (LME.c:150-154)
  HIPCHK(hipMemsetAsync(h->P.d + (size_t)F_LAM * h->P.npad, 0, 7 * h->P.npad * sizeof(double), h->stream));
  if (search_and_lists(h, true, false, 0.0, 0.0)) return 1;
  if (compute_node_mask(h)) return 1;
  return check_status(h, ST_NEWTON | ST_CONNECT | ST_HALO, "initialize__LME__()");
}

// Driver_EigenErosion: compute_Beps__Constitutive__(.., true) runs once before the first step (U-Newmark-beta.c:182-183);
// what it sees -- positions and closest nodes -- is kept for the lists that stay frozen afterwards
static int beps_snapshot(nlps_gpu* h) {
  if (h->beps_snapshot || !h->P.erosion || h->P.softening || h->P.np == 0) return 0;
  hipLaunchKernelGGL(k_beps_snapshot, dim3(nblk(h->P.np)), dim3(BLK), 0, h->stream, h->P, h->nd);
  HIPCHK(hipGetLastError());
  h->beps_snapshot = true;
  return 0;
}

extern "C" int nlps_gpu_local_search(nlps_gpu* h) {
  if (beps_snapshot(h)) return 1;  // (before this search moves any closest node)
  if (search_and_lists(h, false, false, 0.0, 0.0)) return 1;
  if (compute_node_mask(h)) return 1;
  return check_status(h, ST_NEWTON | ST_CONNECT | ST_HALO, "local_search__LME__()");
}

static int ensure_bcs(nlps_gpu* h, const nlps_bcc* bcc, int nbcc) {
  bool same = (int)h->bcs.size() == nbcc;
  for (int i = 0; same && i < nbcc; i++) same = h->bcs[i].host_nodes == bcc[i].nodes && h->bcs[i].n == bcc[i].nnodes;
  if (same) return 0;
  if (materialise_nodal(h)) return 1;  // (the nodal arrays of the last folded step still want the old sets' mask)
  for (auto& b : h->bcs)
    if (b.dnodes) (void)hipFree(b.dnodes);
  h->bcs.clear();
  for (int i = 0; i < nbcc; i++) {
    BcDev b{bcc[i].nodes, bcc[i].nnodes, nullptr};
    if (b.n > 0) {
      HIPCHK(hipMalloc((void**)&b.dnodes, (size_t)b.n * sizeof(int)));
      HIPCHK(hipMemcpy(b.dnodes, bcc[i].nodes, (size_t)b.n * sizeof(int), hipMemcpyHostToDevice));
    }
    h->bcs.push_back(b);
  }
  if (!h->bcmask_d) HIPCHK(hipMalloc((void**)&h->bcmask_d, (size_t)h->g.nnodes * sizeof(unsigned)));
  HIPCHK(hipMemset(h->bcmask_d, 0, (size_t)h->g.nnodes * sizeof(unsigned)));
  if (nbcc <= NLPS_MAX_BC_INLINE)
    for (int i = 0; i < nbcc; i++)
      if (h->bcs[i].n > 0)
        hipLaunchKernelGGL(k_bc_mark, dim3(nblk(h->bcs[i].n)), dim3(BLK), 0, 0, h->bcs[i].dnodes, h->bcs[i].n, 1u << i,
                           h->bcmask_d);
  HIPCHK(hipGetLastError());
  // the copies above are ordered on the default stream only: a caller-provided non-blocking stream (torch
  // streams are) would not wait for their DMA
  HIPCHK(hipDeviceSynchronize());
  return 0;
}

// bcc[i].dir / value are indexed [k * nsteps + step] (Types.h:296-351): a step outside the range given at create
// would read past the caller's arrays
static int check_step(nlps_gpu* h, int step, const char* who) {
  if (step >= 0 && step < h->nsteps) return 0;
  h->err = std::string(who) + ": step outside [0, nsteps) (nsteps is fixed at nlps_gpu_create)";
  return 1;
}

static int dirbits_of(const nlps_bcc& b, int step, int nsteps) {
  int bits = 0;
  for (int k = 0; k < b.dim && k < 3; k++)
    if (b.dir[(size_t)k * nsteps + step] == 1) bits |= 1 << k;
  return bits;
}

extern "C" int nlps_gpu_active_masks(nlps_gpu* h, const nlps_bcc* bcc, int nbcc, int step, int* nactive,
                                     int* nfree_dofs, int* nodes2mask, int* dofs2mask) {
  int ND = h->nd, nn = h->g.nnodes;
  if (nbcc > 0 && check_step(h, step, "nlps_gpu_active_masks")) return 1;
  if (compute_node_mask(h)) return 1;
  HIPCHK(hipMemcpyAsync(&h->nactive, h->total_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  int order = h->nactive * ND;
  if (ensure_bcs(h, bcc, nbcc)) return 1;
  HIPCHK(hipMemsetAsync(h->fixedm_d, 0, (size_t)nn * ND, h->stream));
  for (int i = 0; i < nbcc; i++) {
    if (h->bcs[i].n == 0) continue;
    hipLaunchKernelGGL(k_mark_fixed_masked, dim3(nblk(h->bcs[i].n)), dim3(BLK), 0, h->stream, h->bcs[i].dnodes,
                       h->bcs[i].n, bcc[i].dim, dirbits_of(bcc[i], step, h->nsteps), ND, h->n2m_d, h->fixedm_d);
  }
  int nb = (order + 1023) / 1024;
  if (order > 0) {
    hipLaunchKernelGGL(k_scan_count, dim3(nb), dim3(256), 0, h->stream, h->fixedm_d, order, 1, h->bsum_d);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, h->stream, h->bsum_d, nb, h->total_d + 1);
    hipLaunchKernelGGL(k_scan_write, dim3(nb), dim3(256), 0, h->stream, h->fixedm_d, order, 1, h->bsum_d, h->d2m_d);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(&h->nfree, h->total_d + 1, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (order == 0) h->nfree = 0;
  if (nactive) *nactive = h->nactive;
  if (nfree_dofs) *nfree_dofs = h->nfree;
  // (with a file numbering set, Nodes2Mask is indexed by the file's node, like the reference's array)
  if (nodes2mask)
    HIPCHK(hipMemcpy(nodes2mask, h->canon_d ? h->mask_idx_d : h->n2m_d, (size_t)nn * sizeof(int), hipMemcpyDeviceToHost));
  if (dofs2mask && order) HIPCHK(hipMemcpy(dofs2mask, h->d2m_d, (size_t)order * sizeof(int), hipMemcpyDeviceToHost));
  h->masks_valid = true;
  h->lagr_valid = false;  // (the masked numbering may have changed: cached residual vectors are void)
  return 0;
}

// masked nodal vector of the caller (host or device) -> grid-numbered device array
static int to_grid(nlps_gpu* h, double* grid, const double* masked, int nf) {
  size_t n = (size_t)h->nactive * nf;
  const double* src = masked;
  if (!is_device_ptr(masked)) {
    if (ensure_masked(h, n)) return 1;
    HIPCHK(hipMemcpyAsync(h->maskedA, masked, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    src = h->maskedA;
  }
  hipLaunchKernelGGL(k_expand, dim3(nblk(h->g.nnodes)), dim3(BLK), 0, h->stream, grid, src, h->n2m_d, h->g.nnodes, nf);
  HIPCHK(hipGetLastError());
  return 0;
}

// grid-numbered device array -> caller's masked vector (mode: see k_compact)
static int from_grid(nlps_gpu* h, double* masked, const double* grid, int nf, int gstride, int goff, int bcast, int mode,
                     const double* div_masked) {
  size_t n = (size_t)h->nactive * nf;
  if (n == 0) return 0;
  bool dev_out = is_device_ptr(masked);
  if (ensure_masked(h, 2 * n)) return 1;
  double* dst = dev_out ? masked : h->maskedA;
  if (!dev_out && mode == 1) HIPCHK(hipMemcpyAsync(dst, masked, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  const double* div = div_masked;
  if (div_masked && !is_device_ptr(div_masked)) {
    HIPCHK(hipMemcpyAsync(h->maskedA + n, div_masked, n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    div = h->maskedA + n;
  }
  hipLaunchKernelGGL(k_compact, dim3(nblk(h->g.nnodes)), dim3(BLK), 0, h->stream, dst, grid, h->n2m_d, h->d2m_d, div,
                     h->g.nnodes, nf, gstride, goff, bcast, mode);
  HIPCHK(hipGetLastError());
  if (!dev_out) {
    HIPCHK(hipMemcpyAsync(masked, dst, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
  }
  return 0;
}

static int need_binning(nlps_gpu* h, const char* who) {
  if (h->binned) return 0;
  h->err = std::string(who) + ": call nlps_gpu_local_search() first (particles are not binned to tiles)";
  return 1;
}

static int need_masks(nlps_gpu* h, const char* who) {
  if (need_binning(h, who)) return 1;
  if (h->masks_valid) return 0;
  h->err = std::string(who) + ": call nlps_gpu_active_masks() after the local search first";
  return 1;
}

extern "C" int nlps_gpu_lumped_mass(nlps_gpu* h, double* M) {
  if (need_masks(h, "nlps_gpu_lumped_mass")) return 1;
  int ND = h->nd;
  HIPCHK(hipMemsetAsync(h->gridA, 0, (size_t)h->g.nnodes * sizeof(double), h->stream));
  {
    TileD td = tile_view(h);
    if (ND == 2) hipLaunchKernelGGL((kb_p2g_tile<2, 0>), dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, h->gridA);
    else hipLaunchKernelGGL((kb_p2g_tile<3, 0>), dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, h->gridA);
  }
  HIPCHK(hipGetLastError());
  if (halo(h, h->gridA, 1, 8, 0)) return 1;
  return from_grid(h, M, h->gridA, ND, 1, 0, 1, 0, nullptr);
}

extern "C" int nlps_gpu_nodal_field_n(nlps_gpu* h, double* V, double* A, const double* M) {
  if (need_masks(h, "nlps_gpu_nodal_field_n")) return 1;
  int ND = h->nd;
  HIPCHK(hipMemsetAsync(h->gridA, 0, (size_t)h->g.nnodes * 2 * ND * sizeof(double), h->stream));
  {
    TileD td = tile_view(h);
    if (ND == 2) hipLaunchKernelGGL((kb_p2g_tile<2, 1>), dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, h->gridA);
    else hipLaunchKernelGGL((kb_p2g_tile<3, 1>), dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, h->gridA);
  }
  HIPCHK(hipGetLastError());
  if (halo(h, h->gridA, 2 * ND, 8, 0)) return 1;
  if (from_grid(h, V, h->gridA, ND, 2 * ND, 0, 0, 2, M)) return 1;
  return from_grid(h, A, h->gridA, ND, 2 * ND, ND, 0, 2, M);
}

extern "C" int nlps_gpu_compatibility(nlps_gpu* h, const double* dU, const double* dU_dt) {
  if (need_masks(h, "nlps_gpu_compatibility")) return 1;
  if (materialise_roll(h)) return 1;
  if (to_grid(h, h->N.dU, dU, h->nd)) return 1;
  if (dU_dt && to_grid(h, h->gridB, dU_dt, h->nd)) return 1;
  if (dU_dt) h->level_b_fields = true;
  TileD td = tile_view(h);
  const dim3 grid(h->ntw * K3_SPLIT), blk(K3_BLK);
  const double* dV = dU_dt ? h->gridB : nullptr;
  if (h->nd == 2) {
    if (dV) hipLaunchKernelGGL((k3_tile<2, 0, 2>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, h->gstatus_d, dV);
    else hipLaunchKernelGGL((k3_tile<2, 0, 0>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, h->gstatus_d, dV);
  } else {
    if (dV) hipLaunchKernelGGL((k3_tile<3, 0, 2>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, h->gstatus_d, dV);
    else hipLaunchKernelGGL((k3_tile<3, 0, 0>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, h->gstatus_d, dV);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int nlps_gpu_constitutive(nlps_gpu* h) {
  if (materialise_roll(h)) return 1;
  h->level_b_fields = true;
  if (h->law_present & (1 << NLPS_KLAW_FRICTIONAL))
    LAUNCH_ND((k_stress<2, true>), (k_stress<3, true>), nblk(h->P.np), h->P, h->mats_d, h->prm, h->gstatus_d);
  else
    LAUNCH_ND((k_stress<2, false>), (k_stress<3, false>), nblk(h->P.np), h->P, h->mats_d, h->prm, h->gstatus_d);
  HIPCHK(hipGetLastError());
  return check_status(h, ST_CONSTITUTIVE, "Stress_integration__Constitutive__()");
}

extern "C" int nlps_gpu_internal_forces(nlps_gpu* h, double* R) {
  if (need_masks(h, "nlps_gpu_internal_forces")) return 1;
  if (materialise_roll(h)) return 1;
  int ND = h->nd;
  if (h->P.erosion && h->P.np > 0) {  // damage of every particle, then its Kirchhoff stress scaled in place (:1313-1331)
    const int np = h->P.np;
    const size_t nn = (size_t)h->g.nnodes;
    if (!h->dmg_first_d) {
      HIPCHK(hipMalloc((void**)&h->dmg_first_d, nn * sizeof(int)));
      HIPCHK(hipMalloc((void**)&h->dmg_last_d, nn * sizeof(int)));
    }
    HIPCHK(hipMemsetAsync(h->dmg_first_d, 0, nn * sizeof(int), h->stream));
    HIPCHK(hipMemsetAsync(h->dmg_last_d, 0, nn * sizeof(int), h->stream));
    hipLaunchKernelGGL(k_tangent_keys, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->P, h->skey_d, h->sval_d);
    size_t bytes = h->cub_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(h->cub_tmp, bytes, h->skey_d, h->skey2_d, h->sval_d, h->sval2_d, np, 0, 32,
                                              h->stream));
    hipLaunchKernelGGL(k_node_ranges, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, h->skey2_d, h->dmg_first_d, h->dmg_last_d);
    if (h->P.softening) {
      double* T0 = reinterpret_cast<double*>(h->gather_tmp);  // [npad] scratch of the re-sort, idle here
      LAUNCH_ND((k_soften_pass1<2>), (k_soften_pass1<3>), nblk(np), h->P, h->mats_d, T0);
      LAUNCH_ND((k_soften_pass2<2>), (k_soften_pass2<3>), nblk(np), h->P, h->g, h->mats_d, h->dmg_first_d, h->dmg_last_d, h->sval2_d,
                (const uint8_t*)h->rank1_d, (const int*)h->perm_d, (const double*)T0, h->g.h);
    } else {
      // frozen lists (Beps.c:30-36): the node tables of the snapshot's closest nodes, rebuilt per call like the others
      if (beps_snapshot(h)) return 1;
      if (!h->dmg_first0_d) {
        HIPCHK(hipMalloc((void**)&h->dmg_first0_d, nn * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->dmg_last0_d, nn * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->dmg_sorted0_d, h->P.npad * sizeof(int)));
      }
      std::swap(h->dmg_sorted0_d, h->sval2_d);  // dmg_sorted0_d: the CURRENT order just sorted; sval2_d: free for the next sort
      HIPCHK(hipMemsetAsync(h->dmg_first0_d, 0, nn * sizeof(int), h->stream));
      HIPCHK(hipMemsetAsync(h->dmg_last0_d, 0, nn * sizeof(int), h->stream));
      hipLaunchKernelGGL(k_beps_keys0, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->P, h->skey_d, h->sval_d);
      bytes = h->cub_tmp_bytes;
      HIPCHK(hipcub::DeviceRadixSort::SortPairs(h->cub_tmp, bytes, h->skey_d, h->skey2_d, h->sval_d, h->sval2_d, np, 0, 32,
                                                h->stream));
      hipLaunchKernelGGL(k_node_ranges, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, h->skey2_d, h->dmg_first0_d, h->dmg_last0_d);
      // (current order in dmg_sorted0_d, snapshot order in sval2_d)
      LAUNCH_ND((k_damage<2>), (k_damage<3>), nblk(np), h->P, h->g, h->mats_d, h->dmg_first_d, h->dmg_last_d, h->dmg_sorted0_d,
                h->dmg_first0_d, h->dmg_last0_d, h->sval2_d, h->g.h);
      std::swap(h->dmg_sorted0_d, h->sval2_d);  // the sort buffers go back to where the re-sort expects them
    }
    HIPCHK(hipGetLastError());
  }
  if (materialise_nodal(h)) return 1;  // (the forces of the last explicit step are about to go)
  HIPCHK(hipMemsetAsync(h->N.force, 0, (size_t)h->g.nnodes * ND * sizeof(double), h->stream));
  {
    TileD td = tile_view(h);
    if (ND == 2) hipLaunchKernelGGL(kb_fint_tile<2>, dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, h->N.force, h->gstatus_d);
    else hipLaunchKernelGGL(kb_fint_tile<3>, dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, h->N.force, h->gstatus_d);
  }
  HIPCHK(hipGetLastError());
  if (halo(h, h->N.force, ND, 8, 0)) return 1;
  return from_grid(h, R, h->N.force, ND, ND, 0, 0, 1, nullptr);
}

// __nodal_traction_forces (U-Newmark-beta.c:1376-1500): R_A -= N_pA T A0_p over the particles of the Neumann contours.
// One thread per listed particle (a contour holds few), global atomics into a grid array.
__global__ void k_inverse_perm(const int* __restrict__ perm, int np, int* __restrict__ inv) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s < np) inv[perm[s]] = s;
}
template <int ND>
__global__ void k_traction(PView P, GridD g, int n, const int* __restrict__ ids, const int* __restrict__ inv,
                           const double* __restrict__ T, const double* __restrict__ A0, double thickness,
                           double* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const int p = inv[ids[e]];
  Lme<ND> c;
  double lam[ND], beta;
  if (!load_lme<ND>(P, g, p, c, lam, beta)) return;
  const double a0 = A0 ? A0[e] : PF(P, F_VOL0, p) / thickness;  // :1440-1444
  const double zinv = lme_zinv<ND>(c);
  for (int k = 0; k < Lme<ND>::KN; k++)
    for (int j = 0; j < 5; j++)
      for (int i = 0; i < 5; i++) {
        if (!c.on(i + 5 * j + 25 * k)) continue;
        const double Npa = c.ex[i] * c.ey[j] * (ND == 3 ? c.ez[k % Lme<ND>::KN] : 1.0) * zinv;
        const int node = c.I0 + c.node_offset(g, i, j, k + (ND == 3 ? 0 : 2));
        for (int a = 0; a < ND; a++) atomic_add_f64(out + (size_t)node * ND + a, -Npa * T[(size_t)e * ND + a] * a0);
      }
}

// the traction sums of the Neumann contours in grid numbering: out[nnodes][ND] = - sum N_pA T A0_p (zeroed first);
// *any = 0 when the contours hold no particle (out is then left alone)
static int traction_to_grid(nlps_gpu* h, const char* who, double* out, const nlps_bcc* loads, int nloads, int step,
                            double thickness, const double* area0, int* any) {
  const int ND = h->nd, np = h->P.np;
  *any = 0;
  if (step < 0 || step >= h->nsteps) {
    h->err = std::string(who) + ": step outside [0, nsteps)";
    return 1;
  }
  if (ND == 3 && !area0) {
    h->err = std::string(who) + ": the 3-D build needs Phi.Area_0 (area0)";
    return 1;
  }
  if (h->migrated) {
    h->err = std::string(who) + ": not available after a migration (particle order is by global id)";
    return 1;
  }
  // the traction vector carries over from contour to contour where a direction is switched off (:1457-1461)
  std::vector<int> ids;
  std::vector<double> T, A;
  double Tc[3] = {0.0, 0.0, 0.0};
  for (int l = 0; l < nloads; l++) {
    for (int i = 0; i < ND; i++)
      if (loads[l].dir[(size_t)i * h->nsteps + step] == 1) Tc[i] = loads[l].value[(size_t)i * h->nsteps + step];
    for (int q = 0; q < loads[l].nnodes; q++) {
      const int p = loads[l].nodes[q];
      if (p < 0 || p >= np) {
        h->err = std::string(who) + ": particle index outside the cloud";
        return 1;
      }
      ids.push_back(p);
      for (int i = 0; i < ND; i++) T.push_back(Tc[i]);
      if (ND == 3) A.push_back(area0[p]);
    }
  }
  const int n = (int)ids.size();
  if (n == 0) return 0;
  *any = 1;
  int* ids_d = nullptr;
  double *T_d = nullptr, *A_d = nullptr;
  HIPCHK(hipMalloc((void**)&ids_d, (size_t)n * sizeof(int)));
  HIPCHK(hipMalloc((void**)&T_d, (size_t)n * ND * sizeof(double)));
  if (ND == 3) HIPCHK(hipMalloc((void**)&A_d, (size_t)n * sizeof(double)));
  HIPCHK(hipMemcpyAsync(ids_d, ids.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(T_d, T.data(), (size_t)n * ND * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (ND == 3) HIPCHK(hipMemcpyAsync(A_d, A.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_inverse_perm, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->perm_d, np, h->sval_d);
  HIPCHK(hipMemsetAsync(out, 0, (size_t)h->g.nnodes * ND * sizeof(double), h->stream));
  if (ND == 2) hipLaunchKernelGGL(k_traction<2>, dim3(nblk(n)), dim3(BLK), 0, h->stream, h->P, h->g, n, ids_d, h->sval_d, T_d, A_d, thickness, out);
  else hipLaunchKernelGGL(k_traction<3>, dim3(nblk(n)), dim3(BLK), 0, h->stream, h->P, h->g, n, ids_d, h->sval_d, T_d, A_d, thickness, out);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(h->stream));  // (the host vectors above are about to go)
  (void)hipFree(ids_d);
  (void)hipFree(T_d);
  if (A_d) (void)hipFree(A_d);
  return 0;
}

extern "C" int nlps_gpu_nodal_traction_forces(nlps_gpu* h, double* R, const nlps_bcc* loads, int nloads, int step,
                                              double thickness, const double* area0) {
  if (need_masks(h, "nlps_gpu_nodal_traction_forces")) return 1;
  int any = 0;
  if (traction_to_grid(h, "nlps_gpu_nodal_traction_forces", h->gridA, loads, nloads, step, thickness, area0, &any)) return 1;
  if (!any) return 0;
  return from_grid(h, R, h->gridA, h->nd, h->nd, 0, 0, 1, nullptr);
}

extern "C" int nlps_gpu_roll_state(nlps_gpu* h) {
  if (materialise_roll(h)) return 1;
  LAUNCH_ND((k_roll<2>), (k_roll<3>), nblk(h->P.np), h->P);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int nlps_gpu_update_kinetics(nlps_gpu* h, double alpha_blend, const double* dU, const double* Un_dt,
                                        const double* dU_dt, const double* dU_dt2) {
  if (need_masks(h, "nlps_gpu_update_kinetics")) return 1;
  h->searched = h->ahead = false;  // the particles move: the next search is a search
  int ND = h->nd;
  size_t st = (size_t)h->g.nnodes * ND;
  // Un_dt = dU_dt = dU_dt2 = NULL: the quasi-static driver's __update_Particles (U-Static.c:1380-1470) moves the
  // particles by sum N dU and leaves velocity and acceleration alone
  const bool quasi_static = !Un_dt && !dU_dt && !dU_dt2;
  if (!quasi_static && (!Un_dt || !dU_dt || !dU_dt2)) {
    h->err = "nlps_gpu_update_kinetics: pass all of Un_dt, dU_dt, dU_dt2 or none of them (quasi-static update)";
    return 1;
  }
  if (!dU) {
    h->err = "nlps_gpu_update_kinetics: dU is NULL";
    return 1;
  }
  if (to_grid(h, h->gridB, dU, ND)) return 1;
  if (quasi_static) {
    HIPCHK(hipMemsetAsync(h->gridB + st, 0, 3 * st * sizeof(double), h->stream));
  } else {
    if (to_grid(h, h->gridB + st, Un_dt, ND)) return 1;
    if (to_grid(h, h->gridB + 2 * st, dU_dt, ND)) return 1;
    if (to_grid(h, h->gridB + 3 * st, dU_dt2, ND)) return 1;
  }
  {
    TileD td = tile_view(h);
    const int qs = quasi_static ? 1 : 0;
    if (ND == 2) hipLaunchKernelGGL(kb_kinetics_tile<2>, dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, alpha_blend, h->gridB, h->gridB + st, h->gridB + 2 * st, h->gridB + 3 * st, qs);
    else hipLaunchKernelGGL(kb_kinetics_tile<3>, dim3(h->ntw), dim3(BLK), 0, h->stream, h->P, h->g, td, alpha_blend, h->gridB, h->gridB + st, h->gridB + 2 * st, h->gridB + 3 * st, qs);
  }
  HIPCHK(hipGetLastError());
  return 0;
}

__global__ void k_null_bracket(PView, GridD, NView, TileD, const MatD*, ParamsD, int*, const double*) {}

extern "C" int nlps_gpu_explicit_step(nlps_gpu* h, const nlps_bcc* bcc, int nbcc, int step, double dt, double gamma_nm,
                                      const double* gravity) {
  int ND = h->nd;
  if (h->P.erosion) {
    h->err = "nlps_gpu_explicit_step: the eigenerosion hooks exist in the level-B stages only (the reference defines them in "
             "U-Newmark-beta.c / U-Static.c; its explicit drivers are stubs)";
    return 1;
  }
  if (nbcc > 0 && check_step(h, step, "nlps_gpu_explicit_step")) return 1;
  if (ensure_bcs(h, bcc, nbcc)) return 1;
  if (!h->Pd_alt && h->resort_every > 0 && h->P.np > 0) {
    // the twin block of the periodic re-sort, at the first step of the fused scheme rather than inside the first
    // re-sort: a hipMalloc of this size (1.3 GB per million particles) takes milliseconds, the re-sort itself 0.3
    HIPCHK(hipMalloc((void**)&h->Pd_alt, (size_t)NFD * h->P.npad * sizeof(double)));
    HIPCHK(hipMemsetAsync(h->Pd_alt, 0, (size_t)NFD * h->P.npad * sizeof(double), h->stream));
  }
  if (!h->rolled && h->P.np > 0) {  // entering the fused scheme: rho J of every particle (see F_RHOJ)
    hipLaunchKernelGGL(k_init_rhoj, dim3(nblk(h->P.np)), dim3(BLK), 0, h->stream, h->P);
    HIPCHK(hipGetLastError());
  }
  // periodic re-sort; earlier when enough particles have left the tiles their memory slots were sorted into: the count
  // of a recent step sits in the pinned word (no synchronisation: it may be a step old), its share of the cloud is this
  // step's cost estimate, and the re-sort comes when the estimates since the last one add up to the budget
  bool drifted = false;
  if (h->adaptive_resort > 0.0 && !h->deterministic && h->resort_every > 0 && h->P.np > 0) {  // (a re-sort moment read
    // from an asynchronous word is not reproducible: the deterministic mode keeps the fixed interval)
    h->debt += (double)*(volatile int*)h->foreign_h / (double)h->P.np;
    drifted = h->debt > h->adaptive_resort && h->steps_since_sort >= h->adaptive_min_steps;
  }
  if (h->resort_every > 0 && (h->steps_since_sort >= h->resort_every || drifted)) {
    if (resort(h, nullptr, true)) return 1;
  }
  h->steps_since_sort++;
  h->nwait = 0;
  if (h->timing) HIPCHK(hipEventRecord(h->ev[0], h->stream));
  // With a halo callback and ghost bands set, every exchange is started right after the tiles that touch a
  // ghost band have produced their part and is waited for only before those tiles need the result; the tiles
  // (and nodes) away from the bands run in between, on the handle's stream, while the exchange proceeds on the
  // callee's stream.  Without overlap each stage is one pass over all tiles and the exchange blocks in place.
  const bool ov2 = h->rccl && h->overlap == 2 && !h->deterministic;  // one launch per stage, exchange behind its interior tiles
  const bool ov = !ov2 && (h->halo || h->rccl) && h->overlap;
  double gv[3] = {0, 0, 0};
  if (gravity)
    for (int a = 0; a < ND; a++) gv[a] = gravity[a];
  const bool det = h->deterministic;
  if (det && !h->slab_d) {
    const size_t NWs = ND == 3 ? TileCfg<3>::NWA : TileCfg<2>::NWA;
    const size_t per_tile = std::max<size_t>((size_t)(1 + ND), (size_t)4 * ND) * NWs;  // K2: one slab; K3: one per law
    HIPCHK(hipMalloc((void**)&h->slab_d, (size_t)h->ntiles * per_tile * sizeof(double)));
  }
  // second half of the P2G flush: per node, the window slabs of the tiles that hold it (part: node ranges as below)
  auto gather_nm = [&](int part) {
    const NodeRanges r = node_ranges(h, part);
    if (!det || r.an + r.bn == 0) return;
    TileD td = tile_view(h, 0);
    if (ND == 2) hipLaunchKernelGGL((k_slab_gather<2, 3>), dim3(nblk(r.an + r.bn)), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->g, td, h->N.nm);
    else hipLaunchKernelGGL((k_slab_gather<3, 4>), dim3(nblk(r.an + r.bn)), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->g, td, h->N.nm);
  };
  auto gather_force = [&](int part) {
    const NodeRanges r = node_ranges(h, part);
    if (!det || r.an + r.bn == 0) return;
    TileD td = tile_view(h, 0);
    td.slab_n = __builtin_popcount(h->law_present);
    if (ND == 2) hipLaunchKernelGGL((k_slab_gather<2, 2>), dim3(nblk(r.an + r.bn)), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->g, td, h->N.force);
    else hipLaunchKernelGGL((k_slab_gather<3, 3>), dim3(nblk(r.an + r.bn)), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->g, td, h->N.force);
  };
  const bool fuse_early = h->fuse_search && h->P.np > 0;
  // Folded form (k3_tile_lazy / k5_tile_lazy): one law, the Dirichlet sets small enough to travel as kernel arguments:
  // no nodal kernels between the stages (dU, accelerations, reactions are made later if somebody asks).  With a ghost
  // exchange too, in every overlap form: a K3 / K5 launch comes behind the pick-up of the exchange its nodes need (interior
  // tiles never touch a band node), exactly where the nodal kernel of its node range stood.
  // (measured: 3 % faster per step at 1 M particles -- three launches less -- and equal within the noise at 4 M and 8 M,
  // where the window loads of 17 k tiles redo the two divisions per node 16 times over: on below 2 M particles,
  // NLPS_LAZY_NODAL=2 always)
  const bool lazy = (h->lazy_nodal == 2 || (h->lazy_nodal == 1 && h->P.np <= 2000000)) && fuse_early && h->fuse_search == 1 && !det && h->uniform_law >= 0 &&
                    h->uniform_law <= NLPS_KLAW_FRICTIONAL && nbcc <= NLPS_MAX_BC_INLINE;
  LazyNodal ln;
  if (lazy) {
    memset(&ln, 0, sizeof ln);
    ln.fs.bc.n = nbcc;
    for (int i = 0; i < nbcc; i++) {
      ln.fs.bc.dim[i] = bcc[i].dim;
      ln.fs.bc.bits[i] = h->bcs[i].n > 0 ? dirbits_of(bcc[i], step, h->nsteps) : 0;
      for (int k = 0; k < 3; k++) ln.fs.bc.v[i][k] = (k < bcc[i].dim) ? bcc[i].value[(size_t)k * h->nsteps + step] : 0.0;
    }
    ln.fs.bcmask = nbcc > 0 ? h->bcmask_d : nullptr;
    for (int a = 0; a < 3; a++) ln.fs.gv[a] = gv[a];
    ln.n0 = h->n0;
    ln.nwn = h->nwn;
    ln.node_cnt = node_lists(h) ? h->node_cnt_d : nullptr;
    ln.ntw = h->ntw;  // (ln.tile_count: after search_and_lists, which swaps the two counter arrays)
  }
  auto nodal_dU = [&](int part) {
    if (lazy) return;  // (K3 makes dU of its window nodes itself)
    const NodeRanges r = node_ranges(h, part);
    if (r.an + r.bn == 0) return;
    BcStep bs;
    bs.n = 0;
    const bool inline_bc = nbcc <= NLPS_MAX_BC_INLINE;
    if (inline_bc) {
      bs.n = nbcc;
      for (int i = 0; i < nbcc; i++) {
        bs.dim[i] = bcc[i].dim;
        bs.bits[i] = h->bcs[i].n > 0 ? dirbits_of(bcc[i], step, h->nsteps) : 0;
        for (int k = 0; k < 3; k++) bs.v[i][k] = (k < bcc[i].dim) ? bcc[i].value[(size_t)k * h->nsteps + step] : 0.0;
      }
    }
    const unsigned* bm = (inline_bc && nbcc > 0) ? h->bcmask_d : nullptr;
    if (ND == 2) hipLaunchKernelGGL(k_nodal_dU<2>, dim3(nblk(r.an + r.bn)), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->N, bm, bs);
    else hipLaunchKernelGGL(k_nodal_dU<3>, dim3(nblk(r.an + r.bn)), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->N, bm, bs);
    if (inline_bc) return;
    // Dirichlet values (nodes of the same range only)
    const NodeRanges in = node_ranges(h, 1);
    for (int i = 0; i < nbcc; i++) {
      if (h->bcs[i].n == 0) continue;
      double v[3] = {0, 0, 0};
      for (int k = 0; k < bcc[i].dim && k < 3; k++) v[k] = bcc[i].value[(size_t)k * h->nsteps + step];
      int bits = dirbits_of(bcc[i], step, h->nsteps);
      const int r0 = part == 0 ? 0 : in.a0, r1 = part == 0 ? h->g.nnodes : in.a0 + in.an, inside = part == 2 ? 0 : 1;
      if (ND == 2)
        hipLaunchKernelGGL(k_bc<2>, dim3(nblk(h->bcs[i].n)), dim3(BLK), 0, h->stream, h->bcs[i].dnodes, h->bcs[i].n,
                           bcc[i].dim, bits, v[0], v[1], v[2], h->N, r0, r1, inside);
      else
        hipLaunchKernelGGL(k_bc<3>, dim3(nblk(h->bcs[i].n)), dim3(BLK), 0, h->stream, h->bcs[i].dnodes, h->bcs[i].n,
                           bcc[i].dim, bits, v[0], v[1], v[2], h->N, r0, r1, inside);
    }
  };
  // the search of the next step rides on K5 (k5_tile<., ., true>) unless the lists must come from an exact sort
  const bool fuse = h->fuse_search && h->P.np > 0;
  bool tiles_cleared = false;
  auto nodal_accel = [&](int part) {
    if (lazy) return;  // (K5 makes the accelerations itself; K3's workgroups have reset the search accumulators)
    const NodeRanges r = node_ranges(h, part);
    const bool fbin = fuse && h->fuse_search == 1;
    int* ctile = (fbin && !tiles_cleared) ? h->tile_count2_d + h->tile0 : nullptr;
    if (r.an + r.bn == 0 && !ctile) return;
    tiles_cleared = true;
    int* cnode = (fbin && node_lists(h)) ? h->node_cnt_d : nullptr;
    if (ND == 2) hipLaunchKernelGGL(k_nodal_accel<2>, dim3(nblk(std::max(1, r.an + r.bn))), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->N, gv[0], gv[1], gv[2], fbin ? 1 : 0, cnode, ctile, h->ntw);
    else hipLaunchKernelGGL(k_nodal_accel<3>, dim3(nblk(std::max(1, r.an + r.bn))), dim3(BLK), 0, h->stream, r.a0, r.an, r.b0, r.bn, h->N, gv[0], gv[1], gv[2], fbin ? 1 : 0, cnode, ctile, h->ntw);
  };
  auto launch_k3 = [&](int cls, bool signal = false) {
    TileD td = tile_view(h, cls);
    if (signal) {  // several laws: only the last launch releases the exchange (launches of one stream run in order)
      if (h->uniform_law >= 0 || !h->k3_per_law) arm_signal(h, td, 1);
    }
#define NLPS_K3(NDv, LAWv)                                                                                      \
  do {                                                                                                          \
    hipLaunchKernelGGL((k3_tile<NDv, LAWv, 1>), dim3(h->ntw * K3_SPLIT), dim3(K3_BLK), 0, h->stream, h->P, h->g,          \
                       h->N, td, h->mats_d, h->prm, h->gstatus_d, (const double*)nullptr);                      \
  } while (0)
#define NLPS_K3D(NDv, LAWv)                                                                                     \
  hipLaunchKernelGGL((k3_tile<NDv, LAWv, 1, true, 64>), dim3(h->ntw), dim3(64), 0, h->stream, h->P, h->g, h->N, td, \
                     h->mats_d, h->prm, h->gstatus_d, (const double*)nullptr)
    if (det) {  // one wave per tile and per law present, particles in list order, one slab per (tile, law)
      td.slab_n = __builtin_popcount(h->law_present);
      td.slab_slot = -1;
      for (int l = 0; l <= NLPS_KLAW_FRICTIONAL; l++) {
        if (!(h->law_present & (1 << l))) continue;
        td.slab_slot++;
        if (ND == 2) {
          if (l == 0) NLPS_K3D(2, 0);
          else if (l == 1) NLPS_K3D(2, 1);
          else if (l == 2) NLPS_K3D(2, 2);
          else if (l == 3) NLPS_K3D(2, 3);
          else NLPS_K3D(2, 4);
        } else {
          if (l == 0) NLPS_K3D(3, 0);
          else if (l == 1) NLPS_K3D(3, 1);
          else if (l == 2) NLPS_K3D(3, 2);
          else if (l == 3) NLPS_K3D(3, 3);
          else NLPS_K3D(3, 4);
        }
      }
      return;
    }
#undef NLPS_K3D
#define NLPS_K3F(NDv, LAWv)                                                                                     \
  hipLaunchKernelGGL((k3_tile<NDv, LAWv, 1, true>), dim3(h->ntw * K3_SPLIT), dim3(K3_BLK), 0, h->stream, h->P, h->g, h->N, td, \
                     h->mats_d, h->prm, h->gstatus_d, (const double*)nullptr)
    const int law = h->uniform_law;
#define NLPS_K3U(LAWv)                                                                                          \
  hipLaunchKernelGGL((k3_tile<3, LAWv, 1, false, K3_BLK, true>), dim3(h->ntw * K3_SPLIT), dim3(K3_BLK), 0, h->stream, h->P, \
                     h->g, h->N, td, h->mats_d, h->prm, h->gstatus_d, (const double*)nullptr)
    if (ND == 3 && h->nmats == 1 && law >= 1 && law <= 3) {  // one material (k3_body, UMAT)
      if (law == 1) NLPS_K3U(1);
      else if (law == 2) NLPS_K3U(2);
      else NLPS_K3U(3);
    } else if (ND == 2) {
      if (law == 0) NLPS_K3(2, 0);
      else if (law == 1) NLPS_K3(2, 1);
      else if (law == 2) NLPS_K3(2, 2);
      else if (law == 3) NLPS_K3(2, 3);
      else if (law == 4) NLPS_K3(2, 4);
      else if (!h->k3_per_law) NLPS_K3(2, -1);
      else {  // several laws in the cloud: one launch of the single-law kernel per law present
        const int last = 31 - __builtin_clz((unsigned)h->law_present);
        for (int l = 0; l <= NLPS_KLAW_FRICTIONAL; l++) {
          if (!(h->law_present & (1 << l))) continue;
          if (signal && l == last) arm_signal(h, td, 1);
          if (l == 0) NLPS_K3F(2, 0);
          else if (l == 1) NLPS_K3F(2, 1);
          else if (l == 2) NLPS_K3F(2, 2);
          else if (l == 3) NLPS_K3F(2, 3);
          else NLPS_K3F(2, 4);
        }
      }
    } else {
      if (law == 0) NLPS_K3(3, 0);
      else if (law == 1) NLPS_K3(3, 1);
      else if (law == 2) NLPS_K3(3, 2);
      else if (law == 3) NLPS_K3(3, 3);
      else if (law == 4) NLPS_K3(3, 4);
      else if (!h->k3_per_law) NLPS_K3(3, -1);
      else {
        const int last = 31 - __builtin_clz((unsigned)h->law_present);
        for (int l = 0; l <= NLPS_KLAW_FRICTIONAL; l++) {
          if (!(h->law_present & (1 << l))) continue;
          if (signal && l == last) arm_signal(h, td, 1);
          if (l == 0) NLPS_K3F(3, 0);
          else if (l == 1) NLPS_K3F(3, 1);
          else if (l == 2) NLPS_K3F(3, 2);
          else if (l == 3) NLPS_K3F(3, 3);
          else NLPS_K3F(3, 4);
        }
      }
    }
#undef NLPS_K3F
#undef NLPS_K3
#undef NLPS_K3U
  };
  K5Search ks;
  ks.rank1 = h->rank1_d;
  ks.bin = h->fuse_search == 1;
  bool ks_made = false;
  auto launch_k5 = [&](int cls) {
    TileD td = tile_view(h, cls);
    if (fuse && !ks_made) {  // (one TileCnt per step: it consumes the re-home flag of the adaptive re-sort)
      ks.tc = tile_cnt(h, true);
      ks.tc.count = h->tile_count2_d;  // (tile_count_d sizes the lists this very launch walks)
      ks.tc.defer = (h->defer_ranks && ks.bin && node_lists(h)) ? 1 : 0;
      h->ranks_deferred = ks.tc.defer != 0;
      ks_made = true;
    }
#define NLPS_K5(NDv, LAWv)                                                                               \
  do {                                                                                                   \
    if (fuse)                                                                                            \
      hipLaunchKernelGGL((k5_tile<NDv, LAWv, true>), dim3(h->ntw * K5_SPLIT), dim3(K5_BLK), 0, h->stream, h->P, h->g, \
                         h->N, td, dt, gamma_nm, ks);                                                    \
    else                                                                                                 \
      hipLaunchKernelGGL((k5_tile<NDv, LAWv, false>), dim3(h->ntw * K5_SPLIT), dim3(K5_BLK), 0, h->stream, h->P, h->g, \
                         h->N, td, dt, gamma_nm, ks);                                                    \
  } while (0)
    const int law = h->uniform_law;
    if (ND == 2) {
      if (law == 0 || law == 1) NLPS_K5(2, 0);
      else NLPS_K5(2, 2);
    } else {
      if (law == 0 || law == 1) NLPS_K5(3, 0);
      else NLPS_K5(3, 2);
    }
#undef NLPS_K5
  };
  auto launch_k3_lazy = [&](int cls, bool signal) {
    TileD td = tile_view(h, cls);
    if (signal) arm_signal(h, td, 1);
    ln.tile_count = h->tile_count2_d + h->tile0;  // (after search_and_lists, which swaps the two counter arrays)
#define NLPS_K3L(NDv, LAWv)                                                                                              \
  hipLaunchKernelGGL((k3_tile_lazy<NDv, LAWv>), dim3(h->ntw * K3_SPLIT), dim3(K3_BLK), 0, h->stream, h->P, h->g, h->N, td, \
                     h->mats_d, h->prm, h->gstatus_d, ln)
#define NLPS_K3LU(LAWv)                                                                                                  \
  hipLaunchKernelGGL((k3_tile_lazy<3, LAWv, true>), dim3(h->ntw * K3_SPLIT), dim3(K3_BLK), 0, h->stream, h->P, h->g, h->N, td, \
                     h->mats_d, h->prm, h->gstatus_d, ln)
    const int law = h->uniform_law;
    if (ND == 3 && h->nmats == 1 && law >= 1 && law <= 3) {  // one material: its constants by scalar loads (k3_body, UMAT)
      if (law == 1) NLPS_K3LU(1);
      else if (law == 2) NLPS_K3LU(2);
      else NLPS_K3LU(3);
    } else if (ND == 2) {
      if (law == 0) NLPS_K3L(2, 0);
      else if (law == 1) NLPS_K3L(2, 1);
      else if (law == 2) NLPS_K3L(2, 2);
      else if (law == 3) NLPS_K3L(2, 3);
      else NLPS_K3L(2, 4);
    } else {
      if (law == 0) NLPS_K3L(3, 0);
      else if (law == 1) NLPS_K3L(3, 1);
      else if (law == 2) NLPS_K3L(3, 2);
      else if (law == 3) NLPS_K3L(3, 3);
      else NLPS_K3L(3, 4);
    }
#undef NLPS_K3L
#undef NLPS_K3LU
  };
  auto launch_k5_lazy = [&](int cls) {
    const TileD td = tile_view(h, cls);
    if (!ks_made) {  // (one TileCnt per step: it consumes the re-home flag of the adaptive re-sort)
      ks.tc = tile_cnt(h, true);
      ks.tc.count = h->tile_count2_d;
      ks.tc.defer = (h->defer_ranks && ks.bin && node_lists(h)) ? 1 : 0;
      h->ranks_deferred = ks.tc.defer != 0;
      ks_made = true;
    }
#define NLPS_K5L(NDv, LAWv)                                                                                             \
  hipLaunchKernelGGL((k5_tile_lazy<NDv, LAWv>), dim3(h->ntw * K5_SPLIT), dim3(K5_BLK), 0, h->stream, h->P, h->g, h->N, td, \
                     dt, gamma_nm, ks, ln, h->gstatus_d)
    const int law = h->uniform_law;
    if (ND == 2) {
      if (law == 0 || law == 1) NLPS_K5L(2, 0);
      else NLPS_K5L(2, 2);
    } else {
      if (law == 0 || law == 1) NLPS_K5L(3, 0);
      else NLPS_K5L(3, 2);
    }
#undef NLPS_K5L
  };
  auto k3 = [&](int cls, bool signal = false) {
    if (lazy) launch_k3_lazy(cls, signal);
    else launch_k3(cls, signal);
  };
  auto k5 = [&](int cls) {
    if (lazy) launch_k5_lazy(cls);
    else launch_k5(cls);
  };
  h->nodal_stale = false;
  // S1 + S2 (ev[1] is recorded between the search and the lists/Newton/P2G kernel; the nodal accumulators of
  // the node window are reset by k_step_clear inside search_and_lists)
  if (search_and_lists(h, false, true, dt, gamma_nm, ov2 ? 2 : (ov ? 1 : 0))) return 1;
  if (h->timing) HIPCHK(hipEventRecord(h->ev[2], h->stream));
  if (lazy) {  // (what nlps_gpu_explicit_nodal needs to make the nodal arrays of this step later)
    h->nodal_stale = true;
    h->last_bc = ln.fs.bc;
    h->last_bm = ln.fs.bcmask;
    for (int a = 0; a < 3; a++) h->last_gv[a] = gv[a];
  }
  if (ov2) {
    // mode 2: K2 is in flight as ONE launch; its boundary tiles release the exchange of the shared layers of nm, which
    // runs beside its interior tiles; the handle's stream picks the result up before the nodal kernel
    if (halo(h, h->N.nm, 1 + ND, 8, 0, 3)) return 1;
    if (halo(h, h->N.nm, 1 + ND, 8, 0, 2)) return 1;
    nodal_dU(0);
    if (h->timing) HIPCHK(hipEventRecord(h->ev[3], h->stream));
    k3(0, true);
    HIPCHK(hipGetLastError());
    if (h->timing) HIPCHK(hipEventRecord(h->ev[4], h->stream));
    if (halo(h, h->N.force, ND, 8, 0, 4)) return 1;
    if (halo(h, h->N.force, ND, 8, 0, 2)) return 1;
    nodal_accel(0);
    if (h->timing) HIPCHK(hipEventRecord(h->ev[5], h->stream));
    k5(0);
    HIPCHK(hipGetLastError());
  } else {
  // with ghost bands the shared layers are gathered first (only boundary tiles reach them) and go on their way
  // while the rest is summed
  gather_nm(ov ? 2 : 0);
  if (halo(h, h->N.nm, 1 + ND, 8, 0, ov ? 1 : 0)) return 1;
  // nodal dU = (sum m N dD) / M + Dirichlet values; S3 + S4
  if (ov) {
    gather_nm(1);
    nodal_dU(1);
    // timing brackets: the K3 bucket starts before the interior tiles (the band nodes' dU then counts as K3 time)
    if (h->timing) HIPCHK(hipEventRecord(h->ev[3], h->stream));
    k3(2);
    if (halo(h, h->N.nm, 1 + ND, 8, 0, 2)) return 1;
    nodal_dU(2);
    k3(1);
  } else {
    nodal_dU(0);
    if (h->timing) HIPCHK(hipEventRecord(h->ev[3], h->stream));
    k3(0);
  }
  HIPCHK(hipGetLastError());
  if (h->timing) HIPCHK(hipEventRecord(h->ev[4], h->stream));
  gather_force(ov ? 2 : 0);
  if (halo(h, h->N.force, ND, 8, 0, ov ? 1 : 0)) return 1;
  // nodal acceleration; S5
  if (ov) {
    gather_force(1);
    nodal_accel(1);
    if (h->timing) HIPCHK(hipEventRecord(h->ev[5], h->stream));
    k5(2);
    if (halo(h, h->N.force, ND, 8, 0, 2)) return 1;
    nodal_accel(2);
    k5(1);
  } else {
    nodal_accel(0);
    if (h->timing) HIPCHK(hipEventRecord(h->ev[5], h->stream));
    k5(0);
  }
  HIPCHK(hipGetLastError());
  }  // !ov2
  h->P.flip ^= 1;  // F_n <- F_n+1, b_e,n <- b_e,n+1 by renaming
  h->rolled = true;
  h->searched = fuse;                      // K5 has updated the closest nodes for the positions it wrote ...
  h->ahead = fuse && h->fuse_search == 1;  // ... and binned the particles for the next step
  if (h->timing) {
    HIPCHK(hipEventRecord(h->ev[6], h->stream));
    // calibration bracket: a kernel of K3's grid and argument block that does nothing; what it reads is the part of
    // a one-kernel bracket that is not kernel time (records, dispatch, launch of the empty grid)
    hipLaunchKernelGGL(k_null_bracket, dim3(h->ntw * K3_SPLIT), dim3(BLK), 0, h->stream, h->P, h->g, h->N,
                       tile_view(h, 0), h->mats_d, h->prm, h->gstatus_d, (const double*)nullptr);
    HIPCHK(hipEventRecord(h->ev[7], h->stream));
    HIPCHK(hipEventSynchronize(h->ev[7]));
    float t01, t12, t23, t34, t45, t56, t67;
    HIPCHK(hipEventElapsedTime(&t67, h->ev[6], h->ev[7]));
    h->ms[5] = t67;
    HIPCHK(hipEventElapsedTime(&t01, h->ev[0], h->ev[1]));
    HIPCHK(hipEventElapsedTime(&t12, h->ev[1], h->ev[2]));
    HIPCHK(hipEventElapsedTime(&t23, h->ev[2], h->ev[3]));
    HIPCHK(hipEventElapsedTime(&t34, h->ev[3], h->ev[4]));
    HIPCHK(hipEventElapsedTime(&t45, h->ev[4], h->ev[5]));
    HIPCHK(hipEventElapsedTime(&t56, h->ev[5], h->ev[6]));
    h->ms[0] = t01;
    h->ms[1] = t12;
    h->ms[2] = t34;
    h->ms[3] = t56;
    h->ms[4] = t23 + t45;
    h->ms[7] = 0.f;
    h->ms[6] = 0.f;  // time the handle's stream spent in / waiting for the ghost-layer exchanges of this step
    for (int q = 0; q < h->nwait; q++) {
      float tw = 0.f;
      HIPCHK(hipEventElapsedTime(&tw, h->evw[2 * q], h->evw[2 * q + 1]));
      h->ms[6] += tw;
    }
  }
  return 0;
}

extern "C" int nlps_gpu_num_active(nlps_gpu* h, int* nactive) {
  if (compute_node_mask(h)) return 1;
  HIPCHK(hipMemcpyAsync(&h->nactive, h->total_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  *nactive = h->nactive;
  return 0;
}

extern "C" int nlps_gpu_explicit_nodal(nlps_gpu* h, double* mass, double* dU, double* force, double* accel,
                                       double* reaction) {
  int ND = h->nd;
  if (materialise_nodal(h)) return 1;
  if (compute_node_mask(h)) return 1;
  HIPCHK(hipMemcpyAsync(&h->nactive, h->total_d, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (mass && from_grid(h, mass, h->N.nm, ND, 1 + ND, 0, 1, 0, nullptr)) return 1;
  if (dU && from_grid(h, dU, h->N.dU, ND, ND, 0, 0, 0, nullptr)) return 1;
  if (force && from_grid(h, force, h->N.force, ND, ND, 0, 0, 0, nullptr)) return 1;
  if (accel && from_grid(h, accel, h->N.accel, ND, ND, 0, 0, 0, nullptr)) return 1;
  if (reaction && from_grid(h, reaction, h->N.reaction, ND, ND, 0, 0, 0, nullptr)) return 1;
  return check_status(h, ST_NEWTON | ST_CONNECT | ST_JACOBIAN | ST_CONSTITUTIVE | ST_HALO, "nlps_gpu_explicit_step()");
}

// ------------------------------------------------------------------------------------------------
// a21: the per-dof vector updates of the implicit driver (masked numbering, N_A*d doubles)
// ------------------------------------------------------------------------------------------------
__global__ void k_vec_initial_guess(int n, double* __restrict__ dU, const double* __restrict__ v, const double* __restrict__ a,
                                    double dt, int trial) {  // __form_initial_guess, U-Newmark-beta.c:893-901
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && trial) dU[i] = dt * v[i] + 0.5 * dsqr(dt) * a[i];
}
__global__ void k_vec_guess_bc(const int* __restrict__ nodes, int n, int nd, int dim, int dirbits, double v0, double v1,
                               double v2, const int* __restrict__ n2m, double* __restrict__ dU) {  // :909-950
  int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int m = n2m[nodes[q]];
  if (m == -1) return;
  const double v[3] = {v0, v1, v2};
  for (int k = 0; k < nd; k++)
    if (k < dim && ((dirbits >> k) & 1)) dU[m * nd + k] = v[k];
}
__global__ void k_vec_increments(int n, double* __restrict__ dV, double* __restrict__ dA, const double* __restrict__ dU,
                                 const double* __restrict__ v, const double* __restrict__ a, double a1, double a2,
                                 double a3, double a4, double a5, double a6) {  // :1834-1906
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (dV) dV[i] = a4 * dU[i] + (a5 - 1) * v[i] + a6 * a[i];
  if (dA) dA[i] = a1 * dU[i] - a2 * v[i] - (a3 + 1) * a[i];
}
__global__ void k_vec_inertial(int n, int nd, double* __restrict__ R, const double* __restrict__ M,
                               const double* __restrict__ dU, const double* __restrict__ v, const double* __restrict__ a,
                               const int* __restrict__ d2m, double a1, double a2, double a3, double b0, double b1,
                               double b2) {  // __nodal_inertial_forces, :1519-1557
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || d2m[i] == -1) return;
  const double b[3] = {b0, b1, b2};
  R[i] += M[i] * (a1 * dU[i] - a2 * v[i] - a3 * a[i] - b[i % nd]);
}

// host or device vectors of the caller: inputs are staged into a scratch pool when they live on the host,
// outputs are written in place (device) or copied back (host)
struct VecIO {
  nlps_gpu* h;
  size_t n;
  int slot = 0;
  std::vector<std::pair<double*, double*>> back;  // (host destination, device source)
  const double* in(const double* p) {
    if (!p || is_device_ptr(p)) return p;
    double* d = h->vec_d + (size_t)(slot++) * n;
    (void)hipMemcpyAsync(d, p, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    return d;
  }
  double* out(double* p, bool load) {
    if (!p || is_device_ptr(p)) return p;
    double* d = h->vec_d + (size_t)(slot++) * n;
    if (load) (void)hipMemcpyAsync(d, p, n * sizeof(double), hipMemcpyHostToDevice, h->stream);
    back.push_back({p, d});
    return d;
  }
  int finish() {
    for (auto& b : back)
      if (hipMemcpyAsync(b.first, b.second, n * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 1;
    if (hipGetLastError() != hipSuccess) return 1;
    if (!back.empty() && hipStreamSynchronize(h->stream) != hipSuccess) return 1;
    return 0;
  }
};
static int vec_begin(nlps_gpu* h, const char* who, VecIO& io) {
  if (need_masks(h, who)) return 1;
  io.h = h;
  io.n = (size_t)h->nactive * h->nd;
  if (io.n * 6 > h->vec_cap) {
    if (h->vec_d) HIPCHK(hipFree(h->vec_d));
    h->vec_cap = io.n * 6;
    HIPCHK(hipMalloc((void**)&h->vec_d, h->vec_cap * sizeof(double)));
  }
  return 0;
}

extern "C" int nlps_gpu_form_initial_guess(nlps_gpu* h, double* dU, const double* Un_dt, const double* Un_dt2, double dt,
                                           int use_explicit_trial, const nlps_bcc* bcc, int nbcc, int step) {
  VecIO io;
  if (vec_begin(h, "nlps_gpu_form_initial_guess", io)) return 1;
  if (nbcc > 0 && check_step(h, step, "nlps_gpu_form_initial_guess")) return 1;
  if (ensure_bcs(h, bcc, nbcc)) return 1;
  const int n = (int)io.n, ND = h->nd;
  double* d = io.out(dU, true);
  const double *v = io.in(Un_dt), *a = io.in(Un_dt2);
  if (n > 0) hipLaunchKernelGGL(k_vec_initial_guess, dim3(nblk(n)), dim3(BLK), 0, h->stream, n, d, v, a, dt, use_explicit_trial);
  for (int i = 0; i < nbcc; i++) {
    if (h->bcs[i].n == 0) continue;
    double val[3] = {0, 0, 0};
    for (int k = 0; k < bcc[i].dim && k < 3; k++) val[k] = bcc[i].value[(size_t)k * h->nsteps + step];
    hipLaunchKernelGGL(k_vec_guess_bc, dim3(nblk(h->bcs[i].n)), dim3(BLK), 0, h->stream, h->bcs[i].dnodes, h->bcs[i].n, ND,
                       bcc[i].dim, dirbits_of(bcc[i], step, h->nsteps), val[0], val[1], val[2], h->n2m_d, d);
  }
  if (io.finish()) {
    h->err = "nlps_gpu_form_initial_guess: HIP error";
    return 1;
  }
  return 0;
}

extern "C" int nlps_gpu_nodal_kinetic_increments(nlps_gpu* h, double* dU_dt, double* dU_dt2, const double* dU,
                                                 const double* Un_dt, const double* Un_dt2, const double* alpha) {
  VecIO io;
  if (vec_begin(h, "nlps_gpu_nodal_kinetic_increments", io)) return 1;
  const int n = (int)io.n;
  double *dV = io.out(dU_dt, false), *dA = io.out(dU_dt2, false);
  const double *u = io.in(dU), *v = io.in(Un_dt), *a = io.in(Un_dt2);
  if (n > 0)
    hipLaunchKernelGGL(k_vec_increments, dim3(nblk(n)), dim3(BLK), 0, h->stream, n, dV, dA, u, v, a, alpha[0], alpha[1], alpha[2],
                       alpha[3], alpha[4], alpha[5]);
  if (io.finish()) {
    h->err = "nlps_gpu_nodal_kinetic_increments: HIP error";
    return 1;
  }
  return 0;
}

extern "C" int nlps_gpu_nodal_inertial_forces(nlps_gpu* h, double* R, const double* M, const double* dU,
                                              const double* Un_dt, const double* Un_dt2, const double* alpha,
                                              const double* gravity) {
  VecIO io;
  if (vec_begin(h, "nlps_gpu_nodal_inertial_forces", io)) return 1;
  const int n = (int)io.n, ND = h->nd;
  double* r = io.out(R, true);
  const double *m = io.in(M), *u = io.in(dU), *v = io.in(Un_dt), *a = io.in(Un_dt2);
  double b[3] = {0, 0, 0};
  if (gravity)
    for (int k = 0; k < ND; k++) b[k] = gravity[k];
  if (n > 0)
    hipLaunchKernelGGL(k_vec_inertial, dim3(nblk(n)), dim3(BLK), 0, h->stream, n, ND, r, m, u, v, a, h->d2m_d, alpha[0], alpha[1],
                       alpha[2], b[0], b[1], b[2]);
  if (io.finish()) {
    h->err = "nlps_gpu_nodal_inertial_forces: HIP error";
    return 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// __lagrangian_evaluation (U-Newmark-beta.c:970-1058): the residual of the maintained driver's SNES solve, evaluated at
// every Newton iterate and line-search trial, as one device call
// ------------------------------------------------------------------------------------------------
// Closing nodal kernel: L = f_int (+ traction) + M (alpha_1 dU - alpha_2 v - alpha_3 a - b) on the free dofs, 0 on the
// Dirichlet ones (VecZeroEntries :1003, then the three += of :1028-1036 in that order: __nodal_internal_forces skips the
// Dirichlet dofs at :1359, __nodal_inertial_forces at :1550, k_traction's sums are taken for the free dofs only like
// :1489).  force / trac: grid numbering [nnodes][ND]; everything else masked.
template <int ND>
__global__ void k_lagrangian_nodal(int nnodes, const int* __restrict__ n2m, const int* __restrict__ d2m,
                                   const double* __restrict__ force, const double* __restrict__ trac, double* __restrict__ R,
                                   const double* __restrict__ M, const double* __restrict__ dU, const double* __restrict__ v,
                                   const double* __restrict__ a, double a1, double a2, double a3, double b0, double b1,
                                   double b2, const int* __restrict__ gstatus, int* __restrict__ status_out) {
  const int A = blockIdx.x * blockDim.x + threadIdx.x;
  // the failure flags of the stress update, left where the host reads them after its one synchronise (check_status,
  // mirrored): every kernel that can set them has finished before this one starts
  if (A == 0 && status_out) *status_out = *gstatus;
  if (A >= nnodes) return;
  const int m = n2m[A];
  if (m < 0) return;
  const double b[3] = {b0, b1, b2};
#pragma unroll
  for (int f = 0; f < ND; f++) {
    const size_t i = (size_t)m * ND + f;
    double r = 0.0;
    if (d2m[i] != -1) {
      r += force[(size_t)A * ND + f];
      if (trac) r += trac[(size_t)A * ND + f];
      r += M[i] * (a1 * dU[i] - a2 * v[i] - a3 * a[i] - b[f]);
    }
    R[i] = r;
  }
}

extern "C" int nlps_gpu_lagrangian_evaluation(nlps_gpu* h, double* R, const double* dU, const double* Un_dt,
                                              const double* Un_dt2, const double* M, const double* alpha,
                                              const double* gravity, const nlps_bcc* loads, int nloads, int step,
                                              double thickness, const double* area0, int flags) {
  if (need_masks(h, "nlps_gpu_lagrangian_evaluation")) return 1;
  if (!R || !dU || !Un_dt || !Un_dt2 || !M || !alpha) {
    h->err = "nlps_gpu_lagrangian_evaluation: R, dU, Un_dt, Un_dt2, M and alpha are required";
    return 1;
  }
  const int ND = h->nd;
  const size_t n = (size_t)h->nactive * ND;
  // The composition of the separate stage calls, in the order of :1018-1036 -- on request (NLPS_LAGR_SEPARATE: the form
  // the fused one is measured and tested against), for what the fused kernel does not carry (rate tensors, which only the
  // Newtonian-fluid law reads; the damage hooks, which sit between the stress update and the force scatter and need
  // every particle's stress before any force)
  const bool fused = !(flags & (NLPS_LAGR_SEPARATE | NLPS_LAGR_RATES)) && !h->P.erosion && h->uniform_law <= NLPS_KLAW_FRICTIONAL &&
                     h->P.np > 0;
  if (!fused) {
    double* dV = nullptr;
    if (flags & NLPS_LAGR_RATES) {  // __compute_nodal_velocity_increments, :1018
      HIPCHK(hipMalloc((void**)&dV, std::max<size_t>(n, 1) * sizeof(double)));
      if (nlps_gpu_nodal_kinetic_increments(h, dV, nullptr, dU, Un_dt, Un_dt2, alpha)) {
        (void)hipFree(dV);
        return 1;
      }
    }
    int st = nlps_gpu_compatibility(h, dU, dV);
    if (dV) {
      (void)hipStreamSynchronize(h->stream);
      (void)hipFree(dV);
    }
    if (st) return 1;
    if (nlps_gpu_constitutive(h)) return 1;
    if (is_device_ptr(R)) HIPCHK(hipMemsetAsync(R, 0, n * sizeof(double), h->stream));
    else memset(R, 0, n * sizeof(double));
    if (nlps_gpu_internal_forces(h, R)) return 1;
    if (nloads > 0 && nlps_gpu_nodal_traction_forces(h, R, loads, nloads, step, thickness, area0)) return 1;
    return nlps_gpu_nodal_inertial_forces(h, R, M, dU, Un_dt, Un_dt2, alpha, gravity);
  }
  if (materialise_roll(h)) return 1;
  if (materialise_nodal(h)) return 1;  // (the forces of the last explicit step are about to go)
  h->level_b_fields = true;            // C_ep holds data from here on
  VecIO io;
  if (vec_begin(h, "nlps_gpu_lagrangian_evaluation", io)) return 1;
  if (h->timing) HIPCHK(hipEventRecord(h->ev[0], h->stream));  // slots of nlps_gpu_get_timing: [0] staging, [2] the kernel, [4] nodal + copy back
  double* r = io.out(R, false);
  const double* u = io.in(dU);
  const double* cst[3] = {Un_dt, Un_dt2, M};  // constant over the evaluations of one SNES solve
  if (!is_device_ptr(Un_dt) || !is_device_ptr(Un_dt2) || !is_device_ptr(M)) {
    if (h->lagr_cap < 3 * n) {
      if (h->lagr_d) HIPCHK(hipFree(h->lagr_d));
      h->lagr_cap = 3 * n;
      HIPCHK(hipMalloc((void**)&h->lagr_d, h->lagr_cap * sizeof(double)));
      h->lagr_valid = false;
    }
    if ((flags & NLPS_LAGR_SAME_STEP) && !h->lagr_valid) {
      h->err = "nlps_gpu_lagrangian_evaluation: NLPS_LAGR_SAME_STEP without an earlier evaluation since nlps_gpu_active_masks";
      return 1;
    }
    for (int q = 0; q < 3; q++) {
      if (is_device_ptr(cst[q])) continue;
      double* d = h->lagr_d + (size_t)q * n;
      if (!(flags & NLPS_LAGR_SAME_STEP)) HIPCHK(hipMemcpyAsync(d, cst[q], n * sizeof(double), hipMemcpyHostToDevice, h->stream));
      cst[q] = d;
    }
    h->lagr_valid = true;
  }
  const double *v = cst[0], *a = cst[1], *m = cst[2];
  // the caller's dU in grid numbering (the gather windows read N.dU), the force accumulator of the node window reset
  hipLaunchKernelGGL(k_expand_reset, dim3(nblk(h->g.nnodes)), dim3(BLK), 0, h->stream, h->N.dU, u, h->n2m_d, h->g.nnodes, ND,
                     h->N.force, h->n0, h->nwn);
  if (h->timing) HIPCHK(hipEventRecord(h->ev[2], h->stream));
  {
    TileD td = tile_view(h);
    td.slab = nullptr;  // (level-B semantics: atomics also in deterministic mode, like kb_fint_tile)
    const dim3 grid(h->ntw * K3_SPLIT), blk(K3_BLK);
#define NLPS_K3R(NDv, LAWv)                                                                                          \
  hipLaunchKernelGGL((k3_tile<NDv, LAWv, 3>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, h->gstatus_d, \
                     (const double*)nullptr)
    const int law = h->uniform_law;
    // a cloud of several laws: one launch per law present of the kernel compiled for that law, every workgroup compacting
    // its tile's particles of that law first (FILT, as the explicit step's per-law launches)
#define NLPS_K3RF(NDv, LAWv)                                                                                          \
  hipLaunchKernelGGL((k3_tile<NDv, LAWv, 3, true>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, \
                     h->gstatus_d, (const double*)nullptr)
    if (law < 0) {
      for (int l = 0; l <= NLPS_KLAW_FRICTIONAL; l++) {
        if (!(h->law_present & (1 << l))) continue;
        if (ND == 2) {
          if (l == 0) NLPS_K3RF(2, 0);
          else if (l == 1) NLPS_K3RF(2, 1);
          else if (l == 2) NLPS_K3RF(2, 2);
          else if (l == 3) NLPS_K3RF(2, 3);
          else NLPS_K3RF(2, 4);
        } else {
          if (l == 0) NLPS_K3RF(3, 0);
          else if (l == 1) NLPS_K3RF(3, 1);
          else if (l == 2) NLPS_K3RF(3, 2);
          else if (l == 3) NLPS_K3RF(3, 3);
          else NLPS_K3RF(3, 4);
        }
      }
    } else
#define NLPS_K3RU(LAWv)                                                                                              \
  hipLaunchKernelGGL((k3_tile<3, LAWv, 3, false, K3_BLK, true>), grid, blk, 0, h->stream, h->P, h->g, h->N, td, h->mats_d, h->prm, \
                     h->gstatus_d, (const double*)nullptr)
    if (ND == 3 && h->nmats == 1 && law >= 1 && law <= 3) {  // one material (k3_body, UMAT)
      if (law == 1) NLPS_K3RU(1);
      else if (law == 2) NLPS_K3RU(2);
      else NLPS_K3RU(3);
    } else if (ND == 2) {
      if (law == 0) NLPS_K3R(2, 0);
      else if (law == 1) NLPS_K3R(2, 1);
      else if (law == 2) NLPS_K3R(2, 2);
      else if (law == 3) NLPS_K3R(2, 3);
      else NLPS_K3R(2, 4);
    } else {
      if (law == 0) NLPS_K3R(3, 0);
      else if (law == 1) NLPS_K3R(3, 1);
      else if (law == 2) NLPS_K3R(3, 2);
      else if (law == 3) NLPS_K3R(3, 3);
      else NLPS_K3R(3, 4);
    }
#undef NLPS_K3R
#undef NLPS_K3RU
#undef NLPS_K3RF
  }
  HIPCHK(hipGetLastError());
  if (h->timing) HIPCHK(hipEventRecord(h->ev[3], h->stream));
  if (halo(h, h->N.force, ND, 8, 0)) return 1;
  int any = 0;
  if (nloads > 0 && traction_to_grid(h, "nlps_gpu_lagrangian_evaluation", h->gridA, loads, nloads, step, thickness, area0, &any))
    return 1;
  double b[3] = {0, 0, 0};
  if (gravity)
    for (int k = 0; k < ND; k++) b[k] = gravity[k];
  LAUNCH_ND((k_lagrangian_nodal<2>), (k_lagrangian_nodal<3>), nblk(h->g.nnodes), h->g.nnodes, (const int*)h->n2m_d,
            (const int*)h->d2m_d, (const double*)h->N.force, any ? (const double*)h->gridA : (const double*)nullptr, r, m, u, v,
            a, alpha[0], alpha[1], alpha[2], b[0], b[1], b[2], (const int*)h->gstatus_d, h->status_hd);
  HIPCHK(hipGetLastError());
  if (io.finish()) {
    h->err = "nlps_gpu_lagrangian_evaluation: HIP error";
    return 1;
  }
  if (h->timing) HIPCHK(hipEventRecord(h->ev[6], h->stream));
  if (check_status(h, ST_CONSTITUTIVE, "Stress_integration__Constitutive__()", h->status_hd != nullptr)) return 1;  // (synchronises)
  if (h->timing) {
    for (int q = 0; q < 8; q++) h->ms[q] = 0.f;
    HIPCHK(hipEventElapsedTime(&h->ms[0], h->ev[0], h->ev[2]));
    HIPCHK(hipEventElapsedTime(&h->ms[2], h->ev[2], h->ev[3]));
    HIPCHK(hipEventElapsedTime(&h->ms[4], h->ev[3], h->ev[6]));
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// tangent assembly
// ------------------------------------------------------------------------------------------------
extern "C" int nlps_gpu_tangent_assemble(nlps_gpu* h, long long* nnz) {
  if (need_masks(h, "nlps_gpu_tangent_assemble")) return 1;
  if (materialise_roll(h)) return 1;
  const int ND = h->nd, S = ND == 3 ? TanCfg<3>::S : TanCfg<2>::S;
  const size_t nn = (size_t)h->g.nnodes, nblk_st = nn * S;
  if (!h->kst_d) {
    HIPCHK(hipMalloc((void**)&h->kst_d, nblk_st * ND * ND * sizeof(double)));
    HIPCHK(hipMalloc((void**)&h->ktouched_d, nblk_st));
    HIPCHK(hipMalloc((void**)&h->kcnt_d, (nn + 1) * sizeof(int)));
    HIPCHK(hipMalloc((void**)&h->koffs_d, (nn + 1) * sizeof(int)));
    HIPCHK(hipMalloc((void**)&h->khead_d, h->P.npad * sizeof(int)));
    HIPCHK(hipMalloc((void**)&h->kng_d, sizeof(int)));
    HIPCHK(hipcub::DeviceScan::ExclusiveSum(nullptr, h->kscan_bytes, h->kcnt_d, h->koffs_d, (int)nn + 1, h->stream));
    HIPCHK(hipMalloc(&h->kscan_tmp, h->kscan_bytes + 16));
  }
  HIPCHK(hipMemsetAsync(h->kst_d, 0, nblk_st * ND * ND * sizeof(double), h->stream));
  HIPCHK(hipMemsetAsync(h->ktouched_d, 0, nblk_st, h->stream));
  HIPCHK(hipMemsetAsync(h->kcnt_d, 0, (nn + 1) * sizeof(int), h->stream));
  const int np = h->P.np;
  h->ktan_sym = false;
  if (np > 0 && h->tangent_grouped) {
    // particles grouped by closest node (radix sort of (I0, p) in the re-sort buffers), one workgroup per node
    hipLaunchKernelGGL(k_tangent_keys, dim3(nblk(np)), dim3(BLK), 0, h->stream, h->P, h->skey_d, h->sval_d);
    size_t bytes = h->cub_tmp_bytes;
    HIPCHK(hipcub::DeviceRadixSort::SortPairs(h->cub_tmp, bytes, h->skey_d, h->skey2_d, h->sval_d, h->sval2_d, np, 0, 32,
                                              h->stream));
    HIPCHK(hipMemsetAsync(h->kng_d, 0, sizeof(int), h->stream));
    hipLaunchKernelGGL(k_tangent_groups, dim3(nblk(np)), dim3(BLK), 0, h->stream, np, h->skey2_d, h->khead_d, h->kng_d);
    const int ngrid = std::min(np, h->g.nnodes);  // upper bound of the number of groups; surplus workgroups exit
    // a cloud of Neo-Hookean particles only: the upper half of every row, the rest by symmetry in nlps_gpu_tangent_coo
    h->ktan_sym = h->tangent_symmetric && h->uniform_law == NLPS_MAT_NEO_HOOKEAN;
    if (ND == 2)
      hipLaunchKernelGGL(k_tangent_nh_grouped<2>, dim3(ngrid), dim3(TAN_NT), 0, h->stream, h->P, h->g, h->mats_d, np, h->skey2_d,
                         h->sval2_d, h->khead_d, h->kng_d, h->kst_d, h->ktouched_d, h->gstatus_d, h->ktan_sym ? 1 : 0);
    else
      hipLaunchKernelGGL(k_tangent_nh_grouped<3>, dim3(ngrid), dim3(TAN_NT), 0, h->stream, h->P, h->g, h->mats_d, np, h->skey2_d,
                         h->sval2_d, h->khead_d, h->kng_d, h->kst_d, h->ktouched_d, h->gstatus_d, h->ktan_sym ? 1 : 0);
  } else if (np > 0) {
    h->ktan_sym = false;
    if (ND == 2) hipLaunchKernelGGL(k_tangent_nh<2>, dim3(np), dim3(64), 0, h->stream, h->P, h->g, h->mats_d, h->kst_d, h->ktouched_d, h->gstatus_d);
    else hipLaunchKernelGGL(k_tangent_nh<3>, dim3(np), dim3(64), 0, h->stream, h->P, h->g, h->mats_d, h->kst_d, h->ktouched_d, h->gstatus_d);
  }
  LAUNCH_ND((k_tangent_count<2>), (k_tangent_count<3>), ((int)nn + 3) / 4, (int)nn, h->ktouched_d, h->kcnt_d);  // (one wave per row node)
  HIPCHK(hipGetLastError());
  size_t bytes = h->kscan_bytes;
  HIPCHK(hipcub::DeviceScan::ExclusiveSum(h->kscan_tmp, bytes, h->kcnt_d, h->koffs_d, (int)nn + 1, h->stream));
  int total = 0;
  HIPCHK(hipMemcpyAsync(&total, h->koffs_d + nn, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (check_status(h, ST_NEWTON | ST_CONSTITUTIVE, "nlps_gpu_tangent_assemble() (Neo-Hookean particles only)")) return 1;
  h->knnz_blocks = total;
  if (nnz) *nnz = (long long)total * ND * ND;
  return 0;
}

extern "C" int nlps_gpu_tangent_set_grouped(nlps_gpu* h, int grouped) {
  h->tangent_grouped = grouped != 0;
  return 0;
}

extern "C" int nlps_gpu_tangent_coo(nlps_gpu* h, double alpha_1, const double* lumped_mass, int apply_dirichlet,
                                    int* rows, int* cols, double* vals) {
  if (h->knnz_blocks < 0) {
    h->err = "nlps_gpu_tangent_coo: call nlps_gpu_tangent_assemble() first";
    return 1;
  }
  if (need_masks(h, "nlps_gpu_tangent_coo")) return 1;
  const int ND = h->nd, nn = h->g.nnodes;
  const size_t ne = (size_t)h->knnz_blocks * ND * ND;
  if (ne == 0) return 0;
  const double* mass_d = nullptr;
  if (lumped_mass) {
    if (ensure_masked(h, (size_t)h->nactive * ND)) return 1;
    HIPCHK(hipMemcpyAsync(h->maskedA, lumped_mass, (size_t)h->nactive * ND * sizeof(double), hipMemcpyDefault, h->stream));
    mass_d = h->maskedA;
  }
  if (is_device_ptr(rows) && is_device_ptr(cols) && is_device_ptr(vals)) {  // (MatSetValuesCOO of a GPU matrix type: no staging)
    LAUNCH_ND((k_tangent_emit<2>), (k_tangent_emit<3>), (nn + 3) / 4, nn, h->g, h->ktouched_d, h->kst_d, h->koffs_d, h->n2m_d,
              apply_dirichlet ? h->d2m_d : (const int*)nullptr, alpha_1, mass_d, rows, cols, vals, h->ktan_sym ? 1 : 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
  }
  int *rows_d = nullptr, *cols_d = nullptr;
  double* vals_d = nullptr;
  HIPCHK(hipMalloc((void**)&rows_d, ne * sizeof(int)));
  HIPCHK(hipMalloc((void**)&cols_d, ne * sizeof(int)));
  HIPCHK(hipMalloc((void**)&vals_d, ne * sizeof(double)));
  LAUNCH_ND((k_tangent_emit<2>), (k_tangent_emit<3>), (nn + 3) / 4, nn, h->g, h->ktouched_d, h->kst_d, h->koffs_d, h->n2m_d,
            apply_dirichlet ? h->d2m_d : (const int*)nullptr, alpha_1, mass_d, rows_d, cols_d, vals_d, h->ktan_sym ? 1 : 0);
  int st = 0;
  if (hipGetLastError() != hipSuccess) st = 1;
  if (!st && hipMemcpyAsync(rows, rows_d, ne * sizeof(int), hipMemcpyDefault, h->stream) != hipSuccess) st = 1;
  if (!st && hipMemcpyAsync(cols, cols_d, ne * sizeof(int), hipMemcpyDefault, h->stream) != hipSuccess) st = 1;
  if (!st && hipMemcpyAsync(vals, vals_d, ne * sizeof(double), hipMemcpyDefault, h->stream) != hipSuccess) st = 1;
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(rows_d);
  (void)hipFree(cols_d);
  (void)hipFree(vals_d);
  if (st) h->err = "nlps_gpu_tangent_coo: HIP error while emitting the triplets";
  return st;
}

extern "C" int nlps_gpu_sparsity_pattern(nlps_gpu* h, int* nnz_per_row) {
  if (h->knnz_blocks < 0) {
    h->err = "nlps_gpu_sparsity_pattern: call nlps_gpu_tangent_assemble() first";
    return 1;
  }
  if (need_masks(h, "nlps_gpu_sparsity_pattern")) return 1;
  const int ND = h->nd, nn = h->g.nnodes;
  int* pat_d = nullptr;
  HIPCHK(hipMalloc((void**)&pat_d, (size_t)h->nactive * ND * sizeof(int) + 16));
  LAUNCH_ND((k_tangent_pattern<2>), (k_tangent_pattern<3>), nblk(nn), nn, h->kcnt_d, h->n2m_d, pat_d);
  int st = hipGetLastError() != hipSuccess;
  if (!st && hipMemcpyAsync(nnz_per_row, pat_d, (size_t)h->nactive * ND * sizeof(int), hipMemcpyDefault, h->stream) != hipSuccess) st = 1;
  (void)hipStreamSynchronize(h->stream);
  (void)hipFree(pat_d);
  if (st) h->err = "nlps_gpu_sparsity_pattern: HIP error";
  return st;
}

