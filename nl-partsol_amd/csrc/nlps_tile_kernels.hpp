// nlps_tile_kernels.hpp — tile-binned kernels of the fused explicit step (included by nlps_gpu.hip).
//
// One workgroup = one tile of TB^d closest-nodes (I0).  Every particle whose I0 lies in the tile has
// its whole 5^d LME stencil inside the tile's (TB+4)^d node WINDOW, which is staged in LDS:
//   * P2G: nodal accumulators live in LDS (ds_add_f64), one flush of the window per tile with
//          global_atomic_add_f64, shaped as contiguous (node,field) runs of a window row
//          => ~70x fewer HBM atomics than one per (particle,node,field);
//   * G2P: the nodal fields to gather (dU, acceleration, active flags) are read once per tile with
//          coalesced loads and then served from LDS instead of per-lane L1 gathers.
// Particles are binned to tiles every step by index (order[]), see k_search / k_fill_order; the
// physical SoA order is tile-major from the upload sort, so order[] is close to the identity.
#pragma once

template <int ND>
struct TileCfg;
template <>
struct TileCfg<3> {
  static constexpr int TB = 4, W = 8, NW = 512;
};
template <>
struct TileCfg<2> {
  static constexpr int TB = 16, W = 20, NW = 400;
};

struct TileD {
  int nt[3];
  int ntiles;
  const int* start;
  const int* count;
  const int* order;
};

template <int ND>
__device__ __forceinline__ int tile_of_node(const GridD& g, const int* nt, int I0) {
  constexpr int TB = TileCfg<ND>::TB;
  int i = I0 % g.n[0], j = (I0 / g.n[0]) % g.n[1], k = I0 / (g.n[0] * g.n[1]);
  return (i / TB) + nt[0] * ((j / TB) + nt[1] * (ND == 3 ? k / TB : 0));
}

template <int ND>
__device__ __forceinline__ void tile_origin(const TileD& td, int tile, int* w0) {
  constexpr int TB = TileCfg<ND>::TB;
  int tx = tile % td.nt[0], ty = (tile / td.nt[0]) % td.nt[1], tz = tile / (td.nt[0] * td.nt[1]);
  w0[0] = tx * TB - 2;
  w0[1] = ty * TB - 2;
  w0[2] = (ND == 3) ? tz * TB - 2 : 0;
}

template <int ND>
__device__ __forceinline__ int window_node(const GridD& g, const int* w0, int idx, bool& inside) {
  constexpr int W = TileCfg<ND>::W;
  int li = idx % W, lj = (idx / W) % W, lk = (ND == 3) ? idx / (W * W) : 0;
  int gi = w0[0] + li, gj = w0[1] + lj, gk = (ND == 3) ? w0[2] + lk : 0;
  inside = gi >= 0 && gi < g.n[0] && gj >= 0 && gj < g.n[1] && (ND == 2 || (gk >= 0 && gk < g.n[2]));
  return gi + g.n[0] * (gj + g.n[1] * gk);
}

// window-local index of the stencil member (i,j,k) of a particle whose I0 has local index `base`
template <int ND>
__device__ __forceinline__ int wl(int base, int i, int j, int k) {
  constexpr int W = TileCfg<ND>::W;
  return base + (i - 2) + W * (j - 2) + (ND == 3 ? W * W * (k - 2) : 0);
}

template <int ND>
__device__ __forceinline__ int window_base(const int* ijk, const int* w0) {
  constexpr int W = TileCfg<ND>::W;
  return (ijk[0] - w0[0]) + W * ((ijk[1] - w0[1]) + (ND == 3 ? W * (ijk[2] - w0[2]) : 0));
}

// exclusive scan of the per-tile particle counts (one 1024-thread block)
__global__ void k_tile_scan(const int* __restrict__ count, int* __restrict__ start, int n) {
  __shared__ int sh[1024];
  int chunk = (n + 1023) / 1024;
  int lo = threadIdx.x * chunk, hi = min(n, lo + chunk), c = 0;
  for (int q = lo; q < hi; q++) c += count[q];
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
    start[q] = run;
    run += count[q];
  }
}

__global__ void k_fill_order(int np, const int* __restrict__ tile, const int* __restrict__ rank,
                             const int* __restrict__ start, int* __restrict__ order) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np) return;
  order[start[tile[p]] + rank[p]] = p;
}

// ------------------------------------------------------------------------------------------------
// K2: neighbour mask + beta + Newton + predictor + P2G(mass, m*dD)      (S1b + S2)
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(BLK) void k2_tile(PView P, GridD g, NView N, TileD td, ParamsD prm, double dt,
                                               double gamma_nm, int* __restrict__ gstatus) {
  constexpr int NW = TileCfg<ND>::NW, NF = 1 + ND;
  __shared__ double acc[NF * NW];
  __shared__ unsigned char act[NW];
  const int tile = blockIdx.x;
  const int cnt = td.count[tile];
  if (cnt == 0) return;
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx = threadIdx.x; idx < NW; idx += BLK) {
    bool in;
    int node = window_node<ND>(g, w0, idx, in);
    act[idx] = in ? N.active[node] : 0;
#pragma unroll
    for (int f = 0; f < NF; f++) acc[f * NW + idx] = 0.0;
  }
  __syncthreads();
  const int start = td.start[tile];
  for (int s = threadIdx.x; s < cnt; s += BLK) {
    const int p = td.order[start + s];
    Lme<ND> c;
    double x[ND], lam[ND];
#pragma unroll
    for (int a = 0; a < ND; a++) {
      x[a] = PF(P, F_X + a, p);
      lam[a] = PF(P, F_LAM + a, p);
    }
    const int I0 = P.I0[p];
    c.geom(g, x, I0);
    const int base = window_base<ND>(c.ijk, w0);
    double beta_prev = PF(P, F_BETA, p);
    double Ra = sqrt(prm.neg_log_tol_zero / beta_prev);
    u64 mlo = 0ull, mhi = 0ull;
#pragma unroll
    for (int k = 0; k < Lme<ND>::KN; k++)
#pragma unroll
      for (int j = 0; j < 5; j++)
#pragma unroll
        for (int i = 0; i < 5; i++) {
          double sq = 0.0;
          sq += c.lx[i] * c.lx[i];
          sq += c.ly[j] * c.ly[j];
          if (ND == 3) sq += c.lz[k % Lme<ND>::KN] * c.lz[k % Lme<ND>::KN];
          bool ok = act[wl<ND>(base, i, j, k)] && (sqrt(sq) <= Ra);
          int b = i + 5 * j + 25 * k;
          if (ok) {
            if (b < 64) mlo |= (1ull << b);
            else mhi |= (1ull << (b - 64));
          }
        }
    c.mlo = mlo;
    c.mhi = mhi;
    int nn = __popcll(mlo) + __popcll(mhi);
    if (nn < ND + 1) {
      P.nn[p] = 0;
      P.mlo[p] = 0ull;
      P.mhi[p] = 0ull;
      atomicOr(&P.status[p], ST_CONNECT);
      atomicOr(gstatus, ST_CONNECT);
      continue;
    }
    double hv = N.h_avg[I0];
    double beta = prm.gamma_lme / (hv * hv);
    int st = 0, NumIter = 0;
    double Zinv = 0.0;
    while (NumIter <= prm.max_iter_lme) {
      double r[ND], J[ND * ND], Jm1[ND * ND];
      c.factors(lam, beta);
      lme_moments<ND>(c, Zinv, r, J);
      double aux = 0.0;
#pragma unroll
      for (int a = 0; a < ND; a++) aux += dsqr(r[a]);
      if (sqrt(aux) > prm.tol_wrapper) {
        if (rcond_ref<ND>(J) < 1E-8 || !inverse<ND>(Jm1, J)) {
          st |= ST_NEWTON;
          break;
        }
#pragma unroll
        for (int a = 0; a < ND; a++) {
          double dl = 0.0;
#pragma unroll
          for (int b2 = 0; b2 < ND; b2++) dl += Jm1[a * ND + b2] * r[b2];
          lam[a] -= dl;
        }
        NumIter++;
      } else {
        break;
      }
    }
    if (NumIter >= prm.max_iter_lme) st |= ST_NEWTON;
    P.nn[p] = nn;
    P.mlo[p] = mlo;
    P.mhi[p] = mhi;
    PF(P, F_BETA, p) = beta;
#pragma unroll
    for (int a = 0; a < ND; a++) PF(P, F_LAM + a, p) = lam[a];
    if (st) {
      atomicOr(&P.status[p], st);
      atomicOr(gstatus, st);
    }
    double dd[ND];
#pragma unroll
    for (int a = 0; a < ND; a++) {
      double v = PF(P, F_VEL + a, p), ac = PF(P, F_ACC + a, p);
      dd[a] = dt * v + 0.5 * dsqr(dt) * ac;
      PF(P, F_DDIS + a, p) = dd[a];
      PF(P, F_VEL + a, p) = v + (1 - gamma_nm) * dt * ac;
    }
    double mz = PF(P, F_MASS, p) * Zinv;
    for_each_nb<ND>(c, [&](int, int i, int j, int k, double e) {
      int li = wl<ND>(base, i, j, k);
      double w = mz * e;
      atomicAdd(&acc[li], w);
#pragma unroll
      for (int a = 0; a < ND; a++) atomicAdd(&acc[(1 + a) * NW + li], w * dd[a]);
    });
  }
  __syncthreads();
  for (int q = threadIdx.x; q < NW * NF; q += BLK) {
    int f = q % NF, idx = q / NF;
    double v = acc[f * NW + idx];
    if (v != 0.0) {
      bool in;
      int node = window_node<ND>(g, w0, idx, in);
      if (in) atomic_add_f64(N.nm + (size_t)node * NF + f, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K3: G2P grad(dU) -> DF, F, J, density; stress; P2G of -f_int            (S3 + S4)
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(BLK) void k3_tile(PView P, GridD g, NView N, TileD td, const MatD* __restrict__ mats,
                                               ParamsD prm, int* __restrict__ gstatus) {
  constexpr int NW = TileCfg<ND>::NW;
  __shared__ double du[ND * NW];
  __shared__ double fac[ND * NW];
  const int tile = blockIdx.x;
  const int cnt = td.count[tile];
  if (cnt == 0) return;
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx = threadIdx.x; idx < NW; idx += BLK) {
    bool in;
    int node = window_node<ND>(g, w0, idx, in);
#pragma unroll
    for (int a = 0; a < ND; a++) {
      du[a * NW + idx] = in ? N.dU[(size_t)node * ND + a] : 0.0;
      fac[a * NW + idx] = 0.0;
    }
  }
  __syncthreads();
  const int start = td.start[tile];
  for (int s = threadIdx.x; s < cnt; s += BLK) {
    const int p = td.order[start + s];
    Lme<ND> c;
    double lam[ND], beta;
    if (!load_lme<ND>(P, g, p, c, lam, beta)) continue;
    const int base = window_base<ND>(c.ijk, w0);
    double Z = 0.0, sm[ND], q[ND * ND], G[ND * ND];
#pragma unroll
    for (int a = 0; a < ND; a++) sm[a] = 0.0;
#pragma unroll
    for (int a = 0; a < ND * ND; a++) {
      q[a] = 0.0;
      G[a] = 0.0;
    }
    for_each_nb<ND>(c, [&](int, int i, int j, int k, double e) {
      int li = wl<ND>(base, i, j, k);
      double l[3] = {c.lx[i], c.ly[j], ND == 3 ? c.lz[k % Lme<ND>::KN] : 0.0};
      double u[ND];
#pragma unroll
      for (int a = 0; a < ND; a++) u[a] = du[a * NW + li];
      Z += e;
#pragma unroll
      for (int a = 0; a < ND; a++) {
        double el = e * l[a];
        sm[a] += el;
#pragma unroll
        for (int b2 = a; b2 < ND; b2++) q[a * ND + b2] += el * l[b2];
#pragma unroll
        for (int b2 = 0; b2 < ND; b2++) G[b2 * ND + a] += el * u[b2];
      }
    });
    double Zinv = 1.0 / Z, r[ND], J[ND * ND], Jm1[ND * ND];
#pragma unroll
    for (int a = 0; a < ND; a++) r[a] = sm[a] * Zinv;
#pragma unroll
    for (int a = 0; a < ND; a++)
#pragma unroll
      for (int b2 = a; b2 < ND; b2++) {
        double v = q[a * ND + b2] * Zinv - r[a] * r[b2];
        J[a * ND + b2] = v;
        J[b2 * ND + a] = v;
      }
    int st = 0;
    if (!inverse<ND>(Jm1, J)) st |= ST_NEWTON;
    double DF[ND * ND], Fn[ND * ND], Fn1[ND * ND], fzz;
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < ND; m++) v += (G[i * ND + m] * Zinv) * Jm1[j * ND + m];
        DF[i * ND + j] = ((i == j) ? 1.0 : 0.0) - v;
      }
    load_block<ND>(P, F_FN, p, Fn, fzz);
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) {
        double a2 = 0.0;
#pragma unroll
        for (int k2 = 0; k2 < ND; k2++) a2 += DF[i * ND + k2] * Fn[k2 * ND + j];
        Fn1[i * ND + j] = a2;
      }
    double Jn1 = det<ND>(Fn1);
    if (Jn1 <= 0.0) st |= ST_JACOBIAN;
    store_block<ND>(P, F_DF, p, DF, 0.0, false);
    store_block<ND>(P, F_FN1, p, Fn1, 0.0, false);
    PF(P, F_JN1, p) = Jn1;
    PF(P, F_RHO, p) = PF(P, F_RHO, p) / det<ND>(DF);
    double tau[ND * ND], B[ND * ND];
    st |= stress_update<ND>(P, p, mats, prm, Fn1, DF, Jn1, tau);
    if (force_operator<ND>(B, tau, DF, Jm1, PF(P, F_VOL0, p), -1.0)) {
      for_each_nb<ND>(c, [&](int, int i, int j, int k, double e) {
        int li = wl<ND>(base, i, j, k);
        double l[3] = {c.lx[i], c.ly[j], ND == 3 ? c.lz[k % Lme<ND>::KN] : 0.0};
        double pa = e * Zinv;
#pragma unroll
        for (int a = 0; a < ND; a++) {
          double sv = 0.0;
#pragma unroll
          for (int m = 0; m < ND; m++) sv += B[a * ND + m] * l[m];
          atomicAdd(&fac[a * NW + li], pa * sv);
        }
      });
    } else {
      st |= ST_JACOBIAN;
    }
    if (st) {
      atomicOr(&P.status[p], st);
      atomicOr(gstatus, st);
    }
  }
  __syncthreads();
  for (int qq = threadIdx.x; qq < NW * ND; qq += BLK) {
    int f = qq % ND, idx = qq / ND;
    double v = fac[f * NW + idx];
    if (v != 0.0) {
      bool in;
      int node = window_node<ND>(g, w0, idx, in);
      if (in) atomic_add_f64(N.force + (size_t)node * ND + f, v);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K5: G2P of nodal acceleration and dU, corrector, roll                   (S5)
// ------------------------------------------------------------------------------------------------
template <int ND>
__global__ __launch_bounds__(BLK) void k5_tile(PView P, GridD g, NView N, TileD td, double dt, double gamma_nm) {
  constexpr int NW = TileCfg<ND>::NW;
  __shared__ double du[ND * NW];
  __shared__ double ac[ND * NW];
  const int tile = blockIdx.x;
  const int cnt = td.count[tile];
  if (cnt == 0) return;
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx = threadIdx.x; idx < NW; idx += BLK) {
    bool in;
    int node = window_node<ND>(g, w0, idx, in);
#pragma unroll
    for (int a = 0; a < ND; a++) {
      du[a * NW + idx] = in ? N.dU[(size_t)node * ND + a] : 0.0;
      ac[a * NW + idx] = in ? N.accel[(size_t)node * ND + a] : 0.0;
    }
  }
  __syncthreads();
  const int start = td.start[tile];
  for (int s = threadIdx.x; s < cnt; s += BLK) {
    const int p = td.order[start + s];
    Lme<ND> c;
    double lam[ND], beta;
    if (!load_lme<ND>(P, g, p, c, lam, beta)) continue;
    const int base = window_base<ND>(c.ijk, w0);
    double Z = 0.0, sa[ND], su[ND];
#pragma unroll
    for (int a = 0; a < ND; a++) {
      sa[a] = 0.0;
      su[a] = 0.0;
    }
    for_each_nb<ND>(c, [&](int, int i, int j, int k, double e) {
      int li = wl<ND>(base, i, j, k);
      Z += e;
#pragma unroll
      for (int a = 0; a < ND; a++) {
        sa[a] += e * ac[a * NW + li];
        su[a] += e * du[a * NW + li];
      }
    });
    double Zinv = 1.0 / Z;
#pragma unroll
    for (int a = 0; a < ND; a++) {
      double av = sa[a] * Zinv, dd = su[a] * Zinv;
      PF(P, F_ACC + a, p) = av;
      PF(P, F_DDIS + a, p) = dd;
      PF(P, F_VEL + a, p) = PF(P, F_VEL + a, p) + gamma_nm * dt * av;
      PF(P, F_X + a, p) = PF(P, F_X + a, p) + dd;
      PF(P, F_DIS + a, p) = PF(P, F_DIS + a, p) + dd;
    }
    PF(P, F_JN, p) = PF(P, F_JN1, p);
    PF(P, F_KN, p) = PF(P, F_KN1, p);
    PF(P, F_EN, p) = PF(P, F_EN1, p);
    constexpr int T = (ND == 2) ? 5 : 9;
#pragma unroll
    for (int s2 = 0; s2 < T; s2++) {
      PF(P, F_BEN + s2, p) = PF(P, F_BEN1 + s2, p);
      PF(P, F_FN + s2, p) = PF(P, F_FN1 + s2, p);
    }
  }
}
