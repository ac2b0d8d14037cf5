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

// NLPS_DEV = 1 (tools/build_variant.sh): developer builds -- environment switches, phase timers and the ablation macros
// below (which produce WRONG results by design).  The product build (nl-partsol_amd/build.py) never sets it.
#ifndef NLPS_DEV
#define NLPS_DEV 0
#endif
#if !NLPS_DEV && (defined(NLPS_ABL_ATOM) || defined(NLPS_ABL_GATHER) || defined(NLPS_PHASE_TIMING))
#error "NLPS_ABL_* / NLPS_PHASE_TIMING are developer experiments: build with -DNLPS_DEV=1 (tools/build_variant.sh)"
#endif

template <int ND>
struct TileCfg;
// PS = LDS stride between z-planes of the window.  8x8 planes padded to 68 doubles: the 64 possible
// I0 positions of a tile then spread evenly over the 32 b64 LDS banks (with 64 every z maps to the
// same bank: 4-way conflicts on every window read / atomic).
// WA / PSA / NWA: layout of the ACCUMULATOR windows (LDS f64 atomics).  ds_add_f64 serves a wave in four groups of 16
// consecutive lanes over 16 eight-byte banks: a group is conflict-free iff its 16 slots differ mod 16.  In lattice order
// a group is one z-plane of the tile, slots bx + WA by (bx, by in 0..3): with rows of 8 the planes by and by + 2 collide
// (16.1 cycles per wave-instruction per CU measured, tools/lds_atomic_bench.hip), with rows padded to 12 the residues
// are {0-3, 12-15, 8-11, 4-7}: 8.3 cycles.  The read windows keep rows of 8 (ds_read_b128 is fastest there).
template <>
struct TileCfg<3> {
  static constexpr int TB = 4, W = 8, PS = 68, NW = 8 * 68;
  static constexpr int WA = 12, PSA = 100, NWA = 8 * 100;
};
template <>
struct TileCfg<2> {
  static constexpr int TB = 16, W = 20, PS = 400, NW = 400;
  static constexpr int WA = 20, PSA = 400, NWA = 400;  // a 16-lane group is one row of 16 consecutive slots already
};

// A tile's particle list is shared by TILE_SPLIT workgroups (each builds the window, takes every
// TILE_SPLIT-th chunk of 256 particles and flushes with atomics): twice as many, half as long work
// units shorten the partially filled last round of workgroups (tail) without changing the data flow.
// Measured at 1 M particles: pays for K2 (0.42 -> 0.39 ms), costs for K3 (double window load + flush).
#ifndef NLPS_K2_SPLIT
#define NLPS_K2_SPLIT 1  // r02: 1 and 2 are equal at 1 M particles (0.242 / 0.243 ms), 1 is 2 % faster at 8 M and halves the flush
#endif
// workgroup sizes of the tile kernels (K2 keeps BLK: its work list splits tiles at BLK particles).  Measured at 1 M
// particles: K3 with 64 / 128 / 256 / 512 threads 0.402 / 0.327 / 0.312 / 0.338 ms, K5 with 128 / 256 / 512 0.097 / 0.100 / 0.111 ms.
#ifndef NLPS_K3_BLK
#define NLPS_K3_BLK 256
#endif
#ifndef NLPS_K5_BLK
#define NLPS_K5_BLK 256
#endif
static constexpr int K3_BLK = NLPS_K3_BLK, K5_BLK = NLPS_K5_BLK;
#ifndef NLPS_K3_SPLIT
#define NLPS_K3_SPLIT 1
#endif
#ifndef NLPS_K5_SPLIT
#define NLPS_K5_SPLIT 1
#endif
#ifndef NLPS_K5_PREFETCH
#define NLPS_K5_PREFETCH 1  // corrector operands requested before the gather loop: K5 0.097 -> 0.092 ms (1 M particles)
#endif
static constexpr int K2_SPLIT = NLPS_K2_SPLIT, K3_SPLIT = NLPS_K3_SPLIT, K5_SPLIT = NLPS_K5_SPLIT;

struct TileD {
  int nt[3];
  int ntiles;
  int tile0;  // first tile of the launched range (node window)
  int ntw;    // tiles in that range
  // Deterministic mode (nlps_gpu_set_deterministic), nullptr otherwise: the P2G results leave a workgroup as ONE plain,
  // coalesced copy of its LDS window into the slab of its (tile, part), slab[(tile * SPLIT + part)][field][window slot];
  // k_slab_gather then sums, for every node, the <= 2^d windows that hold it in a fixed order.  No global atomics and
  // an inter-tile summation order that never changes -- measured 7 % slower per step than the atomic flush (the
  // no-return atomics hide behind the other workgroups' arithmetic, the gather is two more passes over the grid).
  double* slab;
  int slab_n, slab_slot;  // slabs per tile (one per law launched on the tile) and the slot this launch writes
  const int* start;
  const int* count;
  const int* order;   // tile lists: canonical (layer, closest node) order when per-tile ordering is on (K2, K3)
  const int* order_m; // tile lists as binned: runs of memory-consecutive particles (K5 and the level-B gathers)
  // compacted work lists (tile_scan_block): work[S-1][b] = (tile, part) for the b-th workgroup of a kernel that
  // splits a tile's particles over S workgroups, only for non-empty (tile, part) pairs; nwork[S-1] entries.
  // Consecutive workgroups go to different XCDs, so a compacted list spreads the populated tiles evenly over
  // the 8 XCDs whatever the shape of the cloud (tile-index order left XCDs 23 % apart for the cube).
  const int2* work[2];
  // Workgroup range of the launch inside work[S-1]: {begin, end} at range[2*(S-1)].  The lists hold the tiles whose
  // window touches a ghost band (nodes shared with a neighbouring rank) first, so a launch can take all tiles,
  // only the "boundary" ones or only the "interior" ones (overlap of the halo exchange with interior work).
  const int* range;
  unsigned long long* phase;  // -DNLPS_PHASE_TIMING=1 only: per-phase wave-cycle sums (developer profiling)
  // Single-launch overlap of the halo exchange (multi-GPU): the work list holds the boundary tiles first; every boundary
  // workgroup counts itself on sig_cnt after its flush, the last one publishes sig_seq on sig_flag, on which the
  // library's exchange stream waits (k_wait_flag): the exchange of the shared layers then runs beside the interior
  // tiles of the SAME launch.  nullptr = off.
  unsigned* sig_cnt;
  unsigned* sig_flag;
  unsigned sig_seq;
};

// see TileD::sig_flag.  nb = number of boundary workgroups of this launch (device-side, from the tile scan.s ranges);
// called by all threads of a workgroup after its global flush
__device__ __forceinline__ void tile_signal(const TileD& td, int wb, int nb) {
  if (!td.sig_flag || wb >= nb) return;
  // producer side of the hand-off (MI355X_MICROARCH.md, "Valid forms"): every wave drains its own flush atomics, the
  // workgroup meets, ONE lane counts.  What is handed over -- the nodal sums of the shared layers -- was written by
  // agent-scope atomics only, which leave no line behind in the XCD's L2 (same table: "atomic DROP it"), so a workgroup
  // has nothing to write back: the agent-scope release every boundary workgroup used to make here (buffer_wbl2, ~900
  // workgroups per stage) cost 30 us per step (0.664 -> 0.634 ms in the one-GPU rehearsal of overlap mode 2).  The last
  // workgroup still publishes with a release, and the consumers are kernels launched after k_wait_flag has seen the
  // flag: a launch starts with an agent acquire.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned old = atomicAdd(td.sig_cnt, 1u);
    if (old == (unsigned)nb - 1u) {
      atomicExch(td.sig_cnt, 0u);  // ready for the next launch
      __hip_atomic_store(td.sig_flag, td.sig_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}
// a launch without boundary tiles still has to release the waiting stream
__device__ __forceinline__ void tile_signal_empty(const TileD& td, int nb) {
  if (td.sig_flag && nb == 0 && blockIdx.x == 0 && threadIdx.x == 0)
    __hip_atomic_store(td.sig_flag, td.sig_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// Work item of this workgroup: tile, part and the number of parts its tile list is dealt into
struct TileWork {
  int tile, part, nparts, wb;
};
template <int SPLIT>
__device__ __forceinline__ bool tile_work_item(const TileD& td, TileWork& w) {
  w.wb = td.range[2 * (SPLIT - 1)] + (int)blockIdx.x;
  if (w.wb >= td.range[2 * (SPLIT - 1) + 1]) return false;
  const int2 wk = td.work[SPLIT - 1][w.wb];
  w.tile = wk.x;
  w.part = wk.y;
  w.nparts = SPLIT;
  return true;
}

// Developer profiling: wall-clock cycles per kernel phase, summed per wave (slot spread over 1024 rows to keep
// the atomics off one address).  Compiled out by default.
#ifndef NLPS_PHASE_TIMING
#define NLPS_PHASE_TIMING 0
#endif
#if NLPS_PHASE_TIMING
#define PH_INIT long long ph_t0 = clock64();
#define PH_COUNT(k, n) atomicAdd(&td.phase[(k) + 16 * (blockIdx.x & 1023)], (unsigned long long)(n));
#define PH(k)                                                                                               \
  {                                                                                                         \
    long long ph_t1 = clock64();                                                                            \
    if ((threadIdx.x & 63) == 0)                                                                            \
      atomicAdd(&td.phase[(k) + 16 * (blockIdx.x & 1023)], (unsigned long long)(ph_t1 - ph_t0));           \
    ph_t0 = ph_t1;                                                                                          \
  }
#else
#define PH_INIT
#define PH(k)
#define PH_COUNT(k, n)
#endif

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

// global node of window slot idx (inside = false for slots outside the grid and for padding slots);
// row = index of the slot's x-row
template <int ND>
__device__ __forceinline__ int window_node(const GridD& g, const int* w0, int idx, bool& inside, int* row = nullptr,
                                           int* col = nullptr) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS;
  const int lk = (ND == 3) ? idx / PS : 0;
  const int rem = (ND == 3) ? idx % PS : idx;
  const int li = rem % W, lj = rem / W;
  const int gi = w0[0] + li, gj = w0[1] + lj, gk = (ND == 3) ? w0[2] + lk : 0;
  inside = (lj < W) && gi >= 0 && gi < g.n[0] && gj >= 0 && gj < g.n[1] && (ND == 2 || (gk >= 0 && gk < g.n[2]));
  if (row) *row = lj + (ND == 3 ? W * lk : 0);
  if (col) *col = li;
  return gi + g.n[0] * (gj + g.n[1] * gk);
}

// the same for a slot of an accumulator window (WA / PSA layout)
template <int ND>
__device__ __forceinline__ int window_node_a(const GridD& g, const int* w0, int idx, bool& inside) {
  constexpr int W = TileCfg<ND>::W, WA = TileCfg<ND>::WA, PSA = TileCfg<ND>::PSA;
  const int lk = (ND == 3) ? idx / PSA : 0;
  const int rem = (ND == 3) ? idx % PSA : idx;
  const int li = rem % WA, lj = rem / WA;
  const int gi = w0[0] + li, gj = w0[1] + lj, gk = (ND == 3) ? w0[2] + lk : 0;
  inside = (li < W) && (lj < W) && gi >= 0 && gi < g.n[0] && gj >= 0 && gj < g.n[1] && (ND == 2 || (gk >= 0 && gk < g.n[2]));
  return gi + g.n[0] * (gj + g.n[1] * gk);
}
template <int ND>
__device__ __forceinline__ int window_base_a(const int* ijk, const int* w0) {
  constexpr int WA = TileCfg<ND>::WA, PSA = TileCfg<ND>::PSA;
  return (ijk[0] - w0[0]) + WA * (ijk[1] - w0[1]) + (ND == 3 ? PSA * (ijk[2] - w0[2]) : 0);
}

// window-local index of the stencil member (i,j,k) of a particle whose I0 has local index `base`
template <int ND>
__device__ __forceinline__ int wl(int base, int i, int j, int k) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS;
  return base + (i - 2) + W * (j - 2) + (ND == 3 ? PS * (k - 2) : 0);
}

template <int ND>
__device__ __forceinline__ int window_base(const int* ijk, const int* w0) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS;
  return (ijk[0] - w0[0]) + W * (ijk[1] - w0[1]) + (ND == 3 ? PS * (ijk[2] - w0[2]) : 0);
}

// block-wide exclusive scan of one int per thread (1024 threads = 16 waves); returns the exclusive prefix,
// *total = sum.  Wave-level shuffles + one pass over the 16 wave totals: two barriers instead of twenty.
__device__ __forceinline__ int block_scan_1024(int v, int* sh, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();  // sh[] may still be read from the previous scan
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const int t = sh[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + incl - v;
}

// the same for four 16-bit counters packed into one 64-bit word (sums below 65536 each)
__device__ __forceinline__ unsigned long long block_scan_1024_u64(unsigned long long v, unsigned long long* sh,
                                                                  unsigned long long* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == 63) sh[wave] = incl;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) {
    const unsigned long long t = sh[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  return base + incl - v;
}

// exclusive scan of the per-tile particle counts + the compacted work lists (one 1024-thread block).
// count/start are already offset to the first tile of the node window; tile0 = that tile's index.
// A tile is "boundary" when its node window reaches a ghost band: slow-axis layers <= band_lo or >= band_hi
// (tpl = tiles per slow-axis tile layer, TB = tile edge).  ranges[cls][S-1] = {begin, end} in work<S> for
// cls 0 = all, 1 = boundary, 2 = interior.
struct TileScanArgs {
  const int* count;
  int* start;
  int n, tile0, tpl, TB, band_lo, band_hi;
  int2 *work1, *work2;
  int* ranges;
  int* cursor;  // [n] or nullptr: a copy of start[] for k_fill_orders to hand out list positions from (deferred ranks)
};
__device__ __forceinline__ void tile_scan_block(const TileScanArgs& a) {
  const int* __restrict__ count = a.count;
  int* __restrict__ start = a.start;
  const int n = a.n, tile0 = a.tile0, tpl = a.tpl, TB = a.TB, band_lo = a.band_lo, band_hi = a.band_hi;
  int2* __restrict__ work1 = a.work1;
  int2* __restrict__ work2 = a.work2;
  int* __restrict__ ranges = a.ranges;
  __shared__ int sh[1024];
  int chunk = (n + 1023) / 1024;
  int lo = threadIdx.x * chunk, hi = min(n, lo + chunk), c = 0, b1 = 0, b2 = 0, i1 = 0, i2 = 0;
  for (int q = lo; q < hi; q++) {
    const int cq = count[q];
    c += cq;
    const int tz = (tile0 + q) / tpl;
    const bool bnd = (tz * TB - 2 <= band_lo) || (tz * TB + TB + 1 >= band_hi);
    const int e1 = cq > 0, e2 = (cq > 0) + (cq > BLK);
    if (bnd) {
      b1 += e1;
      b2 += e2;
    } else {
      i1 += e1;
      i2 += e2;
    }
  }
  int nb1, nb2, ni1, ni2, tot;
  int run = block_scan_1024(c, sh, &tot);
  __shared__ unsigned long long sh64[16];
  int rb1, rb2, ri1, ri2;
  if (n < 32768) {  // the four list counters as 16-bit fields of one scan (each sum is below 2 n)
    unsigned long long wt;
    const unsigned long long wp = block_scan_1024_u64((unsigned long long)b1 | ((unsigned long long)b2 << 16) |
                                                          ((unsigned long long)i1 << 32) | ((unsigned long long)i2 << 48),
                                                      sh64, &wt);
    rb1 = (int)(wp & 0xFFFFull), rb2 = (int)((wp >> 16) & 0xFFFFull), ri1 = (int)((wp >> 32) & 0xFFFFull), ri2 = (int)(wp >> 48);
    nb1 = (int)(wt & 0xFFFFull), nb2 = (int)((wt >> 16) & 0xFFFFull), ni1 = (int)((wt >> 32) & 0xFFFFull), ni2 = (int)(wt >> 48);
  } else {
    rb1 = block_scan_1024(b1, sh, &nb1);
    rb2 = block_scan_1024(b2, sh, &nb2);
    ri1 = block_scan_1024(i1, sh, &ni1);
    ri2 = block_scan_1024(i2, sh, &ni2);
  }
  ri1 += nb1;
  ri2 += nb2;
  if (threadIdx.x == 0) {
    const int r[12] = {0, nb1 + ni1, 0, nb2 + ni2, 0, nb1, 0, nb2, nb1, nb1 + ni1, nb2, nb2 + ni2};
    for (int k = 0; k < 12; k++) ranges[k] = r[k];
  }
  for (int q = lo; q < hi; q++) {
    const int cq = count[q];
    start[q] = run;
    if (a.cursor) a.cursor[q] = run;
    run += cq;
    if (cq > 0) {
      const int tz = (tile0 + q) / tpl;
      const bool bnd = (tz * TB - 2 <= band_lo) || (tz * TB + TB + 1 >= band_hi);
      int& r1 = bnd ? rb1 : ri1;
      int& r2 = bnd ? rb2 : ri2;
      work1[r1++] = make_int2(tile0 + q, 0);
      work2[r2++] = make_int2(tile0 + q, 0);
      if (cq > BLK) work2[r2++] = make_int2(tile0 + q, 1);
    }
  }
}

__global__ void k_fill_order(int np, const int* __restrict__ tile, const int* __restrict__ rank,
                             const int* __restrict__ start, int* __restrict__ order) {
  int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= np || tile[p] < 0) return;
  order[start[tile[p]] + rank[p]] = p;
}

// Canonical order of a tile's particle list, every step: layer r holds the r-th particle of every closest node that has
// one, nodes in lattice order (x fastest).  64 consecutive entries (one wave) are then 64 DISTINCT closest nodes in
// the order for which the window layouts are bank-conflict free (TileCfg, k3_tile), whatever the arrival order of the
// binning was: the window atomics of the scatters and the window reads of the gathers ran with 25 % (K2), 31 % (K3) and
// 49 % (K5) of their LDS cycles lost to bank conflicts on the lists as binned (profiles/r01_sq_counters.md).
// One workgroup per non-empty tile, counting sort in LDS: rank inside the node by an LDS atomic (the order among the
// particles of one node is the arrival order), position = layer offset + number of earlier nodes that reach the layer.
// EXACT (deterministic mode): rank inside the node = number of the node's particles with a smaller slot index, not
// the arrival order of the LDS atomics: the list is then a function of the particle arrays alone.
// order_in: the tile lists as binned (runs of memory-consecutive particles: what the memory-bound K5 wants);
// order_out: the same lists in canonical order (what the LDS-atomic-bound K2 and K3 want); may alias order_in.
template <int ND, bool EXACT = false>
__global__ __launch_bounds__(256) void k_tile_order(PView P, GridD g, TileD td, const int* order_in, int* order) {
  constexpr int TB = TileCfg<ND>::TB, NN = (ND == 3) ? TB * TB * TB : TB * TB;
  constexpr int CAP = EXACT ? 4096 : 1536, LMAX = 32;  // larger tiles / deeper nodes: order of the binning (EXACT: by slot index)
  __shared__ int cnt[NN];
  __shared__ unsigned short tbl[LMAX][NN];
  __shared__ int lsize[LMAX + 1];
  __shared__ int keys[CAP], pp[CAP];
  __shared__ int maxc;
  const int wb = td.range[0] + (int)blockIdx.x;
  if (wb >= td.range[1]) return;
  const int tile = td.work[0][wb].x;
  const int n = td.count[tile];
  const int start = td.start[tile];
  if (n > CAP || n <= 1) {
    if (EXACT && n > 1) {
      // deterministic mode, tile too large for the LDS sort: ascending slot index, ranks counted straight from the
      // binned list (O(n^2) reads of an L1-resident list: slow, but the list is a function of the particle arrays alone)
      for (int s = threadIdx.x; s < n; s += 256) {
        const int p = order_in[start + s];
        int r = 0;
        for (int q = 0; q < n; q++) r += (order_in[start + q] < p) ? 1 : 0;
        order[start + r] = p;
      }
      return;
    }
    if (order_in != order)  // keeps the order of the binning
      for (int s = threadIdx.x; s < n; s += 256) order[start + s] = order_in[start + s];
    return;
  }
  for (int q = threadIdx.x; q < NN; q += 256) cnt[q] = 0;
  if (threadIdx.x == 0) maxc = 0;
  __syncthreads();
  int w0[3];
  tile_origin<ND>(td, tile, w0);  // window origin = tile origin - 2
  for (int s = threadIdx.x; s < n; s += 256) {
    const int p = order_in[start + s];
    const int I0 = P.I0[p];
    const int bx = I0 % g.n[0] - (w0[0] + 2), by = (I0 / g.n[0]) % g.n[1] - (w0[1] + 2);
    const int bz = (ND == 3) ? I0 / (g.n[0] * g.n[1]) - (w0[2] + 2) : 0;
    const int node = bx + TB * (by + TB * bz);
    const int r = atomicAdd(&cnt[node], 1);
    keys[s] = node | (r << 16);
    pp[s] = p;
  }
  __syncthreads();
  if (EXACT) {
    for (int s = threadIdx.x; s < n; s += 256) {
      const int node = keys[s] & 0xFFFF, p = pp[s];
      int r = 0;
      for (int q = 0; q < n; q++) r += ((keys[q] & 0xFFFF) == node && pp[q] < p) ? 1 : 0;  // LDS broadcast reads
      keys[s] = node | (r << 16);  // only this thread's own entries change: the node bits the others read stay
    }
    __syncthreads();
  }
  for (int q = threadIdx.x; q < NN; q += 256) atomicMax(&maxc, cnt[q]);
  __syncthreads();
  const int nl = maxc;
  if (nl > LMAX) {  // uniform: every thread reads the same maxc
    if (EXACT) {  // a closest node deeper than the layer table: ascending slot index (deterministic, see above)
      for (int s = threadIdx.x; s < n; s += 256) {
        const int p = pp[s];
        int r = 0;
        for (int q = 0; q < n; q++) r += (pp[q] < p) ? 1 : 0;  // LDS broadcast reads
        order[start + r] = p;
      }
      return;
    }
    if (order_in != order)
      for (int s = threadIdx.x; s < n; s += 256) order[start + s] = pp[s];
    return;
  }
  // tbl[r][node] = number of earlier nodes that reach layer r; lsize[r] = nodes in layer r
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < nl; r += 4) {
    int run = 0;
    for (int q0 = 0; q0 < NN; q0 += 64) {
      const bool f = cnt[q0 + lane] > r;
      const unsigned long long m = __ballot(f);
      tbl[r][q0 + lane] = (unsigned short)(run + (int)__popcll(m & ((1ull << lane) - 1ull)));
      run += (int)__popcll(m);
    }
    if (lane == 0) lsize[r] = run;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int run = 0;
    for (int r = 0; r < nl; r++) {
      const int v = lsize[r];
      lsize[r] = run;
      run += v;
    }
  }
  __syncthreads();
  for (int s = threadIdx.x; s < n; s += 256) {
    const int node = keys[s] & 0xFFFF, r = keys[s] >> 16;
    order[start + lsize[r] + tbl[r][node]] = pp[s];
  }
}

// Developer ablations (tools/kbench.py, never in the product build): NLPS_ABL_ATOM = 1 keeps the arithmetic of the window
// scatters but issues no LDS atomic (the test value never occurs); NLPS_ABL_GATHER = 1 replaces the window reads of the
// gathers by a register constant.  They bound what the LDS work of a kernel costs beside its arithmetic.
#ifndef NLPS_K3_TWOPASS_ALL
#define NLPS_K3_TWOPASS_ALL 0
#endif
#ifndef NLPS_K3_PRELOAD_FN
#define NLPS_K3_PRELOAD_FN 0  // measured: 40 B of scratch appear, K3 0.213 -> 0.217 ms
#endif
#ifndef NLPS_K3_RELOAD
#define NLPS_K3_RELOAD 1
#endif
#ifndef NLPS_ABL_ATOM
#define NLPS_ABL_ATOM 0
#endif
#ifndef NLPS_ABL_GATHER
#define NLPS_ABL_GATHER 0
#endif
__device__ __forceinline__ void lds_add(double* a, double v) {
#if NLPS_ABL_ATOM
  if (v == 1.2345e300)
#endif
    atomicAdd(a, v);
}

// ---- 3-D window helpers without divisions: the 8 x 8 x 8 window is walked as r = lx + 8 (ly + 8 lz) -----------------
#ifndef NLPS_FAST_WINDOWS
#define NLPS_FAST_WINDOWS 1
#endif
// One z value of a gather window as its OWN ds_read_b64.  The back end pairs the reads of two neighbouring slots into
// ds_read2_b64, which the LDS serves in 8 cycles over 32 banks (two ds_read_b64: 4 cycles over 64 banks) and which puts
// the rows by and by + 2 of the tile on the same banks (rows of 8 doubles = 16 of its 32 slots): K5 spent 38 % of its
// LDS cycles on those conflicts.  A volatile access is not paired; with planes of 68 the 32 lanes of a ds_read_b64
// group (bx + 8 by + 68 bz, bz in {0, 1}) fall on 32 different 8-byte slots.
#ifndef NLPS_Z_SINGLE_READS
#define NLPS_Z_SINGLE_READS 1
#endif
__device__ __forceinline__ double lds_z(const double* z, int i) {
#if NLPS_Z_SINGLE_READS
  return *(const volatile __attribute__((address_space(3))) double*)(z + i);  // (z is a __shared__ array: keep the access a ds_read)
#else
  return z[i];
#endif
}

// global node of window cell r (3-D); inside = false outside the grid
__device__ __forceinline__ int window_cell3(const GridD& g, const int* w0, int r, bool& inside) {
  const int gi = w0[0] + (r & 7), gj = w0[1] + ((r >> 3) & 7), gk = w0[2] + (r >> 6);
  inside = gi >= 0 && gi < g.n[0] && gj >= 0 && gj < g.n[1] && gk >= 0 && gk < g.n[2];
  return gi + g.n[0] * (gj + g.n[1] * gk);
}
// active flags of the 3-D window as one bit row per (y,z) line: four threads share a line (two nodes each), the bits meet
// through two lane exchanges -- no LDS atomics, no zeroing pass, byte loads of runs of the node array
template <int NT>
__device__ __forceinline__ void window_actrows3(const GridD& g, const int* w0, const unsigned char* __restrict__ active,
                                                unsigned* actrow) {
  for (int t = threadIdx.x; t < 256; t += NT) {
    const int row = t >> 2, q = t & 3;
    const int gj = w0[1] + (row & 7), gk = w0[2] + (row >> 3), gi = w0[0] + 2 * q;
    const bool rin = gj >= 0 && gj < g.n[1] && gk >= 0 && gk < g.n[2];
    const size_t n0 = (size_t)g.n[0] * ((size_t)gj + (size_t)g.n[1] * (size_t)gk);
    unsigned bits = 0u;
    if (rin && gi >= 0 && gi < g.n[0] && active[n0 + gi]) bits |= 1u << (2 * q);
    if (rin && gi + 1 >= 0 && gi + 1 < g.n[0] && active[n0 + gi + 1]) bits |= 2u << (2 * q);
    bits |= __shfl_xor(bits, 1);
    bits |= __shfl_xor(bits, 2);
    if (q == 0) actrow[row] = bits;
  }
}
// flush of a 3-D accumulator window (WA / PSA layout, NF fields per node contiguous in `out`): consecutive lanes take
// consecutive doubles of `out` (field fastest, then the 8 nodes of a window row), so one wave-instruction of atomics
// covers runs of 8 NF doubles
template <int NF, int NT>
__device__ __forceinline__ void window_flush3(const GridD& g, const int* w0, const double* acc, double* __restrict__ out) {
  constexpr int WA = TileCfg<3>::WA, PSA = TileCfg<3>::PSA, NWA = TileCfg<3>::NWA;
#pragma unroll 2
  for (int e = threadIdx.x; e < 512 * NF; e += NT) {
    const int r = e / NF, f = e - r * NF;
    const double v = acc[f * NWA + (r & 7) + WA * ((r >> 3) & 7) + PSA * (r >> 6)];
    if (v != 0.0) {
      bool in;
      const int node = window_cell3(g, w0, r, in);
      if (in) atomic_add_f64(out + (size_t)node * NF + f, v);
    }
  }
}

// membership bits of row (j,k) into the 125-bit mask
__device__ __forceinline__ void put_row(u64& mlo, u64& mhi, unsigned bits, int s) {
  if (s + 5 <= 64) mlo |= (u64)bits << s;
  else if (s >= 64) mhi |= (u64)bits << (s - 64);
  else {
    mlo |= (u64)bits << s;
    mhi |= (u64)bits >> (64 - s);
  }
}

// 25 membership bits of plane k (run-time k) into the 125-bit mask
__device__ __forceinline__ void put_plane(u64& mlo, u64& mhi, unsigned bits, int k) {
  const int s = 25 * k;
  if (s < 64) {
    mlo |= (u64)bits << s;
    if (s + 25 > 64) mhi |= (u64)bits >> (64 - s);
  } else {
    mhi |= (u64)bits << (s - 64);
  }
}

template <int ND>
struct WinRows {  // active flags of the window as one bit row per (y[,z]) line
  static constexpr int W = TileCfg<ND>::W;
  static constexpr int NROWS = (ND == 3) ? W * W : W;
};

// ------------------------------------------------------------------------------------------------
// The folded explicit step (k3_tile_lazy / k5_tile_lazy): the nodal kernels between the stages (dU = sum m N dD / M
// with the Dirichlet values; a = g + f / M) become part of the window loads of K3 and K5.  What those loads need
// beside the nodal sums: the Dirichlet sets of the step and gravity.
// (Round 3 also held k_step_fused here -- K2, K3, K5 as ONE launch of persistent workgroups with per-tile hand-off
// flags.  Measured slower than the three launches, 0.67 against 0.56 ms; its first revision hung on a barrier that waves
// with an empty exec mask skipped (tools/isa_barriers.py).  Taken out of the library in round 4; last
// revision that holds it: 7527459.)
// ------------------------------------------------------------------------------------------------
struct NodalFold {
  const unsigned* bcmask;  // Dirichlet sets per node (k_bc_mark) or nullptr
  BcStep bc;               // their components and values at this step
  double gv[3];            // gravity
};
// what k_nodal_dU makes of a node (U-Verlet.c:357-362 + :455-527), from the nodal sums as they stand in L2
template <int ND>
__device__ __forceinline__ void fused_nodal_dU(const NView& N, const NodalFold& fs, int A, double* val, bool* fix) {
  // every operand is requested before the first one is looked at (one round of latency instead of three)
  const double M = *(N.nm + (size_t)A * (1 + ND));
  double mom[ND];
#pragma unroll
  for (int a = 0; a < ND; a++) mom[a] = *(N.nm + (size_t)A * (1 + ND) + 1 + a);
  const bool active = N.active[A];
  const unsigned bm0 = fs.bcmask ? fs.bcmask[A] : 0u;
  const bool act = active && M != 0.0;
#pragma unroll
  for (int a = 0; a < ND; a++) {
    val[a] = act ? mom[a] / M : 0.0;
    fix[a] = false;
  }
  const unsigned bm = active ? bm0 : 0u;
  if (bm) {
    for (int i = 0; i < fs.bc.n; i++) {
      if (!((bm >> i) & 1u)) continue;
#pragma unroll
      for (int k = 0; k < ND; k++)
        if (k < fs.bc.dim[i] && ((fs.bc.bits[i] >> k) & 1)) {
          val[k] = fs.bc.v[i][k];
          fix[k] = true;
        }
    }
  }
}
// what k_nodal_accel makes of it (U-Verlet.c:947-957)
template <int ND>
__device__ __forceinline__ void fused_nodal_accel(const NView& N, const NodalFold& fs, int A, double* acc) {
  double dU[ND];
  bool fix[ND];
  const double M = *(N.nm + (size_t)A * (1 + ND));
  double f[ND];
#pragma unroll
  for (int a = 0; a < ND; a++) f[a] = *(N.force + (size_t)A * ND + a);
  const bool active = N.active[A];
  const unsigned bm0 = fs.bcmask ? fs.bcmask[A] : 0u;
  const bool act = active && M != 0.0;
  (void)dU;
#pragma unroll
  for (int a = 0; a < ND; a++) fix[a] = false;
  const unsigned bm = active ? bm0 : 0u;
  if (bm) {
    for (int i = 0; i < fs.bc.n; i++) {
      if (!((bm >> i) & 1u)) continue;
#pragma unroll
      for (int k = 0; k < ND; k++)
        if (k < fs.bc.dim[i] && ((fs.bc.bits[i] >> k) & 1)) fix[k] = true;
    }
  }
#pragma unroll
  for (int a = 0; a < ND; a++)
    acc[a] = (act && !fix[a]) ? fs.gv[a] + f[a] / M : 0.0;
}

// ------------------------------------------------------------------------------------------------
// K2: neighbour mask + beta + Newton + predictor + P2G(mass, m*dD)      (S1b + S2)
// ------------------------------------------------------------------------------------------------
// body of k2_tile for one work item; acc [NF * NWA] doubles and actrow [NROWS] words of LDS are the caller's.
// fs: unused by K2 (kept so that the three stage bodies share one signature)
template <int ND, bool P2G, int NT>
__device__ __forceinline__ void k2_body(const PView& P, const GridD& g, const NView& N, const TileD& td, const ParamsD& prm,
                                        double dt, double gamma_nm, int* __restrict__ gstatus, const TileWork& tw, int nbnd,
                                        double* acc, unsigned* actrow, const NodalFold* fs) {
  constexpr int W = TileCfg<ND>::W, NW = TileCfg<ND>::NW, NF = 1 + ND, NROWS = WinRows<ND>::NROWS;
  constexpr int WA = TileCfg<ND>::WA, PSA = TileCfg<ND>::PSA, NWA = TileCfg<ND>::NWA;
  constexpr int KN = Lme<ND>::KN;
  const int wb = tw.wb, tile = tw.tile, part = tw.part, nparts = tw.nparts;
  const int cnt = td.count[tile];
  PH_INIT
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  if (ND == 3 && NLPS_FAST_WINDOWS) {
    if (P2G)
      for (int idx = threadIdx.x; idx < NF * NWA; idx += NT) acc[idx] = 0.0;
    window_actrows3<NT>(g, w0, N.active, actrow);
  } else {
    for (int r = threadIdx.x; r < NROWS; r += NT) actrow[r] = 0u;
    if (P2G)
      for (int idx = threadIdx.x; idx < NF * NWA; idx += NT) acc[idx] = 0.0;
    __syncthreads();
    for (int idx = threadIdx.x; idx < NW; idx += NT) {
      bool in;
      int row, col;
      int node = window_node<ND>(g, w0, idx, in, &row, &col);
      if (in && N.active[node]) atomicOr(&actrow[row], 1u << col);
    }
  }
  __syncthreads();
  const int start = td.start[tile];
  PH(0)
  for (int s = part * NT + threadIdx.x; s < cnt; s += NT * nparts) {
    const int p = td.order[start + s];
    Lme<ND> c;
    double x[ND], lam[ND];
#pragma unroll
    for (int a = 0; a < ND; a++) {
      x[a] = PF(P, F_X + a, p);
      lam[a] = PF(P, F_LAM + a, p);
#if NLPS_LAMBDA_EXTRAPOLATE
      if (P2G) {  // start Newton from 2 lambda_n - lambda_{n-1}: same root, usually one iteration fewer
        const double lp = PF(P, F_LAMP + a, p);
        PF(P, F_LAMP + a, p) = lam[a];
        lam[a] = 2.0 * lam[a] - lp;
      }
#endif
    }
    const int I0 = P.I0[p];
    c.geom(g, x, I0);
    const int bx = c.ijk[0] - w0[0], by = c.ijk[1] - w0[1], bz = (ND == 3) ? c.ijk[2] - w0[2] : 0;
    const int base = bx + WA * by + (ND == 3 ? PSA * bz : 0);  // slot of I0 in the accumulator window
    const double beta_prev = PF(P, F_BETA, p);
    // beta of this step and the squared cut-off that belongs to it come from the node table; the list of THIS step is
    // cut with the beta of the previous one (LME.c:973,983-984), which is the same number whenever the old and the new
    // closest node have the same h_avg -- everywhere but beside the grid boundary and at the first search (beta = 0)
    const double4 bt = N.beta_t2[I0];
    const double beta = bt.x;
    double Ra, T2;
    if (__builtin_amdgcn_ballot_w64(beta_prev != beta) == 0ull) {
      T2 = bt.y;
      Ra = bt.z;
    } else {
      Ra = sqrt(prm.neg_log_tol_zero / beta_prev);  // LME.c:1052
      T2 = sqrt_threshold(Ra);                       // sqrt(|l|^2) <= Ra  <=>  |l|^2 <= T2
    }
    double lx2[5], ly2[5], lz2[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 5; i++) {
      lx2[i] = c.lx[i] * c.lx[i];
      ly2[i] = c.ly[i] * c.ly[i];
      if (ND == 3) lz2[i] = c.lz[i % KN] * c.lz[i % KN];
    }
    u64 mlo = 0ull, mhi = 0ull;
#if NLPS_MASK_BY_COLUMNS
    {
      // Radius test of the 125 stencil nodes, one (i, j) column of five planes at a time: the left-to-right sum of
      // generalised_Euclidean_distance (MatrixOp.c:895-920), (lx^2 + ly^2) + lz^2, shares its first addition over the
      // five planes (150 additions instead of 250), and every outcome is shifted into the 25-bit word of its plane,
      // word = 2 word + (sq <= T2) -- one compare and one add-with-carry per node instead of compare, select and or.
      // Columns run from (4, 4) down to (0, 0) so that bit i + 5 j ends up in its place.  Same booleans, bit for bit.
      unsigned pbk[KN];
#pragma unroll
      for (int k = 0; k < KN; k++) pbk[k] = 0u;
#pragma unroll
      for (int j = 4; j >= 0; j--)
#pragma unroll
        for (int i = 4; i >= 0; i--) {
          double s = 0.0;
          s += lx2[i];
          s += ly2[j];
#pragma unroll
          for (int k = 0; k < KN; k++) {
            double sq = s;
            if (ND == 3) sq += lz2[k];
            // (v_cmp_le_f64 is false for a NaN like the C comparison; the back end does not form the add-with-carry itself)
            asm("v_cmp_le_f64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(pbk[k]) : "v"(sq), "v"(T2) : "vcc");
          }
        }
#pragma unroll
      for (int k = 0; k < KN; k++) {
        unsigned act = 0u;
#pragma unroll
        for (int j = 0; j < 5; j++) {
          const int row = (by + j - 2) + (ND == 3 ? W * (bz + k - 2) : 0);
          act |= ((actrow[row] >> (bx - 2)) & 31u) << (5 * j);
        }
        put_plane(mlo, mhi, pbk[k] & act, k);
      }
    }
#else
#pragma unroll NLPS_KUNROLL_MASK
    for (int k = 0; k < KN; k++) {
      const double lz2k = (ND == 3) ? lz2[k] : 0.0;
      unsigned pbits = 0u;
#pragma unroll NLPS_JUNROLL_MASK
      for (int j = 0; j < 5; j++) {
        const int row = (by + j - 2) + (ND == 3 ? W * (bz + k - 2) : 0);
        const unsigned actbits = (actrow[row] >> (bx - 2)) & 31u;
        unsigned rb = 0u;
#pragma unroll
        for (int i = 0; i < 5; i++) {
          double sq = 0.0;  // same left-to-right sum as generalised_Euclidean_distance (MatrixOp.c:895-920)
          sq += lx2[i];
          sq += ly2[j];
          if (ND == 3) sq += lz2k;
          rb |= (sq <= T2) ? (1u << i) : 0u;
        }
        pbits |= (rb & actbits) << (5 * j);
      }
      put_plane(mlo, mhi, pbits, k);
    }
#endif
    c.mlo = mlo;
    c.mhi = mhi;
    const int nn = __popcll(mlo) + __popcll(mhi);
    if (nn < ND + 1) {  // LME.c:1087-1092
      P.nn[p] = 0;
      P.mlo[p] = 0ull;
      P.mhi[p] = 0ull;
      atomicOr(&P.status[p], ST_CONNECT);
      atomicOr(gstatus, ST_CONNECT);
      continue;
    }
    PH(1)
    int st = 0, NumIter = 0;
    double Zinv = 0.0;
    while (NumIter <= prm.max_iter_lme) {  // __lambda_Newton_Rapson, LME.c:272-353
      NLPS_FP_CONTRACT
      double r[ND], J[ND * ND], Jm1[ND * ND];
      c.factors(lam, beta, g.h);
      lme_moments_h<ND>(c, Zinv, r, J);
      PH_COUNT(7, 1)
      double aux = 0.0;
#pragma unroll
      for (int a = 0; a < ND; a++) aux += dsqr(r[a]);
      if (sqrt(aux) > prm.tol_wrapper) {
        if (rcond_below_gate<ND>(J) || !inverse<ND>(Jm1, J)) {
          st |= ST_NEWTON;
          break;
        }
        double dl[ND], dlr = 0.0, dl2 = 0.0;
#pragma unroll
        for (int a = 0; a < ND; a++) {
          double d = 0.0;
#pragma unroll
          for (int b2 = 0; b2 < ND; b2++) d = fma(Jm1[a * ND + b2], r[b2], d);
          dl[a] = d;
          lam[a] -= d;
          dlr = fma(d, r[a], dlr);
          dl2 = fma(d, d, dl2);
        }
        NumIter++;
#if NLPS_NEWTON_PREDICT_LAST
        // The reference's next pass would only confirm convergence: with D = -J^-1 r exactly, r(lambda + D) =
        // (1/2) T[D,D] + ..., T the third central moment of l under p, and |T_s[D,D]| <= max|l_a - r| D.J_s.D with
        // D.J.D = |D.r|.  Members satisfy |l_a| <= Ra, so |r_next| <= (1/2)(Ra + |r|) |D.r| (1 + O(|D| Ra)).  A hundred
        // times that below TOL_wrapper_LME means the reference stops at this lambda too; the pass that would have
        // told it so is replaced by the second-order update of what the scatter needs at the new lambda:
        // Z' = Z exp(D.r + D.J.D / 2) = Z exp(-dl.r / 2) and e'(l) = e(l) exp(D.l), both to < 1e-16 relative because
        // |D| Ra <= 1e-3 is required as well (then |D.r| <= 2e-12 / Ra forces |D| Ra ~ 1e-5 in practice).
        {
          const double nr = sqrt(aux);
          if (100.0 * 0.5 * (Ra + nr) * fabs(dlr) <= prm.tol_wrapper && dl2 * (Ra * Ra) <= 1.0e-6) {
            Zinv *= fma(0.5, dlr, 1.0);
            // l(i) = a - h (i - 2) along every axis (Lme::geom): D.l(i) = -dl a + (i - 2) dl h; only the centre
            // values of l stay live through the iteration
            const double hh = c.lx[2] - c.lx[3];
            const double sx = dl[0] * hh, sy = dl[1] * hh, sz = (ND == 3) ? dl[ND - 1] * hh : 0.0;
            const double bx0 = -dl[0] * c.lx[2], by0 = -dl[1] * c.ly[2], bz0 = (ND == 3) ? -dl[ND - 1] * c.lz[2 % KN] : 0.0;
#pragma unroll
            for (int i = 0; i < 5; i++) {
              const double u = (double)(i - 2);
              const double tx = fma(u, sx, bx0), ty = fma(u, sy, by0);
              c.ex[i] *= fma(tx, fma(tx, fma(tx, 1.0 / 6.0, 0.5), 1.0), 1.0);
              c.ey[i] *= fma(ty, fma(ty, fma(ty, 1.0 / 6.0, 0.5), 1.0), 1.0);
              if (ND == 3) {
                const double tz = fma(u, sz, bz0);
                c.ez[i % KN] *= fma(tz, fma(tz, fma(tz, 1.0 / 6.0, 0.5), 1.0), 1.0);
              }
            }
            break;
          }
        }
#endif
      } else {
        break;
      }
    }
    if (NumIter >= prm.max_iter_lme) st |= ST_NEWTON;
    PH(2)
    // The accesses below use their own copy of the particle index, opaque to the optimiser: it otherwise forms the
    // 64-bit addresses of all these components at the top of the loop and carries them (two VGPRs each) through the
    // mask build and the Newton iteration -- the 11 doubles K2 used to spill
    int pl = p;
    asm volatile("" : "+v"(pl));
    P.nn[pl] = nn;
    P.mlo[pl] = mlo;
    P.mhi[pl] = mhi;
    PF(P, F_BETA, pl) = beta;
#pragma unroll
    for (int a = 0; a < ND; a++) PF(P, F_LAM + a, pl) = lam[a];
    if (st) {
      atomicOr(&P.status[pl], st);
      atomicOr(gstatus, st);
    }
    if (!P2G) continue;  // local_search__LME__ alone stops here (LME.c:895-1015)
    double dd[ND];
#pragma unroll
    for (int a = 0; a < ND; a++) {
      double v = PF(P, F_VEL + a, pl), ac = PF(P, F_ACC + a, pl);
      dd[a] = dt * v + 0.5 * dsqr(dt) * ac;  // (lives in registers only: K3 writes the particle increment K5 reads)
      PF(P, F_VEL + a, pl) = v + (1 - gamma_nm) * dt * ac;
    }
    const double mz = PF(P, F_MASS, pl) * Zinv;
    NLPS_YZ_LOCALS(c);
#pragma unroll NLPS_KUNROLL_K2S
    for (int k = 0; k < KN; k++) {
      const unsigned pb = plane_bits<ND>(c, k);
      const double wz = mz * ez5[k];
      const int basek = base + (ND == 3 ? PSA * (k - 2) : 0);
#if NLPS_SCATTER_POP && !NLPS_SCATTER_BRANCHFREE
      unsigned pbs = pb << 7;  // pop_member: bit 24 (j = 4, i = 4) first
#pragma unroll
      for (int j = 4; j >= 0; j--) {
        const double w = wz * ey5[j];
#pragma unroll
        for (int i = 4; i >= 0; i--)
          if (pop_member(pbs)) {
            const int li = basek + (i - 2) + WA * (j - 2);
            const double v0 = w * c.ex[i];
            lds_add(&acc[li], v0);
#pragma unroll
            for (int a = 0; a < ND; a++) lds_add(&acc[(1 + a) * NWA + li], v0 * dd[a]);
          }
      }
      continue;
#endif
#pragma unroll NLPS_JUNROLL_SCATTER
      for (int j = 0; j < 5; j++) {
        const unsigned bits = (pb >> (5 * j)) & 31u;
        if (!wave_row_used(bits)) continue;
        const double w = wz * ey5[j];
#pragma unroll
        for (int i = 0; i < 5; i++)
#if NLPS_SCATTER_BRANCHFREE
        {  // every window slot of the stencil exists: non-members add an exact zero instead of branching around
          const int li = basek + (i - 2) + WA * (j - 2);
          const double v0 = w * masked_zero(c.ex[i], bits, i);
          lds_add(&acc[li], v0);
#pragma unroll
          for (int a = 0; a < ND; a++) lds_add(&acc[(1 + a) * NWA + li], v0 * dd[a]);
        }
#else
          if ((bits >> i) & 1u) {
            const int li = basek + (i - 2) + WA * (j - 2);
            const double v0 = w * c.ex[i];
            lds_add(&acc[li], v0);
#pragma unroll
            for (int a = 0; a < ND; a++) lds_add(&acc[(1 + a) * NWA + li], v0 * dd[a]);
          }
#endif
      }
    }
    PH(3)
  }
  if (!P2G) return;
  __syncthreads();
  PH(4)
  if (td.slab) {
    double* out = td.slab + ((size_t)tile * td.slab_n + td.slab_slot) * (NF * NWA);  // slab mode runs SPLIT = 1
    for (int q = threadIdx.x; q < NWA * NF; q += NT) out[q] = acc[q];
    return;
  }
  if (ND == 3 && NLPS_FAST_WINDOWS) {
    window_flush3<NF, NT>(g, w0, acc, N.nm);
  } else {
    for (int q = threadIdx.x; q < NWA * NF; q += NT) {
      int f = q % NF, idx = q / NF;
      double v = acc[f * NWA + idx];
      if (v != 0.0) {
        bool in;
        int node = window_node_a<ND>(g, w0, idx, in);
        if (in) atomic_add_f64(N.nm + (size_t)node * NF + f, v);
      }
    }
  }
  PH(5)
  tile_signal(td, wb, nbnd);
}
// NT / SPLIT: threads per workgroup and workgroups per tile.  The default is (BLK, K2_SPLIT); deterministic mode runs one
// wave per tile (64, 1): the sorted tile list is then accumulated in list order by a single instruction stream.
template <int ND, bool P2G, int NT = BLK, int SPLIT = K2_SPLIT>
__global__ __launch_bounds__(NT, NT == 64 ? 1 : (ND == 2 ? NLPS_K2_WAVES_2D : NLPS_K2_WAVES)) void k2_tile(PView P, GridD g, NView N, TileD td, ParamsD prm, double dt,
                                               double gamma_nm, int* __restrict__ gstatus) {
  __shared__ double acc[(1 + ND) * TileCfg<ND>::NWA];
  __shared__ unsigned actrow[WinRows<ND>::NROWS];
  const int nbnd = td.sig_flag ? td.range[4 + 2 * (SPLIT - 1) + 1] : 0;  // boundary workgroups come first (cls 0 view)
  tile_signal_empty(td, nbnd);
  TileWork tw;
  if (!tile_work_item<SPLIT>(td, tw)) return;
  k2_body<ND, P2G, NT>(P, g, N, td, prm, dt, gamma_nm, gstatus, tw, nbnd, acc, actrow, nullptr);
}

// ------------------------------------------------------------------------------------------------
// K3: G2P grad(dU) -> DF, F, J, density; stress; P2G of -f_int            (S3 + S4)
// ------------------------------------------------------------------------------------------------
// MODE 1: fused explicit stage (S3+S4).  MODE 0: __local_compatibility_conditions only (level B):
// DF, F_n1, J_n1 with the implicit driver's clamp of J <= 0 (U-Newmark-beta.c:1137-1142).
// MODE 3: the implicit driver's residual, __lagrangian_evaluation (U-Newmark-beta.c:970-1058), as ONE pass: the gather of the
// caller's dU -> DF, F_n1, J_n1 with the implicit clamp -> Stress_integration__Constitutive__ with everything the level-B
// constitutive stage stores (tau, W, b_e,n+1, kappa_n+1, eps_n+1, C_ep: the n state stays untouched, the residual is
// evaluated many times from it) -> P2G of the internal force.  Same registers-only hand-over of DF, tau between the
// stages as MODE 1, same kernel skeleton and occupancy; what it adds are the stores of the n+1 state.
// MODE 2: MODE 0 plus the rate tensors dt_DF = sum dV_A (x) grad N_A and dt_F_n1 = dt_DF F_n + DF dt_F_n
// (compute-Strains.c:48-72, 176-207) from a second gather window dV.
// FILT (clouds with several laws): the launch handles only the tile's particles whose material follows LAW; they are
// compacted into an LDS list first, so every lane works and the kernel is the single-law specialisation (one launch per
// law present; the run-time dispatch over all laws in one kernel needed 436 B of scratch per lane and 0.51 ms).
// waves per SIMD the register budget is set for: the fused 3-D Neo-Hookean stage fits three (two-pass gather), the
// same goes for Hencky once its LME factors are rebuilt after the stress update (RELOAD below), the plastic laws keep two
template <int ND, int LAW, int MODE>
struct K3Waves {
  static constexpr int value = ND == 2 ? NLPS_K3_WAVES_2D
                               : ((MODE == 1 || MODE == 3) && LAW == NLPS_MAT_NEO_HOOKEAN) ? NLPS_K3_WAVES_NH
                               : ((MODE == 1 || MODE == 3) && LAW == NLPS_MAT_HENCKY)      ? NLPS_K3_WAVES_HENCKY
                               : ((MODE == 1 || MODE == 3) && LAW == NLPS_MAT_DRUCKER_PRAGER) ? NLPS_K3_WAVES_DP
                                                                               : NLPS_K3_WAVES;
};
// the LDS of k3_tile, owned by the caller of k3_body (the kernels below)
template <int ND, int MODE, bool FILT>
struct K3Lds {
  static constexpr int NW = TileCfg<ND>::NW, NWA = TileCfg<ND>::NWA;
  static constexpr bool RATES = (MODE == 2);
  static constexpr int SELCAP = FILT ? 4096 : 1;
  static constexpr int N_DVXY = RATES ? 2 * NW : 2, N_DVZ = (RATES && ND == 3) ? NW : 1, N_DUZ = (ND == 3) ? NW : 1;
  double* dvxy;  // [N_DVXY], 16-byte aligned
  double* dvz;   // [N_DVZ]
  double* duxy;  // [2 NW], 16-byte aligned
  double* duz;   // [N_DUZ]
  double* fac;   // [ND NWA]
  int* sel;      // [SELCAP]
  int* nsel;     // [1]
  int* wcnt;     // [NT / 64]
};
// UMAT: the cloud holds ONE material (nlps_gpu_create: nmats == 1): its constants are read through a compile-time index,
// i.e. by scalar loads into scalar registers, instead of through the per-lane MatIdx into vector registers that then live
// through the whole stress update (9 doubles for Drucker-Prager: 72 -> 0 B of scratch at two waves per SIMD, Von-Mises
// 32 -> 0).  Tried first without a second set of kernels: one trip of the update per distinct material of the wave with
// the index from readfirstlane -- the loop-carried state made every law spill (Drucker-Prager 256 B).
template <int ND, int LAW, int MODE, bool FILT, int NT, bool UMAT = false>
__device__ __forceinline__ void k3_body(const PView& P, const GridD& g, const NView& N, const TileD& td,
                                        const MatD* __restrict__ mats, const ParamsD& prm, int* __restrict__ gstatus,
                                        const double* __restrict__ dVgrid, const TileWork& tw, int nbnd,
                                        const K3Lds<ND, MODE, FILT>& lds, const NodalFold* fs) {
  NLPS_FP_CONTRACT
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS, NW = TileCfg<ND>::NW, KN = Lme<ND>::KN;
  constexpr bool RATES = (MODE == 2);
  constexpr bool SCATTER = (MODE == 1 || MODE == 3);  // the stress update and the force scatter follow the F update
  // gather window of dU: {x,y} as one 16-B double2 per node (ds_read_b128) + z as a separate 8-B array
  // (ds_read_b64): with node strides of 16 B and 8 B the tile's 64 I0 positions hit distinct banks; a
  // padded 32-B AoS row put every second node on the same banks (41 % conflict cycles measured).
  double* const dvxy = lds.dvxy;
  double* const dvz = lds.dvz;
  double* const duxy = lds.duxy;
  double* const duz = lds.duz;
  double* const fac = lds.fac;
  int* const sel = lds.sel;
  int& nsel = *lds.nsel;
  int* const wcnt = lds.wcnt;
  constexpr int WA = TileCfg<ND>::WA, PSA = TileCfg<ND>::PSA, NWA = TileCfg<ND>::NWA;
  const int wb = tw.wb, tile = tw.tile, part = tw.part, nparts = tw.nparts;
  int cnt = td.count[tile];
  PH_INIT
  constexpr int SELCAP = FILT ? 4096 : 1;
  bool listed = false;  // sel[] holds this launch's particles
  if (FILT) {
    const int start0 = td.start[tile];
    if (threadIdx.x == 0) nsel = 0;
    __syncthreads();
    if (cnt <= SELCAP) {
      // ordered compaction (the selected particles keep the order of the tile list: runs of memory-consecutive
      // particles stay together, and the accumulation order is a function of the list alone)
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      for (int s0 = 0; s0 < cnt; s0 += NT) {  // uniform trip count
        const int s = s0 + threadIdx.x;
        int p = -1;
        bool f = false;
        if (s < cnt) {
          p = td.order[start0 + s];
          f = mats[P.mat[p]].type == LAW;
        }
        const unsigned long long m = __ballot(f);
        if (lane == 0) wcnt[wave] = (int)__popcll(m);
        __syncthreads();
        int base = nsel, tot = 0;
#pragma unroll
        for (int w = 0; w < NT / 64; w++) {
          if (w < wave) base += wcnt[w];
          tot += wcnt[w];
        }
        if (f) sel[base + (int)__popcll(m & ((1ull << lane) - 1ull))] = p;
        __syncthreads();
        if (threadIdx.x == 0) nsel += tot;
        __syncthreads();
      }
      cnt = nsel;
      listed = true;
      if (cnt == 0) {  // uniform: no particle of this law in the tile (its slab must still read as zeros)
        if (SCATTER && td.slab) {
          double* out = td.slab + ((size_t)tile * td.slab_n + td.slab_slot) * (ND * TileCfg<ND>::NWA);
          for (int qq = threadIdx.x; qq < TileCfg<ND>::NWA * ND; qq += NT) out[qq] = 0.0;
        }
        if (SCATTER) tile_signal(td, wb, nbnd);
        return;
      }
    }
  }
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx0 = threadIdx.x; idx0 < ((ND == 3 && NLPS_FAST_WINDOWS) ? 512 : NW); idx0 += NT) {
    bool in;
    int node, idx;
    if (ND == 3 && NLPS_FAST_WINDOWS) {
      node = window_cell3(g, w0, idx0, in);
      idx = (idx0 & 63) + PS * (idx0 >> 6);
    } else {
      idx = idx0;
      node = window_node<ND>(g, w0, idx, in);
    }
    if (fs) {  // (folded step: the nodal kernel's job, on the sums the K2 launch flushed)
      double val[ND];
      bool fix[ND];
#pragma unroll
      for (int a = 0; a < ND; a++) val[a] = 0.0;
      if (in) fused_nodal_dU<ND>(N, *fs, node, val, fix);
      duxy[2 * idx] = val[0];
      duxy[2 * idx + 1] = val[1];
      if (ND == 3) duz[(ND == 3) ? idx : 0] = val[2 % ND];
    } else {
      duxy[2 * idx] = in ? N.dU[(size_t)node * ND + 0] : 0.0;
      duxy[2 * idx + 1] = in ? N.dU[(size_t)node * ND + 1] : 0.0;
      if (ND == 3) duz[(ND == 3) ? idx : 0] = in ? N.dU[(size_t)node * ND + (2 % ND)] : 0.0;
    }
    if (RATES) {
      dvxy[RATES ? 2 * idx : 0] = in ? dVgrid[(size_t)node * ND + 0] : 0.0;
      dvxy[RATES ? 2 * idx + 1 : 1] = in ? dVgrid[(size_t)node * ND + 1] : 0.0;
      if (ND == 3) dvz[(RATES && ND == 3) ? idx : 0] = in ? dVgrid[(size_t)node * ND + (2 % ND)] : 0.0;
    }
  }
  for (int idx = threadIdx.x; idx < ND * NWA; idx += NT) fac[idx] = 0.0;
  __syncthreads();
  const double2* du2 = reinterpret_cast<const double2*>(duxy);
  const double2* dv2 = reinterpret_cast<const double2*>(dvxy);
  const int start = td.start[tile];
  PH(8)
  for (int s = part * NT + threadIdx.x; s < cnt; s += NT * nparts) {
    const int p = (FILT && listed) ? sel[FILT ? s : 0] : td.order[start + s];
    if (FILT && !listed && mats[P.mat[p]].type != LAW) continue;  // oversized tile: filter per lane
    Lme<ND> c;
    double lam[ND], beta;
    if (!load_lme<ND>(P, g, p, c, lam, beta)) continue;
    // (requested with the particle's other operands: the constants of its material are then one load away, not two, when
    // the stress update asks for them behind the gather)
    const int mat_idx = UMAT ? 0 : (SCATTER ? P.mat[p] : -1);
    const int base = window_base<ND>(c.ijk, w0);
    NLPS_YZ_LOCALS(c);
    PH(9)
    // pass 1: moments (rows -> planes) and G[a][m] = sum e dU_a l_m (rows)
    double Z = 0.0, rx = 0.0, ry = 0.0, rz = 0.0, Jxx = 0.0, Jxy = 0.0, Jxz = 0.0, Jyy = 0.0, Jyz = 0.0, Jzz = 0.0;
    double G[ND * ND], Gv[RATES ? ND * ND : 1];
    double U[ND];  // sum e dU: the value gather of U-Verlet.c:962-1010, by-product of the gradient rows
    double Uv[RATES ? ND : 1];
#pragma unroll
    for (int a = 0; a < ND; a++) U[a] = 0.0;
#pragma unroll
    for (int a = 0; a < (RATES ? ND : 1); a++) Uv[a] = 0.0;
#pragma unroll
    for (int a = 0; a < ND * ND; a++) G[a] = 0.0;
#pragma unroll
    for (int a = 0; a < (RATES ? ND * ND : 1); a++) Gv[a] = 0.0;
    // TWOPASS (3-D, no rate tensors): the gather of dU (LDS reads; G, U) and the moments (no memory traffic; Z, r, J)
    // run as two passes over the rows, each with its own masked weights.  The two live sets never overlap -- 12 + 17
    // doubles of accumulators and row temporaries in the first, 10 + 14 in the second, instead of 37 + 25 at once --
    // which is what lets the elastic laws run at three waves per SIMD; it costs ~350 more VALU instructions a particle.
    // (the level-B modes keep the single pass: with and without rate tensors they must give the same F bit for bit)
    // and the laws that stay at two waves per SIMD keep it too: there the second set of masked weights only costs)
    constexpr bool TWOPASS = (NLPS_K3_TWOPASS != 0) && ND == 3 && SCATTER && (K3Waves<ND, LAW, MODE>::value >= 3 || NLPS_K3_TWOPASS_ALL);
    // NLPS_K3_PRELOAD_FN (off): F_n requested between the two passes -- the moments pass touches no memory and keeps fewer
    // values alive than the gather, so the nine loads would land under it instead of in front of the F update; at 168
    // registers the nine values do not fit beside it (40 B of scratch, 2 % slower)
    constexpr bool PRELOAD_FN = TWOPASS && (NLPS_K3_PRELOAD_FN != 0);
    double Fn[ND * ND], fzz = 0.0;
    if (TWOPASS) {
#pragma unroll 1
      for (int k = 0; k < KN; k++) {
        const unsigned pb = plane_bits<ND>(c, k);
        const int basek = base + PS * (k - 2);
        const double zd0 = ez5[k], zd1 = zd0 * (double)(k - 2);
#pragma unroll NLPS_JUNROLL_K3
        for (int j = 0; j < 5; j++) {
          const unsigned bits = (pb >> (5 * j)) & 31u;
          double m[5], R0[ND], R1[ND];
#pragma unroll
          for (int a = 0; a < ND; a++) R0[a] = R1[a] = 0.0;
          masked_row(m, c.ex, bits);
          const double mu[5] = {-2.0 * m[0], -m[1], 0.0, m[3], 2.0 * m[4]};
#pragma unroll
          for (int i = 0; i < 5; i++) {
            const int li = basek + (i - 2) + W * (j - 2);
#if NLPS_ABL_GATHER
            const double2 u01 = make_double2(c.lx[2] + (double)li, c.ly[2]);
            const double u2 = c.lx[3];
#else
            const double2 u01 = du2[li];
            const double u2 = lds_z(duz, (ND == 3) ? li : 0);
#endif
            const double uu[3] = {u01.x, u01.y, u2};
#pragma unroll
            for (int a = 0; a < ND; a++) {
              R0[a] = fma(m[i], uu[a], R0[a]);
              if (i != 2) R1[a] = fma(mu[i], uu[a], R1[a]);
            }
          }
          const double y0 = ey5[j];
          const double w00 = y0 * zd0, w10 = (y0 * (double)(j - 2)) * zd0, w01 = y0 * zd1;
#pragma unroll
          for (int a = 0; a < ND; a++) {
            G[a * ND + 0] = fma(w00, R1[a], G[a * ND + 0]);
            G[a * ND + 1] = fma(w10, R0[a], G[a * ND + 1]);
            G[a * ND + (2 % ND)] = fma(w01, R0[a], G[a * ND + (2 % ND)]);
            U[a] = fma(w00, R0[a], U[a]);
          }
        }
      }
      if (PRELOAD_FN) {
        int pm = p;
        asm volatile("" : "+v"(pm));
        load_block<ND>(P, fFN(P), pm, Fn, fzz);
      }
#pragma unroll 1
      for (int k = 0; k < KN; k++) {
        const unsigned pb = plane_bits<ND>(c, k);
        double P00 = 0.0, P10 = 0.0, P20 = 0.0, P01 = 0.0, P11 = 0.0, P02 = 0.0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
          const unsigned bits = (pb >> (5 * j)) & 31u;
          double mw[5];
          masked_row(mw, c.ex, bits);
          const double m0 = mw[0], m1 = mw[1], m2 = mw[2], m3 = mw[3], m4 = mw[4];
          const double s13 = m1 + m3, s04 = m0 + m4;
          const double A0 = m2 + s13 + s04, A1 = fma(2.0, m4 - m0, m3 - m1), A2 = fma(4.0, s04, s13);
          const double cj = (double)(j - 2);
          const double y0 = ey5[j], y1 = y0 * cj, y2 = y1 * cj;
          P00 = fma(y0, A0, P00);
          P10 = fma(y0, A1, P10);
          P20 = fma(y0, A2, P20);
          P01 = fma(y1, A0, P01);
          P11 = fma(y1, A1, P11);
          P02 = fma(y2, A0, P02);
        }
        const double z0 = ez5[k], ck = (double)(k - 2), z1 = z0 * ck, z2 = z1 * ck;
        Z = fma(z0, P00, Z);
        rx = fma(z0, P10, rx);
        ry = fma(z0, P01, ry);
        rz = fma(z1, P00, rz);
        Jxx = fma(z0, P20, Jxx);
        Jxy = fma(z0, P11, Jxy);
        Jxz = fma(z1, P10, Jxz);
        Jyy = fma(z0, P02, Jyy);
        Jyz = fma(z1, P01, Jyz);
        Jzz = fma(z2, P00, Jzz);
      }
    }
#pragma unroll NLPS_KUNROLL_K3G
    for (int k = 0; k < (TWOPASS ? 0 : KN); k++) {
      const unsigned pb = plane_bits<ND>(c, k);
      const int basek = base + (ND == 3 ? PS * (k - 2) : 0);
      double P00 = 0.0, P10 = 0.0, P20 = 0.0, P01 = 0.0, P11 = 0.0, P02 = 0.0;
      double Gx[ND], Gy[ND], Gz[ND];  // plane partial sums of G[.][x], G[.][y], G[.][z]/lz
      double Hx[ND], Hy[ND], Hz[ND];  // the same for the velocity increments (RATES)
#pragma unroll
      for (int a = 0; a < ND; a++) Gx[a] = Gy[a] = Gz[a] = Hx[a] = Hy[a] = Hz[a] = 0.0;
      // DIRECT (3-D, no rate tensors): every row goes straight into the totals with its y*z weight -- 15 doubles of
      // plane partials less to keep alive (the kernel then fits three waves per SIMD), for 5 more FMAs per row
      constexpr bool DIRECT = (NLPS_K3_DIRECT != 0) && ND == 3 && SCATTER && LAW == NLPS_MAT_NEO_HOOKEAN;
      const double zd0 = ez5[k], zd1 = zd0 * (double)(k - 2), zd2 = zd1 * (double)(k - 2);
#pragma unroll NLPS_JUNROLL_K3
      for (int j = 0; j < 5; j++) {
        const unsigned bits = (pb >> (5 * j)) & 31u;
        if (!wave_row_used(bits)) continue;
        // INDEX space (as lme_moments_h): with l_x(i) = a_x - h (i - 2) the row weights are the integers
        // u = i - 2 in {-2..2}: no l arrays live in the loop, u = 0 terms vanish, +-1 are sign modifiers.
        double m[5], R0[ND], R1[ND], V0r[ND], V1r[ND];
#pragma unroll
        for (int a = 0; a < ND; a++) R0[a] = R1[a] = V0r[a] = V1r[a] = 0.0;
        masked_row(m, c.ex, bits);  // non-members weigh 0
        const double w0 = -2.0 * m[0], w4 = 2.0 * m[4];
        const double mu[5] = {w0, -m[1], 0.0, m[3], w4};  // u_i m_i
#pragma unroll
        for (int i = 0; i < 5; i++) {  // branch-free (the window slot of a non-member exists)
          const int li = basek + (i - 2) + W * (j - 2);
          const double2 u01 = du2[li];
          const double u2 = (ND == 3) ? lds_z(duz, (ND == 3) ? li : 0) : 0.0;
          const double uu[3] = {u01.x, u01.y, u2};
#pragma unroll
          for (int a = 0; a < ND; a++) {
            R0[a] = fma(m[i], uu[a], R0[a]);
            if (i != 2) R1[a] = fma(mu[i], uu[a], R1[a]);
          }
          if (RATES) {
            const double2 v01 = dv2[RATES ? li : 0];
            const double v2 = (ND == 3) ? lds_z(dvz, (RATES && ND == 3) ? li : 0) : 0.0;
            const double vv[3] = {v01.x, v01.y, v2};
#pragma unroll
            for (int a = 0; a < ND; a++) {
              V0r[a] = fma(m[i], vv[a], V0r[a]);
              if (i != 2) V1r[a] = fma(mu[i], vv[a], V1r[a]);
            }
          }
        }
        const double s13 = m[1] + m[3], s04 = m[0] + m[4];
        const double A0 = m[2] + s13 + s04;                    // sum e
        const double A1 = (m[3] - m[1]) + (w4 + w0);           // sum e u
        const double A2 = fma(4.0, s04, s13);                  // sum e u^2
        const double cj = (double)(j - 2);
        const double y0 = ey5[j], y1 = y0 * cj, y2 = y1 * cj;
        if (DIRECT) {
          const double w00 = y0 * zd0, w10 = y1 * zd0, w20 = y2 * zd0, w01 = y0 * zd1, w11 = y1 * zd1, w02 = y0 * zd2;
          Z = fma(w00, A0, Z);
          rx = fma(w00, A1, rx);
          ry = fma(w10, A0, ry);
          rz = fma(w01, A0, rz);
          Jxx = fma(w00, A2, Jxx);
          Jxy = fma(w10, A1, Jxy);
          Jxz = fma(w01, A1, Jxz);
          Jyy = fma(w20, A0, Jyy);
          Jyz = fma(w11, A0, Jyz);
          Jzz = fma(w02, A0, Jzz);
#pragma unroll
          for (int a = 0; a < ND; a++) {
            G[a * ND + 0] = fma(w00, R1[a], G[a * ND + 0]);
            G[a * ND + 1] = fma(w10, R0[a], G[a * ND + 1]);
            G[a * ND + (2 % ND)] = fma(w01, R0[a], G[a * ND + (2 % ND)]);
            U[a] = fma(w00, R0[a], U[a]);
          }
          continue;
        }
        P00 = fma(y0, A0, P00);
        P10 = fma(y0, A1, P10);
        P20 = fma(y0, A2, P20);
        P01 = fma(y1, A0, P01);
        P11 = fma(y1, A1, P11);
        P02 = fma(y2, A0, P02);
#pragma unroll
        for (int a = 0; a < ND; a++) {
          Gx[a] = fma(y0, R1[a], Gx[a]);
          Gy[a] = fma(y1, R0[a], Gy[a]);
          Gz[a] = fma(y0, R0[a], Gz[a]);
          if (RATES) {
            Hx[a] = fma(y0, V1r[a], Hx[a]);
            Hy[a] = fma(y1, V0r[a], Hy[a]);
            Hz[a] = fma(y0, V0r[a], Hz[a]);
          }
        }
      }
      if (DIRECT) {
        // totals already updated row by row
      } else if (ND == 3) {
        const double z0 = ez5[k], ck = (double)(k - 2), z1 = z0 * ck, z2 = z1 * ck;
        Z = fma(z0, P00, Z);
        rx = fma(z0, P10, rx);
        ry = fma(z0, P01, ry);
        rz = fma(z1, P00, rz);
        Jxx = fma(z0, P20, Jxx);
        Jxy = fma(z0, P11, Jxy);
        Jxz = fma(z1, P10, Jxz);
        Jyy = fma(z0, P02, Jyy);
        Jyz = fma(z1, P01, Jyz);
        Jzz = fma(z2, P00, Jzz);
#pragma unroll
        for (int a = 0; a < ND; a++) {
          G[a * ND + 0] = fma(z0, Gx[a], G[a * ND + 0]);
          G[a * ND + 1] = fma(z0, Gy[a], G[a * ND + 1]);
          G[a * ND + (2 % ND)] = fma(z1, Gz[a], G[a * ND + (2 % ND)]);
          U[a] = fma(z0, Gz[a], U[a]);
          if (RATES) Uv[a % (RATES ? ND : 1)] = fma(z0, Hz[a], Uv[a % (RATES ? ND : 1)]);
          if (RATES) {
            Gv[(a * ND + 0) % (RATES ? ND * ND : 1)] = fma(z0, Hx[a], Gv[(a * ND + 0) % (RATES ? ND * ND : 1)]);
            Gv[(a * ND + 1) % (RATES ? ND * ND : 1)] = fma(z0, Hy[a], Gv[(a * ND + 1) % (RATES ? ND * ND : 1)]);
            Gv[(a * ND + (2 % ND)) % (RATES ? ND * ND : 1)] = fma(z1, Hz[a], Gv[(a * ND + (2 % ND)) % (RATES ? ND * ND : 1)]);
          }
        }
      } else {
        Z = P00;
        rx = P10;
        ry = P01;
        Jxx = P20;
        Jxy = P11;
        Jyy = P02;
#pragma unroll
        for (int a = 0; a < ND; a++) {
          G[a * ND + 0] = Gx[a];
          G[a * ND + 1] = Gy[a];
          U[a] = Gz[a];
          if (RATES) Uv[a % (RATES ? ND : 1)] = Hz[a];
          if (RATES) {
            Gv[(a * ND + 0) % (RATES ? ND * ND : 1)] = Hx[a];
            Gv[(a * ND + 1) % (RATES ? ND * ND : 1)] = Hy[a];
          }
        }
      }
    }
    PH(10)
    // from here on the particle index is a copy the optimiser cannot see through: the 64-bit addresses of the ~40
    // components read and written below would otherwise be formed at the top of the loop and carried through the gather
    int pl = p;
    asm volatile("" : "+v"(pl));
    const double Zinv = 1.0 / Z;
    // index moments -> moments of l = a - h u (a = l of the centre node): J = h^2 (<u u> - <u><u>), the a-terms cancel;
    // G[a][m] = sum e dU_a l_m = a_m sum e dU_a - h sum e dU_a u_m
    const double hx = c.lx[2] - c.lx[3];
    const double al[3] = {c.lx[2], c.ly[2], (ND == 3) ? c.lz[2 % KN] : 0.0};
    rx *= Zinv;
    ry *= Zinv;
    rz *= Zinv;
    double J[ND * ND], Jm1[ND * ND];
    const double h2 = hx * hx;
    if (ND == 2) {
      J[0] = h2 * (Jxx * Zinv - rx * rx);
      J[1] = J[2] = h2 * (Jxy * Zinv - rx * ry);
      J[3] = h2 * (Jyy * Zinv - ry * ry);
    } else {
      J[0] = h2 * (Jxx * Zinv - rx * rx);
      J[1] = J[3 % (ND * ND)] = h2 * (Jxy * Zinv - rx * ry);
      J[2] = J[6 % (ND * ND)] = h2 * (Jxz * Zinv - rx * rz);
      J[4 % (ND * ND)] = h2 * (Jyy * Zinv - ry * ry);
      J[5 % (ND * ND)] = J[7 % (ND * ND)] = h2 * (Jyz * Zinv - ry * rz);
      J[8 % (ND * ND)] = h2 * (Jzz * Zinv - rz * rz);
    }
#pragma unroll
    for (int a = 0; a < ND; a++)
#pragma unroll
      for (int mm = 0; mm < ND; mm++) {
        G[a * ND + mm] = fma(al[mm], U[a], -hx * G[a * ND + mm]);
        if (RATES)
          Gv[(a * ND + mm) % (RATES ? ND * ND : 1)] =
              fma(al[mm], Uv[a % (RATES ? ND : 1)], -hx * Gv[(a * ND + mm) % (RATES ? ND * ND : 1)]);
      }
    int st = 0;
    if (!inverse<ND>(Jm1, J)) st |= ST_NEWTON;
    // DF = I + sum_A dU_A (x) grad N_A = I - (G/Z) J^-T            (compute-Strains.c:20-44)
    double DF[ND * ND], Fn1[ND * ND];
#pragma unroll
    for (int i = 0; i < ND; i++)
#pragma unroll
      for (int j = 0; j < ND; j++) {
        double v = 0.0;
#pragma unroll
        for (int m = 0; m < ND; m++) v = fma(G[i * ND + m] * Zinv, Jm1[j * ND + m], v);
        DF[i * ND + j] = ((i == j) ? 1.0 : 0.0) - v;
      }
    if (!PRELOAD_FN) load_block<ND>(P, fFN(P), pl, Fn, fzz);
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
    if (Jn1 <= 0.0) {
      st |= ST_JACOBIAN;  // fatal in the explicit scheme (U-Verlet.c:608-613), clamped in the implicit one
      if (MODE != 1) Jn1 = 0.0;
    }
    // the fused explicit step (MODE 1) keeps DF in registers: nothing reads it before the next step rewrites it, and a
    // level-B stage or a download gets it back as F_n+1 F_n^-1 from the two slots of F (k_copy_n_to_n1): 72 B per
    // particle less to store (K3 0.260 -> 0.251 ms at 1 M particles)
    if (MODE != 1) store_block<ND>(P, F_DF, pl, DF, 0.0, false);
    store_block<ND>(P, fFN1(P), pl, Fn1, 0.0, false);
    // fused step: J goes straight to its n slot (nothing reads J_n inside the step; J_n+1 = J_n is restored with the
    // roll, k_copy_n_to_n1), so K5 has no copy to make
    PF(P, MODE == 1 ? F_JN : F_JN1, pl) = Jn1;
    if (RATES) {
      double dDF[ND * ND], dFn[ND * ND], dFn1[ND * ND], zz;
#pragma unroll
      for (int i = 0; i < ND; i++)
#pragma unroll
        for (int j = 0; j < ND; j++) {
          double v = 0.0;
#pragma unroll
          for (int m = 0; m < ND; m++) v = fma(Gv[(i * ND + m) % (RATES ? ND * ND : 1)] * Zinv, Jm1[j * ND + m], v);
          dDF[i * ND + j] = -v;
        }
      load_block<ND>(P, F_DTFN, pl, dFn, zz);
#pragma unroll
      for (int i = 0; i < ND; i++)
#pragma unroll
        for (int j = 0; j < ND; j++) {
          double a2 = 0.0;
#pragma unroll
          for (int k2 = 0; k2 < ND; k2++) a2 += dDF[i * ND + k2] * Fn[k2 * ND + j] + DF[i * ND + k2] * dFn[k2 * ND + j];
          dFn1[i * ND + j] = a2;
        }
      store_block<ND>(P, F_DTDF, pl, dDF, 0.0, false);
      store_block<ND>(P, F_DTFN1, pl, dFn1, 0.0, false);
    }
    if (!SCATTER) {
      if (st) {
        atomicOr(&P.status[pl], st);
        atomicOr(gstatus, st);
      }
      continue;
    }
    // (the density update rho <- rho / det DF of U-Verlet.c:630-632 costs no traffic: rho J is invariant, F_RHOJ)
    if (MODE == 1) {
#pragma unroll
      for (int a = 0; a < ND; a++) PF(P, F_DDIS + a, pl) = U[a] * Zinv;  // d_dis_p = sum N dU (used by K5)
    }
    double tau[ND * ND], B[ND * ND];
    // (MODE 3: C_ep kept for the tangent that may follow, everything to the n+1 slots)
    st |= stress_update<ND, LAW, MODE == 3, (LAW == NLPS_KLAW_FRICTIONAL), MODE == 1>(P, pl, mats, prm, Fn1, DF, Jn1, tau, mat_idx);
    // (MODE 1 accumulates -f_int, what the explicit scheme divides by the mass; MODE 3 the +f_int of the Lagrangian, :1359)
    // (measured and dropped, round 4: DF^-T J^-1 formed BEFORE the stress update so that DF and J^-1 need not live through
    // it -- the allocator then spills more, not less: Drucker-Prager at three waves 152 -> 192 B of scratch, Hencky 8 -> 44)
    const bool fo_ok = force_operator<ND>(B, tau, DF, Jm1, PF(P, F_VOL0, pl), MODE == 3 ? 1.0 : -1.0);
    PH(11)
    if (fo_ok) {
      // RELOAD (the laws whose stress update sets the register budget): the LME factors are rebuilt from the particle's
      // stored x, lambda, beta and masks after the stress update instead of living through it (15 exponentials again, for
      // 40 registers less across the most register-hungry part of the kernel)
      // Measured at 1 M particles: Hencky at three waves per SIMD without scratch 0.290 ms (0.295 at two); Drucker-Prager
      // gains nothing from it (two waves with the reload 0.363 ms, three 0.358 ms with 212 B of scratch, 0.348 ms as it
      // was): those kernels are bound by their instruction count (4100 / 5900 static), not by latency.
      constexpr bool RELOAD = (NLPS_K3_RELOAD != 0) && ND == 3 && SCATTER && LAW != NLPS_MAT_NEO_HOOKEAN && K3Waves<ND, LAW, MODE>::value >= 3;
      Lme<ND> cs;
      if (RELOAD) {
        int p2 = pl;
        asm volatile("" : "+v"(p2));
        double lam2[ND], beta2;
        load_lme<ND>(P, g, p2, cs, lam2, beta2);
      } else {
        cs = c;
      }
      const Lme<ND>& c = cs;
      NLPS_YZ_LOCALS(c);
      (void)ly5;
      (void)lz5;
      const double hx = c.lx[2] - c.lx[3];
      const double al[3] = {c.lx[2], c.ly[2], (ND == 3) ? c.lz[2 % KN] : 0.0};
      // pass 2: -f_A = p_A * (B l_A) with l = a - h u:  B l = B a - h (B[.][x] u_i + B[.][y] v_j + B[.][z] w_k)
      double Ba[ND], hB[ND * ND];
#pragma unroll
      for (int a = 0; a < ND; a++) {
        double v = 0.0;
#pragma unroll
        for (int mm = 0; mm < ND; mm++) {
          v = fma(B[a * ND + mm], al[mm], v);
          hB[a * ND + mm] = -hx * B[a * ND + mm];
        }
        Ba[a] = v;
      }
      const int basea = window_base_a<ND>(c.ijk, w0);
#pragma unroll NLPS_KUNROLL_K3S
      for (int k = 0; k < KN; k++) {
        const unsigned pb = plane_bits<ND>(c, k);
        const int basek = basea + (ND == 3 ? PSA * (k - 2) : 0);
        const double wz = Zinv * ez5[k];
        const double ck = (double)(k - 2);
        double cz[ND];
#pragma unroll
        for (int a = 0; a < ND; a++) cz[a] = (ND == 3) ? fma(hB[a * ND + (2 % ND)], ck, Ba[a]) : Ba[a];
#if NLPS_SCATTER_POP && !NLPS_SCATTER_BRANCHFREE
        unsigned pbs = pb << 7;  // pop_member: bit 24 (j = 4, i = 4) first
#pragma unroll
        for (int j = 4; j >= 0; j--) {
          const double w = wz * ey5[j];
          double cr[ND];
#pragma unroll
          for (int a = 0; a < ND; a++) cr[a] = fma(hB[a * ND + 1], (double)(j - 2), cz[a]);
#pragma unroll
          for (int i = 4; i >= 0; i--)
            if (pop_member(pbs)) {
              const int li = basek + (i - 2) + WA * (j - 2);
              const double we = w * c.ex[i];
#pragma unroll
              for (int a = 0; a < ND; a++) lds_add(&fac[a * NWA + li], we * fma(hB[a * ND + 0], (double)(i - 2), cr[a]));
            }
        }
        continue;
#endif
#pragma unroll NLPS_JUNROLL_SCATTER
        for (int j = 0; j < 5; j++) {
          const unsigned bits = (pb >> (5 * j)) & 31u;
          if (!wave_row_used(bits)) continue;
          const double w = wz * ey5[j];
          double cr[ND];
#pragma unroll
          for (int a = 0; a < ND; a++) cr[a] = fma(hB[a * ND + 1], (double)(j - 2), cz[a]);
#pragma unroll
          for (int i = 0; i < 5; i++)
#if NLPS_SCATTER_BRANCHFREE
          {
            const int li = basek + (i - 2) + WA * (j - 2);
            const double we = w * masked_zero(c.ex[i], bits, i);
#pragma unroll
            for (int a = 0; a < ND; a++) lds_add(&fac[a * NWA + li], we * fma(hB[a * ND + 0], (double)(i - 2), cr[a]));
          }
#else
            if ((bits >> i) & 1u) {
              const int li = basek + (i - 2) + WA * (j - 2);
              const double we = w * c.ex[i];
#pragma unroll
              for (int a = 0; a < ND; a++) lds_add(&fac[a * NWA + li], we * fma(hB[a * ND + 0], (double)(i - 2), cr[a]));
            }
#endif
        }
      }
    } else {
      st |= ST_JACOBIAN;
    }
    if (st) {
      atomicOr(&P.status[p], st);
      atomicOr(gstatus, st);
    }
    PH(12)
  }
  if (!SCATTER) return;
  __syncthreads();
  PH(13)
  if (td.slab) {
    double* out = td.slab + ((size_t)tile * td.slab_n + td.slab_slot) * (ND * NWA);
    for (int qq = threadIdx.x; qq < NWA * ND; qq += NT) out[qq] = fac[qq];
    return;
  }
  if (ND == 3 && NLPS_FAST_WINDOWS) {
    window_flush3<ND, NT>(g, w0, fac, N.force);
  } else {
    for (int qq = threadIdx.x; qq < NWA * ND; qq += NT) {
      int f = qq % ND, idx = qq / ND;
      double v = fac[f * NWA + idx];
      if (v != 0.0) {
        bool in;
        int node = window_node_a<ND>(g, w0, idx, in);
        if (in) atomic_add_f64(N.force + (size_t)node * ND + f, v);
      }
    }
  }
  PH(14)
  tile_signal(td, wb, nbnd);
}
template <int ND, int LAW, int MODE, bool FILT = false, int NT = K3_BLK, bool UMAT = false>
__global__ __launch_bounds__(NT, (NT == 64 ? 1 : (K3Waves<ND, LAW, MODE>::value))) void k3_tile(PView P, GridD g, NView N, TileD td, const MatD* __restrict__ mats,
                                               ParamsD prm, int* __restrict__ gstatus,
                                               const double* __restrict__ dVgrid) {
  using L = K3Lds<ND, MODE, FILT>;
  __shared__ __attribute__((aligned(16))) double dvxy[L::N_DVXY];
  __shared__ double dvz[L::N_DVZ];
  __shared__ __attribute__((aligned(16))) double duxy[2 * L::NW];
  __shared__ double duz[L::N_DUZ];
  __shared__ double fac[ND * L::NWA];
  __shared__ int sel[L::SELCAP];
  __shared__ int nsel;
  __shared__ int wcnt[NT / 64];
  const int nbnd = ((MODE == 1 || MODE == 3) && td.sig_flag) ? td.range[4 + 2 * (K3_SPLIT - 1) + 1] : 0;
  if (MODE == 1 || MODE == 3) tile_signal_empty(td, nbnd);
  TileWork tw;
  if (!tile_work_item<K3_SPLIT>(td, tw)) return;
  const L lds{dvxy, dvz, duxy, duz, fac, sel, &nsel, wcnt};
  k3_body<ND, LAW, MODE, FILT, NT, UMAT>(P, g, N, td, mats, prm, gstatus, dVgrid, tw, nbnd, lds, nullptr);
}

// The explicit step of one GPU without a ghost exchange, "folded" form: the nodal kernels between the stages are gone.
// K3 makes dU = sum m N dD / M (+ Dirichlet values) of its window nodes from the sums K2 flushed, K5 makes a = g + f / M
// from the forces K3 flushed (fused_nodal_dU / fused_nodal_accel: what k_nodal_dU / k_nodal_accel do per node, here per
// window slot -- 16 x the divisions, two per thread and tile), and K3's workgroups reset on the side what the search
// inside K5 accumulates into.  Three launches and ~35 us per step less at 1 M particles; the nodal arrays nobody reads
// during the step are made when somebody asks (nlps_gpu_explicit_nodal, nodal_stale).
struct LazyNodal {
  NodalFold fs;  // bc, bcmask, gv
  int n0, nwn;   // node window
  int* node_cnt;
  int* tile_count;
  int ntw;
};
template <int ND, int LAW, bool UMAT = false>
__global__ __launch_bounds__(K3_BLK, (K3Waves<ND, LAW, 1>::value)) void k3_tile_lazy(PView P, GridD g, NView N, TileD td, const MatD* __restrict__ mats,
                                                                                  ParamsD prm, int* __restrict__ gstatus, LazyNodal ln) {
  using L = K3Lds<ND, 1, false>;
  __shared__ __attribute__((aligned(16))) double dvxy[L::N_DVXY];
  __shared__ double dvz[L::N_DVZ];
  __shared__ __attribute__((aligned(16))) double duxy[2 * L::NW];
  __shared__ double duz[L::N_DUZ];
  __shared__ double fac[ND * L::NWA];
  __shared__ int sel[L::SELCAP];
  __shared__ int nsel;
  __shared__ int wcnt[K3_BLK / 64];
  for (int i = blockIdx.x * K3_BLK + threadIdx.x; i < max(ln.nwn, ln.ntw); i += gridDim.x * K3_BLK) {
    if (i < ln.ntw) ln.tile_count[i] = 0;
    if (i < ln.nwn) {
      N.seed[ln.n0 + i] = 0;
      if (ln.node_cnt) ln.node_cnt[ln.n0 + i] = 0;
    }
  }
  const int nbnd = td.sig_flag ? td.range[4 + 2 * (K3_SPLIT - 1) + 1] : 0;  // (overlap mode 2, as k3_tile)
  tile_signal_empty(td, nbnd);
  TileWork tw;
  if (!tile_work_item<K3_SPLIT>(td, tw)) return;
  const L lds{dvxy, dvz, duxy, duz, fac, sel, &nsel, wcnt};
  k3_body<ND, LAW, 1, false, K3_BLK, UMAT>(P, g, N, td, mats, prm, gstatus, nullptr, tw, nbnd, lds, &ln.fs);
}

// Sums, for every node of two node ranges, the window slabs of the tiles whose window holds the node (<= 2 per axis)
// in a fixed order and writes out[node][NF]: the second half of the P2G flush (see TileD::slab).  A tile's slab is
// valid iff the tile was launched this step (inside [tile0, tile0 + ntw) and count > 0); it then holds td.slab_n slabs.
template <int ND, int NF>
__global__ void k_slab_gather(int a0, int an, int b0, int bn, GridD g, TileD td, double* __restrict__ out) {
  constexpr int TB = TileCfg<ND>::TB, W = TileCfg<ND>::WA, PS = TileCfg<ND>::PSA, NW = TileCfg<ND>::NWA;
  int A = blockIdx.x * blockDim.x + threadIdx.x;
  if (A >= an + bn) return;
  A = A < an ? a0 + A : b0 + (A - an);
  const int ijk[3] = {A % g.n[0], (A / g.n[0]) % g.n[1], A / (g.n[0] * g.n[1])};
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < ND; a++) {
    lo[a] = ijk[a] >= 2 ? (ijk[a] - 2) / TB : 0;  // window of tile t covers nodes [TB t - 2, TB t + TB + 1]
    hi[a] = min(td.nt[a] - 1, (ijk[a] + 2) / TB);
  }
  double s[NF];
#pragma unroll
  for (int f = 0; f < NF; f++) s[f] = 0.0;
  for (int tz = lo[2]; tz <= hi[2]; tz++)
    for (int ty = lo[1]; ty <= hi[1]; ty++)
      for (int tx = lo[0]; tx <= hi[0]; tx++) {
        const int t = tx + td.nt[0] * (ty + td.nt[1] * tz);
        if (t < td.tile0 || t >= td.tile0 + td.ntw) continue;
        const int cnt = td.count[t];
        if (cnt <= 0) continue;
        const int idx = (ijk[0] - (tx * TB - 2)) + W * (ijk[1] - (ty * TB - 2)) + (ND == 3 ? PS * (ijk[2] - (tz * TB - 2)) : 0);
        for (int q = 0; q < td.slab_n; q++) {
          const double* src = td.slab + ((size_t)t * td.slab_n + q) * (NF * NW) + idx;
#pragma unroll
          for (int f = 0; f < NF; f++) s[f] += src[f * NW];
        }
      }
#pragma unroll
  for (int f = 0; f < NF; f++) out[(size_t)A * NF + f] = s[f];
}

// ------------------------------------------------------------------------------------------------
// K5: G2P of the nodal acceleration, corrector, roll                     (S5)
// (the value gather of dU is a by-product of K3's gradient rows and arrives in F_DDIS).  The gather window
// keeps {ax, ay} as one 16-B double2 per node (ds_read_b128) and az in a separate 8-B array, the layout that
// is conflict-free for the tile's 64 I0 positions (see K3).
// ------------------------------------------------------------------------------------------------
// SEARCH: the kernel also does the first stage of the NEXT step for the particles it has just moved -- closest-node
// update from the new position (LME.c:924-930), activation seed, binning to tiles and per-node ranks (k_search) --
// while x, dis and I0 are in registers: one launch and one pass over the particle arrays less per step.  The seeds and
// counters it writes were reset by the nodal kernel in front of it (k_nodal_accel).
struct K5Search {
  const uint8_t* rank1;  // chain positions of the 3^d candidates per boundary class (get_closest_node tie-break)
  TileCnt tc;
  int bin;  // 0: closest-node update only (I0n); the seeds and bins are left to k_search in its adopt form
};
template <int ND, int LAW, bool SEARCH>
__device__ __forceinline__ void k5_body(const PView& P, const GridD& g, const NView& N, const TileD& td, double dt,
                                        double gamma_nm, const K5Search& ks, const TileWork& tw, double* axy, double* az,
                                        const NodalFold* fs, int* __restrict__ gstatus) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS, NW = TileCfg<ND>::NW, KN = Lme<ND>::KN;
  const int tile = tw.tile, part = tw.part, nparts = tw.nparts;
  const int cnt = td.count[tile];
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx0 = threadIdx.x; idx0 < ((ND == 3 && NLPS_FAST_WINDOWS) ? 512 : NW); idx0 += K5_BLK) {
    bool in;
    int node, idx;
    if (ND == 3 && NLPS_FAST_WINDOWS) {
      node = window_cell3(g, w0, idx0, in);
      idx = (idx0 & 63) + PS * (idx0 >> 6);
    } else {
      idx = idx0;
      node = window_node<ND>(g, w0, idx, in);
    }
    if (fs) {  // (folded step: the nodal kernel's job, on the forces the K3 launch flushed)
      double av[ND];
#pragma unroll
      for (int a = 0; a < ND; a++) av[a] = 0.0;
      if (in) fused_nodal_accel<ND>(N, *fs, node, av);
      axy[2 * idx] = av[0];
      axy[2 * idx + 1] = av[1];
      if (ND == 3) az[(ND == 3) ? idx : 0] = av[2 % ND];
    } else {
      axy[2 * idx] = in ? N.accel[(size_t)node * ND + 0] : 0.0;
      axy[2 * idx + 1] = in ? N.accel[(size_t)node * ND + 1] : 0.0;
      if (ND == 3) az[(ND == 3) ? idx : 0] = in ? N.accel[(size_t)node * ND + (2 % ND)] : 0.0;
    }
  }
  __syncthreads();
  const double2* a2 = reinterpret_cast<const double2*>(axy);
  const int start = td.start[tile];
  // (SEARCH: the binning is wave-cooperative, so every lane of a wave makes the same number of trips)
  for (int s0 = part * K5_BLK; s0 < cnt; s0 += K5_BLK * nparts) {
    const int s = s0 + (int)threadIdx.x;
    const bool have = s < cnt;
    if (!SEARCH && !have) continue;
    const int p = have ? td.order_m[start + s] : 0;
    int I0n = 0;
    bool binned = false;
    do {
      if (!have) break;
      Lme<ND> c;
      double lam[ND], beta;
      if (!load_lme<ND>(P, g, p, c, lam, beta)) {
        if (SEARCH) {  // a particle without a neighbourhood does not move, but the next step still lists it
          I0n = P.I0[p];
          binned = true;
        }
        break;
      }
      const int base = window_base<ND>(c.ijk, w0);
      NLPS_YZ_LOCALS(c);
      double Z = 0.0, sv[ND];
#pragma unroll
      for (int a = 0; a < ND; a++) sv[a] = 0.0;
      // the operands of the corrector are requested before the gather loop, so that their latency runs under it
      double dd_[ND], vel_[ND], dis_[ND];
#pragma unroll
      for (int a = 0; a < ND; a++) {
        dd_[a] = PF(P, F_DDIS + a, p);
        vel_[a] = PF(P, F_VEL + a, p);
        dis_[a] = PF(P, F_DIS + a, p);
      }
#pragma unroll NLPS_KUNROLL_K5
      for (int k = 0; k < KN; k++) {
        const unsigned pb = plane_bits<ND>(c, k);
        const int basek = base + (ND == 3 ? PS * (k - 2) : 0);
        const double z0 = ez5[k];
#pragma unroll NLPS_JUNROLL_K5
        for (int j = 0; j < 5; j++) {
          const unsigned bits = (pb >> (5 * j)) & 31u;
          if (!wave_row_used(bits)) continue;
          double A0 = 0.0, R[ND], mw[5];
#pragma unroll
          for (int a = 0; a < ND; a++) R[a] = 0.0;
          masked_row(mw, c.ex, bits);
#pragma unroll
          for (int i = 0; i < 5; i++) {  // branch-free: non-members weigh 0 (their window slot exists)
            const int li = basek + (i - 2) + W * (j - 2);
            const double m0 = mw[i];
            A0 += m0;
            const double2 v01 = a2[li];
            R[0] = fma(m0, v01.x, R[0]);
            R[1] = fma(m0, v01.y, R[1]);
            if (ND == 3) R[2 % ND] = fma(m0, lds_z(az, (ND == 3) ? li : 0), R[2 % ND]);
          }
          const double w = ey5[j] * z0;
          Z = fma(w, A0, Z);
#pragma unroll
          for (int a = 0; a < ND; a++) sv[a] = fma(w, R[a], sv[a]);
        }
      }
      const double Zinv = 1.0 / Z;
      double xn[ND], dn2 = 0.0;
#pragma unroll
      for (int a = 0; a < ND; a++) {
        const double dd = dd_[a], av = sv[a] * Zinv;
        PF(P, F_ACC + a, p) = av;
        PF(P, F_VEL + a, p) = vel_[a] + gamma_nm * dt * av;
        xn[a] = PF(P, F_X + a, p) + dd;
        PF(P, F_X + a, p) = xn[a];
        const double dn = dis_[a] + dd;
        PF(P, F_DIS + a, p) = dn;
        dn2 += dsqr(dn);
      }
      // The roll of U-Verlet.c:1062-1075 costs no traffic here: F and b_e roll by renaming (the host swaps the roles of
      // their two slots after this kernel, PView::flip; the stale slot is rewritten in full by the next K3), J, kappa and
      // eps-bar were written to their n slots by K3 in the first place (stress_update LAZY).
      if (SEARCH) {
        I0n = c.I0;
        if (sqrt(dn2) > 0.0) {  // norm__MatrixLib__(dis_p,2) > 0, LME.c:924
          I0n = closest_node_update<ND>(g, ks.rank1, xn, c.I0);
        }
        P.I0n[p] = I0n;  // (P.I0 stays the node of the last search until the next one adopts this: downloads see it)
        binned = true;
      }
    } while (false);
    if (SEARCH && ks.bin) {
      bin_particle<ND>(P, g, ks.tc, binned ? p : P.np, I0n, binned);
      if (binned && P.tile[p] >= 0) N.seed[I0n] = 1;
    }
  }
}

template <int ND, int LAW, bool SEARCH = false>
__global__ __launch_bounds__(K5_BLK) void k5_tile(PView P, GridD g, NView N, TileD td, double dt, double gamma_nm, K5Search ks) {
  __shared__ __attribute__((aligned(16))) double axy[2 * TileCfg<ND>::NW];
  __shared__ double az[(ND == 3) ? TileCfg<ND>::NW : 1];
  TileWork tw;
  if (!tile_work_item<K5_SPLIT>(td, tw)) return;
  k5_body<ND, LAW, SEARCH>(P, g, N, td, dt, gamma_nm, ks, tw, axy, az, nullptr, nullptr);
}

template <int ND, int LAW>
__global__ __launch_bounds__(K5_BLK) void k5_tile_lazy(PView P, GridD g, NView N, TileD td, double dt, double gamma_nm, K5Search ks,
                                                       LazyNodal ln, int* __restrict__ gstatus) {
  __shared__ __attribute__((aligned(16))) double axy[2 * TileCfg<ND>::NW];
  __shared__ double az[(ND == 3) ? TileCfg<ND>::NW : 1];
  TileWork tw;
  if (!tile_work_item<K5_SPLIT>(td, tw)) return;
  k5_body<ND, LAW, true>(P, g, N, td, dt, gamma_nm, ks, tw, axy, az, &ln.fs, gstatus);
}

// ------------------------------------------------------------------------------------------------
// Level-B stage kernels in tile form (the implicit driver's stage functions)
// ------------------------------------------------------------------------------------------------

// Z^-1 alone (rows of masked ex, then ey, ez)
template <int ND>
__device__ __forceinline__ double lme_zinv(const Lme<ND>& c) {
  NLPS_YZ_LOCALS(c);
  (void)ly5;
  (void)lz5;
  double Z = 0.0;
#pragma unroll 1
  for (int k = 0; k < Lme<ND>::KN; k++) {
    const unsigned pb = plane_bits<ND>(c, k);
    double P0 = 0.0;
#pragma unroll
    for (int j = 0; j < 5; j++) {
      const unsigned bits = (pb >> (5 * j)) & 31u;
      double A0 = 0.0;
#pragma unroll
      for (int i = 0; i < 5; i++) A0 += masked_weight(c.ex[i], bits, i);
      P0 = fma(ey5[j], A0, P0);
    }
    Z = fma(ez5[k], P0, Z);
  }
  return 1.0 / Z;
}

// __compute_nodal_lumped_mass (MODE 0, NF = 1) and the accumulation of __get_nodal_field_n
// (MODE 1, NF = 2d: m N v, m N a) into out[nnodes][NF]        (U-Newmark-beta.c:528-597, 615-696)
template <int ND, int MODE>
__global__ __launch_bounds__(BLK) void kb_p2g_tile(PView P, GridD g, TileD td, double* __restrict__ out) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS, NW = TileCfg<ND>::NW, KN = Lme<ND>::KN;
  constexpr int NF = MODE == 0 ? 1 : 2 * ND;
  __shared__ double acc[NF * NW];
  const int wb = td.range[0] + (int)blockIdx.x;
  if (wb >= td.range[1]) return;
  const int tile = td.work[0][wb].x;
  const int cnt = td.count[tile];
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx = threadIdx.x; idx < NW * NF; idx += BLK) acc[idx] = 0.0;
  __syncthreads();
  const int start = td.start[tile];
  for (int s = threadIdx.x; s < cnt; s += BLK) {
    const int p = td.order_m[start + s];
    Lme<ND> c;
    double lam[ND], beta;
    if (!load_lme<ND>(P, g, p, c, lam, beta)) continue;
    const int base = window_base<ND>(c.ijk, w0);
    double vals[NF];
    if (MODE == 0) vals[0] = 1.0;
    else {
#pragma unroll
      for (int a = 0; a < ND; a++) {
        vals[a % NF] = PF(P, F_VEL + a, p);
        vals[(ND + a) % NF] = PF(P, F_ACC + a, p);
      }
    }
    const double mz = PF(P, F_MASS, p) * lme_zinv<ND>(c);
    NLPS_YZ_LOCALS(c);
    (void)ly5;
    (void)lz5;
#pragma unroll 1
    for (int k = 0; k < KN; k++) {
      const unsigned pb = plane_bits<ND>(c, k);
      const double wz = mz * ez5[k];
      const int basek = base + (ND == 3 ? PS * (k - 2) : 0);
#pragma unroll 1
      for (int j = 0; j < 5; j++) {
        const unsigned bits = (pb >> (5 * j)) & 31u;
        const double w = wz * ey5[j];
#pragma unroll
        for (int i = 0; i < 5; i++)
          if ((bits >> i) & 1u) {
            const int li = basek + (i - 2) + W * (j - 2);
            const double v0 = w * c.ex[i];
#pragma unroll
            for (int f = 0; f < NF; f++) atomicAdd(&acc[f * NW + li], v0 * vals[f]);
          }
      }
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < NW * NF; q += BLK) {
    int f = q % NF, idx = q / NF;
    double v = acc[f * NW + idx];
    if (v != 0.0) {
      bool in;
      int node = window_node<ND>(g, w0, idx, in);
      if (in) atomic_add_f64(out + (size_t)node * NF + f, v);
    }
  }
}

// __nodal_internal_forces (U-Newmark-beta.c:1257-1374): +V0 tau (DF^-T grad N) from the stored tau, DF
template <int ND>
__global__ __launch_bounds__(BLK) void kb_fint_tile(PView P, GridD g, TileD td, double* __restrict__ force,
                                                    int* __restrict__ gstatus) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS, NW = TileCfg<ND>::NW, KN = Lme<ND>::KN;
  __shared__ double fac[ND * NW];
  const int wb = td.range[0] + (int)blockIdx.x;
  if (wb >= td.range[1]) return;
  const int tile = td.work[0][wb].x;
  const int cnt = td.count[tile];
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx = threadIdx.x; idx < NW * ND; idx += BLK) fac[idx] = 0.0;
  __syncthreads();
  const int start = td.start[tile];
  for (int s = threadIdx.x; s < cnt; s += BLK) {
    const int p = td.order_m[start + s];
    Lme<ND> c;
    double lam[ND], beta;
    if (!load_lme<ND>(P, g, p, c, lam, beta)) continue;
    const int base = window_base<ND>(c.ijk, w0);
    double Zinv, r[ND], J[ND * ND], Jm1[ND * ND], tau[ND * ND], DF[ND * ND], B[ND * ND], z;
    lme_moments_h<ND>(c, Zinv, r, J);
    load_block<ND>(P, F_TAU, p, tau, z);
    load_block<ND>(P, F_DF, p, DF, z);
    if (!(inverse<ND>(Jm1, J) && force_operator<ND>(B, tau, DF, Jm1, PF(P, F_VOL0, p), 1.0))) {
      atomicOr(&P.status[p], ST_JACOBIAN);
      atomicOr(gstatus, ST_JACOBIAN);
      continue;
    }
    NLPS_YZ_LOCALS(c);
#pragma unroll 1
    for (int k = 0; k < KN; k++) {
      const unsigned pb = plane_bits<ND>(c, k);
      const int basek = base + (ND == 3 ? PS * (k - 2) : 0);
      const double wz = Zinv * ez5[k];
      const double lzk = lz5[k];
#pragma unroll 1
      for (int j = 0; j < 5; j++) {
        const unsigned bits = (pb >> (5 * j)) & 31u;
        const double w = wz * ey5[j];
        double cr[ND];
#pragma unroll
        for (int a = 0; a < ND; a++)
          cr[a] = (ND == 3) ? fma(B[a * ND + 1], ly5[j], B[a * ND + (2 % ND)] * lzk) : B[a * ND + 1] * ly5[j];
#pragma unroll
        for (int i = 0; i < 5; i++)
          if ((bits >> i) & 1u) {
            const int li = basek + (i - 2) + W * (j - 2);
            const double we = w * c.ex[i];
#pragma unroll
            for (int a = 0; a < ND; a++) atomicAdd(&fac[a * NW + li], we * fma(B[a * ND + 0], c.lx[i], cr[a]));
          }
      }
    }
  }
  __syncthreads();
  for (int qq = threadIdx.x; qq < NW * ND; qq += BLK) {
    int f = qq % ND, idx = qq / ND;
    double v = fac[f * NW + idx];
    if (v != 0.0) {
      bool in;
      int node = window_node<ND>(g, w0, idx, in);
      if (in) atomic_add_f64(force + (size_t)node * ND + f, v);
    }
  }
}

// __update_particles_kinetics_FLIP_PIC (U-Newmark-beta.c:1993-2072): gathers 4 nodal vectors
// [nnodes][ND] (dU, Un_dt, dU_dt, dU_dt2) through one AoS window of 4d doubles per node
template <int ND>
__global__ __launch_bounds__(BLK) void kb_kinetics_tile(PView P, GridD g, TileD td, double alpha_blend,
                                                        const double* __restrict__ dU, const double* __restrict__ Un_dt,
                                                        const double* __restrict__ dU_dt,
                                                        const double* __restrict__ dU_dt2, int quasi_static) {
  constexpr int W = TileCfg<ND>::W, PS = TileCfg<ND>::PS, NW = TileCfg<ND>::NW, KN = Lme<ND>::KN;
  constexpr int NV = 4 * ND, NP = NV / 2;
  __shared__ __attribute__((aligned(16))) double win[NW * NV];
  const int wb = td.range[0] + (int)blockIdx.x;
  if (wb >= td.range[1]) return;
  const int tile = td.work[0][wb].x;
  const int cnt = td.count[tile];
  int w0[3];
  tile_origin<ND>(td, tile, w0);
  for (int idx = threadIdx.x; idx < NW; idx += BLK) {
    bool in;
    int node = window_node<ND>(g, w0, idx, in);
#pragma unroll
    for (int a = 0; a < ND; a++) {
      const size_t o = (size_t)node * ND + a;
      win[idx * NV + a] = in ? dU[o] : 0.0;
      win[idx * NV + ND + a] = in ? Un_dt[o] : 0.0;
      win[idx * NV + 2 * ND + a] = in ? dU_dt[o] : 0.0;
      win[idx * NV + 3 * ND + a] = in ? dU_dt2[o] : 0.0;
    }
  }
  __syncthreads();
  const double2* win2 = reinterpret_cast<const double2*>(win);
  const int start = td.start[tile];
  for (int s = threadIdx.x; s < cnt; s += BLK) {
    const int p = td.order_m[start + s];
    Lme<ND> c;
    double lam[ND], beta;
    if (!load_lme<ND>(P, g, p, c, lam, beta)) continue;
    const int base = window_base<ND>(c.ijk, w0);
    NLPS_YZ_LOCALS(c);
    (void)ly5;
    (void)lz5;
    double Z = 0.0, sv[NV];
#pragma unroll
    for (int a = 0; a < NV; a++) sv[a] = 0.0;
#pragma unroll 1
    for (int k = 0; k < KN; k++) {
      const unsigned pb = plane_bits<ND>(c, k);
      const int basek = base + (ND == 3 ? PS * (k - 2) : 0);
      const double z0 = ez5[k];
#pragma unroll 1
      for (int j = 0; j < 5; j++) {
        const unsigned bits = (pb >> (5 * j)) & 31u;
        double A0 = 0.0, R[NV];
#pragma unroll
        for (int a = 0; a < NV; a++) R[a] = 0.0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
          const int li = basek + (i - 2) + W * (j - 2);
          const double m0 = masked_weight(c.ex[i], bits, i);
          A0 += m0;
#pragma unroll
          for (int q = 0; q < NP; q++) {
            const double2 v = win2[li * NP + q];
            R[2 * q] = fma(m0, v.x, R[2 * q]);
            R[2 * q + 1] = fma(m0, v.y, R[2 * q + 1]);
          }
        }
        const double w = ey5[j] * z0;
        Z = fma(w, A0, Z);
#pragma unroll
        for (int a = 0; a < NV; a++) sv[a] = fma(w, R[a], sv[a]);
      }
    }
    const double Zinv = 1.0 / Z, beta_blend = 1 - alpha_blend;
#pragma unroll
    for (int a = 0; a < ND; a++) {
      const double du = sv[a] * Zinv, vn = sv[ND + a] * Zinv, dv = sv[2 * ND + a] * Zinv, da = sv[3 * ND + a] * Zinv;
      if (!quasi_static) {  // U-Static.c:1380-1470 touches dis and x_GC only
        PF(P, F_ACC + a, p) = PF(P, F_ACC + a, p) + da;
        PF(P, F_VEL + a, p) = alpha_blend * PF(P, F_VEL + a, p) + (dv + beta_blend * vn);
      }
      PF(P, F_DIS + a, p) = PF(P, F_DIS + a, p) + du;
      PF(P, F_X + a, p) = PF(P, F_X + a, p) + du;
    }
  }
}
