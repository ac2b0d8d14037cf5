// nlps_tables.hpp — host-side stencil-order tables for the structured background grid.
//
// The reference keeps node neighbourhoods as linked lists whose ORDER is a by-product of how
// GramsBox builds them (InOutFun/Read_GramsBox.c:293-456: per-node element lists filled by
// ascending element index with prepend-push, Matlib/ChainOp.c:163-182; unions that prepend new
// members, ChainOp.c:275-293; GiD connectivity pushed node by node, Nodes/Read-GID-Mesh.c:406-416).
// Two results of the hot path depend on that order and must be bit-exact:
//   * get_closest_node__MeshTools__ (Nodes/Nodes-Tools.c:476-538) keeps the FIRST minimum in chain
//     order of the 1-ring  => `rank1`: position of each 3^d offset in the chain;
//   * ListNodes[p] (Nodes/LME.c:1057-1082) is the 2-ring walk, filtered, reversed => `order2`.
// On a lattice the order only depends on which neighbours exist, i.e. on the boundary class of the
// node (3 classes per axis for the 1-ring, 5 for the 2-ring); the tables are derived once per class
// by replaying the list construction on a tiny local lattice.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <vector>

namespace nlps_host {

struct Lattice {
  int nd;
  int n[3];
  int nc(int a) const { return a < nd ? n[a] - 1 : 1; }
  int id(int i, int j, int k) const { return i + n[0] * (j + n[1] * k); }
};

// a "chain" is represented front-first; prepend = insert at begin
using Chain = std::vector<int>;

inline Chain element_chain(const Lattice& L, int ci, int cj, int ck) {
  std::vector<int> file;
  int layers = L.nd == 3 ? 2 : 1;
  for (int t = 0; t < layers; t++) {
    int k = L.nd == 3 ? ck + t : 0;
    file.push_back(L.id(ci, cj, k));
    file.push_back(L.id(ci + 1, cj, k));
    file.push_back(L.id(ci + 1, cj + 1, k));
    file.push_back(L.id(ci, cj + 1, k));
  }
  return Chain(file.rbegin(), file.rend());  // each read node is prepended
}

inline bool contains(const Chain& c, int x) { return std::find(c.begin(), c.end(), x) != c.end(); }

// 1-ring of node I (self included) in chain order
inline Chain one_ring(const Lattice& L, int I) {
  int i = I % L.n[0], j = (I / L.n[0]) % L.n[1], k = I / (L.n[0] * L.n[1]);
  // elements around I in ascending index, prepended one by one => visited in descending index
  std::vector<std::array<int, 3>> elems;
  for (int dk = (L.nd == 3 ? -1 : 0); dk <= 0; dk++)
    for (int dj = -1; dj <= 0; dj++)
      for (int di = -1; di <= 0; di++) {
        int ci = i + di, cj = j + dj, ck = (L.nd == 3 ? k + dk : 0);
        if (ci < 0 || ci >= L.nc(0) || cj < 0 || cj >= L.nc(1)) continue;
        if (L.nd == 3 && (ck < 0 || ck >= L.nc(2))) continue;
        elems.push_back({ci, cj, ck});
      }
  Chain out;
  for (auto it = elems.rbegin(); it != elems.rend(); ++it)
    for (int node : element_chain(L, (*it)[0], (*it)[1], (*it)[2]))
      if (!contains(out, node)) out.insert(out.begin(), node);
  return out;
}

// 2-ring (self included) in chain order: two sweeps of "add the 1-rings of the last sweep's finds"
inline Chain two_ring(const Lattice& L, int I) {
  Chain set, search{I};
  for (int ring = 0; ring < 2; ring++) {
    Chain fresh;
    for (int s : search)
      for (int a : one_ring(L, s))
        if (!contains(set, a)) {
          set.insert(set.begin(), a);
          fresh.insert(fresh.begin(), a);
        }
    search = fresh;
  }
  return set;
}

struct StencilTables {
  // rank1[c27][o27]: chain position of 1-ring offset o (lexicographic, (di+1)+3(dj+1)+9(dk+1));
  // 255 = neighbour does not exist
  uint8_t rank1[27][27];
  // order2[c125][q]: q-th node of the 2-ring walk as lexicographic 5^d offset index
  // ((di+2)+5(dj+2)+25(dk+2)); count2[c125] entries
  uint8_t order2[125][125];
  uint8_t count2[125];
  double h_avg1[27];  // mean 1-ring neighbour distance per 1-ring class, for h = 1
};

inline int class3(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 2 : 1); }
inline int class5(int i, int n) { return i < 2 ? i : (i > n - 3 ? 4 - (n - 1 - i) : 2); }

// Works for any lattice with >= 5 nodes per used axis (the classes are then well defined).
inline StencilTables build_tables(int nd) {
  StencilTables T;
  for (auto& r : T.rank1)
    for (auto& v : r) v = 255;
  for (auto& r : T.order2)
    for (auto& v : r) v = 255;
  for (auto& v : T.count2) v = 0;
  Lattice L{nd, {7, 7, nd == 3 ? 7 : 1}};
  const int rep3[3] = {0, 3, 6};
  const int rep5[5] = {0, 1, 3, 5, 6};
  for (int cz = 0; cz < (nd == 3 ? 3 : 1); cz++)
    for (int cy = 0; cy < 3; cy++)
      for (int cx = 0; cx < 3; cx++) {
        int i = rep3[cx], j = rep3[cy], k = nd == 3 ? rep3[cz] : 0;
        int cls = cx + 3 * cy + 9 * (nd == 3 ? cz : 1);
        Chain c = one_ring(L, L.id(i, j, k));
        double avg = 0.0;
        int cnt = 0;
        for (size_t q = 0; q < c.size(); q++) {
          int J = c[q];
          int di = J % L.n[0] - i, dj = (J / L.n[0]) % L.n[1] - j, dk = J / (L.n[0] * L.n[1]) - k;
          T.rank1[cls][(di + 1) + 3 * (dj + 1) + 9 * (dk + 1)] = (uint8_t)q;
          if (di || dj || dk) {
            double a = 0.0;  // norm__MatrixLib__: pow(sum DSQR, 0.5), MatrixOp.c:843-870
            a += (double)(di * di);
            a += (double)(dj * dj);
            if (nd == 3) a += (double)(dk * dk);
            avg += std::pow(a, 0.5);
            cnt++;
          }
        }
        T.h_avg1[cls] = avg / (double)cnt;
      }
  for (int cz = 0; cz < (nd == 3 ? 5 : 1); cz++)
    for (int cy = 0; cy < 5; cy++)
      for (int cx = 0; cx < 5; cx++) {
        int i = rep5[cx], j = rep5[cy], k = nd == 3 ? rep5[cz] : 0;
        int cls = cx + 5 * cy + 25 * (nd == 3 ? cz : 2);
        Chain c = two_ring(L, L.id(i, j, k));
        T.count2[cls] = (uint8_t)c.size();
        for (size_t q = 0; q < c.size(); q++) {
          int J = c[q];
          int di = J % L.n[0] - i, dj = (J / L.n[0]) % L.n[1] - j, dk = J / (L.n[0] * L.n[1]) - k;
          T.order2[cls][q] = (uint8_t)((di + 2) + 5 * (dj + 2) + 25 * (nd == 3 ? dk + 2 : 0));
        }
      }
  return T;
}

}  // namespace nlps_host
