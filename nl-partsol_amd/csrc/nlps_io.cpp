// nlps_io.cpp — input side of the path, host only: the GiD ASCII meshes the reference reads
// (Nodes/Read-GID-Mesh.c), the recognition of the structured GramsBox lattice behind a background mesh (what the
// library needs instead of GramsBox's O(N_nodes x N_elem) neighbour construction, InOutFun/Read_GramsBox.c:293-456,
// and of initialize__LME__'s element search, Nodes/LME.c:63-115), and the particles a body mesh generates
// (InOutFun/Analysis/Generate-One-Phase-Analysis.c:569-625, Particles/Particles-Tools.c:8-28, Nodes/T3.c:337-440,
// Nodes/Q4.c:342-452, Nodes/T4.c:322-420, Nodes/H8.c:389-575).  No GPU, no torch; entry points declared in include/nlps_gpu.h.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "../../include/nlps_gpu.h"

namespace {

thread_local std::string io_error;

int fail(const std::string& msg) {
  io_error = msg;
  return 1;
}

// parse(), InOutFun/Parser.c:6-40: strtok on the delimiter set of Read-GID-Mesh.c:21-31 (the _WIN32 set, a superset
// of the __linux__ one, so that files with CR LF line ends read the same).
int split(char* line, std::vector<char*>& words) {
  words.clear();
  char* save = nullptr;  // strtok_r: the error string is per thread, the tokenizer state has to be as well
  for (char* p = strtok_r(line, " \r\n\t", &save); p; p = strtok_r(nullptr, " \r\n\t", &save)) words.push_back(p);
  return (int)words.size();
}

struct LineReader {
  FILE* f;
  std::vector<char> buf;
  explicit LineReader(const char* path) : f(fopen(path, "r")), buf(1 << 16) {}
  ~LineReader() {
    if (f) fclose(f);
  }
  bool next() { return f && fgets(buf.data(), (int)buf.size(), f) != nullptr; }
};

enum Block { NONE, COORDS, ELEMS };

// Block tracking shared by the three passes of the reference (Read_Mesh_Information :225-300,
// Fill_Coordinates :304-352, Fill_Linear_Conectivity :356-408).
Block track(Block cur, const std::vector<char*>& w) {
  if (w.size() == 1 && !strcmp(w[0], "Coordinates")) return COORDS;
  if (w.size() == 1 && !strcmp(w[0], "Elements")) return ELEMS;
  if (w.size() == 2 && !strcmp(w[0], "End") && (!strcmp(w[1], "Coordinates") || !strcmp(w[1], "Elements"))) return NONE;
  return cur;
}

// ---- reference elements -------------------------------------------------------------------------------------

// N__Q4__, Nodes/Q4.c:112-124
void shape_q4(const double* xi, double* N) {
  N[0] = 0.25 * (1 - xi[0]) * (1 - xi[1]);
  N[1] = 0.25 * (1 + xi[0]) * (1 - xi[1]);
  N[2] = 0.25 * (1 + xi[0]) * (1 + xi[1]);
  N[3] = 0.25 * (1 - xi[0]) * (1 + xi[1]);
}
// dN_Ref__Q4__, Nodes/Q4.c:129-155
void dshape_q4(const double* xi, double (*dN)[3]) {
  dN[0][0] = -0.25 * (1 - xi[1]);
  dN[0][1] = -0.25 * (1 - xi[0]);
  dN[1][0] = +0.25 * (1 - xi[1]);
  dN[1][1] = -0.25 * (1 + xi[0]);
  dN[2][0] = +0.25 * (1 + xi[1]);
  dN[2][1] = +0.25 * (1 + xi[0]);
  dN[3][0] = -0.25 * (1 + xi[1]);
  dN[3][1] = +0.25 * (1 - xi[0]);
}
double clamp(double hi, double v) { return std::fmin(hi, std::fmax(0.0, v)); }
// N__H8__, Nodes/H8.c:97-128 (with its DMIN/DMAX clamps)
void shape_h8(const double* x, double* N) {
  static const int s[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
  for (int a = 0; a < 8; a++) N[a] = 0.125 * clamp(8, (1. + s[a][0] * x[0]) * (1. + s[a][1] * x[1]) * (1. + s[a][2] * x[2]));
}
// dN_Ref__H8__, Nodes/H8.c:133-198
void dshape_h8(const double* x, double (*dN)[3]) {
  static const int s[8][3] = {{-1, -1, -1}, {1, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {-1, -1, 1}, {1, -1, 1}, {1, 1, 1}, {-1, 1, 1}};
  for (int a = 0; a < 8; a++) {
    dN[a][0] = s[a][0] * 0.125 * clamp(4, (1. + s[a][1] * x[1]) * (1. + s[a][2] * x[2]));
    dN[a][1] = s[a][1] * 0.125 * clamp(4, (1. + s[a][0] * x[0]) * (1. + s[a][2] * x[2]));
    dN[a][2] = s[a][2] * 0.125 * clamp(4, (1. + s[a][0] * x[0]) * (1. + s[a][1] * x[1]));
  }
}

// N__T3__ / dN_Ref__T3__ (Nodes/T3.c:100-132) and N__T4__ / dN_Ref__T4__ (Nodes/T4.c:96-128)
void shape_t3(const double* x, double* N) {
  N[0] = 1 - x[0] - x[1];
  N[1] = x[0];
  N[2] = x[1];
}
void dshape_t3(double (*dN)[3]) {
  dN[0][0] = -1, dN[0][1] = -1;
  dN[1][0] = +1, dN[1][1] = +0;
  dN[2][0] = +0, dN[2][1] = +1;
}
void shape_t4(const double* x, double* N) {
  N[0] = x[0];
  N[1] = x[1];
  N[2] = x[2];
  N[3] = 1. - x[0] - x[1] - x[2];
}
void dshape_t4(double (*dN)[3]) {
  for (int a = 0; a < 4; a++)
    for (int j = 0; j < 3; j++) dN[a][j] = a < 3 ? (a == j ? 1. : 0.) : -1.;
}

enum Elem { Q4, H8, T3, T4 };

// I3__MatrixLib__, Matlib/MatrixOp.c:290-300
double det(const double F[3][3], int nd) {
  if (nd == 2) return F[0][0] * F[1][1] - F[0][1] * F[1][0];
  return F[0][0] * F[1][1] * F[2][2] - F[0][0] * F[1][2] * F[2][1] + F[0][1] * F[1][2] * F[2][0] -
         F[0][1] * F[1][0] * F[2][2] + F[0][2] * F[1][0] * F[2][1] - F[0][2] * F[1][1] * F[2][0];
}

// volume__Q4__ (Nodes/Q4.c:493-530) / volume__H8__ (Nodes/H8.c:643-690): sum of |det F_ref| over the 2^d
// Gauss points 0.5773502692 (the reference's 10-digit constant), unit weights; volume__T3__ (T3.c:506-540): three
// points of weight 1/6; volume__T4__ (T4.c:488-524): four points of weight 1/24 (7-digit coordinates).
// F_ref = sum_I x_I (x) dN_I (F_Ref__Q4__, Q4.c:160-208, and its siblings).
double element_volume(Elem el, int nd, int npe, const double (*X)[3]) {
  // (the simplices' quadrature points do not matter: their dN, and with it F_ref, is constant over the element)
  const double g = 0.577350269200000;
  const int nq = el == T3 ? 3 : el == T4 ? 4 : (1 << nd);
  const double w = el == T3 ? 1. / 6. : el == T4 ? 1. / 24. : 1.0;
  double vol = 0.0;
  for (int q = 0; q < nq; q++) {
    double xi[3] = {(q & 1) ? g : -g, (q & 2) ? g : -g, (q & 4) ? g : -g};
    double dN[8][3];
    if (el == Q4) dshape_q4(xi, dN);
    else if (el == H8) dshape_h8(xi, dN);
    else if (el == T3) dshape_t3(dN);
    else dshape_t4(dN);
    double F[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int I = 0; I < npe; I++)
      for (int i = 0; i < nd; i++)
        for (int j = 0; j < nd; j++) F[i][j] += X[I][i] * dN[I][j];
    vol += std::fabs(det(F, nd)) * w;
  }
  return vol;
}

// natural coordinates of the particles of one element: element_to_particles__Q4__ (Q4.c:353-416: 1, 4, 5, 9) and
// element_to_particles__H8__ (H8.c:399-548: 1, 8, 27), with the reference's truncated constants
bool particle_sites(Elem el, int nd, int gp, std::vector<double>& xi) {
  xi.assign((size_t)gp * 3, 0.0);
  auto set = [&](int j, double a, double b, double c) {
    xi[3 * j] = a;
    xi[3 * j + 1] = b;
    xi[3 * j + 2] = c;
  };
  if (el == T3) {  // element_to_particles__T3__, T3.c:348-412 (one particle sits on the first vertex, as there)
    const double a = 0.16666666666, b = 0.66666666666, c = 0.33333333333;
    const double n1 = 0.11111111111, n2 = 0.22222222222, n4 = 0.44444444444, n5 = 0.55555555555, n7 = 0.77777777777;
    switch (gp) {
      case 1: return true;
      case 3: set(0, a, a, 0), set(1, b, a, 0), set(2, a, b, 0); return true;
      case 4: set(0, a, a, 0), set(1, b, a, 0), set(2, a, b, 0), set(3, c, c, 0); return true;
      case 9:
        set(0, n1, n1, 0), set(1, n4, n1, 0), set(2, n7, n1, 0), set(3, n2, n2, 0), set(4, n5, n2, 0);
        set(5, n1, n4, 0), set(6, n4, n4, 0), set(7, n2, n5, 0), set(8, n1, n7, 0);
        return true;
      default: return false;
    }
  }
  if (el == T4) {  // element_to_particles__T4__, T4.c:333-392
    const double a = 0.138196601125010, b = 0.585410196624968;
    const double c = 0.108103018168070, d = 0.816847572980459, e = 0.445948490915965;
    switch (gp) {
      case 1: set(0, .25, .25, .25); return true;
      case 4: set(0, a, a, a), set(1, b, a, a), set(2, a, b, a), set(3, a, a, b); return true;
      case 10:
        set(0, c, c, c), set(1, d, c, c), set(2, c, d, c), set(3, c, c, d), set(4, e, c, c), set(5, e, e, c);
        set(6, c, e, c), set(7, c, c, e), set(8, e, c, e), set(9, c, e, e);
        return true;
      default: return false;
    }
  }
  if (nd == 2) {
    const double s = 1. / std::sqrt(3.0), t = 0.6666666666666;
    switch (gp) {
      case 1: return true;
      case 4: set(0, s, s, 0), set(1, s, -s, 0), set(2, -s, s, 0), set(3, -s, -s, 0); return true;
      case 5: set(0, .5, .5, 0), set(1, .5, -.5, 0), set(2, -.5, .5, 0), set(3, -.5, -.5, 0), set(4, 0, 0, 0); return true;
      case 9:
        set(0, 0, 0, 0), set(1, t, 0, 0), set(2, t, t, 0), set(3, 0, t, 0), set(4, -t, t, 0), set(5, -t, 0, 0);
        set(6, -t, -t, 0), set(7, 0, -t, 0), set(8, t, -t, 0);
        return true;
      default: return false;
    }
  }
  const double t = 0.66666666666;
  switch (gp) {
    case 1: return true;
    case 8:
      set(0, -.5, -.5, .5), set(1, .5, -.5, .5), set(2, .5, .5, .5), set(3, -.5, .5, .5);
      set(4, -.5, -.5, -.5), set(5, .5, -.5, -.5), set(6, .5, .5, -.5), set(7, -.5, .5, -.5);
      return true;
    case 27: {
      static const int ring[9][2] = {{0, 0}, {1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
      static const int level[3] = {0, 1, -1};
      for (int l = 0; l < 3; l++)
        for (int r = 0; r < 9; r++) set(9 * l + r, ring[r][0] * t, ring[r][1] * t, level[l] * t);
      return true;
    }
    default: return false;
  }
}

}  // namespace

extern "C" const char* nlps_host_io_last_error(void) { return io_error.c_str(); }

extern "C" int nlps_host_gid_mesh_info(const char* path, nlps_gid_info* info) {
  if (!path || !info) return fail("null argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  memset(info, 0, sizeof(*info));
  std::vector<char*> w;
  Block blk = NONE;
  long line_no = 0;
  while (in.next()) {
    const int nw = split(in.buf.data(), w);
    if (line_no == 0) {  // Read-GID-Mesh.c:254-268: exactly "MESH dimension d ElemType T Nnode n"
      if (!(nw == 7 && !strcmp(w[0], "MESH") && !strcmp(w[1], "dimension") && !strcmp(w[3], "ElemType") &&
            !strcmp(w[5], "Nnode")))
        return fail("The header of a GID has a non-suported structure");
      info->ndim = atoi(w[2]);
      snprintf(info->elem_type, sizeof(info->elem_type), "%s", w[4]);
      info->nodes_per_elem = atoi(w[6]);
      if (info->ndim != 2 && info->ndim != 3) return fail("mesh dimension must be 2 or 3");
      if (info->nodes_per_elem < 1 || info->nodes_per_elem > 27) return fail("Nnode out of range");
    } else {
      const Block nb = track(blk, w);
      if (nb == blk && blk == COORDS) {
        // the reference counts every line of the block (:272-283) but only reads those with 4 words (:334-340):
        // anything else would leave it with uninitialised nodes, so it is refused here
        if (nw != 4) return fail("line " + std::to_string(line_no + 1) + ": a Coordinates line needs 4 words (id x y z)");
        info->nnodes++;
      } else if (nb == blk && blk == ELEMS) {
        if (nw != info->nodes_per_elem + 1)
          return fail("line " + std::to_string(line_no + 1) + ": an Elements line needs Nnode + 1 words");
        info->nelem++;
      }
      blk = nb;
    }
    line_no++;
  }
  if (line_no == 0) return fail("empty mesh file");
  if (blk != NONE) return fail("unterminated Coordinates / Elements block");
  return 0;
}

extern "C" int nlps_host_gid_mesh_read(const char* path, const nlps_gid_info* info, double* coords, int* conn) {
  if (!path || !info || !coords || !conn) return fail("null argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  std::vector<char*> w;
  Block blk = NONE;
  int node = 0, elem = 0;
  const int nd = info->ndim, npe = info->nodes_per_elem;
  long line_no = 0;
  while (in.next()) {
    const int nw = split(in.buf.data(), w);
    if (line_no++ == 0) continue;
    const Block nb = track(blk, w);
    if (nb == blk && blk == COORDS && nw == 4) {  // Fill_Coordinates :334-340: words 1..d, the id is not used
      if (node >= info->nnodes) return fail("more nodes than nlps_host_gid_mesh_info counted");
      for (int j = 0; j < nd; j++) coords[(size_t)node * nd + j] = atof(w[j + 1]);
      node++;
    } else if (nb == blk && blk == ELEMS && nw == npe + 1) {
      // Fill_Linear_Conectivity :392-398 pushes (prepends, ChainOp.c:163-182) the nodes in file order: the chain,
      // and with it every later walk over the element, runs in REVERSED file order
      if (elem >= info->nelem) return fail("more elements than nlps_host_gid_mesh_info counted");
      for (int j = 0; j < npe; j++) {
        const int id = atoi(w[j + 1]) - 1;
        if (id < 0 || id >= info->nnodes) return fail("element " + std::to_string(elem) + ": node id out of range");
        conn[(size_t)elem * npe + (npe - 1 - j)] = id;
      }
      elem++;
    }
    blk = nb;
  }
  if (node != info->nnodes || elem != info->nelem) return fail("the file changed between info and read");
  return 0;
}

extern "C" int nlps_host_lattice_from_nodes(int ndim, int nnodes, const double* coords, double* h_out, int n[3],
                                            double origin[3], int* canon) {
  if ((ndim != 2 && ndim != 3) || nnodes < (1 << ndim) || !coords || !h_out || !n || !origin)
    return fail("bad argument");
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int a = 0; a < ndim; a++) lo[a] = hi[a] = coords[a];
  for (int i = 1; i < nnodes; i++)
    for (int a = 0; a < ndim; a++) {
      const double v = coords[(size_t)i * ndim + a];
      lo[a] = std::fmin(lo[a], v);
      hi[a] = std::fmax(hi[a], v);
    }
  // spacing: distance from node 0 to its nearest node
  double h2 = INFINITY;
  for (int i = 1; i < nnodes; i++) {
    double d2 = 0.0;
    for (int a = 0; a < ndim; a++) {
      const double d = coords[(size_t)i * ndim + a] - coords[a];
      d2 += d * d;
    }
    if (d2 > 0.0 && d2 < h2) h2 = d2;
  }
  if (!(h2 < INFINITY)) return fail("coincident nodes");
  const double h = std::sqrt(h2);
  long long total = 1;
  n[0] = n[1] = n[2] = 1;
  origin[0] = origin[1] = origin[2] = 0.0;
  for (int a = 0; a < ndim; a++) {
    n[a] = (int)std::llround((hi[a] - lo[a]) / h) + 1;
    origin[a] = lo[a];
    total *= n[a];
  }
  if (total != nnodes) return fail("the background mesh is not a structured lattice of one spacing (node count)");
  std::vector<unsigned char> seen((size_t)nnodes, 0);
  for (int i = 0; i < nnodes; i++) {
    long long id = 0, stride = 1;
    for (int a = 0; a < ndim; a++) {
      const double t = (coords[(size_t)i * ndim + a] - lo[a]) / h;
      const long long k = std::llround(t);
      if (std::fabs(t - (double)k) > 1e-6 || k < 0 || k >= n[a])
        return fail("node " + std::to_string(i) + " is off the lattice");
      id += stride * k;
      stride *= n[a];
    }
    if (seen[(size_t)id]) return fail("two nodes on one lattice site");
    seen[(size_t)id] = 1;
    if (canon) canon[i] = (int)id;
  }
  *h_out = h;
  return 0;
}

extern "C" int nlps_host_particles_from_mesh(const nlps_gid_info* info, const double* coords, const int* conn,
                                             int gp_per_elem, double thickness, double* x, double* vol0) {
  if (!info || !coords || !conn || !x || !vol0) return fail("null argument");
  const int nd = info->ndim, npe = info->nodes_per_elem;
  Elem el;
  if (nd == 2 && npe == 4 && !strcmp(info->elem_type, "Quadrilateral")) el = Q4;
  else if (nd == 3 && npe == 8 && !strcmp(info->elem_type, "Hexahedra")) el = H8;
  else if (nd == 2 && npe == 3 && !strcmp(info->elem_type, "Triangle")) el = T3;
  else if (nd == 3 && npe == 4 && !strcmp(info->elem_type, "Tetrahedra")) el = T4;
  else return fail("particles are generated from linear Triangle / Quadrilateral / Tetrahedra / Hexahedra body meshes");
  std::vector<double> xi;
  if (!particle_sites(el, nd, gp_per_elem, xi)) return fail("Wrong number of particles per element");
  for (int e = 0; e < info->nelem; e++) {
    double X[8][3];
    for (int k = 0; k < npe; k++)
      for (int l = 0; l < 3; l++) X[k][l] = l < nd ? coords[(size_t)conn[(size_t)e * npe + k] * nd + l] : 0.0;
    double vol = element_volume(el, nd, npe, X);
    if (nd == 2) vol *= thickness;  // Thickness_Plain_Stress, Q4.c:524
    if (vol <= 0.0) return fail("Element with negative volume");  // Generate-One-Phase-Analysis.c:591-595
    for (int j = 0; j < gp_per_elem; j++) {
      double N[8];
      if (el == Q4) shape_q4(&xi[3 * j], N);
      else if (el == H8) shape_h8(&xi[3 * j], N);
      else if (el == T3) shape_t3(&xi[3 * j], N);
      else shape_t4(&xi[3 * j], N);
      const size_t p = (size_t)e * gp_per_elem + j;
      for (int l = 0; l < nd; l++) x[p * nd + l] = 0.0;
      for (int k = 0; k < npe; k++)  // Q4.c:433-442 / H8.c:560-569
        for (int l = 0; l < nd; l++) x[p * nd + l] += N[k] * X[k][l];
      vol0[p] = vol / gp_per_elem;  // Generate-One-Phase-Analysis.c:607-612
    }
  }
  return 0;
}

// ---- output side: the particle file of particle_results_vtk__InOutFun__ (InOutFun/Outputs/WriteVtk.c:95-266), legacy
// ASCII VTK with every number as %.20g, one vertex cell per particle.  Blocks appear in the reference's order; a block
// whose array is NULL (or whose flag is off) is left out, as with the reference's Out_* switches.
namespace {
void vtk_vectors(FILE* f, const char* name, const double* v, int np, int nd) {  // vtk_Out_vel & co., :516-566
  fprintf(f, "VECTORS %s double \n", name);
  for (int i = 0; i < np; i++) {
    for (int j = 0; j < 3; j++) fprintf(f, "%.20g ", j < nd ? v[(size_t)i * nd + j] : 0.0);
    fprintf(f, "\n");
  }
}
void vtk_scalars(FILE* f, const char* name, const double* v, int np) {  // vtk_Out_mass & co., :466-481
  fprintf(f, "SCALARS %s double \n", name);
  fprintf(f, "LOOKUP_TABLE default \n");
  for (int i = 0; i < np; i++) fprintf(f, "%.20g \n", v[i]);
}
void vtk_integers(FILE* f, const char* name, const int* v, int np) {  // vtk_Out_nodal_idx, :496-512
  fprintf(f, "SCALARS %s integer \n", name);
  fprintf(f, "LOOKUP_TABLE default \n");
  for (int i = 0; i < np; i++) fprintf(f, "%i \n", v[i]);
}
// vtk_Out_Stress (:570-590, the 2-D build puts the out-of-plane component, slot 4, at zz) and
// vtk_Out_Deformation_Gradient (:662-680, zeros outside the d x d block)
void vtk_tensors(FILE* f, const char* name, const double* t, int np, int nd, bool zz_from_slot4) {
  const int T = nd == 2 ? 5 : 9;
  fprintf(f, "TENSORS %s double \n", name);
  for (int i = 0; i < np; i++) {
    for (int j = 0; j < 3; j++) {
      for (int k = 0; k < 3; k++) {
        double v = 0.0;
        if (j < nd && k < nd) v = t[(size_t)i * T + j * nd + k];
        else if (zz_from_slot4 && j == 2 && k == 2) v = t[(size_t)i * T + 4];
        fprintf(f, "%.20g ", v);
      }
      fprintf(f, "\n");
    }
    fprintf(f, "\n");
  }
}
}  // namespace

extern "C" int nlps_host_write_particles_vtk(const char* path, int results_time_step, int ndim, int np,
                                             const nlps_vtk_fields* a, int flags) {
  if (!path || !a || !a->x || (ndim != 2 && ndim != 3) || np < 0) return fail("bad argument");
  FILE* f = fopen(path, "w");
  if (!f) return fail(std::string("cannot open ") + path);
  fprintf(f, "# vtk DataFile Version 3.0 \n");
  fprintf(f, "Results time step %i \n", results_time_step);
  fprintf(f, "ASCII \n");
  fprintf(f, "DATASET UNSTRUCTURED_GRID \n");
  fprintf(f, "POINTS %i double \n", np);
  for (int i = 0; i < np; i++) {
    for (int j = 0; j < 3; j++) fprintf(f, "%.20g ", j < ndim ? a->x[(size_t)i * ndim + j] : 0.0);
    fprintf(f, "\n");
  }
  fprintf(f, "CELLS %i %i \n", np, 2 * np);
  for (int i = 0; i < np; i++) fprintf(f, "%i %i \n", 1, i);
  fprintf(f, "CELL_TYPES %i \n", np);
  for (int i = 0; i < np; i++) fprintf(f, "%i \n", 1);
  fprintf(f, "POINT_DATA %i \n", np);
  if (flags & NLPS_VTK_X_GC) vtk_vectors(f, "X_GC", a->x, np, ndim);
  fprintf(f, "CELL_DATA %i \n", np);
  if (a->mass) vtk_scalars(f, "MASS", a->mass, np);
  if (a->rho) vtk_scalars(f, "DENSITY", a->rho, np);
  if (a->I0) vtk_integers(f, "ELEM_i", a->I0, np);
  if (a->matidx) vtk_integers(f, "MatIdx", a->matidx, np);
  if (a->vel) vtk_vectors(f, "VELOCITY", a->vel, np, ndim);
  if (a->acc) vtk_vectors(f, "ACCELERATION", a->acc, np, ndim);
  if (a->dis) vtk_vectors(f, "DISPLACEMENT", a->dis, np, ndim);
  if (a->stress) vtk_tensors(f, "STRESS", a->stress, np, ndim, true);
  if (a->stress && (flags & NLPS_VTK_P)) {  // vtk_Out_Stress_P, :594-612
    const int T = ndim == 2 ? 5 : 9;
    fprintf(f, "SCALARS P double \n");
    fprintf(f, "LOOKUP_TABLE default \n");
    for (int i = 0; i < np; i++) {
      const double* s = a->stress + (size_t)i * T;
      const double p = ndim == 2 ? (1.0 / 3.0) * (s[0] + s[3] + s[4]) : (1.0 / 3.0) * (s[0] + s[4] + s[8]);
      fprintf(f, "%.20g \n", p);
    }
  }
  if (a->F_n) vtk_tensors(f, "DEFORMATION-GRADIENT", a->F_n, np, ndim, false);
  if ((flags & NLPS_VTK_ENERGY) && a->W && a->vel && a->mass) {  // :816-843
    vtk_scalars(f, "Energy-Potential", a->W, np);
    fprintf(f, "SCALARS Energy-Kinetic double \n");
    fprintf(f, "LOOKUP_TABLE default \n");
    for (int i = 0; i < np; i++) {
      double K = 0;
      for (int j = 0; j < ndim; j++) {
        const double v = a->vel[(size_t)i * ndim + j];
        K += (v == 0.0 ? 0.0 : v * v);  // DSQR, Macros.h:49-50
      }
      fprintf(f, "%.20g \n", 0.5 * K * a->mass[i]);
    }
  }
  if (a->eps) vtk_scalars(f, "EPS", a->eps, np);
  if (fclose(f)) return fail(std::string("write error on ") + path);
  return 0;
}

// The nodal file of nodal_results_vtk__InOutFun__ (WriteVtk.c:269-405): mesh in FILE numbering (coordinates as %f,
// cells with their chains, cell types only for triangles (5) and quadrilaterals (9) as in the reference), the
// active-node mask and the reactions (%.20g; an inactive node writes "0 0 0 " with the reference's trailing blank).
// active / reactions are the library's lattice-numbered arrays (nlps_gpu_download_active, nlps_gpu_explicit_nodal);
// canon[file node] = lattice node (nlps_host_lattice_from_nodes), NULL = identity.
extern "C" int nlps_host_write_nodes_vtk(const char* path, const nlps_gid_info* info, const double* coords,
                                         const int* conn, const int* canon, const unsigned char* active,
                                         const double* reactions) {
  if (!path || !info || !coords || !conn || !active || !reactions) return fail("null argument");
  FILE* f = fopen(path, "w");
  if (!f) return fail(std::string("cannot open ") + path);
  const int nd = info->ndim, nn = info->nnodes, ne = info->nelem, npe = info->nodes_per_elem;
  fprintf(f, "# vtk DataFile Version 3.0 \n");
  fprintf(f, "vtk output \n");
  fprintf(f, "ASCII \n");
  fprintf(f, "DATASET UNSTRUCTURED_GRID \n");
  fprintf(f, "POINTS %i float \n", nn);
  for (int i = 0; i < nn; i++)
    fprintf(f, "%f %f %f\n", coords[(size_t)i * nd], coords[(size_t)i * nd + 1], nd == 3 ? coords[(size_t)i * nd + 2] : 0.0);
  fprintf(f, "\n");
  fprintf(f, "CELLS %i %i \n", ne, ne + ne * npe);
  for (int e = 0; e < ne; e++) {
    fprintf(f, "%i", npe);
    for (int k = 0; k < npe; k++) fprintf(f, " %i", conn[(size_t)e * npe + k]);
    fprintf(f, "\n");
  }
  fprintf(f, "\n");
  fprintf(f, "CELL_TYPES %i \n", ne);
  const int type = (!strcmp(info->elem_type, "Triangle") && npe == 3) ? 5
                   : (!strcmp(info->elem_type, "Quadrilateral") && npe == 4) ? 9 : 0;
  if (type)
    for (int e = 0; e < ne; e++) fprintf(f, "%i \n", type);
  fprintf(f, "\n");
  fprintf(f, "POINT_DATA %i \n", nn);
  fprintf(f, "SCALARS Mask int \n");
  fprintf(f, "LOOKUP_TABLE default \n");
  for (int i = 0; i < nn; i++) fprintf(f, "%i\n", active[canon ? canon[i] : i] ? 1 : 0);
  fprintf(f, "VECTORS %s float \n", "REACTIONS");
  for (int i = 0; i < nn; i++) {
    const size_t a = (size_t)(canon ? canon[i] : i);
    if (active[a]) fprintf(f, "%.20g %.20g %.20g\n", reactions[a * nd], reactions[a * nd + 1], nd == 3 ? reactions[a * nd + 2] : 0.0);
    else fprintf(f, "%.20g %.20g %.20g \n", 0.0, 0.0, 0.0);
  }
  if (fclose(f)) return fail(std::string("write error on ") + path);
  return 0;
}

// ---- command file (.nlp), the subset the path needs before its first step -----------------------------------------
// Blocks "Keyword (Key=value,...) {" + "Property = value" lines + "}" as the reference's readers take them:
//   NLPS-Solver (Type=scheme) { CFL= Cel= N= i0= Epsilon= ... }          InOutFun/Read_GramsTime.c:44-230
//   GramsShapeFun (Type=LME) { gamma= TOL-Zero= MaxIter= wrapper= TOL-Wrapper= }   InOutFun/Read_GramsShapeFun.c:20-200
//   GramsBox (Type=GID,File=box.msh) { ... }                              InOutFun/Read_GramsBox.c:235-262
//   One-Phase-Analysis (File=body.msh, GPxElement=n) ...                  Analysis/Generate-One-Phase-Analysis.c:386-445
// Materials, initial values, boundary conditions and outputs are other blocks of the same file and are not read here.
namespace {
std::string dir_of(const char* path) {  // generate_route: the directory of the command file, with the separator
  std::string p(path);
  const size_t k = p.find_last_of('/');
  return k == std::string::npos ? std::string("./") : p.substr(0, k + 1);
}
int tokens(char* line, const char* delims, std::vector<char*>& w) {
  w.clear();
  char* save = nullptr;
  for (char* p = strtok_r(line, delims, &save); p; p = strtok_r(nullptr, delims, &save)) w.push_back(p);
  return (int)w.size();
}
// "Property = value" lines up to the closing brace; fn returns false for a property it does not know
template <class F>
int read_properties(LineReader& in, const char* who, F fn) {
  std::vector<char*> w;
  while (true) {
    if (!in.next()) return fail(std::string(who) + ": you forget to put a } !!!");
    const int n = tokens(in.buf.data(), " =\t\r\n", w);
    if (n > 0 && !strcmp(w[0], "}")) return 0;
    if (n == 0) continue;  // (the reference stops on an empty line; tolerated here)
    if (n != 2) return fail(std::string(who) + ": Use this format -> Propertie = value !!!");
    if (!fn(w[0], w[1])) return fail(std::string(who) + ": Undefined " + w[0]);
  }
}
}  // namespace

extern "C" int nlps_host_read_deck(const char* path, nlps_deck* d) {
  if (!path || !d) return fail("null argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  memset(d, 0, sizeof(*d));
  // defaults of Initialise_Parameters (Read_GramsTime.c:246-262) and of GramsShapeFun (Read_GramsShapeFun.c:86-90)
  d->CFL = 0.8;
  d->epsilon_mass_matrix = 1.0;
  d->beta_newmark = 0.25;
  d->gamma_newmark = 0.5;
  d->tol_newmark = 1E-10;
  d->rb_generalized_alpha = 0.6;
  d->tol_generalized_alpha = 1E-10;
  d->max_iter = 10;
  d->gamma_lme = 3;
  d->tol_zero_lme = 1e-6;
  d->tol_wrapper_lme = 1e-10;
  d->max_iter_lme = 10;
  snprintf(d->wrapper_lme, sizeof(d->wrapper_lme), "Newton-Raphson");
  const std::string route = dir_of(path);
  int n_solver = 0;
  bool has_N = false, has_cel = false, has_cfl = false, has_beta = false, has_gamma = false, has_tol = false,
       has_eps = false, has_rb = false, has_tolga = false;
  std::vector<char*> w, kv;
  std::vector<char> copy;
  while (in.next()) {
    copy.assign(in.buf.begin(), in.buf.begin() + strlen(in.buf.data()) + 1);
    if (tokens(copy.data(), " \r\n\t", w) < 1) continue;
    if (!strcmp(w[0], "NLPS-Solver")) {
      n_solver++;
      if (w.size() < 3 || tokens(w[1], "(=)", kv) != 2 || strcmp(kv[0], "Type"))
        return fail("NLPS-Solver: Use this format -> NLPS-Solver (Type=string) { !!!");
      snprintf(d->scheme, sizeof(d->scheme), "%s", kv[1]);
      if (strcmp(w[2], "{")) return fail("NLPS-Solver: Use this format -> NLPS-Solver (Type=string) { !!!");
      if (read_properties(in, "NLPS-Solver", [&](const char* k, const char* v) {
            if (!strcmp(k, "CFL")) d->CFL = atof(v), has_cfl = true;
            else if (!strcmp(k, "Cel")) d->Cel = atof(v), has_cel = true;
            else if (!strcmp(k, "i0")) d->i0 = atoi(v);
            else if (!strcmp(k, "N")) d->N = atoi(v), has_N = true;
            else if (!strcmp(k, "Epsilon")) d->epsilon_mass_matrix = atof(v), has_eps = true;
            else if (!strcmp(k, "rb-Generalized-alpha")) d->rb_generalized_alpha = atof(v), has_rb = true;
            else if (!strcmp(k, "TOL-Generalized-alpha")) d->tol_generalized_alpha = atof(v), has_tolga = true;
            else if (!strcmp(k, "Beta-Newmark-beta")) d->beta_newmark = atof(v), has_beta = true;
            else if (!strcmp(k, "Gamma-Newmark-beta")) d->gamma_newmark = atof(v), has_gamma = true;
            else if (!strcmp(k, "TOL-Newmark-beta")) d->tol_newmark = atof(v), has_tol = true;
            else if (!strcmp(k, "Max-Iter")) d->max_iter = atoi(v);
            else if (!strcmp(k, "Explicit-trial")) d->explicit_trial = atoi(v);
            else return false;
            return true;
          }))
        return 1;
    } else if (!strcmp(w[0], "GramsShapeFun")) {
      if (w.size() < 2 || tokens(w[1], "(=)", kv) != 2 || strcmp(kv[0], "Type"))
        return fail("GramsShapeFun: Use this format -> (Type=string) !!!");
      snprintf(d->shape_fun, sizeof(d->shape_fun), "%s", kv[1]);
      if (w.size() < 3 || strcmp(w[2], "{")) return fail("GramsShapeFun: Use this format -> GramsShapeFun (Type=string) { !!!");
      if (!(w.size() == 4 && !strcmp(w[3], "}")) &&
          read_properties(in, "GramsShapeFun", [&](const char* k, const char* v) {
            if (!strcmp(k, "gamma")) d->gamma_lme = atof(v);
            else if (!strcmp(k, "TOL-Zero")) d->tol_zero_lme = atof(v);
            else if (!strcmp(k, "MaxIter")) d->max_iter_lme = atoi(v);
            else if (!strcmp(k, "wrapper")) snprintf(d->wrapper_lme, sizeof(d->wrapper_lme), "%s", v);
            else if (!strcmp(k, "TOL-Wrapper")) d->tol_wrapper_lme = atof(v);
            else return false;
            return true;
          }))
        return 1;
      if ((!strcmp(d->shape_fun, "LME") || !strcmp(d->shape_fun, "aLME")) && d->gamma_lme == 0)
        return fail("GramsShapeFun: gamma parameter required for LME !!!");
      // wrapper=Nelder-Mead makes initialize__LME__ start lambda from initialise_lambda__LME__ (LME.c:97-101, 189-268:
      // a simplex of the particle's element in the file's connectivity order; exit(0) for 8-node elements), which this
      // path does not restate: say so instead of running Newton from lambda = 0 as if nothing had been asked
      if (d->wrapper_lme[0] && strcmp(d->wrapper_lme, "Newton-Raphson"))
        return fail("GramsShapeFun: wrapper=Newton-Raphson is the only LME wrapper of the GPU path (Nelder-Mead stays on the CPU path)");
    } else if (!strcmp(w[0], "GramsBox")) {
      copy.assign(in.buf.begin(), in.buf.begin() + strlen(in.buf.data()) + 1);
      if (tokens(copy.data(), " =,()\r\n\t", w) < 5 || strcmp(w[1], "Type") || strcmp(w[3], "File"))
        return fail("GramsBox: GramsBox (Type=GID,File=mesh.msh) {");
      if (strcmp(w[2], "GID")) return fail("GramsBox: Unrecognised kind of mesh");
      snprintf(d->box_mesh, sizeof(d->box_mesh), "%s%s", route.c_str(), w[4]);
    } else if (!strcmp(w[0], "One-Phase-Analysis")) {
      copy.assign(in.buf.begin(), in.buf.begin() + strlen(in.buf.data()) + 1);
      if (tokens(copy.data(), " (,)\r\n\t", w) < 3) return fail("One-Phase-Analysis (File=Mesh.msh, GPxElement=int)");
      if (tokens(w[1], "=", kv) != 2 || strcmp(kv[0], "File")) return fail("One-Phase-Analysis (File=Mesh.msh, *)");
      snprintf(d->body_mesh, sizeof(d->body_mesh), "%s%s", route.c_str(), kv[1]);
      if (tokens(w[2], "=", kv) != 2 || strcmp(kv[0], "GPxElement")) return fail("One-Phase-Analysis (*, GPxElement=int)");
      d->gp_per_elem = atoi(kv[1]);
    }
  }
  // check_Solver, Read_GramsTime.c:265-378
  if (n_solver != 1) return fail(n_solver ? "NLPS-Solver: more than one solver defined" : "NLPS-Solver: no solver defined");
  if (!(has_N && has_cel && has_cfl)) return fail("NLPS-Solver: N, Cel and CFL are required");
  if (!strcmp(d->scheme, "Newmark-beta-Finite-Strains") && !(has_beta && has_gamma && has_tol && has_eps))
    return fail("NLPS-Solver: Newmark-beta-Finite-Strains needs Beta-Newmark-beta, Gamma-Newmark-beta, TOL-Newmark-beta, Epsilon");
  if (!strcmp(d->scheme, "Generalized-alpha") && !(has_rb && has_tolga))
    return fail("NLPS-Solver: Generalized-alpha needs rb-Generalized-alpha and TOL-Generalized-alpha");
  if (!strcmp(d->scheme, "Discrete-Energy-Momentum") && !(has_rb && has_tolga && has_eps))
    return fail("NLPS-Solver: Discrete-Energy-Momentum needs rb-Generalized-alpha, TOL-Generalized-alpha and Epsilon");
  return 0;
}

// Define-Material blocks (InOutFun/Material/Read_GramsMaterials2.c:51-175) of the six laws of this path:
//   Define-Material(idx=0,Model=Neo-Hookean-Wriggers | Hencky | Drucker-Prager | Von-Mises | Matsuoka-Nakai |
//                   Lade-Duncan) { property = value ... }
// with the property names, defaults and completeness checks of Material/Hyperelastic/Neo-Hookean.c, Hencky.c and
// Material/Plasticity/Drucker-Prager.c, Von-Mises.c, Matsuoka-Nakai.c, Lade-Duncan.c (the last two also set the
// globals TOL_Radial_Returning / Max_Iterations_Radial_Returning to 1e-10 / 20 and 1e-14 / 10: the caller's
// nlps_params).  The eigenerosion / eigensoftening constants (Ceps, Gf, ft,
// heps, wcrit) are accepted and dropped, as the reference does when those drivers are off.
extern "C" int nlps_host_read_materials(const char* path, int max_materials, nlps_material* mats, double* rho,
                                        int* idx, int* nmats) {
  if (!path || !mats || !rho || !nmats || max_materials < 1) return fail("bad argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  *nmats = 0;
  std::vector<char*> w, kv;
  while (in.next()) {
    if (tokens(in.buf.data(), " ,()\r\n\t", w) < 1 || strcmp(w[0], "Define-Material")) continue;
    if (w.size() < 3) return fail("Define-Material: Define-Material(idx=int,Model=string)");
    if (*nmats >= max_materials) return fail("Define-Material: more materials than the caller has room for");
    // Read_Index_and_Model, :178-210: "idx=0" or "0", "Model=X" or "X"
    int n = tokens(w[1], "=", kv);
    if (n < 1) return fail("Define-Material: Define-Material(idx=int,Model=string)");
    const int id = atoi(kv[n == 2 ? 1 : 0]);
    n = tokens(w[2], "=", kv);
    if (n < 1) return fail("Define-Material: Define-Material(idx=int,Model=string)");
    const std::string model = kv[n == 2 ? 1 : 0];
    nlps_material m;
    memset(&m, 0, sizeof(m));
    m.theta_voce = 1.0;  // Von-Mises.c:70-75
    if (model == "Neo-Hookean-Wriggers") m.type = NLPS_MAT_NEO_HOOKEAN;
    else if (model == "Hencky") m.type = NLPS_MAT_HENCKY;
    else if (model == "Drucker-Prager") m.type = NLPS_MAT_DRUCKER_PRAGER;
    else if (model == "Von-Mises") m.type = NLPS_MAT_VON_MISES;
    else if (model == "Matsuoka-Nakai") m.type = NLPS_MAT_MATSUOKA_NAKAI;
    else if (model == "Lade-Duncan") m.type = NLPS_MAT_LADE_DUNCAN;
    else return fail("Define-Material: model " + model + " is not one of the laws of this path");
    const std::string who = "Define-Material(" + model + ")";
    bool open = false, closed = false, has_rho = false, has_E = false, has_nu = false, has_phi = false, has_psi = false,
         has_eps0 = false, has_yield = false, fbar = false, has_alpha = false, has_a1 = false, has_a2 = false,
         has_a3 = false, has_kappa0 = false;
    double H = 0.0, r = 0.0;
    while (!closed) {
      if (!in.next()) return fail(who + ": the block is not closed");
      const int np_ = tokens(in.buf.data(), " =\t\r\n", kv);
      if (np_ == 0) continue;
      const char* k = kv[0];
      const double v = np_ > 1 ? atof(kv[1]) : 0.0;
      const bool dp = m.type == NLPS_MAT_DRUCKER_PRAGER, vm = m.type == NLPS_MAT_VON_MISES;
      const bool ld = m.type == NLPS_MAT_LADE_DUNCAN, fr = ld || m.type == NLPS_MAT_MATSUOKA_NAKAI;
      if (!strcmp(k, "{") && np_ == 1) open = true;
      else if (!strcmp(k, "}") && np_ == 1) closed = true;
      else if (np_ != 2) return fail(who + ": Use this format -> Propertie = value");
      else if (!strcmp(k, "rho")) r = v, has_rho = true;
      else if (!strcmp(k, "E")) m.E = v, has_E = true;
      else if (!strcmp(k, "nu")) m.nu = v, has_nu = true;
      else if (!strcmp(k, "Ceps") || !strcmp(k, "Gf") || !strcmp(k, "ft") || !strcmp(k, "heps") || !strcmp(k, "wcrit")) {
      } else if (m.type == NLPS_MAT_NEO_HOOKEAN && !strcmp(k, "Fbar"))
        fbar = !strcmp(kv[1], "true") || !strcmp(kv[1], "True") || !strcmp(kv[1], "TRUE") || !strcmp(kv[1], "1");
      else if (m.type == NLPS_MAT_NEO_HOOKEAN && !strcmp(k, "Fbar-alpha")) {
      } else if (dp && !strcmp(k, "m")) m.exponent_ortiz = v;
      else if (dp && !strcmp(k, "Hardening-modulus")) H = v;
      else if (dp && !strcmp(k, "Reference-pressure")) m.p_ref = v;
      else if (dp && !strcmp(k, "Reference-plastic-strain")) m.eps_0 = v, has_eps0 = true;
      else if (dp && !strcmp(k, "kappa-0")) m.kappa_0 = v;
      else if (dp && !strcmp(k, "Friction-angle")) m.phi_deg = v, has_phi = true;
      else if (dp && !strcmp(k, "Dilatancy-angle")) m.psi_deg = v, has_psi = true;
      else if (dp && !strcmp(k, "J2-degradated")) {
      } else if (vm && !strcmp(k, "Yield-stress")) m.kappa_0 = v, has_yield = true;
      else if (vm && !strcmp(k, "Hardening-Modulus")) m.hardening_modulus = v;
      else if (vm && !strcmp(k, "theta")) m.theta_voce = v;
      else if (vm && !strcmp(k, "K-0")) m.K0_voce = v;
      else if (vm && !strcmp(k, "K-inf")) m.Kinf_voce = v;
      else if (vm && !strcmp(k, "delta")) m.delta_voce = v;
      else if (fr && !strcmp(k, "alpha")) m.alpha_borja = v, has_alpha = true;
      else if (fr && !strcmp(k, "a1")) m.a_borja[0] = v, has_a1 = true;
      else if (fr && !strcmp(k, "a2")) m.a_borja[1] = v, has_a2 = true;
      else if (fr && !strcmp(k, "a3")) m.a_borja[2] = v, has_a3 = true;
      else if (fr && !strcmp(k, "Reference-pressure")) m.p_ref = v;
      else if (fr && !strcmp(k, "Friction-angle")) m.phi_deg = v, has_phi = true;
      else if (fr && !strcmp(k, "EPS-0")) m.eps_0 = v, has_eps0 = true;
      else if (fr && !strcmp(k, "kappa-0")) m.kappa_0 = v, has_kappa0 = true;
      else if (fr && !strcmp(k, "Cohesion")) m.cohesion = v;
      else if (ld && !strcmp(k, "Atmospheric-pressure")) {
      } else return fail(who + ": Undefined " + k);
    }
    (void)open;  // the reference notes the opening brace and never asks for it either
    if (!(has_rho && has_E && has_nu)) return fail(who + ": rho, E and nu are required");
    if (fbar) return fail(who + ": Fbar needs the quadratic-triangle patches this path does not cover");
    if (m.type == NLPS_MAT_DRUCKER_PRAGER) {
      if (!(has_phi && has_psi)) return fail(who + ": Friction-angle and Dilatancy-angle are required");
      const double mo = m.exponent_ortiz;
      if (!((mo > 0 && H > 0) || (mo < 0 && H < 0))) return fail(who + ": m and Hardening-modulus must have one sign");
      if (!has_eps0) m.eps_0 = (m.kappa_0 / (mo * H)) * std::pow(1, (1.0 / mo - 1.0));  // Drucker-Prager.c:200-209
    }
    if (m.type == NLPS_MAT_VON_MISES && !has_yield) return fail(who + ": Yield-stress is required");
    if (m.type == NLPS_MAT_MATSUOKA_NAKAI || m.type == NLPS_MAT_LADE_DUNCAN) {
      const double PI = 3.14159265358979323846;
      if (!(has_alpha && has_a1 && has_a2 && has_a3)) return fail(who + ": alpha, a1, a2 and a3 are required");
      if (m.type == NLPS_MAT_MATSUOKA_NAKAI) {
        // Matsuoka-Nakai.c:207-211: without a friction angle, the one that matches kappa_0 (degrees)
        if (!has_phi) m.phi_deg = (180.0 / PI) * std::asin(std::sqrt(m.kappa_0 / (m.kappa_0 + 8.0)));
      } else if (has_phi) {
        // Lade-Duncan.c:210-237: kappa_0 from the friction angle, EPS-0 from kappa_0 = a1 EPS exp(a2 I1) exp(-a3 EPS)
        // at I1 = 3 (p_ref - c cot phi) by Newton (tolerance = the reader's TOL_Radial_Returning, 1e-14; 10 iterations)
        const double rad = (PI / 180.0) * m.phi_deg, c_cotphi = m.cohesion / std::tan(rad);
        const double a1 = m.a_borja[0], a2 = m.a_borja[1], a3 = m.a_borja[2], I1 = 3 * (m.p_ref - c_cotphi);
        const double kappa_0 = 8.0 * std::sin(rad) * std::sin(rad) / (1.0 - std::sin(rad) * std::sin(rad));
        double EPS_0 = 0.0, f = kappa_0 - a1 * EPS_0 * std::exp(a2 * I1) * std::exp(-a3 * EPS_0);
        int iter = 0;
        while (std::fabs(f) > 1E-14) {
          iter++;
          const double df = (a3 * EPS_0 - 1) * a1 * std::exp(a2 * I1) * std::exp(-a3 * EPS_0);
          EPS_0 += -f / df;
          f = kappa_0 - a1 * EPS_0 * std::exp(a2 * I1) * std::exp(-a3 * EPS_0);
          if (iter > 10) return fail(who + ": Iter > 10");
        }
        m.kappa_0 = kappa_0;
        m.eps_0 = EPS_0;
      } else if (has_kappa0) {
        // :238-241 (its first test is an assignment, so kappa-0 alone decides; the angle is left in radians there)
        m.phi_deg = std::asin(std::sqrt(m.kappa_0 / (m.kappa_0 + 8)));
      } else {
        return fail(who + ": Some parameters are missed for Lade-Duncan initialization");
      }
    }
    mats[*nmats] = m;
    rho[*nmats] = r;
    if (idx) idx[*nmats] = id;
    (*nmats)++;
  }
  return 0;
}

// ---- Dirichlet boundaries of the command file ------------------------------------------------------------------------
//   GramsBoundary (File=nodes.txt) { BcDirichlet V.x curve.txt | NULL ... }
// (Boundary-Conditions/NLPS-Read-u-Dirichlet-Boundary-Conditions.c:46-300): the node list is one id per line
// (File2Chain.c: first word of every line, pushed, so the list runs in REVERSED file order), every direction with a
// curve is active for the first min(NumTimeStep, curve.Num) steps (active_direction, :367-371) and carries the
// curve's values.  Curves: ReadCurve.c (DAT_CURVE NUM#n, then CONSTANT_CURVE SCALE#s | RAMP_CURVE SCALE#s |
// HEAVISIDE_CURVE SCALE#s Tc#t | DELTA_CURVE SCALE#s Tc#t | HAT_CURVE SCALE#s T0#a T1#b | CUSTOM_CURVE + n lines).
namespace {
int read_curve(const std::string& file, std::vector<double>& fx) {
  LineReader in(file.c_str());
  if (!in.f) return fail("ReadCurve: during the lecture of " + file);
  fx.clear();
  std::vector<char*> w, a, b, c;
  auto kv = [&](char* word, std::vector<char*>& out, const char* key) {
    return tokens(word, "#\r\n", out) == 2 && !strcmp(out[0], key);
  };
  while (in.next()) {
    if (tokens(in.buf.data(), " \r\n\t", w) < 1) continue;
    const int n = (int)fx.size();
    if (!strcmp(w[0], "DAT_CURVE")) {
      for (size_t i = 1; i < w.size(); i++)
        if (kv(w[i], a, "NUM")) fx.assign((size_t)std::max(0, atoi(a[1])), 0.0);
    } else if (!strcmp(w[0], "CUSTOM_CURVE")) {
      for (int i = 0; i < n && in.next(); i++)
        if (tokens(in.buf.data(), " \r\n\t", a) > 0) fx[i] += atof(a[0]);
    } else if (!strcmp(w[0], "CONSTANT_CURVE")) {
      if (w.size() < 2 || !kv(w[1], a, "SCALE")) return fail("fill_ConstantCurve: Wrong parameters");
      for (int i = 0; i < n; i++) fx[i] += atof(a[1]);
    } else if (!strcmp(w[0], "RAMP_CURVE")) {
      if (w.size() < 2 || !kv(w[1], a, "SCALE")) return fail("fill_RampCurve: Wrong parameters");
      for (int i = 0; i < n; i++) fx[i] = atof(a[1]) * (double)i / n;
    } else if (!strcmp(w[0], "HEAVISIDE_CURVE") || !strcmp(w[0], "DELTA_CURVE")) {
      if (w.size() < 3 || !kv(w[1], a, "SCALE") || !kv(w[2], b, "Tc")) return fail(std::string(w[0]) + ": Wrong parameters");
      const double s = atof(a[1]);
      const int tc = atoi(b[1]);
      if (tc > n || tc < 0) return fail(std::string(w[0]) + ": Tc outside the curve");
      const bool step = !strcmp(w[0], "HEAVISIDE_CURVE");
      for (int i = 0; i < n; i++) fx[i] += step ? (i <= tc ? 0.0 : s) : (i == tc ? s : 0.0);
    } else if (!strcmp(w[0], "HAT_CURVE")) {
      if (w.size() < 4 || !kv(w[1], a, "SCALE") || !kv(w[2], b, "T0") || !kv(w[3], c, "T1")) return fail("fill_HatCurve: Wrong parameters");
      const int t0 = atoi(b[1]), t1 = atoi(c[1]);
      if (t0 > t1 || t1 > n || t0 < 0) return fail("fill_HatCurve: T0, T1 outside the curve");
      // as written (ReadCurve.c:433-441) the hat never comes down: (i >= T0) || (i <= T1) holds for every i >= T0
      for (int i = 0; i < n; i++) fx[i] += i < t0 ? 0.0 : atof(a[1]);
    }
  }
  return 0;
}
}  // namespace

namespace {
// gp = 0: GramsBoundary / BcDirichlet V.* with a node list; gp > 0: Define-Neumann-Boundary / T.* with a list of body
// elements, each standing for its gp particles e * gp + j (NLPS-Read-u-Neumann-Boundary-Conditions.c:150-185, 318-352)
int read_bounds(const char* path, int ndim, int nsteps, int gp, int max_bounds, int node_cap, int* nbounds, int* nnodes,
                int* nodes, int* dir, double* value) {
  if (!path || !nbounds || (ndim != 2 && ndim != 3) || nsteps < 1 || gp < 0) return fail("bad argument");
  const char* keyword = gp ? "Define-Neumann-Boundary" : "GramsBoundary";
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  const std::string route = dir_of(path);
  const bool fill = nnodes && nodes && dir && value;  // nnodes alone: only the node counts are wanted
  *nbounds = 0;
  int used = 0;
  std::vector<char*> w, kv;
  std::vector<double> fx;
  while (in.next()) {
    if (tokens(in.buf.data(), " ,()\r\n\t", w) < 1 || strcmp(w[0], keyword)) continue;
    if (w.size() < 2 || tokens(w[1], "=", kv) != 2 || strcmp(kv[0], "File"))
      return fail(std::string(keyword) + ": use this format -> " + keyword + " (File=Nodes.txt)");
    const int b = *nbounds;
    if (nnodes && b >= max_bounds) return fail("GramsBoundary: more boundaries than the caller has room for");
    // File2Chain: first word of every line; the chain, and with it the node array, is in reversed file order
    std::vector<int> ids;
    {
      LineReader nf((route + kv[1]).c_str());
      if (!nf.f) return fail("File2Chain: Incorrect lecture of " + route + kv[1]);
      std::vector<char*> t;
      while (nf.next())
        if (tokens(nf.buf.data(), " \r\n\t", t) > 0) ids.push_back(atoi(t[0]));
    }
    if (gp) {  // the chain (reversed file order) of elements, each expanded to its particles
      std::vector<int> parts;
      for (size_t i = 0; i < ids.size(); i++)
        for (int j = 0; j < gp; j++) parts.push_back(ids[ids.size() - 1 - i] * gp + j);
      ids.assign(parts.rbegin(), parts.rend());  // (the copy below reverses once more)
    }
    if (nnodes) nnodes[b] = (int)ids.size();
    if (fill) {
      if (used + (int)ids.size() > node_cap) return fail("GramsBoundary: more boundary nodes than the caller has room for");
      for (size_t i = 0; i < ids.size(); i++) nodes[used + (int)i] = ids[ids.size() - 1 - i];
      for (int k = 0; k < ndim * nsteps; k++) {
        dir[(size_t)b * ndim * nsteps + k] = 0;
        value[(size_t)b * ndim * nsteps + k] = 0.0;
      }
    }
    used += (int)ids.size();
    bool closed = false;
    while (!closed) {
      if (!in.next()) return fail("GramsBoundary: the block is not closed");
      const int n = tokens(in.buf.data(), " =\t\r\n", w);
      if (n == 0) continue;
      if (n == 1 && !strcmp(w[0], "{")) continue;
      if (n == 1 && !strcmp(w[0], "}")) {
        closed = true;
      } else if ((!gp && n == 3 && !strcmp(w[0], "BcDirichlet")) || (gp && n == 2 && w[0][0] == 'T' && w[0][1] == '.')) {
        const char* comp = gp ? w[0] + 2 : (w[1][0] == 'V' && w[1][1] == '.' ? w[1] + 2 : "?");
        const char* file = gp ? w[1] : w[2];
        const int k = !strcmp(comp, "x") ? 0 : !strcmp(comp, "y") ? 1 : !strcmp(comp, "z") ? 2 : -1;
        if (k < 0) return fail(std::string(keyword) + ": Velocity component " + (gp ? w[0] : w[1]) + " is not available");
        if (!strcmp(file, "NULL") || k >= ndim) continue;
        if (read_curve(route + file, fx)) return 1;
        if (fill) {
          const int na = std::min(nsteps, (int)fx.size());
          for (int t = 0; t < na; t++) {
            dir[((size_t)b * ndim + k) * nsteps + t] = 1;
            value[((size_t)b * ndim + k) * nsteps + t] = fx[t];
          }
        }
      } else {
        return fail(std::string(keyword) + ": undefined property " + w[0]);
      }
    }
    (*nbounds)++;
  }
  (void)used;
  return 0;
}
}  // namespace

extern "C" int nlps_host_read_boundaries(const char* path, int ndim, int nsteps, int max_bounds, int node_cap,
                                         int* nbounds, int* nnodes, int* nodes, int* dir, double* value) {
  return read_bounds(path, ndim, nsteps, 0, max_bounds, node_cap, nbounds, nnodes, nodes, dir, value);
}
extern "C" int nlps_host_read_neumann(const char* path, int ndim, int nsteps, int gp_per_elem, int max_bounds,
                                      int node_cap, int* nbounds, int* nnodes, int* nodes, int* dir, double* value) {
  if (gp_per_elem < 1) return fail("bad argument");
  return read_bounds(path, ndim, nsteps, gp_per_elem, max_bounds, node_cap, nbounds, nnodes, nodes, dir, value);
}

// Assign-material-to-particles (MatIdx=i, Particles=list.txt)  (Generate-One-Phase-Analysis.c:458-566): the list
// names body elements; their particles e * GPxElement + j get material i.  matidx[nparticles] is updated in place.
extern "C" int nlps_host_read_material_assignment(const char* path, int gp_per_elem, int nmaterials, int nparticles,
                                                  int* matidx) {
  if (!path || !matidx || gp_per_elem < 1) return fail("bad argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  const std::string route = dir_of(path);
  std::vector<char*> w, kv;
  while (in.next()) {
    if (tokens(in.buf.data(), " (,)\r\n\t", w) < 3 || strcmp(w[0], "Assign-material-to-particles")) continue;
    if (tokens(w[1], "=", kv) != 2 || strcmp(kv[0], "MatIdx")) return fail("Assign-material-to-particles (MatIdx=Int, *)");
    const int mi = atoi(kv[1]);
    if (mi < 0 || mi >= nmaterials) return fail("Assign-material-to-particles: MatIdx should go from 0 to " + std::to_string(nmaterials - 1));
    if (tokens(w[2], "=", kv) != 2 || strcmp(kv[0], "Particles")) return fail("Assign-material-to-particles (*, Particles=List-Particles.txt)");
    LineReader nf((route + kv[1]).c_str());
    if (!nf.f) return fail("File2Chain: Incorrect lecture of " + route + kv[1]);
    std::vector<char*> t;
    while (nf.next()) {
      if (tokens(nf.buf.data(), " \r\n\t", t) < 1) continue;
      const int e = atoi(t[0]);
      if (e < 0 || (long long)(e + 1) * gp_per_elem > nparticles) return fail("Assign-material-to-particles: element " + std::to_string(e) + " is outside the particle set");
      for (int j = 0; j < gp_per_elem; j++) matidx[(size_t)e * gp_per_elem + j] = mi;
    }
  }
  return 0;
}

// GramsInitials (Nodes=list.txt) { Value=[vx,vy,vz] }  (Initial-Conditions/Read_GramsInitials.c:7-186): the list
// holds ELEMENT ids of the body mesh (0-based, File2Chain), every particle e * GPxElement + j of a listed element
// gets the velocity.  vel[nparticles][ndim] is updated in place; elements outside the cloud are an error here
// (the reference writes out of bounds).
extern "C" int nlps_host_read_initials(const char* path, int ndim, int gp_per_elem, int nparticles, double* vel) {
  if (!path || !vel || (ndim != 2 && ndim != 3) || gp_per_elem < 1) return fail("bad argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  const std::string route = dir_of(path);
  std::vector<char*> w, kv, val;
  while (in.next()) {
    if (tokens(in.buf.data(), " \r\n\t", w) < 1 || strcmp(w[0], "GramsInitials")) continue;
    if (w.size() < 3 || tokens(w[1], "(=)", kv) != 2 || strcmp(kv[0], "Nodes"))
      return fail("GramsInitials: Use this format -> (Nodes=str) !!!");
    if (strcmp(w[2], "{")) return fail("GramsInitials: Use this format -> GramsInitials (Nodes=str) { !!!");
    std::vector<int> ids;
    {
      LineReader nf((route + kv[1]).c_str());
      if (!nf.f) return fail("File2Chain: Incorrect lecture of " + route + kv[1]);
      std::vector<char*> t;
      while (nf.next())
        if (tokens(nf.buf.data(), " \r\n\t", t) > 0) ids.push_back(atoi(t[0]));
    }
    bool any = false;
    while (true) {
      if (!in.next()) return fail("GramsInitials: you forget to put a } !!!");
      const int n = tokens(in.buf.data(), " =\t\r\n", kv);
      if (n > 0 && !strcmp(kv[0], "}")) break;
      if (n == 0) continue;
      if (n != 2) return fail("GramsInitials: Use this format -> Propertie = value !!!");
      if (strcmp(kv[0], "Value")) return fail(std::string("GramsInitials: Undefined ") + kv[0]);
      if (tokens(kv[1], "[,]", val) != ndim) return fail("GramsInitials: Use this format -> Value=[vx,vy,vz] with one entry per dimension");
      for (int e : ids) {
        if (e < 0 || (long long)(e + 1) * gp_per_elem > nparticles) return fail("GramsInitials: element " + std::to_string(e) + " is outside the particle set");
        for (int j = 0; j < gp_per_elem; j++)
          for (int k = 0; k < ndim; k++) vel[((size_t)e * gp_per_elem + j) * ndim + k] = atof(val[k]);
      }
      any = true;
    }
    if (!any) return fail("GramsInitials: Undefined initial condition !!!");
  }
  return 0;
}

// generate-gravity-field-constant { g.x = .. g.y = .. g.z = .. } and generate-gravity-field-curve { g = file.csv }
// (Read_Generate_Gravity_Field.c:170-371): the gravity vector of every time step, g[nsteps][ndim] (the argument of
// nlps_gpu_explicit_step per step).  The csv holds one "gx,gy[,gz]" line per step and is opened by the name as written
// (the reference does not join it to the directory of the command file).  No block: zeros, *found = 0.
extern "C" int nlps_host_read_gravity(const char* path, int ndim, int nsteps, double* g, int* found) {
  if (!path || !g || (ndim != 2 && ndim != 3) || nsteps < 1) return fail("bad argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  for (int k = 0; k < nsteps * ndim; k++) g[k] = 0.0;
  if (found) *found = 0;
  std::vector<char*> w, col;
  int kind = 0;
  while (!kind && in.next()) {
    if (tokens(in.buf.data(), " \r\n\t", w) < 1) continue;
    if (!strcmp(w[0], "generate-gravity-field-constant")) kind = 1;
    else if (!strcmp(w[0], "generate-gravity-field-curve")) kind = 2;
  }
  if (!kind) return 0;
  if (found) *found = 1;
  bool open = false, closed = false;
  while (!closed && in.next()) {
    const int n = tokens(in.buf.data(), "= \r\n\t", w);
    if (n < 1) continue;
    if (!strcmp(w[0], "{")) open = true;
    else if (open && !strcmp(w[0], "}")) closed = true;
    else if (open && kind == 1 && n > 1) {
      const int a = !strcmp(w[0], "g.x") ? 0 : !strcmp(w[0], "g.y") ? 1 : !strcmp(w[0], "g.z") ? 2 : -1;
      if (a >= 0 && a < ndim)
        for (int t = 0; t < nsteps; t++) g[(size_t)t * ndim + a] = atof(w[1]);
    } else if (open && kind == 2 && n > 1 && !strcmp(w[0], "g")) {
      LineReader csv(w[1]);
      if (!csv.f) return fail(std::string("gravity field: ") + w[1] + " is not a file");
      for (int t = 0; t < nsteps; t++) {
        if (!csv.next()) return fail(std::string("gravity field: ") + w[1] + " ends before step " + std::to_string(t + 1));
        if (tokens(csv.buf.data(), ",\r\n", col) != ndim)
          return fail(std::string("gravity field: ") + w[1] + ": wrong number of columns in line " + std::to_string(t + 1));
        for (int a = 0; a < ndim; a++) g[(size_t)t * ndim + a] = atof(col[a]);
      }
    }
  }
  if (!open || !closed) return fail("gravity field: the block needs its braces");
  return 0;
}

// GramsOutputs (i=int) { DIR=dir  Particles-file=name  Nodes-file=name  Out-...=true|false }
// (Outputs/Read_GramsOutputs.c:25-345): output interval, directory (joined to the command file's, must exist), file
// names and the Out_* switches.  The switches the particle writer of this library knows come back as the selection
// nlps_host_write_particles_vtk understands; the others (water pressure, strains, damage ...) are accepted and
// counted in `unsupported` when they are switched on.
extern "C" int nlps_host_read_outputs(const char* path, nlps_outputs* o) {
  if (!path || !o) return fail("null argument");
  LineReader in(path);
  if (!in.f) return fail(std::string("cannot open ") + path);
  memset(o, 0, sizeof(*o));
  const std::string route = dir_of(path);
  std::vector<char*> w, kv;
  bool have_dir = false;
  auto on = [&](const char* v, bool& bad) {
    if (!strcmp(v, "true")) return 1;
    if (!strcmp(v, "false")) return 0;
    bad = true;
    return 0;
  };
  while (in.next()) {
    if (tokens(in.buf.data(), " \r\n\t", w) < 1 || strcmp(w[0], "GramsOutputs")) continue;
    o->found = 1;
    if (w.size() < 3 || tokens(w[1], "(=)", kv) != 2 || strcmp(kv[0], "i")) return fail("GramsOutputs: Use this format -> (i=int)");
    o->results_time_step = atoi(kv[1]);
    if (strcmp(w[2], "{")) return fail("GramsOutputs: Use this format -> GramsOutputs (Type=string) { ");
    while (true) {
      if (!in.next()) return fail("GramsOutputs: you forget to put a }");
      const int n = tokens(in.buf.data(), " =\t\r\n", kv);
      if (n > 0 && !strcmp(kv[0], "}")) break;
      if (n == 0) continue;
      if (n != 2) return fail("GramsOutputs: Use this format -> Propertie = value");
      const char *k = kv[0], *v = kv[1];
      bool bad = false;
      if (!strcmp(k, "DIR")) {
        snprintf(o->dir, sizeof(o->dir), "%s%s", route.c_str(), v);
        struct stat info;  // Check_Output_directory, :333-341: it has to exist
        have_dir = stat(o->dir, &info) == 0 && S_ISDIR(info.st_mode);
      } else if (!strcmp(k, "Particles-file")) snprintf(o->particles_file, sizeof(o->particles_file), "%s", v);
      else if (!strcmp(k, "Nodes-file")) snprintf(o->nodes_file, sizeof(o->nodes_file), "%s", v);
      else if (!strcmp(k, "Out-global-coordinates")) o->global_coordinates = on(v, bad);
      else if (!strcmp(k, "Out-mass")) o->mass = on(v, bad);
      else if (!strcmp(k, "Out-density")) o->density = on(v, bad);
      else if (!strcmp(k, "Out-nodal-idx")) o->nodal_idx = on(v, bad);
      else if (!strcmp(k, "Out-material-idx")) o->material_idx = on(v, bad);
      else if (!strcmp(k, "Out-velocity")) o->velocity = on(v, bad);
      else if (!strcmp(k, "Out-acceleration")) o->acceleration = on(v, bad);
      else if (!strcmp(k, "Out-displacement")) o->displacement = on(v, bad);
      else if (!strcmp(k, "Out-stress")) o->stress = on(v, bad);
      else if (!strcmp(k, "Out-volumetric-stress")) o->volumetric_stress = on(v, bad);
      else if (!strcmp(k, "Out-deformation-gradient")) o->deformation_gradient = on(v, bad);
      else if (!strcmp(k, "Out-energy")) o->energy = on(v, bad);
      else if (!strcmp(k, "Out-Equivalent-Plastic-Strain")) o->eps = on(v, bad);
      else if (!strcmp(k, "Out-element-coordinates") || !strcmp(k, "Out-damage") || !strcmp(k, "Out-eigenvalues-stress") ||
               !strcmp(k, "Out-water-pressure") || !strcmp(k, "Out-Pore-water-pressure") ||
               !strcmp(k, "Out-Rate-Pore-water-pressure") || !strcmp(k, "Out-strain") || !strcmp(k, "Out-eigenvalues-strain") ||
               !strcmp(k, "Out-green-lagrange") || !strcmp(k, "Out-plastic-deformation-gradient") || !strcmp(k, "Out-Metric") ||
               !strcmp(k, "Out-plastic-jacobian") || !strcmp(k, "Out-Von-Mises") || !strcmp(k, "Out-Check-Partition-Unity"))
        o->unsupported += on(v, bad);
      else return fail(std::string("GramsOutputs: the output ") + k + " is not available");
      if (bad) return fail(std::string("GramsOutputs: the input was ") + v + ". Please, use : true/false");
    }
    if (!have_dir) return fail("GramsOutputs: No output dir was defined");
    break;
  }
  return 0;
}
