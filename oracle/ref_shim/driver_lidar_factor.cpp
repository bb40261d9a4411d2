// oracle/ref_shim/driver_lidar_factor.cpp — evaluates the reference's OWN residual functors (/root/reference/src/
// lidarFactor.hpp, included where it lies) through ceres::AutoDiffCostFunction of the stand-in Ceres header, for a list of
// factors with a general interpolation ratio s.  TEST INFRASTRUCTURE ONLY: this is how the s != 1 (DISTORTION 1) branch of
// the functors is pinned — the reference's nodes compile it out (#define DISTORTION 0, src/laserOdometry.cpp:59).
//   usage: ref_lidar_factor <in.bin> <out.bin>
//   in : int32 n, then n records of float64: kind (0 edge, 1 plane), s, q[4] (x,y,z,w), t[3], 12 constants
//        (edge: cp, a, b, 3 unused; plane: cp, j, l, m)
//   out: per record float64 residual[3] (plane: first entry), d r / d q [3][4], d r / d t [3][3]  (unused rows zero)
#include <cstdio>
#include <memory>

#include "lidarFactor.hpp"
#include "ref_io.hpp"

int main(int argc, char** argv) {
  ref_io::must(argc == 3, "usage: ref_lidar_factor <in.bin> <out.bin>");
  FILE* fin = std::fopen(argv[1], "rb");
  FILE* fout = std::fopen(argv[2], "wb");
  ref_io::must(fin && fout, "cannot open files");
  const int n = ref_io::read_i32(fin);
  for (int i = 0; i < n; ++i) {
    double rec[21];
    ref_io::must(std::fread(rec, 8, 21, fin) == 21, "short read");
    const int kind = static_cast<int>(rec[0]);
    const double s = rec[1];
    const double* q = rec + 2;
    const double* t = rec + 6;
    const double* c = rec + 9;
    const Eigen::Vector3d cp(c[0], c[1], c[2]), p1(c[3], c[4], c[5]), p2(c[6], c[7], c[8]), p3(c[9], c[10], c[11]);
    std::unique_ptr<ceres::CostFunction> f(kind == 0 ? LidarEdgeFactor::Create(cp, p1, p2, s) : LidarPlaneFactor::Create(cp, p1, p2, p3, s));
    double r[3] = {0, 0, 0}, jq[12] = {0}, jt[9] = {0};
    const double* params[2] = {q, t};
    double* jac[2] = {jq, jt};
    ref_io::must(f->Evaluate(params, r, jac), "functor returned false");
    ref_io::write_f64(fout, r, 3);
    ref_io::write_f64(fout, jq, 12);
    ref_io::write_f64(fout, jt, 9);
  }
  std::fclose(fin);
  std::fclose(fout);
  return 0;
}
