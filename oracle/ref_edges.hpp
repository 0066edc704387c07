// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).  Restatement of the g2o
// edge classes used by the batch graph (fp64):
//   EdgeSE3PointXYZ            dependencies/g2o/g2o/types/edge_se3_pointxyz.cpp:99-140
//   CacheSE3Offset             dependencies/g2o/g2o/types/parameter_se3_offset.cpp:77-82
//   LandmarkMotionTernaryEdge  dependencies/g2o/g2o/types/types_dyn_slam3d.cpp:53-85   (quirk F4 kept)
//   EdgeSE3 / EdgeSE3Prior     dependencies/g2o/g2o/types/edge_se3.cpp:77-104, edge_se3_prior.cpp:89-102
//   computeEdgeSE3Gradient     dependencies/g2o/g2o/types/isometry3d_gradients.h:191-261
//   computeEdgeSE3PriorGradient                                    ...:264-325
//   compute_dq_dR              dependencies/g2o/g2o/types/dquat2mat.cpp:35-84 (+ maxima tables, re-derived)
#pragma once
#include "ref_math.hpp"

namespace vdo_oracle {

// EdgeSE3PointXYZ.  Jpose 3x6 row-major, Jpoint 3x3 row-major.
inline void edge_eb(const Iso& X, V3 p, V3 z, double e[3], double Jpose[18], double Jpoint[9]) {
  Iso w2l = iso_inv(X);               // CacheSE3Offset::updateImpl, offset = I
  V3 zc = iso_apply(w2l, p);          // perr = w2n * point ; Zcam = w2l * point (same for I offset)
  e[0] = zc.x - z.x; e[1] = zc.y - z.y; e[2] = zc.z - z.z;
  if (Jpose) {
    for (int i = 0; i < 18; ++i) Jpose[i] = 0;
    Jpose[0 * 6 + 0] = -1; Jpose[1 * 6 + 1] = -1; Jpose[2 * 6 + 2] = -1;
    Jpose[0 * 6 + 4] = -2 * zc.z; Jpose[0 * 6 + 5] = 2 * zc.y;
    Jpose[1 * 6 + 3] = 2 * zc.z;  Jpose[1 * 6 + 5] = -2 * zc.x;
    Jpose[2 * 6 + 3] = -2 * zc.y; Jpose[2 * 6 + 4] = 2 * zc.x;
  }
  if (Jpoint) for (int i = 0; i < 9; ++i) Jpoint[i] = w2l.R.m[i];
}

// LandmarkMotionTernaryEdge: e = p1 - H^-1 p2 - z.   J_H has no factor 2 on the
// rotation columns (F4, types_dyn_slam3d.cpp:73-78) — kept as in the reference.
inline void edge_et(const Iso& H, V3 p1, V3 p2, V3 z, double e[3], double Jp1[9], double Jp2[9], double Jh[18]) {
  Iso Hi = iso_inv(H);
  V3 v = iso_apply(Hi, p2);
  e[0] = p1.x - v.x - z.x; e[1] = p1.y - v.y - z.y; e[2] = p1.z - v.z - z.z;
  if (Jp1) { for (int i = 0; i < 9; ++i) Jp1[i] = 0; Jp1[0] = Jp1[4] = Jp1[8] = 1; }
  if (Jp2) for (int i = 0; i < 9; ++i) Jp2[i] = -Hi.R.m[i];
  if (Jh) {
    for (int i = 0; i < 18; ++i) Jh[i] = 0;
    Jh[0] = Jh[7] = Jh[14] = 1;
    Jh[0 * 6 + 4] = v.z;  Jh[0 * 6 + 5] = -v.y;
    Jh[1 * 6 + 3] = -v.z; Jh[1 * 6 + 5] = v.x;
    Jh[2 * 6 + 3] = v.y;  Jh[2 * 6 + 4] = -v.x;
  }
}

// d(q_xyz)/d(R) for q = compact quaternion of R; dq[3][9], column index = i + 3 j
// (column-major flattening of R).  Branch selection follows _q2m (dquat2mat.cpp:35-64);
// the entries are re-derived analytically from q_d = 1/2 sqrt(1 + sum s_ii r_ii),
// q_k = (r_dk +- r_kd) / (4 q_d).
inline void dq_dR(const M3& R, double dq[3][9]) {
  for (int a = 0; a < 3; ++a) for (int c = 0; c < 9; ++c) dq[a][c] = 0;
  const double r00 = R(0, 0), r11 = R(1, 1), r22 = R(2, 2);
  const double tr = r00 + r11 + r22;
  double qw;
  if (tr > 0) {
    double S = std::sqrt(tr + 1.0) * 2;
    qw = 0.25 * S;
    const double a = 0.25 / qw, d = -0.03125 / (qw * qw * qw);
    // q_x = (r21 - r12)/(4 qw), q_y = (r02 - r20)/(4 qw), q_z = (r10 - r01)/(4 qw)
    const int hi[3][2] = {{2, 1}, {0, 2}, {1, 0}};
    for (int k = 0; k < 3; ++k) {
      int i = hi[k][0], j = hi[k][1];
      double num = R(i, j) - R(j, i);
      dq[k][0] = dq[k][4] = dq[k][8] = num * d;
      dq[k][i + 3 * j] = a;
      dq[k][j + 3 * i] = -a;
    }
  } else {
    int dmn;
    if ((r00 > r11) & (r00 > r22)) dmn = 0; else if (r11 > r22) dmn = 1; else dmn = 2;
    double s[3] = {-1, -1, -1};
    s[dmn] = 1;
    double S = std::sqrt(1.0 + s[0] * r00 + s[1] * r11 + s[2] * r22) * 2;
    const int j = (dmn + 1) % 3, k = (dmn + 2) % 3;
    qw = (R(k, j) - R(j, k)) / S;
    const double qd = 0.25 * S;
    const double a = 0.25 / qd, g = 0.125 / qd, d3 = 0.03125 / (qd * qd * qd);
    for (int i = 0; i < 3; ++i) dq[dmn][i + 3 * i] = s[i] * g;
    for (int o = 0; o < 3; ++o) {
      if (o == dmn) continue;
      double num = R(dmn, o) + R(o, dmn);
      for (int i = 0; i < 3; ++i) dq[o][i + 3 * i] = -s[i] * d3 * num;
      dq[o][dmn + 3 * o] = a;
      dq[o][o + 3 * dmn] = a;
    }
  }
  if (qw <= 0) for (int a = 0; a < 3; ++a) for (int c = 0; c < 9; ++c) dq[a][c] = -dq[a][c];
}

// isometry3d_gradients.h skew(S,v) / skewT(S,v): 2*[v]x^T and 2*[v]x (note the factor 2)
inline M3 skew2(V3 v)  { double x = 2 * v.x, y = 2 * v.y, z = 2 * v.z; return M3{{0, z, -y, -z, 0, x, y, -x, 0}}; }
inline M3 skew2T(V3 v) { double x = 2 * v.x, y = 2 * v.y, z = 2 * v.z; return M3{{0, -z, y, z, 0, -x, -y, x, 0}}; }
// skew(Sx,Sy,Sz,R) / skewT(...) (isometry3d_gradients.h:57-85)
inline void skew3(const M3& R, double sgn, M3& Sx, M3& Sy, M3& Sz) {
  double r[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r[i][j] = sgn * 2 * R(i, j);
  Sx = M3{{0, 0, 0, -r[2][0], -r[2][1], -r[2][2], r[1][0], r[1][1], r[1][2]}};
  Sy = M3{{r[2][0], r[2][1], r[2][2], 0, 0, 0, -r[0][0], -r[0][1], -r[0][2]}};
  Sz = M3{{-r[1][0], -r[1][1], -r[1][2], r[0][0], r[0][1], r[0][2], 0, 0, 0}};
}
// out(3x3 block at rows 3..5, cols 3..5 of a 6x6 row-major J) = dq_dR * [vec(A Sx) vec(A Sy) vec(A Sz)]
inline void rot_block(const double dq[3][9], const M3& A, const M3& Sx, const M3& Sy, const M3& Sz, double J[36]) {
  M3 P[3] = {mul(A, Sx), mul(A, Sy), mul(A, Sz)};
  for (int a = 0; a < 3; ++a)
    for (int c = 0; c < 3; ++c) {
      double s = 0;
      for (int col = 0; col < 3; ++col)
        for (int row = 0; row < 3; ++row) s += dq[a][row + 3 * col] * P[c](row, col);
      J[(3 + a) * 6 + 3 + c] = s;
    }
}

// EdgeSE3: e = toVectorMQT(Z^-1 Xi^-1 Xj); Ji, Jj 6x6 row-major.
inline void edge_se3(const Iso& Z, const Iso& Xi, const Iso& Xj, double e[6], double Ji[36], double Jj[36]) {
  Iso A = iso_inv(Z);
  // computeError (edge_se3.cpp:77-82) multiplies left to right: (Z^-1 Xi^-1) Xj; the gradient (isometry3d_gradients.h:203-206) forms E = A (Xi^-1 Xj).
  // Found by compiling the reference's own file (oracle/_ref, tests/test_ref_g2o.py): the two differ in the last bit.
  toVectorMQT(iso_mul(iso_mul(A, iso_inv(Xi)), Xj), e);
  if (!Ji) return;
  Iso B = iso_mul(iso_inv(Xi), Xj);
  Iso E = iso_mul(A, B);
  for (int i = 0; i < 36; ++i) Ji[i] = Jj[i] = 0;
  double dq[3][9];
  dq_dR(E.R, dq);
  M3 RaS = mul(A.R, skew2T(B.t));
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      Ji[r * 6 + c] = -A.R(r, c);
      Jj[r * 6 + c] = E.R(r, c);
      Ji[r * 6 + 3 + c] = RaS(r, c);
    }
  M3 Sx, Sy, Sz;
  skew3(B.R, -1.0, Sx, Sy, Sz);      // skewT(Sxt,Syt,Szt,Rb)
  rot_block(dq, A.R, Sx, Sy, Sz, Ji);
  skew3(m3_identity(), 1.0, Sx, Sy, Sz);
  rot_block(dq, E.R, Sx, Sy, Sz, Jj);
}

// EdgeSE3Prior with identity offset: e = toVectorMQT(Z^-1 X)
inline void edge_prior(const Iso& Z, const Iso& X, double e[6], double J[36]) {
  Iso A = iso_mul(iso_inv(Z), X);
  toVectorMQT(A, e);
  if (!J) return;
  for (int i = 0; i < 36; ++i) J[i] = 0;
  double dq[3][9];
  dq_dR(A.R, dq);
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) J[r * 6 + c] = A.R(r, c);
  // dte/dq = Ra * skew(tb) with tb = 0  -> exactly zero
  M3 Sx, Sy, Sz;
  skew3(m3_identity(), 1.0, Sx, Sy, Sz);
  rot_block(dq, A.R, Sx, Sy, Sz, J);
}

// VertexSE3::oplusImpl (vertex_se3.h:105-114).  `calls` is the vertex' _numOplusCalls.
inline void iso_oplus(Iso& X, const double d[6], int& calls) {
  Iso inc = fromVectorMQT(d);
  X = iso_mul(X, inc);
  if (++calls > 1000) { calls = 0; approximateNearestOrthogonalMatrix(X.R); }
}

}  // namespace vdo_oracle
