// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never linked or called by the product library.
//
// Independent checker for the camera-model gates of the matchers.  It shares no code with the library's
// orb_slam3_amd/csrc/orbx_camera_gates.h: every step is evaluated in binary64, and the null vector of the
// triangulation system comes from a cyclic Jacobi eigen-decomposition of A^T A (the library runs a one-sided Jacobi
// SVD in binary32, the reference Eigen's two-sided JacobiSVD in binary32).  Restated from:
//   KannalaBrandt8::project(Vector3f)      src/CameraModels/KannalaBrandt8.cpp:66-83
//   KannalaBrandt8::unproject              src/CameraModels/KannalaBrandt8.cpp:117-146   (precision 1e-6, KannalaBrandt8.h:43-49)
//   KannalaBrandt8::epipolarConstrain      src/CameraModels/KannalaBrandt8.cpp:227-232
//   KannalaBrandt8::TriangulateMatches     src/CameraModels/KannalaBrandt8.cpp:323-397
//   KannalaBrandt8::Triangulate            src/CameraModels/KannalaBrandt8.cpp:418-430
//   Pinhole::project / unprojectEig        src/CameraModels/Pinhole.cpp:46-53, :66-70
//   Pinhole::epipolarConstrain             src/CameraModels/Pinhole.cpp:118-141
//
// PARITY UNPINNED: the reference holds no fixtures for these functions and cannot be built here (Eigen, OpenCV and
// Sophus are absent).  The reference itself evaluates them in binary32 with Eigen and the C library, so its accept /
// reject decision is only defined up to rounding when a tested quantity lies at its threshold.  This checker
// therefore reports, with every decision, a MARGIN: the smallest distance of any tested quantity from its threshold,
// in units of the band inside which binary32 rounding may legitimately flip the test.  margin >= 1 means the decision
// does not depend on rounding, and the library must agree; margin < 1 marks an in-band case, which the tests count
// and report instead of comparing.  Bands (the figure each tested quantity is divided by):
//   cosParallaxRays > 0.9998                  |cos - 0.9998|                 / 2e-6
//   z1 <= 0, z2 <= 0                          |z| / |x3D|                    / 1e-4      (and |h3| / |h| / 1e-5)
//   reprojection e^2 > 5.991 sigma            |e^2 - 5.991 sigma| / (5.991 sigma)  / 2e-3
//   depth > 0.0001 (callers' test)            |z1 - 1e-4| / 1e-4             / 1e-2
//   Pinhole: den == 0                         never in band (exact zero only)
//   Pinhole: dsqr < 3.84 unc                  |dsqr - 3.84 unc| / (3.84 unc) / 5e-3
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "camera_oracle.h"

namespace camo {

static const double kPi = 3.14159265358979323846;

void unproject(const Cam& c, double u, double v, double ray[3]) {
    const double x = (u - c.p[2]) / c.p[0], y = (v - c.p[3]) / c.p[1];
    ray[0] = x; ray[1] = y; ray[2] = 1.0;
    if (c.type == kPinhole) return;                                  // Pinhole.cpp:66-70
    double theta_d = std::sqrt(x * x + y * y);                       // KannalaBrandt8.cpp:120-123
    theta_d = std::min(std::max(-kPi / 2.0, theta_d), kPi / 2.0);
    double scale = 1.0;
    if (theta_d > 1e-8) {                                            // :125
        double theta = theta_d;
        for (int j = 0; j < 10; ++j) {                               // :130-140
            const double t2 = theta * theta, t4 = t2 * t2, t6 = t4 * t2, t8 = t4 * t4;
            const double a = c.p[4] * t2, b = c.p[5] * t4, cc = c.p[6] * t6, d = c.p[7] * t8;
            const double fix = (theta * (1 + a + b + cc + d) - theta_d) / (1 + 3 * a + 5 * b + 7 * cc + 9 * d);
            theta -= fix;
            if (std::fabs(fix) < 1e-6) break;
        }
        scale = std::tan(theta) / theta_d;                           // :142
    }
    ray[0] = x * scale; ray[1] = y * scale;
}

void project(const Cam& c, const double P[3], double uv[2]) {
    if (c.type == kPinhole) {                                        // Pinhole.cpp:46-53
        uv[0] = c.p[0] * P[0] / P[2] + c.p[2];
        uv[1] = c.p[1] * P[1] / P[2] + c.p[3];
        return;
    }
    const double theta = std::atan2(std::sqrt(P[0] * P[0] + P[1] * P[1]), P[2]);   // KannalaBrandt8.cpp:68-70
    const double psi = std::atan2(P[1], P[0]);
    const double t2 = theta * theta, t3 = theta * t2, t5 = t3 * t2, t7 = t5 * t2, t9 = t7 * t2;
    const double r = theta + c.p[4] * t3 + c.p[5] * t5 + c.p[6] * t7 + c.p[7] * t9;
    uv[0] = c.p[0] * r * std::cos(psi) + c.p[2];
    uv[1] = c.p[1] * r * std::sin(psi) + c.p[3];
}

// Unit eigenvector of the smallest eigenvalue of the symmetric 4x4 matrix S (cyclic Jacobi rotations).
static void smallest_eigenvector4(double S[4][4], double out[4]) {
    double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0, diag = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) (i == j ? diag : off) += S[i][j] * S[i][j];
        if (off <= 1e-40 * diag || off == 0) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (S[p][q] == 0) continue;
                const double tau = (S[q][q] - S[p][p]) / (2 * S[p][q]);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1 + tau * tau));
                const double cs = 1 / std::sqrt(1 + t * t), sn = t * cs;
                for (int k = 0; k < 4; ++k) {                         // S <- S J
                    const double skp = S[k][p], skq = S[k][q];
                    S[k][p] = cs * skp - sn * skq; S[k][q] = sn * skp + cs * skq;
                }
                for (int k = 0; k < 4; ++k) {                         // S <- J^T S
                    const double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = cs * spk - sn * sqk; S[q][k] = sn * spk + cs * sqk;
                }
                for (int k = 0; k < 4; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    int best = 0;
    for (int j = 1; j < 4; ++j)
        if (S[j][j] < S[best][best]) best = j;
    for (int i = 0; i < 4; ++i) out[i] = V[i][best];
}

static inline void note(double& margin, double distance, double band) { margin = std::min(margin, distance / band); }

double triangulate_matches(const Cam& c1, const Cam& c2, double u1, double v1, double u2, double v2, const double R12[9],
                           const double t12[3], double sigmaLevel, double unc, double p3D[3], double* margin_out) {
    double margin = 1e300;
    double r1[3], r2[3];
    unproject(c1, u1, v1, r1);                                       // :326-327
    unproject(c2, u2, v2, r2);
    double r21[3];
    for (int i = 0; i < 3; ++i) r21[i] = R12[3 * i] * r2[0] + R12[3 * i + 1] * r2[1] + R12[3 * i + 2] * r2[2];   // :330
    const double n1 = std::sqrt(r1[0] * r1[0] + r1[1] * r1[1] + r1[2] * r1[2]);
    const double n21 = std::sqrt(r21[0] * r21[0] + r21[1] * r21[1] + r21[2] * r21[2]);
    const double cosParallax = (r1[0] * r21[0] + r1[1] * r21[1] + r1[2] * r21[2]) / (n1 * n21);                 // :332
    note(margin, std::fabs(cosParallax - 0.9998), 2e-6);
    if (cosParallax > 0.9998) { if (margin_out) *margin_out = margin; return -1; }                            // :334-337
    // Tcw1 = [I | 0], Tcw2 = [R21 | -R21 t12]  (:349-355)
    double T2[3][4];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T2[i][j] = R12[3 * j + i];
        T2[i][3] = -(R12[i] * t12[0] + R12[3 + i] * t12[1] + R12[6 + i] * t12[2]);
    }
    // Triangulate (:418-430): rows p.x * T.row(2) - T.row(0), p.y * T.row(2) - T.row(1)
    double A[4][4] = {{-1, 0, r1[0], 0}, {0, -1, r1[1], 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int j = 0; j < 4; ++j) {
        A[2][j] = r2[0] * T2[2][j] - T2[0][j];
        A[3][j] = r2[1] * T2[2][j] - T2[1][j];
    }
    double S[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[k][i] * A[k][j];
            S[i][j] = s;
        }
    double h[4];
    smallest_eigenvector4(S, h);
    const double hn = std::sqrt(h[0] * h[0] + h[1] * h[1] + h[2] * h[2] + h[3] * h[3]);
    note(margin, std::fabs(h[3]) / hn, 1e-5);
    const double x3D[3] = {h[0] / h[3], h[1] / h[3], h[2] / h[3]};                                            // :429
    const double xn = std::sqrt(x3D[0] * x3D[0] + x3D[1] * x3D[1] + x3D[2] * x3D[2]);
    const double z1 = x3D[2];                                                                                 // :360
    note(margin, std::fabs(z1) / xn, 1e-4);
    if (z1 <= 0) { if (margin_out) *margin_out = margin; return -2; }
    const double z2 = T2[2][0] * x3D[0] + T2[2][1] * x3D[1] + T2[2][2] * x3D[2] + T2[2][3];                   // :366
    note(margin, std::fabs(z2) / xn, 1e-4);
    if (z2 <= 0) { if (margin_out) *margin_out = margin; return -3; }
    double uv[2];
    project(c1, x3D, uv);                                                                                     // :373
    const double e1 = (uv[0] - u1) * (uv[0] - u1) + (uv[1] - v1) * (uv[1] - v1);
    note(margin, std::fabs(e1 - 5.991 * sigmaLevel) / (5.991 * sigmaLevel), 2e-3);
    if (e1 > 5.991 * sigmaLevel) { if (margin_out) *margin_out = margin; return -4; }                         // :378
    double x2[3];
    for (int i = 0; i < 3; ++i) x2[i] = T2[i][0] * x3D[0] + T2[i][1] * x3D[1] + T2[i][2] * x3D[2] + T2[i][3];   // :383
    project(c2, x2, uv);
    const double e2 = (uv[0] - u2) * (uv[0] - u2) + (uv[1] - v2) * (uv[1] - v2);
    note(margin, std::fabs(e2 - 5.991 * unc) / (5.991 * unc), 2e-3);
    if (e2 > 5.991 * unc) { if (margin_out) *margin_out = margin; return -5; }                                // :389
    note(margin, std::fabs(z1 - 1e-4) / 1e-4, 1e-2);                  // the callers' `> 0.0001f`
    if (p3D) { p3D[0] = x3D[0]; p3D[1] = x3D[1]; p3D[2] = x3D[2]; }
    if (margin_out) *margin_out = margin;
    return z1;
}

bool epipolar_constrain(const Cam& c1, const Cam& c2, double u1, double v1, double u2, double v2, const double R12[9],
                        const double t12[3], double sigmaLevel, double unc, double* margin_out) {
    if (c1.type == kKannalaBrandt8)                                  // KannalaBrandt8.cpp:227-232
        return triangulate_matches(c1, c2, u1, v1, u2, v2, R12, t12, sigmaLevel, unc, nullptr, margin_out) > 0.0001;
    // Pinhole.cpp:121-124: F12 = K1^-T [t12]x R12 K2^-1.  With x~ = K^-1 (u, v, 1) the epipolar line is
    // l = x1~^T [t12]x R12 K2^-1, evaluated here without forming F12.
    const double x1[3] = {(u1 - c1.p[2]) / c1.p[0], (v1 - c1.p[3]) / c1.p[1], 1.0};
    // w = x1~^T [t]x  =  (x1~ x t)^T ... row vector times the skew matrix: (x^T [t]x)_j = sum_i x_i [t]x_ij
    const double tx[3][3] = {{0, -t12[2], t12[1]}, {t12[2], 0, -t12[0]}, {-t12[1], t12[0], 0}};
    double w[3], m[3];
    for (int j = 0; j < 3; ++j) w[j] = x1[0] * tx[0][j] + x1[1] * tx[1][j] + x1[2] * tx[2][j];
    for (int j = 0; j < 3; ++j) m[j] = w[0] * R12[j] + w[1] * R12[3 + j] + w[2] * R12[6 + j];
    // l = m^T K2^-1, K2^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1]
    const double a = m[0] / c2.p[0], b = m[1] / c2.p[1];
    const double cc = -m[0] * c2.p[2] / c2.p[0] - m[1] * c2.p[3] / c2.p[1] + m[2];
    const double num = a * u2 + b * v2 + cc;                         // :131
    const double den = a * a + b * b;                                // :133
    if (den == 0) { if (margin_out) *margin_out = 1e300; return false; }
    const double dsqr = num * num / den;                             // :138
    if (margin_out) *margin_out = std::fabs(dsqr - 3.84 * unc) / (3.84 * unc) / 5e-3;
    return dsqr < 3.84 * unc;                                        // :140
}

Cam make_cam(int type, const float* params) {
    Cam c;
    c.type = type;
    for (int k = 0; k < 8; ++k) c.p[k] = params[k];
    return c;
}

}  // namespace camo

extern "C" {

// The gates on their own.  Inputs are the reference's binary32 values; results are rounded to binary32 once.
void orc_camera_project(int type, const float* params, const float* xyz, float* uv) {
    const camo::Cam c = camo::make_cam(type, params);
    const double P[3] = {xyz[0], xyz[1], xyz[2]};
    double r[2];
    camo::project(c, P, r);
    uv[0] = (float)r[0]; uv[1] = (float)r[1];
}
void orc_camera_unproject(int type, const float* params, const float* uv, float* xyz) {
    const camo::Cam c = camo::make_cam(type, params);
    double r[3];
    camo::unproject(c, uv[0], uv[1], r);
    xyz[0] = (float)r[0]; xyz[1] = (float)r[1]; xyz[2] = (float)r[2];
}
// TriangulateMatches: depth in camera 1 (or the negative rejection code); p3d (3 doubles) and margin may be NULL.
double orc_triangulate_matches(int type1, const float* cam1, int type2, const float* cam2, float u1, float v1, float u2,
                               float v2, const float* R12, const float* t12, float sigmaLevel, float unc, double* p3d,
                               double* margin) {
    const camo::Cam c1 = camo::make_cam(type1, cam1), c2 = camo::make_cam(type2, cam2);
    double R[9], t[3];
    for (int i = 0; i < 9; ++i) R[i] = R12[i];
    for (int i = 0; i < 3; ++i) t[i] = t12[i];
    return camo::triangulate_matches(c1, c2, u1, v1, u2, v2, R, t, sigmaLevel, unc, p3d, margin);
}
float orc_kb8_triangulate_matches(const float* cam1, const float* cam2, float u1, float v1, float u2, float v2,
                                  const float* R12, const float* t12, float sigmaLevel, float unc) {
    return (float)orc_triangulate_matches(camo::kKannalaBrandt8, cam1, camo::kKannalaBrandt8, cam2, u1, v1, u2, v2, R12, t12,
                                          sigmaLevel, unc, nullptr, nullptr);
}
int orc_epipolar_constrain_margin(int type1, const float* cam1, int type2, const float* cam2, float u1, float v1, float u2,
                                  float v2, const float* R12, const float* t12, float sigmaLevel, float unc, double* margin) {
    const camo::Cam c1 = camo::make_cam(type1, cam1), c2 = camo::make_cam(type2, cam2);
    double R[9], t[3];
    for (int i = 0; i < 9; ++i) R[i] = R12[i];
    for (int i = 0; i < 3; ++i) t[i] = t12[i];
    return camo::epipolar_constrain(c1, c2, u1, v1, u2, v2, R, t, sigmaLevel, unc, margin) ? 1 : 0;
}
int orc_epipolar_constrain(int type1, const float* cam1, int type2, const float* cam2, float u1, float v1, float u2, float v2,
                           const float* R12, const float* t12, float sigmaLevel, float unc) {
    return orc_epipolar_constrain_margin(type1, cam1, type2, cam2, u1, v1, u2, v2, R12, t12, sigmaLevel, unc, nullptr);
}

// n pairs at once: depth[i] / tri_margin[i] = TriangulateMatches (KannalaBrandt8 pairs only, else untouched),
// accepted[i] / epi_margin[i] = pCamera1->epipolarConstrain.  p3d: n x 3 doubles, may be NULL.
void orc_camera_gates_many(int type1, const float* cam1, int type2, const float* cam2, const float* uv1, const float* uv2,
                           int n, const float* R12, const float* t12, const float* sigmaLevel, const float* unc,
                           double* depth, double* tri_margin, double* p3d, uint8_t* accepted, double* epi_margin) {
    const camo::Cam c1 = camo::make_cam(type1, cam1), c2 = camo::make_cam(type2, cam2);
    double R[9], t[3];
    for (int i = 0; i < 9; ++i) R[i] = R12[i];
    for (int i = 0; i < 3; ++i) t[i] = t12[i];
    for (int i = 0; i < n; ++i) {
        if (type1 == camo::kKannalaBrandt8)
            depth[i] = camo::triangulate_matches(c1, c2, uv1[2 * i], uv1[2 * i + 1], uv2[2 * i], uv2[2 * i + 1], R, t,
                                                 sigmaLevel[i], unc[i], p3d ? p3d + 3 * i : nullptr, &tri_margin[i]);
        accepted[i] = camo::epipolar_constrain(c1, c2, uv1[2 * i], uv1[2 * i + 1], uv2[2 * i], uv2[2 * i + 1], R, t,
                                               sigmaLevel[i], unc[i], &epi_margin[i]) ? 1 : 0;
    }
}

}  // extern "C"
