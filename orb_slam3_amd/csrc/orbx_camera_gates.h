// Per-pair camera-model gates of ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:1139):
//   Pinhole::epipolarConstrain          src/CameraModels/Pinhole.cpp:118-141
//   KannalaBrandt8::epipolarConstrain   src/CameraModels/KannalaBrandt8.cpp:227-232
//     -> TriangulateMatches :323-397, unproject :117-146, project :66-83, Triangulate :418-430
// A few dozen float operations per candidate pair that survives the descriptor test: host code (plain C++, no HIP),
// evaluated inside the in-order replay loop.  This header is shared with the CPU checker of the tests in the way the
// sampling pattern table is: the reference evaluates these gates with Eigen (3x3 inverse, 4x4 JacobiSVD) and libm, neither of
// which fixes the last bits, so no independent restatement could be compared bit for bit anyway.  What is restated is
// the sequence of tests, their thresholds and the precision (binary32) of every step; the null vector of the 4x4
// triangulation system comes from a one-sided Jacobi SVD.
#ifndef ORBX_CAMERA_GATES_H_
#define ORBX_CAMERA_GATES_H_

#include <cmath>

namespace orbx_gates {

enum { kPinhole = 0, kKannalaBrandt8 = 1 };

// GeometricCamera::mvParameters: {fx, fy, cx, cy} (+ {k0, k1, k2, k3} for KannalaBrandt8)
struct Camera {
    int type;
    float p[8];
};

struct Vec3 { float x, y, z; };

inline Vec3 mat_vec(const float* R, const Vec3 v) {   // row-major 3x3
    return Vec3{R[0] * v.x + R[1] * v.y + R[2] * v.z, R[3] * v.x + R[4] * v.y + R[5] * v.z,
                R[6] * v.x + R[7] * v.y + R[8] * v.z};
}

// KannalaBrandt8::unproject (Newton iterations on theta, precision 1e-6); Pinhole::unproject
inline Vec3 unproject(const Camera& c, float u, float v) {
    const float pwx = (u - c.p[2]) / c.p[0], pwy = (v - c.p[3]) / c.p[1];
    if (c.type == kPinhole) return Vec3{pwx, pwy, 1.f};
    const float halfPi = (float)(3.1415926535897932384626433832795 / 2.f);
    float scale = 1.f;
    float theta_d = sqrtf(pwx * pwx + pwy * pwy);
    theta_d = fminf(fmaxf(-halfPi, theta_d), halfPi);
    if (theta_d > 1e-8) {
        float theta = theta_d;
        for (int j = 0; j < 10; j++) {
            const float theta2 = theta * theta, theta4 = theta2 * theta2, theta6 = theta4 * theta2, theta8 = theta4 * theta4;
            const float k0_theta2 = c.p[4] * theta2, k1_theta4 = c.p[5] * theta4;
            const float k2_theta6 = c.p[6] * theta6, k3_theta8 = c.p[7] * theta8;
            const float theta_fix = (theta * (1 + k0_theta2 + k1_theta4 + k2_theta6 + k3_theta8) - theta_d) /
                                    (1 + 3 * k0_theta2 + 5 * k1_theta4 + 7 * k2_theta6 + 9 * k3_theta8);
            theta = theta - theta_fix;
            if (fabsf(theta_fix) < 1e-6f) break;
        }
        scale = tanf(theta) / theta_d;
    }
    return Vec3{pwx * scale, pwy * scale, 1.f};
}

// KannalaBrandt8::project(const Eigen::Vector3f&); Pinhole::project
inline void project(const Camera& c, const Vec3 P, float* u, float* v) {
    if (c.type == kPinhole) {
        *u = c.p[0] * P.x / P.z + c.p[2];
        *v = c.p[1] * P.y / P.z + c.p[3];
        return;
    }
    const float x2_plus_y2 = P.x * P.x + P.y * P.y;
    const float theta = atan2f(sqrtf(x2_plus_y2), P.z);
    const float psi = atan2f(P.y, P.x);
    const float theta2 = theta * theta, theta3 = theta * theta2, theta5 = theta3 * theta2;
    const float theta7 = theta5 * theta2, theta9 = theta7 * theta2;
    const float r = theta + c.p[4] * theta3 + c.p[5] * theta5 + c.p[6] * theta7 + c.p[7] * theta9;
    *u = c.p[0] * r * cosf(psi) + c.p[2];
    *v = c.p[1] * r * sinf(psi) + c.p[3];
}

// Right singular vector of the smallest singular value of a 4x4 matrix (what
// JacobiSVD<Matrix4f>(A, ComputeFullV).matrixV().col(3) returns, up to sign): one-sided Jacobi on the columns.
inline void null_vector4(const float A[4][4], float out[4]) {
    float U[4][4], V[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) { U[i][j] = A[i][j]; V[i][j] = (i == j) ? 1.f : 0.f; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                float alpha = 0.f, beta = 0.f, gamma = 0.f;
                for (int i = 0; i < 4; ++i) { alpha += U[i][p] * U[i][p]; beta += U[i][q] * U[i][q]; gamma += U[i][p] * U[i][q]; }
                if (gamma == 0.f || fabsf(gamma) <= 1e-7f * sqrtf(alpha * beta)) continue;
                rotated = true;
                const float zeta = (beta - alpha) / (2.f * gamma);
                const float t = copysignf(1.f, zeta) / (fabsf(zeta) + sqrtf(1.f + zeta * zeta));
                const float cs = 1.f / sqrtf(1.f + t * t), sn = cs * t;
                for (int i = 0; i < 4; ++i) {
                    const float up = U[i][p], uq = U[i][q];
                    U[i][p] = cs * up - sn * uq; U[i][q] = sn * up + cs * uq;
                    const float vp = V[i][p], vq = V[i][q];
                    V[i][p] = cs * vp - sn * vq; V[i][q] = sn * vp + cs * vq;
                }
            }
        if (!rotated) break;
    }
    int best = 0;
    float bestNorm = 0.f;
    for (int j = 0; j < 4; ++j) {
        float n = 0.f;
        for (int i = 0; i < 4; ++i) n += U[i][j] * U[i][j];
        if (j == 0 || n < bestNorm) { bestNorm = n; best = j; }
    }
    for (int i = 0; i < 4; ++i) out[i] = V[i][best];
}

// KannalaBrandt8::TriangulateMatches: the depth in camera 1 of the triangulated point, or a negative code; the point
// itself (camera-1 coordinates) through p3D when the depth is positive.
inline float triangulate_matches(const Camera& c1, const Camera& c2, float u1, float v1, float u2, float v2,
                                 const float* R12, const float* t12, float sigmaLevel, float unc, Vec3* p3D = nullptr) {
    const Vec3 r1 = unproject(c1, u1, v1), r2 = unproject(c2, u2, v2);
    const Vec3 r21 = mat_vec(R12, r2);
    const float n1 = sqrtf(r1.x * r1.x + r1.y * r1.y + r1.z * r1.z), n21 = sqrtf(r21.x * r21.x + r21.y * r21.y + r21.z * r21.z);
    const float cosParallaxRays = (r1.x * r21.x + r1.y * r21.y + r1.z * r21.z) / (n1 * n21);
    if (cosParallaxRays > 0.9998) return -1;
    // Tcw1 = [I | 0], Tcw2 = [R21 | -R21 t12]
    float R21[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R21[3 * i + j] = R12[3 * j + i];
    const Vec3 mt = mat_vec(R21, Vec3{t12[0], t12[1], t12[2]});
    const float T2[3][4] = {{R21[0], R21[1], R21[2], -mt.x}, {R21[3], R21[4], R21[5], -mt.y}, {R21[6], R21[7], R21[8], -mt.z}};
    const float T1[3][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}};
    float A[4][4];
    for (int j = 0; j < 4; ++j) {
        A[0][j] = r1.x * T1[2][j] - T1[0][j];
        A[1][j] = r1.y * T1[2][j] - T1[1][j];
        A[2][j] = r2.x * T2[2][j] - T2[0][j];
        A[3][j] = r2.y * T2[2][j] - T2[1][j];
    }
    float h[4];
    null_vector4(A, h);
    const Vec3 x3D{h[0] / h[3], h[1] / h[3], h[2] / h[3]};
    const float z1 = x3D.z;
    if (z1 <= 0) return -2;
    const float z2 = R21[6] * x3D.x + R21[7] * x3D.y + R21[8] * x3D.z + T2[2][3];
    if (z2 <= 0) return -3;
    float pu, pv;
    project(c1, x3D, &pu, &pv);
    const float errX1 = pu - u1, errY1 = pv - v1;
    if ((errX1 * errX1 + errY1 * errY1) > 5.991 * sigmaLevel) return -4;
    const Vec3 rx = mat_vec(R21, x3D);
    const Vec3 x3D2{rx.x + T2[0][3], rx.y + T2[1][3], rx.z + T2[2][3]};
    project(c2, x3D2, &pu, &pv);
    const float errX2 = pu - u2, errY2 = pv - v2;
    if ((errX2 * errX2 + errY2 * errY2) > 5.991 * unc) return -5;
    if (p3D) *p3D = x3D;
    return z1;
}

// F12 = K1^-T [t12]x R12 K2^-1 (Pinhole.cpp:121-124), row-major.
inline void fundamental_matrix(const Camera& c1, const Camera& c2, const float* R12, const float* t12, float* F) {
    const float tx[9] = {0.f, -t12[2], t12[1], t12[2], 0.f, -t12[0], -t12[1], t12[0], 0.f};
    // K^-1 = [1/fx 0 -cx/fx; 0 1/fy -cy/fy; 0 0 1]
    const float k1it[9] = {1.f / c1.p[0], 0.f, 0.f, 0.f, 1.f / c1.p[1], 0.f, -c1.p[2] / c1.p[0], -c1.p[3] / c1.p[1], 1.f};   // (K1^T)^-1
    const float k2i[9] = {1.f / c2.p[0], 0.f, -c2.p[2] / c2.p[0], 0.f, 1.f / c2.p[1], -c2.p[3] / c2.p[1], 0.f, 0.f, 1.f};
    float a[9], b[9];
    auto mul = [](const float* X, const float* Y, float* Z) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) Z[3 * i + j] = X[3 * i] * Y[j] + X[3 * i + 1] * Y[3 + j] + X[3 * i + 2] * Y[6 + j];
    };
    mul(k1it, tx, a);
    mul(a, R12, b);
    mul(b, k2i, F);
}

// Pinhole::epipolarConstrain with the fundamental matrix given.
inline bool pinhole_epipolar(const float* F12, float u1, float v1, float u2, float v2, float unc) {
    const float a = u1 * F12[0] + v1 * F12[3] + F12[6];
    const float b = u1 * F12[1] + v1 * F12[4] + F12[7];
    const float c = u1 * F12[2] + v1 * F12[5] + F12[8];
    const float num = a * u2 + b * v2 + c;
    const float den = a * a + b * b;
    if (den == 0) return false;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * unc;
}

// pCamera1->epipolarConstrain(pCamera2, kp1, kp2, R12, t12, sigmaLevel, unc), dispatched on camera 1's type.
inline bool epipolar_constrain(const Camera& c1, const Camera& c2, float u1, float v1, float u2, float v2, const float* R12,
                               const float* t12, float sigmaLevel, float unc) {
    if (c1.type == kKannalaBrandt8)
        return triangulate_matches(c1, c2, u1, v1, u2, v2, R12, t12, sigmaLevel, unc) > 0.0001f;
    float F[9];
    fundamental_matrix(c1, c2, R12, t12, F);
    return pinhole_epipolar(F, u1, v1, u2, v2, unc);
}

}  // namespace orbx_gates
#endif
