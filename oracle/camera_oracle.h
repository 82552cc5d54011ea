// CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Declarations of oracle/camera_oracle.cpp for the oracle's matcher loops.
#ifndef ORACLE_CAMERA_ORACLE_H_
#define ORACLE_CAMERA_ORACLE_H_

namespace camo {

enum { kPinhole = 0, kKannalaBrandt8 = 1 };

// GeometricCamera::mvParameters widened to binary64: {fx, fy, cx, cy} (+ {k0, k1, k2, k3} for KannalaBrandt8)
struct Cam {
    int type;
    double p[8];
};

Cam make_cam(int type, const float* params);
void unproject(const Cam& c, double u, double v, double ray[3]);
void project(const Cam& c, const double P[3], double uv[2]);
// Depth in camera 1 or the reference's negative rejection code; *margin_out (may be null) = distance of the nearest
// tested quantity from its threshold in units of its rounding band (>= 1: the decision does not depend on rounding).
double triangulate_matches(const Cam& c1, const Cam& c2, double u1, double v1, double u2, double v2, const double R12[9],
                           const double t12[3], double sigmaLevel, double unc, double p3D[3], double* margin_out);
bool epipolar_constrain(const Cam& c1, const Cam& c2, double u1, double v1, double u2, double v2, const double R12[9],
                        const double t12[3], double sigmaLevel, double unc, double* margin_out);

}  // namespace camo
#endif
