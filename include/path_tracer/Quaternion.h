// Quaternion.h — double-precision rotation quaternion q = w + xi + yj + zk
// (same interface as the reference's class Quaternion, reference
// include/Quaternion.h:14-190, src/Quaternion.cpp).
//
// The renderer consumes orientations as float (x, y, z, w) in
// GPUEllipsoid::rotation; the demo builds them with fromAxisAngle
// (reference src/main_vulkan.cpp:79-81, 349-356).
#ifndef PATH_TRACER_QUATERNION_H
#define PATH_TRACER_QUATERNION_H

#include "Vec3.h"

#include <cmath>
#include <stdexcept>

class Quaternion {
public:
  double w, x, y, z;

  Quaternion() : w(1.0), x(0.0), y(0.0), z(0.0) {}
  Quaternion(double w_, double x_, double y_, double z_) : w(w_), x(x_), y(y_), z(z_) {}

  // Rotation by angleRadians about `axis` (normalised here).
  static Quaternion fromAxisAngle(const Vec3 &axis, double angleRadians) {
    const Vec3 n = axis.normalized();
    const double s = std::sin(0.5 * angleRadians);
    return Quaternion(std::cos(0.5 * angleRadians), n.x * s, n.y * s, n.z * s);
  }

  // ZYX convention: roll about X, pitch about Y, yaw about Z.
  static Quaternion fromEulerAngles(double roll, double pitch, double yaw) {
    const double cr = std::cos(0.5 * roll), sr = std::sin(0.5 * roll);
    const double cp = std::cos(0.5 * pitch), sp = std::sin(0.5 * pitch);
    const double cy = std::cos(0.5 * yaw), sy = std::sin(0.5 * yaw);
    return Quaternion(cr * cp * cy + sr * sp * sy, sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy,
                      cr * cp * sy - sr * sp * cy);
  }

  // From a row-major rotation matrix m[row][col] (Shepperd's method: divide by
  // the largest of the four candidate components).
  static Quaternion fromRotationMatrix(double m00, double m01, double m02, double m10, double m11, double m12,
                                       double m20, double m21, double m22) {
    Quaternion q;
    const double trace = m00 + m11 + m22;
    if (trace > 0.0) {
      const double s = 0.5 / std::sqrt(trace + 1.0);
      q.w = 0.25 / s;
      q.x = (m21 - m12) * s;
      q.y = (m02 - m20) * s;
      q.z = (m10 - m01) * s;
    } else if (m00 > m11 && m00 > m22) {
      const double s = 2.0 * std::sqrt(1.0 + m00 - m11 - m22);
      q.w = (m21 - m12) / s;
      q.x = 0.25 * s;
      q.y = (m01 + m10) / s;
      q.z = (m02 + m20) / s;
    } else if (m11 > m22) {
      const double s = 2.0 * std::sqrt(1.0 + m11 - m00 - m22);
      q.w = (m02 - m20) / s;
      q.x = (m01 + m10) / s;
      q.y = 0.25 * s;
      q.z = (m12 + m21) / s;
    } else {
      const double s = 2.0 * std::sqrt(1.0 + m22 - m00 - m11);
      q.w = (m10 - m01) / s;
      q.x = (m02 + m20) / s;
      q.y = (m12 + m21) / s;
      q.z = 0.25 * s;
    }
    return q;
  }

  Quaternion operator+(const Quaternion &q) const { return Quaternion(w + q.w, x + q.x, y + q.y, z + q.z); }
  Quaternion operator-(const Quaternion &q) const { return Quaternion(w - q.w, x - q.x, y - q.y, z - q.z); }
  // Hamilton product.
  Quaternion operator*(const Quaternion &q) const {
    return Quaternion(w * q.w - x * q.x - y * q.y - z * q.z, w * q.x + x * q.w + y * q.z - z * q.y,
                      w * q.y - x * q.z + y * q.w + z * q.x, w * q.z + x * q.y - y * q.x + z * q.w);
  }
  Quaternion operator*(double s) const { return Quaternion(w * s, x * s, y * s, z * s); }
  Quaternion operator/(double s) const {
    if (s == 0.0) throw std::invalid_argument("Division by zero");
    return Quaternion(w / s, x / s, y / s, z / s);
  }
  Quaternion &operator+=(const Quaternion &q) { return *this = *this + q; }
  Quaternion &operator*=(const Quaternion &q) { return *this = *this * q; }
  Quaternion &operator*=(double s) { return *this = *this * s; }

  double magnitudeSquared() const { return w * w + x * x + y * y + z * z; }
  double magnitude() const { return std::sqrt(magnitudeSquared()); }

  // A zero quaternion is left as it is.
  Quaternion &normalize() {
    const double m = magnitude();
    if (m > 0.0) { w /= m; x /= m; y /= m; z /= m; }
    return *this;
  }
  Quaternion normalized() const { Quaternion q = *this; q.normalize(); return q; }

  Quaternion conjugate() const { return Quaternion(w, -x, -y, -z); }
  Quaternion inverse() const {
    const double m2 = magnitudeSquared();
    if (m2 == 0.0) throw std::invalid_argument("Cannot invert a zero quaternion");
    return conjugate() / m2;
  }

  // q * (0, v) * q^-1
  Vec3 rotateVector(const Vec3 &v) const {
    const Quaternion r = (*this) * Quaternion(0.0, v.x, v.y, v.z) * inverse();
    return Vec3(r.x, r.y, r.z);
  }

  static double dot(const Quaternion &a, const Quaternion &b) { return a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z; }

  // Normalised linear blend along the shorter arc; t is clamped to [0, 1].
  static Quaternion lerp(const Quaternion &q1, const Quaternion &q2, double t) {
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    const Quaternion target = dot(q1, q2) < 0.0 ? q2 * -1.0 : q2;
    return (q1 * (1.0 - t) + target * t).normalized();
  }

  // Spherical interpolation along the shorter arc; nearly parallel inputs
  // (cos > 0.9995) fall back to lerp.
  static Quaternion slerp(const Quaternion &q1, const Quaternion &q2, double t) {
    t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
    double c = dot(q1, q2);
    Quaternion target = q2;
    if (c < 0.0) { target = q2 * -1.0; c = -c; }
    if (c > 0.9995) return lerp(q1, target, t);
    const double theta = std::acos(c);
    const double s = std::sin(theta);
    return (q1 * (std::sin((1.0 - t) * theta) / s) + target * (std::sin(t * theta) / s)).normalized();
  }

  // Unit rotation axis; (1, 0, 0) when the angle is (nearly) zero.
  Vec3 getAxis() const {
    const double s = std::sqrt(x * x + y * y + z * z);
    if (s < 1e-10) return Vec3(1.0, 0.0, 0.0);
    return Vec3(x, y, z) / s;
  }
  // Rotation angle in [0, 2*pi].
  double getAngle() const {
    const double m = magnitude();
    if (m == 0.0) return 0.0;
    double c = w / m;
    c = c < -1.0 ? -1.0 : (c > 1.0 ? 1.0 : c);
    return 2.0 * std::acos(c);
  }

  // ZYX convention; pitch saturates at +-pi/2 at the gimbal poles.
  void toEulerAngles(double &roll, double &pitch, double &yaw) const {
    roll = std::atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y));
    const double sp = 2.0 * (w * y - z * x);
    pitch = std::fabs(sp) >= 1.0 ? std::copysign(M_PI / 2.0, sp) : std::asin(sp);
    yaw = std::atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z));
  }
};

inline Quaternion operator*(double s, const Quaternion &q) { return q * s; }

#endif
