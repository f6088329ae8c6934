// Vec3.h — double-precision 3-vector of the scene API
// (same interface as the reference's class Vec3, reference include/Vec3.h:4-37).
#ifndef PATH_TRACER_VEC3_H
#define PATH_TRACER_VEC3_H

#include <cmath>

class Vec3 {
public:
  double x, y, z;

  Vec3() : x(0.0), y(0.0), z(0.0) {}
  Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}

  Vec3 operator+(const Vec3 &v) const { return Vec3(x + v.x, y + v.y, z + v.z); }
  Vec3 operator-(const Vec3 &v) const { return Vec3(x - v.x, y - v.y, z - v.z); }
  Vec3 operator*(double t) const { return Vec3(x * t, y * t, z * t); }
  Vec3 operator/(double t) const { return Vec3(x / t, y / t, z / t); }

  Vec3 &operator+=(const Vec3 &v) { x += v.x; y += v.y; z += v.z; return *this; }
  Vec3 &operator*=(double t) { x *= t; y *= t; z *= t; return *this; }
  Vec3 &operator/=(double t) { x /= t; y /= t; z /= t; return *this; }

  double dot(const Vec3 &v) const { return x * v.x + y * v.y + z * v.z; }
  Vec3 cross(const Vec3 &v) const { return Vec3(y * v.z - z * v.y, z * v.x - x * v.z, x * v.y - y * v.x); }

  double lengthSquared() const { return dot(*this); }
  double length() const { return std::sqrt(lengthSquared()); }

  // A zero vector stays zero (nothing to normalise).
  Vec3 normalized() const {
    const double len = length();
    return len > 0.0 ? *this / len : *this;
  }
  void normalize() { *this = normalized(); }
};

inline Vec3 operator*(double t, const Vec3 &v) { return v * t; }

#endif
