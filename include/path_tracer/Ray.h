// Ray.h — origin + direction (reference include/Ray.h:6-17).
#ifndef PATH_TRACER_RAY_H
#define PATH_TRACER_RAY_H

#include "Vec3.h"

class Ray {
public:
  Vec3 origin;
  Vec3 direction;

  Ray() : origin(), direction() {}
  Ray(const Vec3 &origin_, const Vec3 &direction_) : origin(origin_), direction(direction_) {}

  // Point at parameter t along the ray.
  Vec3 at(double t) const { return origin + direction * t; }
};

#endif
