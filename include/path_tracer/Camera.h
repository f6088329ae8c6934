// Camera.h — double-precision look-at camera of the scene API
// (same interface as the reference's class Camera, reference include/Camera.h:7-33,
// src/Camera.cpp:5-56).
//
// Only `origin` reaches the renderer (as PushConstants::cameraPos, reference
// src/main_vulkan.cpp:363-371); the ray loop rebuilds its own basis from it with
// a fixed target (2, 1.5, 0) and spans tan(fov/2) over the full image height,
// half the extent getRay() spans (reference shaders/raytrace.comp:121-140).
#ifndef PATH_TRACER_CAMERA_H
#define PATH_TRACER_CAMERA_H

#include "Ray.h"
#include "Vec3.h"

#include <cmath>

class Camera {
  double m_viewportHeight = 2.0;
  double m_viewportWidth = 4.0;
  Vec3 m_lookFrom;
  Vec3 m_lookAt;
  Vec3 m_vup;

  void computeCoords() {
    const Vec3 back = (m_lookFrom - m_lookAt).normalized(); // camera looks along -back
    const Vec3 side = m_vup.cross(back).normalized();
    const Vec3 upward = back.cross(side);
    origin = m_lookFrom;
    horizontal = m_viewportWidth * side;
    vertical = m_viewportHeight * upward;
    lowerLeftCorner = origin - horizontal / 2.0 - vertical / 2.0 - back;
  }

public:
  Vec3 origin;
  Vec3 lowerLeftCorner;
  Vec3 horizontal;
  Vec3 vertical;

  // Default: pinhole at the origin looking down -z through a 4 x 2 viewport.
  Camera() : origin(0, 0, 0), lowerLeftCorner(-2, -1, -1), horizontal(4, 0, 0), vertical(0, 2, 0) {}

  // vfov is the vertical field of view in degrees.
  Camera(const Vec3 &lookFrom, const Vec3 &lookAt, const Vec3 &vup, double vfov, double aspectRatio)
      : m_lookFrom(lookFrom), m_lookAt(lookAt), m_vup(vup) {
    const double halfHeight = std::tan(vfov * M_PI / 180.0 / 2.0);
    m_viewportHeight = 2.0 * halfHeight;
    m_viewportWidth = aspectRatio * m_viewportHeight;
    computeCoords();
  }

  // u, v in [0, 1] across the viewport, (0, 0) = lower left.
  Ray getRay(double u, double v) const {
    const Vec3 through = lowerLeftCorner + u * horizontal + v * vertical - origin;
    return Ray(origin, through.normalized());
  }

  void setLookFrom(const Vec3 &lookFrom) { m_lookFrom = lookFrom; computeCoords(); }
  void setLookAt(const Vec3 &lookAt) { m_lookAt = lookAt; computeCoords(); }
  void setVup(const Vec3 &vup) { m_vup = vup; computeCoords(); }
};

#endif
