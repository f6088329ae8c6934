// Ellipsoid.h — ellipsoid scene object and its packed record
// (same interface as the reference's include/Ellipsoid.h:11-69, src/Ellipsoid.cpp:3-13).
#ifndef PATH_TRACER_ELLIPSOID_H
#define PATH_TRACER_ELLIPSOID_H

#include "glm_compat.h"

class Material;

// 64 bytes; rotation is a unit quaternion stored (x, y, z, w).
struct GPUEllipsoid {
  glm::vec3 center;
  float padding1;
  glm::vec3 radii;
  int materialIndex;
  glm::vec3 color;
  float padding2;
  glm::vec4 rotation;
};

class Ellipsoid {
public:
  Ellipsoid() noexcept
      : center_(0.0f), radii_(1.0f), color_(1.0f), material_(nullptr), materialIndex_(-1),
        rotation_(0.0f, 0.0f, 0.0f, 1.0f) {}

  Ellipsoid(const glm::vec3 &center, const glm::vec3 &radii, const glm::vec3 &color,
            const Material &material) noexcept
      : center_(center), radii_(radii), color_(color), material_(&material), materialIndex_(-1),
        rotation_(0.0f, 0.0f, 0.0f, 1.0f) {}

  glm::vec3 getCenter() const noexcept { return center_; }
  glm::vec3 getRadii() const noexcept { return radii_; }
  glm::vec3 getColor() const noexcept { return color_; }
  int getMaterialIndex() const noexcept { return materialIndex_; }
  const Material *getMaterial() const noexcept { return material_; }
  glm::vec4 getRotation() const noexcept { return rotation_; }

  void setCenter(const glm::vec3 &center) noexcept { center_ = center; }
  void setRadii(const glm::vec3 &radii) noexcept { radii_ = radii; }
  void setColor(const glm::vec3 &color) noexcept { color_ = color; }
  // A new material invalidates the resolved index.
  void setMaterial(const Material &material) noexcept {
    material_ = &material;
    materialIndex_ = -1;
  }
  void setRotation(const glm::vec4 &quat) noexcept { rotation_ = quat; }

  // Set by SceneManager::prepareForRender.
  void setMaterialIndex(int index) noexcept { materialIndex_ = index; }

  GPUEllipsoid toGPU() const noexcept {
    GPUEllipsoid g{};
    g.center = center_;
    g.padding1 = 0.0f;
    g.radii = radii_;
    g.materialIndex = materialIndex_;
    g.color = color_;
    g.padding2 = 0.0f;
    g.rotation = rotation_;
    return g;
  }

private:
  glm::vec3 center_;
  glm::vec3 radii_;
  glm::vec3 color_;
  const Material *material_;
  int materialIndex_;
  glm::vec4 rotation_;
};

#endif
