// SceneWrappers.h — host-side scene objects and their flattening to the packed
// records the renderer uploads (same interface as the reference's
// include/SceneWrappers.h:16-354 and src/SceneWrappers.cpp:5-65).
#ifndef PATH_TRACER_SCENE_WRAPPERS_H
#define PATH_TRACER_SCENE_WRAPPERS_H

#include "HipRenderer.h"
#include "glm_compat.h"

#include <iostream>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>

// Surface description.  Defaults: white, ambient 0.1, diffuse 0.5, specular 0.5,
// shininess 32, everything else 0, built-in Phong shading.
class Material {
public:
  Material() noexcept = default;

  static Material Diffuse(const glm::vec3 &color, float diffuse = 0.7f, float ambient = 0.1f) noexcept {
    Material m;
    m.color_ = color;
    m.ambient_ = ambient;
    m.diffuse_ = diffuse;
    m.specular_ = 0.1f;
    m.shininess_ = 16.0f;
    return m;
  }
  static Material Specular(const glm::vec3 &color, float specular, float shininess,
                           float reflectivity = 0.0f) noexcept {
    Material m;
    m.color_ = color;
    m.specular_ = specular;
    m.shininess_ = shininess;
    m.reflectivity_ = reflectivity;
    return m;
  }
  static Material Mirror(const glm::vec3 &color = glm::vec3(0.9f), float reflectivity = 0.7f) noexcept {
    Material m;
    m.color_ = color;
    m.diffuse_ = 0.2f;
    m.specular_ = 0.8f;
    m.shininess_ = 128.0f;
    m.reflectivity_ = reflectivity;
    return m;
  }
  static Material Volumetric(const glm::vec3 &scatterColor, float absorptionCoeff) noexcept {
    Material m;
    m.scatterColor_ = scatterColor;
    m.absorptionCoeff_ = absorptionCoeff;
    return m;
  }
  // Emissive surfaces take no ambient or diffuse light.
  static Material Emissive(const glm::vec3 &color, float strength = 1.0f) noexcept {
    Material m;
    m.color_ = color;
    m.ambient_ = 0.0f;
    m.diffuse_ = 0.0f;
    m.emissiveStrength_ = strength;
    return m;
  }
  // shaderPath must be one of the strings in HipRenderer::getShaderPathToIndexMap().
  static Material CustomShader(const glm::vec3 &color, const std::string &shaderPath) noexcept {
    Material m;
    m.color_ = color;
    m.ambient_ = 0.1f;
    m.diffuse_ = 0.7f;
    m.specular_ = 0.3f;
    m.shininess_ = 32.0f;
    m.shaderPath_ = shaderPath;
    return m;
  }

  glm::vec3 getColor() const noexcept { return color_; }
  float getAmbient() const noexcept { return ambient_; }
  float getDiffuse() const noexcept { return diffuse_; }
  float getSpecular() const noexcept { return specular_; }
  float getShininess() const noexcept { return shininess_; }
  float getReflectivity() const noexcept { return reflectivity_; }
  float getTransparency() const noexcept { return transparency_; }
  float getEmissiveStrength() const noexcept { return emissiveStrength_; }
  glm::vec3 getEmissive() const noexcept { return emissive_; }
  glm::vec3 getScatterColor() const noexcept { return scatterColor_; }
  float getAbsorptionCoeff() const noexcept { return absorptionCoeff_; }
  const std::string &getShaderPath() const noexcept { return shaderPath_; }

  void setColor(const glm::vec3 &color) noexcept { color_ = color; }
  void setAmbient(float v) noexcept { ambient_ = v; }
  void setDiffuse(float v) noexcept { diffuse_ = v; }
  void setSpecular(float v) noexcept { specular_ = v; }
  void setShininess(float v) noexcept { shininess_ = v; }
  void setReflectivity(float v) noexcept { reflectivity_ = v; }
  void setTransparency(float v) noexcept { transparency_ = v; }
  void setEmissiveStrength(float v) noexcept { emissiveStrength_ = v; }
  void setEmissive(const glm::vec3 &v) noexcept { emissive_ = v; }
  void setScatterColor(const glm::vec3 &v) noexcept { scatterColor_ = v; }
  void setAbsorptionCoeff(float v) noexcept { absorptionCoeff_ = v; }
  void setShaderPath(const std::string &path) noexcept { shaderPath_ = path; }

  // Packs the five vec4 of GPUMaterial; the shader path becomes its index as a
  // float, 0 (Phong) when the path is empty or unknown.
  GPUMaterial toGPU(const std::unordered_map<std::string, int> &shaderPathToIndex) const noexcept {
    int shaderIndex = 0;
    if (!shaderPath_.empty()) {
      const auto it = shaderPathToIndex.find(shaderPath_);
      if (it != shaderPathToIndex.end()) {
        shaderIndex = it->second;
      } else {
        std::cerr << "Material: unknown surface shader '" << shaderPath_ << "', using Phong; known shaders:";
        for (const auto &kv : shaderPathToIndex) std::cerr << " '" << kv.first << "'->" << kv.second;
        std::cerr << std::endl;
      }
    }
    GPUMaterial g{};
    g.colorAndAmbient = glm::vec4(color_, ambient_);
    g.diffuseSpecularShiny = glm::vec4(diffuse_, specular_, shininess_, reflectivity_);
    g.transparencyEmissive = glm::vec4(transparency_, emissiveStrength_, static_cast<float>(shaderIndex), 0.0f);
    g.emissive = glm::vec4(emissive_, 0.0f);
    g.scatterAndAbsorption = glm::vec4(scatterColor_, absorptionCoeff_);
    return g;
  }

private:
  glm::vec3 color_{1.0f};
  float ambient_ = 0.1f;
  float diffuse_ = 0.5f;
  float specular_ = 0.5f;
  float shininess_ = 32.0f;
  float reflectivity_ = 0.0f;
  float transparency_ = 0.0f;
  float emissiveStrength_ = 0.0f;
  glm::vec3 emissive_{0.0f};
  glm::vec3 scatterColor_{0.0f};
  float absorptionCoeff_ = 0.0f;
  std::string shaderPath_; // empty = built-in Phong
};

// Objects refer to their material by address; SceneManager::prepareForRender
// turns the addresses into indices into one shared material list.
class Sphere {
public:
  Sphere() noexcept : center_(0.0f), radius_(1.0f), color_(1.0f), material_(nullptr), materialIndex_(-1) {}
  Sphere(const glm::vec3 &center, float radius, const glm::vec3 &color, const Material &material) noexcept
      : center_(center), radius_(radius), color_(color), material_(&material), materialIndex_(-1) {}

  glm::vec3 getCenter() const noexcept { return center_; }
  float getRadius() const noexcept { return radius_; }
  glm::vec3 getColor() const noexcept { return color_; }
  int getMaterialIndex() const noexcept { return materialIndex_; }
  const Material *getMaterial() const noexcept { return material_; }

  void setCenter(const glm::vec3 &center) noexcept { center_ = center; }
  void setRadius(float radius) noexcept { radius_ = radius; }
  void setColor(const glm::vec3 &color) noexcept { color_ = color; }
  void setMaterial(const Material &material) noexcept {
    material_ = &material;
    materialIndex_ = -1;
  }
  void setMaterialIndex(int index) noexcept { materialIndex_ = index; }

  GPUSphere toGPU() const noexcept {
    GPUSphere g{};
    g.center = center_;
    g.radius = radius_;
    g.color = color_;
    g.materialIndex = materialIndex_;
    return g;
  }

private:
  glm::vec3 center_;
  float radius_;
  glm::vec3 color_;
  const Material *material_;
  int materialIndex_;
};

class Light {
public:
  Light() noexcept : position_(0.0f), intensity_(1.0f), color_(1.0f) {}
  Light(const glm::vec3 &position, float intensity, const glm::vec3 &color) noexcept
      : position_(position), intensity_(intensity), color_(color) {}

  glm::vec3 getPosition() const noexcept { return position_; }
  float getIntensity() const noexcept { return intensity_; }
  glm::vec3 getColor() const noexcept { return color_; }
  void setPosition(const glm::vec3 &p) noexcept { position_ = p; }
  void setIntensity(float i) noexcept { intensity_ = i; }
  void setColor(const glm::vec3 &c) noexcept { color_ = c; }

  GPULight toGPU() const noexcept {
    GPULight g{};
    g.position = position_;
    g.intensity = intensity_;
    g.color = color_;
    g.padding = 0.0f;
    return g;
  }

private:
  glm::vec3 position_;
  float intensity_;
  glm::vec3 color_;
};

// An axis-aligned box [position + v0, position + v1] filled with a
// resolution.x * y * z grid of uint8 densities (x fastest).
class VolumetricData {
public:
  VolumetricData() noexcept
      : position_(0.0f), scale_(1.0f), v0_(0.0f), v1_(1.0f), resolution_(1), material_(nullptr), materialIndex_(-1) {}
  VolumetricData(const glm::vec3 &position, float scale, const glm::vec3 &v0, const glm::vec3 &v1,
                 const glm::ivec3 &resolution, const Material &material) noexcept
      : position_(position), scale_(scale), v0_(v0), v1_(v1), resolution_(resolution), material_(&material),
        materialIndex_(-1) {}

  glm::vec3 getPosition() const noexcept { return position_; }
  float getScale() const noexcept { return scale_; }
  glm::vec3 getV0() const noexcept { return v0_; }
  glm::vec3 getV1() const noexcept { return v1_; }
  glm::ivec3 getResolution() const noexcept { return resolution_; }
  int getMaterialIndex() const noexcept { return materialIndex_; }
  const Material *getMaterial() const noexcept { return material_; }

  void setPosition(const glm::vec3 &p) noexcept { position_ = p; }
  void setScale(float s) noexcept { scale_ = s; }
  void setV0(const glm::vec3 &v) noexcept { v0_ = v; }
  void setV1(const glm::vec3 &v) noexcept { v1_ = v; }
  void setResolution(const glm::ivec3 &r) noexcept { resolution_ = r; }
  void setMaterial(const Material &material) noexcept {
    material_ = &material;
    materialIndex_ = -1;
  }
  void setMaterialIndex(int index) noexcept { materialIndex_ = index; }

  GPUVolumetricData toGPU() const noexcept {
    GPUVolumetricData g{};
    g.position = position_;
    g.scale = scale_;
    g.v0 = v0_;
    g.v1 = v1_;
    g.resolution_x = resolution_.x;
    g.resolution_y = resolution_.y;
    g.resolution_z = resolution_.z;
    g.materialIndex = materialIndex_;
    return g;
  }

private:
  glm::vec3 position_;
  float scale_;
  glm::vec3 v0_;
  glm::vec3 v1_;
  glm::ivec3 resolution_;
  const Material *material_;
  int materialIndex_;
};

class SceneManager {
public:
  // Collects the distinct materials (by address, in first-seen order over the
  // containers as given) into outMaterials and stores each object's index.
  // Objects without a material are skipped and keep index -1.
  template <typename... ObjectContainers>
  static void prepareForRender(std::vector<Material> &outMaterials, ObjectContainers &...objectContainers) {
    outMaterials.clear();
    std::unordered_map<const Material *, int> seen;
    auto visit = [&](auto &container) {
      for (auto &object : container) {
        const Material *mat = object.getMaterial();
        if (!mat) continue;
        auto found = seen.find(mat);
        if (found == seen.end()) {
          found = seen.emplace(mat, static_cast<int>(outMaterials.size())).first;
          outMaterials.push_back(*mat);
        }
        object.setMaterialIndex(found->second);
      }
    };
    (visit(objectContainers), ...);
  }
};

#endif
