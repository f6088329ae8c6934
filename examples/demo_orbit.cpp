// demo_orbit.cpp — the reference demo (reference src/main_vulkan.cpp:83-395)
// on the HIP renderer: same scene, same orbiting camera and spinning
// ellipsoids, frames written as PPM files instead of shown in a window.
//
//   demo_orbit [frames=4] [width=800] [height=600] [outdir=.] [dt_ms=8.3333]
//
// Assets are looked up relative to the working directory like the reference
// does: volume/walnut.dat + volume/walnut.raw (without them the scene renders
// without the volume, as the reference does).
#include "path_tracer/Camera.h"
#include "path_tracer/Ellipsoid.h"
#include "path_tracer/HipRenderer.h"
#include "path_tracer/Quaternion.h"
#include "path_tracer/SceneWrappers.h"
#include "path_tracer/Vec3.h"
#include "path_tracer/VolumeLoader.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <vector>

static glm::vec4 quaternionToGlm(const Quaternion &q) {
  return glm::vec4((float)q.x, (float)q.y, (float)q.z, (float)q.w);
}

int main(int argc, char *argv[]) {
  const int frames = argc > 1 ? std::atoi(argv[1]) : 4;
  const int WINDOW_WIDTH = argc > 2 ? std::atoi(argv[2]) : 800;
  const int WINDOW_HEIGHT = argc > 3 ? std::atoi(argv[3]) : 600;
  const std::string outDir = argc > 4 ? argv[4] : ".";
  const float dtMs = argc > 5 ? (float)std::atof(argv[5]) : 1000.0f / 120.0f;
  const int MAX_DEPTH = 5; // the reference passes 5 and its shader ignores it (10 bounces); so does HipRenderer

  HipRenderer renderer;
  if (!renderer.initialize(nullptr, WINDOW_WIDTH, WINDOW_HEIGHT)) return 1;

  const double aspectRatio = static_cast<double>(WINDOW_WIDTH) / WINDOW_HEIGHT;
  Camera camera(Vec3(0, 1.5, 6), Vec3(2, 1.5, 0), Vec3(0, 1, 0), 60.0, aspectRatio);

  std::vector<Sphere> spheres;
  std::vector<Ellipsoid> ellipsoids;
  std::vector<Light> lights;
  std::vector<VolumetricData> volumes;

  Material redMat = Material::Diffuse(glm::vec3(0.8f, 0.2f, 0.2f), 0.7f, 0.1f);
  redMat.setSpecular(0.3f);
  redMat.setShininess(32.0f);
  Material yellowMat = Material::Diffuse(glm::vec3(0.8f, 0.8f, 0.2f), 0.6f, 0.1f);
  yellowMat.setSpecular(0.1f);
  yellowMat.setShininess(16.0f);
  Material greenMat = Material::Emissive(glm::vec3(0.2f, 0.8f, 0.2f), 2.0f);
  Material blueMat = Material::CustomShader(glm::vec3(0.2f, 0.2f, 0.8f), "shaders/surface_shaders/rainbow.glsl");
  blueMat.setDiffuse(0.7f);
  blueMat.setAmbient(0.1f);
  blueMat.setSpecular(0.5f);
  blueMat.setShininess(64.0f);
  blueMat.setEmissiveStrength(1.0f);
  Material mirrorMat = Material::Mirror(glm::vec3(0.9f, 0.9f, 0.9f), 0.9f);
  Material volumetricMat = Material::Volumetric(glm::vec3(0.8f, 0.6f, 0.4f), 8.0f);

  spheres.push_back(Sphere(glm::vec3(0.0f, 0.0f, -1.0f), 0.5f, glm::vec3(0.8f, 0.2f, 0.2f), redMat));
  spheres.push_back(Sphere(glm::vec3(-1.0f, 0.0f, -1.0f), 0.3f, glm::vec3(0.2f, 0.8f, 0.2f), greenMat));
  spheres.push_back(Sphere(glm::vec3(1.0f, 0.0f, -1.0f), 0.3f, glm::vec3(0.2f, 0.2f, 0.8f), blueMat));
  spheres.push_back(Sphere(glm::vec3(0.0f, -100.5f, -1.0f), 100.0f, glm::vec3(0.8f, 0.8f, 0.2f), yellowMat));
  spheres.push_back(Sphere(glm::vec3(0.5f, 0.3f, -0.5f), 0.2f, glm::vec3(0.9f, 0.9f, 0.9f), mirrorMat));
  spheres.push_back(Sphere(glm::vec3(2.0f, 0.5f, 1.5f), 1.0f, glm::vec3(0.8f, 0.8f, 0.2f), yellowMat));

  ellipsoids.push_back(Ellipsoid(glm::vec3(-2.0f, 0.8f, -1.0f), glm::vec3(0.5f, 0.8f, 0.3f),
                                 glm::vec3(0.8f, 0.4f, 0.8f), mirrorMat));
  ellipsoids.push_back(Ellipsoid(glm::vec3(0.0f, 1.2f, -2.0f), glm::vec3(0.6f, 0.4f, 0.4f),
                                 glm::vec3(0.4f, 0.8f, 0.8f), blueMat));

  lights.push_back(Light(glm::vec3(2.0f, 2.0f, 1.0f), 0.5f, glm::vec3(1.0f, 0.9f, 0.8f)));
  lights.push_back(Light(glm::vec3(-2.0f, 1.0f, 0.0f), 0.5f, glm::vec3(0.3f, 0.5f, 1.0f)));
  lights.push_back(Light(glm::vec3(0.0f, -0.2f, 0.5f), 0.5f, glm::vec3(1.0f, 0.4f, 0.4f)));

  std::vector<uint8_t> volumeData;
  glm::ivec3 volumeResolution;
  if (loadVolumetricData("volume/walnut.dat", "volume/walnut.raw", volumeData, volumeResolution)) {
    volumes.push_back(VolumetricData(glm::vec3(1.5f, 1.0f, -0.5f), 0.001f, glm::vec3(0.0f, 0.0f, 0.0f),
                                     glm::vec3(2.0f, 2.0f, 2.0f), volumeResolution, volumetricMat));
  } else {
    std::cerr << "Warning: Failed to load volumetric data, continuing without volume" << std::endl;
  }

  std::vector<Material> materials;
  SceneManager::prepareForRender(materials, spheres, ellipsoids, volumes);

  std::vector<GPUSphere> gpuSpheres(spheres.size());
  std::vector<GPUEllipsoid> gpuEllipsoids(ellipsoids.size());
  std::vector<GPUMaterial> gpuMaterials(materials.size());
  std::vector<GPULight> gpuLights(lights.size());
  std::vector<GPUVolumetricData> gpuVolumes(volumes.size());
  std::transform(spheres.begin(), spheres.end(), gpuSpheres.begin(), [](const Sphere &s) { return s.toGPU(); });
  std::transform(ellipsoids.begin(), ellipsoids.end(), gpuEllipsoids.begin(),
                 [](const Ellipsoid &e) { return e.toGPU(); });
  const auto &shaderPathToIndex = renderer.getShaderPathToIndexMap();
  std::transform(materials.begin(), materials.end(), gpuMaterials.begin(),
                 [&](const Material &m) { return m.toGPU(shaderPathToIndex); });
  std::transform(lights.begin(), lights.end(), gpuLights.begin(), [](const Light &l) { return l.toGPU(); });
  std::transform(volumes.begin(), volumes.end(), gpuVolumes.begin(),
                 [](const VolumetricData &v) { return v.toGPU(); });

  renderer.updateScene(gpuSpheres, gpuEllipsoids, gpuMaterials, gpuLights, gpuVolumes, volumeData);

  PushConstants pushConst{};
  pushConst.numSpheres = static_cast<int>(gpuSpheres.size());
  pushConst.numEllipsoids = static_cast<int>(gpuEllipsoids.size());
  pushConst.numLights = static_cast<int>(gpuLights.size());
  pushConst.numVolumes = static_cast<int>(gpuVolumes.size());
  pushConst.maxDepth = MAX_DEPTH;
  pushConst.bgColorBottom = glm::vec3(1.0f, 1.0f, 1.0f);
  pushConst.bgColorTop = glm::vec3(0.4f, 0.45f, 1.0f);

  double theta = 0.0;
  const double orbitRadius = 3.0;
  float time = 0.0f;
  const auto t0 = std::chrono::steady_clock::now();
  for (int frame = 0; frame < frames; ++frame) {
    camera.setLookFrom(Vec3(2.0 + orbitRadius * std::cos(theta), 1.5, 6.0 + orbitRadius * std::sin(theta)));
    theta += 1.0 / 180.0;
    for (size_t i = 0; i < ellipsoids.size(); i++) {
      ellipsoids[i].setRotation(quaternionToGlm(Quaternion::fromAxisAngle(Vec3(2, 1, 0), theta)));
      gpuEllipsoids[i] = ellipsoids[i].toGPU();
    }
    renderer.updateScene(gpuSpheres, gpuEllipsoids, gpuMaterials, gpuLights, gpuVolumes, volumeData);

    const glm::vec3 eye((float)camera.origin.x, (float)camera.origin.y, (float)camera.origin.z);
    pushConst.cameraMatrix = glm::perspective(glm::radians(60.0f), static_cast<float>(aspectRatio), 0.1f, 100.0f) *
                             glm::lookAt(eye, glm::vec3(2.0f, 1.5f, 0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
    pushConst.cameraPos = eye;
    pushConst.time = time;

    renderer.render(pushConst);
    renderer.present();
    char name[64];
    std::snprintf(name, sizeof name, "/frame_%04d.ppm", frame);
    if (!renderer.writePPM(outDir + name)) {
      std::cerr << "could not write " << outDir << name << std::endl;
      return 1;
    }
    time += dtMs;
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::cout << frames << " frames, " << WINDOW_WIDTH << "x" << WINDOW_HEIGHT << ", " << sec * 1e3 / std::max(frames, 1)
            << " ms per frame including scene upload, readback and PPM write" << std::endl;
  renderer.shutdown();
  return 0;
}
