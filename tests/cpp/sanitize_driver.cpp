// sanitize_driver.cpp — host-side checks meant to run under AddressSanitizer + UndefinedBehaviorSanitizer on a
// machine without a GPU (`make -C tests/cpp asan`): every C-ABI entry point's argument validation (what a
// caller can reach before any device work), the scene API's flattening, and a small oracle frame with a volume.
// Prints "sanitize ok" and exits 0 when every expectation holds; the sanitizers abort on a finding.
#include "path_tracer/Ellipsoid.h"
#include "path_tracer/HipRenderer.h"
#include "path_tracer/SceneWrappers.h"
#include "path_tracer/VolumeLoader.h"

#include "../../oracle/pt_oracle.h"

#include <cstdio>
#include <cstring>
#include <vector>

static int failures = 0;
#define EXPECT(cond)                                                        \
  do {                                                                      \
    if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
  } while (0)

int main() {
  // ---- C-ABI argument validation (no renderer can exist without a GPU: every path below returns early) ----
  PtRenderer *r = nullptr;
  EXPECT(pt_create(0, 10, 0, &r) == PT_ERR_INVALID && r == nullptr);
  EXPECT(pt_create(10, -1, 0, &r) == PT_ERR_INVALID);
  EXPECT(pt_create(1 << 16, 1 << 16, 0, &r) == PT_ERR_INVALID);
  EXPECT(pt_create(16, 16, 0, nullptr) == PT_ERR_INVALID);
  EXPECT(std::strlen(pt_last_error()) > 0);
  const int rc = pt_create(16, 16, 0, &r); // a GPU box creates one; a CPU box reports the HIP error
  if (rc == PT_OK) {
    PtPushConstants pc{};
    EXPECT(pt_render(r, &pc) == PT_ERR_STATE); // no scene yet
    EXPECT(pt_render_static(r, &pc, nullptr, nullptr, nullptr) == PT_ERR_INVALID);
    PtGPUSphere s{};
    s.materialIndex = -1; // what SceneManager leaves on an object without a material
    EXPECT(pt_update_scene(r, &s, 1, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0) == PT_ERR_INVALID);
    EXPECT(pt_update_scene(r, nullptr, 1, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0) == PT_ERR_INVALID);
    EXPECT(pt_set_row_bands(r, 0, 0, 2) == PT_ERR_INVALID);
    EXPECT(pt_set_option(r, 12345, 0) == PT_ERR_INVALID);
    pt_destroy(r);
  } else {
    EXPECT(rc == PT_ERR_HIP && r == nullptr);
  }
  PtPushConstants pc{};
  uint64_t counters[PT_CNT_COUNT];
  float ms = 0;
  uint8_t byte = 0;
  EXPECT(pt_render(nullptr, &pc) == PT_ERR_INVALID);
  EXPECT(pt_render_static(nullptr, &pc, &byte, nullptr, nullptr) == PT_ERR_INVALID);
  EXPECT(pt_render_samples(nullptr, &pc, 0, 1, 0) == PT_ERR_INVALID);
  EXPECT(pt_render_counted(nullptr, &pc, counters) == PT_ERR_INVALID);
  EXPECT(pt_update_scene(nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0) == PT_ERR_INVALID);
  EXPECT(pt_reset_voxels(nullptr) == PT_ERR_INVALID);
  EXPECT(pt_present(nullptr) == PT_ERR_INVALID);
  EXPECT(pt_readback_rgba8(nullptr, &byte, 1) == PT_ERR_INVALID);
  EXPECT(pt_readback_f32(nullptr, &ms, 4) == PT_ERR_INVALID);
  EXPECT(pt_set_row_bands(nullptr, 16, 0, 2) == PT_ERR_INVALID);
  EXPECT(pt_local_rows(nullptr) == 0);
  EXPECT(pt_assemble_bands(nullptr, &byte, &byte, 4, 16, 2, 16) == PT_ERR_INVALID);
  EXPECT(pt_device_rgba8(nullptr) == nullptr && pt_device_f32(nullptr) == nullptr);
  EXPECT(pt_set_output(nullptr, nullptr, nullptr) == PT_ERR_INVALID);
  EXPECT(pt_set_stream(nullptr, nullptr) == PT_ERR_INVALID);
  EXPECT(pt_set_slot(nullptr, 0, nullptr, nullptr, nullptr) == PT_ERR_INVALID);
  EXPECT(pt_last_slot(nullptr) == 0);
  EXPECT(pt_timer_begin(nullptr) == PT_ERR_INVALID && pt_timer_end(nullptr) == PT_ERR_INVALID);
  EXPECT(pt_timer_elapsed_ms(nullptr, &ms) == PT_ERR_INVALID);
  EXPECT(pt_set_option(nullptr, PT_OPT_FLOAT_OUTPUT, 1) == PT_ERR_INVALID);
  EXPECT(pt_device_info(nullptr, nullptr, 0, nullptr) == PT_ERR_INVALID);
  EXPECT(pt_last_launch(nullptr, nullptr, nullptr, nullptr) == PT_ERR_INVALID);
  pt_destroy(nullptr);
  // the shader registry (reference src/ShaderCompiler.cpp:93-118)
  EXPECT(pt_shader_count() == 2);
  EXPECT(pt_shader_path(-1) == nullptr && pt_shader_path(pt_shader_count()) == nullptr);
  for (int i = 0; i < pt_shader_count(); ++i) EXPECT(pt_shader_index(pt_shader_path(i)) == i + 1);
  EXPECT(pt_shader_index("shaders/surface_shaders/nothing.glsl") == 0 && pt_shader_index(nullptr) == 0);

  // ---- scene API: dedupe by pointer, first-seen order, toGPU ----
  Material red = Material::Diffuse(glm::vec3(0.8f, 0.2f, 0.2f), 0.7f, 0.1f);
  Material toon = Material::CustomShader(glm::vec3(0.9f, 0.4f, 0.6f), "shaders/surface_shaders/toon.glsl");
  Material vol = Material::Volumetric(glm::vec3(0.8f, 0.6f, 0.4f), 8.0f);
  std::vector<Sphere> spheres{Sphere(glm::vec3(2.0f, 1.5f, 0.0f), 1.0f, glm::vec3(1.0f), red),
                              Sphere(glm::vec3(0.0f, 0.5f, 0.0f), 0.5f, glm::vec3(1.0f), toon)};
  std::vector<Ellipsoid> ellipsoids{Ellipsoid(glm::vec3(-1.0f, 0.8f, -1.0f), glm::vec3(0.5f, 0.8f, 0.3f), glm::vec3(1.0f), red)};
  std::vector<VolumetricData> volumes{VolumetricData(glm::vec3(1.5f, 1.0f, -0.5f), 0.001f, glm::vec3(0.0f), glm::vec3(2.0f),
                                                     glm::ivec3(6, 5, 4), vol)};
  std::vector<Material> materials;
  SceneManager::prepareForRender(materials, spheres, ellipsoids, volumes);
  EXPECT(materials.size() == 3);
  std::unordered_map<std::string, int> shaderIndex;
  for (int i = 0; i < pt_shader_count(); ++i) shaderIndex[pt_shader_path(i)] = i + 1;
  std::vector<PtGPUSphere> gs;
  std::vector<PtGPUEllipsoid> ge;
  std::vector<PtGPUMaterial> gm;
  std::vector<PtGPUVolumetricData> gv;
  auto as = [](const auto &src, auto &dstVec) {
    typename std::remove_reference<decltype(dstVec)>::type::value_type d;
    static_assert(sizeof d == sizeof src, "record layout");
    std::memcpy(&d, &src, sizeof d);
    dstVec.push_back(d);
  };
  for (const Sphere &s : spheres) as(s.toGPU(), gs);
  for (const Ellipsoid &e : ellipsoids) as(e.toGPU(), ge);
  for (const Material &m : materials) as(m.toGPU(shaderIndex), gm);
  for (const VolumetricData &v : volumes) as(v.toGPU(), gv);
  EXPECT(gs[0].materialIndex == 0 && gs[1].materialIndex == 1 && ge[0].materialIndex == 0 && gv[0].materialIndex == 2);
  EXPECT((int)gm[1].transparencyEmissive[2] == 2);

  // ---- the oracle on that scene (8 x 6 pixels, volume included), twice: threads must not change it ----
  std::vector<uint8_t> voxels(6 * 5 * 4);
  for (size_t i = 0; i < voxels.size(); ++i) voxels[i] = (uint8_t)(37 * i);
  PtGPULight light{};
  light.position[0] = 4; light.position[1] = 4; light.position[2] = 4; light.intensity = 1.0f;
  PtoScene sc{gs.data(), ge.data(), gm.data(), &light, gv.data(), voxels.data(), voxels.size()};
  pc.cameraPos[0] = 5.0f; pc.cameraPos[1] = 1.5f; pc.cameraPos[2] = 6.0f;
  pc.numSpheres = 2; pc.numEllipsoids = 1; pc.numLights = 1; pc.numVolumes = 1; pc.maxDepth = 10;
  pc.bgColorTop[0] = 0.4f; pc.bgColorTop[1] = 0.45f; pc.bgColorTop[2] = 1.0f;
  pc.bgColorBottom[0] = pc.bgColorBottom[1] = pc.bgColorBottom[2] = 1.0f;
  const int W = 8, H = 6;
  std::vector<float> f1(W * H * 4), f2(W * H * 4);
  std::vector<uint8_t> b1(W * H * 4), b2(W * H * 4);
  uint64_t c1[PT_CNT_COUNT] = {0}, c2[PT_CNT_COUNT] = {0};
  EXPECT(pto_render(&sc, &pc, W, H, 10, 0, H, f1.data(), b1.data(), c1, 1) == 0);
  EXPECT(pto_render(&sc, &pc, W, H, 10, 0, H, f2.data(), b2.data(), c2, 3) == 0);
  EXPECT(std::memcmp(f1.data(), f2.data(), f1.size() * sizeof(float)) == 0 && b1 == b2);
  EXPECT(c1[0] == c2[0] && c1[0] >= (uint64_t)(W * H));

  // ---- the .dat/.raw loader on files that do not exist ----
  std::vector<uint8_t> data;
  glm::ivec3 res;
  EXPECT(!loadVolumetricData("/nonexistent/walnut.dat", "/nonexistent/walnut.raw", data, res));

  if (failures) { std::printf("%d expectation(s) failed\n", failures); return 1; }
  std::printf("sanitize ok\n");
  return 0;
}
