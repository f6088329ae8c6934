// HipRenderer.h — the renderer boundary of the scene API on MI355X.
//
// Same public interface as the reference's class VulkanRenderer (reference
// include/VulkanRenderer.h:73-97) and the same record types (reference
// include/VulkanRenderer.h:16-71); a program written against the reference
// compiles against this header after replacing the type name.  Everything
// forwards to the C-ABI of pt_hip.h (libpt_hip.so), so no HIP header is needed
// to use it.
//
// Differences, all forced by leaving Vulkan/SDL behind:
//   * initialize() takes an opaque window pointer that is ignored (there is
//     no swapchain); the frame is read back with readbackRGBA8()/readbackFloat()
//     or written with writePPM().
//   * present() waits for the frame instead of blitting it.
//   * render() runs 10 bounces whatever PushConstants::maxDepth says, like the
//     reference shader, which ignores the field (reference
//     shaders/raytrace.comp:603); setHonourMaxDepth(true) makes maxDepth the limit.
//   * updateScene() reports a rejected scene on std::cerr and render() then refuses to
//     draw until a scene is accepted (the reference fails silently and draws garbage).
//   * getOutputImage() returns the device pointer of the rgba8 frame.
#ifndef PATH_TRACER_HIP_RENDERER_H
#define PATH_TRACER_HIP_RENDERER_H

#include "../pt_hip.h"
#include "Ellipsoid.h"
#include "glm_compat.h"

#include <cstdint>
#include <cstdio>
#include <iostream>
#include <string>
#include <unordered_map>
#include <vector>

struct GPUSphere {
  glm::vec3 center;
  float radius;
  glm::vec3 color;
  int materialIndex;
};

// GPUEllipsoid is defined in Ellipsoid.h

struct GPUMaterial {
  glm::vec4 colorAndAmbient;      // color.xyz, ambient.w
  glm::vec4 diffuseSpecularShiny; // diffuse.x, specular.y, shininess.z, reflectivity.w
  glm::vec4 transparencyEmissive; // transparency.x, emissiveStrength.y, shaderFunctionIndex.z (as float), pad.w
  glm::vec4 emissive;             // emissive.xyz, pad.w
  glm::vec4 scatterAndAbsorption; // scatterColor.xyz, absorptionCoeff.w
};

struct GPULight {
  glm::vec3 position;
  float intensity;
  glm::vec3 color;
  float padding;
};

struct GPUVolumetricData {
  glm::vec3 position;
  float scale;
  glm::vec3 v0;
  int resolution_x;
  glm::vec3 v1;
  int resolution_y;
  int resolution_z;
  int materialIndex;
  int padding[2];
};

struct PushConstants {
  glm::mat4 cameraMatrix;  // offset 0   (not used by the ray loop)
  glm::vec3 cameraPos;     // offset 64
  float time;              // offset 76  (milliseconds)
  int numSpheres;          // offset 80
  int numEllipsoids;       // offset 84
  int numLights;           // offset 88
  int numVolumes;          // offset 92
  int maxDepth;            // offset 96
  int padding1;
  int padding2;
  int padding3;
  glm::vec3 bgColorTop;    // offset 112
  float padding4;
  glm::vec3 bgColorBottom; // offset 128
  float padding5;
};

static_assert(sizeof(GPUSphere) == sizeof(PtGPUSphere), "GPUSphere layout");
static_assert(sizeof(GPUEllipsoid) == sizeof(PtGPUEllipsoid), "GPUEllipsoid layout");
static_assert(sizeof(GPUMaterial) == sizeof(PtGPUMaterial), "GPUMaterial layout");
static_assert(sizeof(GPULight) == sizeof(PtGPULight), "GPULight layout");
static_assert(sizeof(GPUVolumetricData) == sizeof(PtGPUVolumetricData), "GPUVolumetricData layout");
static_assert(sizeof(PushConstants) == sizeof(PtPushConstants), "PushConstants layout");

class HipRenderer {
public:
  HipRenderer() = default;
  ~HipRenderer() { shutdown(); }
  HipRenderer(const HipRenderer &) = delete;
  HipRenderer &operator=(const HipRenderer &) = delete;

  // Extension: make PushConstants::maxDepth (> 0) the bounce limit.  Off by default: the reference
  // shader always runs 10 bounces (reference shaders/raytrace.comp:603-605).
  void setHonourMaxDepth(bool on) {
    honourMaxDepth_ = on;
    if (handle_) pt_set_option(handle_, PT_OPT_HONOUR_MAX_DEPTH, on ? 1 : 0);
  }
  bool honourMaxDepth() const { return honourMaxDepth_; }
  // Extension: present() after a render() waits for the frames before that one only and the readbacks show the
  // frame before it, so a loop of updateScene / render / present keeps one frame drawing while the host prepares the
  // next (PT_OPT_PRESENT_LAG; the reference's present() drains the queue, src/VulkanRenderer.cpp:1355-1356, which
  // remains the default).  A present() without a render() since the last one shows the last frame.
  void setPresentLag(bool on) {
    presentLag_ = on;
    if (handle_) pt_set_option(handle_, PT_OPT_PRESENT_LAG, on ? 1 : 0);
  }
  // False after an updateScene() that the library rejected (until one is accepted).
  bool sceneOk() const { return sceneOk_; }

  // `window` is ignored.  Failure: false + message on std::cerr, like the reference.
  bool initialize(void *window, int width, int height, int device = 0) {
    (void)window;
    shutdown();
    if (pt_create(width, height, device, &handle_) != PT_OK) {
      std::cerr << "HipRenderer initialization failed: " << pt_last_error() << std::endl;
      handle_ = nullptr;
      return false;
    }
    width_ = width;
    height_ = height;
    sceneOk_ = true;
    pt_set_option(handle_, PT_OPT_HONOUR_MAX_DEPTH, honourMaxDepth_ ? 1 : 0);
    pt_set_option(handle_, PT_OPT_PRESENT_LAG, presentLag_ ? 1 : 0);
    shaderPathToIndex_.clear();
    for (int i = 0; i < pt_shader_count(); ++i) shaderPathToIndex_[pt_shader_path(i)] = i + 1;
    return true;
  }

  void shutdown() {
    leaveGather();
    if (handle_) pt_destroy(handle_);
    handle_ = nullptr;
  }

  // Copies the scene; may be called every frame.  Voxels are uploaded once.  An empty vector keeps
  // what the previous call supplied for it, as the reference's buffers do (it skips the copy of
  // an empty vector, reference src/VulkanRenderer.cpp:824-987).  A scene the library rejects (an
  // object without a material: SceneManager leaves index -1, which the reference shader reads out
  // of bounds) is reported on std::cerr and blocks render() until a scene is accepted.
  void updateScene(const std::vector<GPUSphere> &spheres, const std::vector<GPUEllipsoid> &ellipsoids,
                   const std::vector<GPUMaterial> &materials, const std::vector<GPULight> &lights,
                   const std::vector<GPUVolumetricData> &volumes,
                   const std::vector<uint8_t> &voxelData = std::vector<uint8_t>()) {
    if (!handle_) return;
    if (!spheres.empty()) spheres_ = spheres;
    if (!ellipsoids.empty()) ellipsoids_ = ellipsoids;
    if (!materials.empty()) materials_ = materials;
    if (!lights.empty()) lights_ = lights;
    if (!volumes.empty()) volumes_ = volumes;
    const int rc = pt_update_scene(
        handle_, reinterpret_cast<const PtGPUSphere *>(spheres_.data()), (int)spheres_.size(),
        reinterpret_cast<const PtGPUEllipsoid *>(ellipsoids_.data()), (int)ellipsoids_.size(),
        reinterpret_cast<const PtGPUMaterial *>(materials_.data()), (int)materials_.size(),
        reinterpret_cast<const PtGPULight *>(lights_.data()), (int)lights_.size(),
        reinterpret_cast<const PtGPUVolumetricData *>(volumes_.data()), (int)volumes_.size(), voxelData.data(),
        voxelData.size());
    sceneOk_ = rc == PT_OK;
    if (!sceneOk_) std::cerr << "HipRenderer::updateScene: " << pt_last_error() << " (render() is blocked until a scene is accepted)" << std::endl;
  }

  // Enqueues one frame and returns.
  void render(const PushConstants &pushConstants) {
    if (!handle_) return;
    if (!sceneOk_) {
      std::cerr << "HipRenderer::render: the last updateScene() was rejected; nothing drawn" << std::endl;
      return;
    }
    const PushConstants &pc = pushConstants;
    if (pt_render(handle_, reinterpret_cast<const PtPushConstants *>(&pc)) != PT_OK)
      std::cerr << "HipRenderer::render: " << pt_last_error() << std::endl;
  }

  // Not in the reference: mean of sub-pixel samples firstSample .. firstSample + numSamples - 1
  // (sample 0 is the reference's pixel-corner sample); keep continues the previous call's sum.
  void renderSamples(const PushConstants &pushConstants, int firstSample, int numSamples, bool keep = false) {
    if (!handle_ || !sceneOk_) return;
    const PushConstants &pc = pushConstants;
    if (pt_render_samples(handle_, reinterpret_cast<const PtPushConstants *>(&pc), firstSample, numSamples, keep ? 1 : 0) != PT_OK)
      std::cerr << "HipRenderer::renderSamples: " << pt_last_error() << std::endl;
  }

  // Waits for the frames enqueued so far.
  void present() {
    if (handle_ && pt_present(handle_) != PT_OK) std::cerr << "HipRenderer::present: " << pt_last_error() << std::endl;
  }

  // Device pointer of the rgba8 frame (the reference returns its VkImage).
  void *getOutputImage() const { return handle_ ? pt_device_rgba8(handle_) : nullptr; }

  const std::unordered_map<std::string, int> &getShaderPathToIndexMap() const { return shaderPathToIndex_; }

  // ---- multi-GPU: one process (or thread) per GPU, interleaved row bands, gather to rank 0 over RCCL ----
  // Rank 0 makes the id and hands it to the other ranks (file, socket, MPI ...); then every rank calls
  // joinGather (collective).  After that render() draws this rank's bands only, gather() sends them to rank 0
  // (rank 0: receives, de-interleaves) on the frame's stream, and rank 0 reads the whole frame with
  // readbackAssembledRGBA8().  The reference is single-GPU (reference src/main_vulkan.cpp:105-389).
  static bool gatherUniqueId(std::vector<uint8_t> &id) {
    id.assign(PT_COMM_ID_BYTES, 0);
    return pt_comm_unique_id(id.data(), id.size()) == PT_OK;
  }
  bool joinGather(const std::vector<uint8_t> &id, int rank, int world, int device = 0, int bandRows = 16) {
    if (!handle_) return false;
    leaveGather();
    if (pt_comm_create(id.data(), id.size(), rank, world, device, &comm_) != PT_OK ||
        pt_set_row_bands(handle_, bandRows, rank, world) != PT_OK) {
      std::cerr << "HipRenderer::joinGather: " << pt_last_error() << std::endl;
      leaveGather();
      return false;
    }
    rank_ = rank;
    return true;
  }
  void leaveGather() {
    if (comm_) pt_comm_destroy(comm_);
    comm_ = nullptr;
  }
  void gather() {
    if (handle_ && comm_ && pt_gather_bands(handle_, comm_, 4, nullptr) != PT_OK)
      std::cerr << "HipRenderer::gather: " << pt_last_error() << std::endl;
  }
  bool readbackAssembledRGBA8(std::vector<uint8_t> &out) { // rank 0
    if (!handle_ || rank_ != 0) return false;
    out.resize((size_t)width_ * height_ * 4);
    return pt_readback_assembled(handle_, out.data(), out.size()) == PT_OK;
  }

  // ---- beyond the reference interface ----
  int width() const { return width_; }
  int height() const { return height_; }
  PtRenderer *handle() const { return handle_; }

  bool readbackRGBA8(std::vector<uint8_t> &out) {
    if (!handle_) return false;
    out.resize((size_t)width_ * pt_local_rows(handle_) * 4);
    return pt_readback_rgba8(handle_, out.data(), out.size()) == PT_OK;
  }
  bool readbackFloat(std::vector<float> &out) {
    if (!handle_) return false;
    out.resize((size_t)width_ * pt_local_rows(handle_) * 4);
    return pt_readback_f32(handle_, out.data(), out.size() * sizeof(float)) == PT_OK;
  }
  // Binary PPM (P6) of the current frame.
  bool writePPM(const std::string &path) {
    std::vector<uint8_t> px;
    if (!readbackRGBA8(px)) return false;
    std::FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const int rows = pt_local_rows(handle_);
    std::fprintf(f, "P6\n%d %d\n255\n", width_, rows);
    for (size_t i = 0; i < (size_t)width_ * rows; ++i) std::fwrite(&px[4 * i], 1, 3, f);
    std::fclose(f);
    return true;
  }

private:
  PtRenderer *handle_ = nullptr;
  PtComm *comm_ = nullptr;
  int rank_ = 0;
  int width_ = 0, height_ = 0;
  bool honourMaxDepth_ = false, sceneOk_ = true, presentLag_ = false;
  std::vector<GPUSphere> spheres_;
  std::vector<GPUEllipsoid> ellipsoids_;
  std::vector<GPUMaterial> materials_;
  std::vector<GPULight> lights_;
  std::vector<GPUVolumetricData> volumes_;
  std::unordered_map<std::string, int> shaderPathToIndex_;
};

#endif
