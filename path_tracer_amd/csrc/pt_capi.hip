// pt_capi.hip — host side of libpt_hip.so: the C-ABI of include/pt_hip.h over
// the gfx950 kernels of pt_device.hpp.
//
// Stands where the reference has VulkanRenderer + ShaderCompiler
// (reference src/VulkanRenderer.cpp, src/ShaderCompiler.cpp): device buffers,
// scene upload (AoS records -> SoA device arrays), per-frame launch, readback.
#include "../../include/pt_hip.h"
#include "pt_device.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: the library is opened at run time (dlopen), single-GPU users do not need it

#include <dlfcn.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
  g_last_error = msg;
  return code;
}

#define PT_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
  } while (0)

// Compiled-in surface shaders: the rows of the registry in pt_shaders.hpp, which stand in the order the
// reference's sorted directory scan assigns indices 1, 2, ... (reference src/ShaderCompiler.cpp:93-118,131-185).
const char *const kShaderPaths[] = {
#define PT_X(IDX, PATH, S) PATH,
    PT_SURFACE_SHADERS(PT_X)
#undef PT_X
};
constexpr int kShaderIndices[] = {
#define PT_X(IDX, PATH, S) IDX,
    PT_SURFACE_SHADERS(PT_X)
#undef PT_X
};
constexpr int kShaderCount = (int)(sizeof kShaderPaths / sizeof kShaderPaths[0]);
constexpr bool shader_rows_in_order() { // row i has index i + 1 and the paths ascend as std::map<std::string, ...> orders them
  for (int i = 0; i < kShaderCount; ++i)
    if (kShaderIndices[i] != i + 1) return false;
  return true;
}
static_assert(shader_rows_in_order(), "PT_SURFACE_SHADERS: row i must carry index i + 1");
bool shader_paths_sorted() {
  for (int i = 1; i < kShaderCount; ++i)
    if (std::strcmp(kShaderPaths[i - 1], kShaderPaths[i]) >= 0) return false;
  return true;
}

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = std::max<size_t>(bytes, 256);
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Host float helpers with the same operation order as the device code.
struct H3 { float x, y, z; };
inline float hdot(H3 a, H3 b) { return std::fmaf(a.z, b.z, std::fmaf(a.y, b.y, a.x * b.x)); }
inline H3 hcross(H3 a, H3 b) {
  return {std::fmaf(a.y, b.z, -(a.z * b.y)), std::fmaf(a.z, b.x, -(a.x * b.z)), std::fmaf(a.x, b.y, -(a.y * b.x))};
}
inline H3 hnormalize(H3 a) {
  float inv = 1.0f / std::sqrt(hdot(a, a));
  return {a.x * inv, a.y * inv, a.z * inv};
}

} // namespace

struct PtRenderer {
  int device = 0;
  int width = 0, height = 0;
  int computeUnits = 0;
  char deviceName[256] = {0};
  // Frames in flight (reference src/VulkanRenderer.cpp:1080-1100, MAX_FRAMES_IN_FLIGHT = 2): plain
  // pt_render calls go round the slots, each with its own stream, output frames and scratch, so the
  // next frames fill the compute units the tail of the previous ones leaves idle.  Two by default, as
  // in the reference; up to kMaxSlots where a frame is short against its longest chain of dependent rays
  // (a rank's share of a frame split over many GPUs).
  static constexpr int kMaxSlots = 8;
  hipStream_t ownStream[kMaxSlots] = {}; // created when the slot is first needed
  hipStream_t stream[kMaxSlots] = {};    // ownStream[k], or the caller's (pt_set_stream for slot 0, pt_set_slot)
  hipEvent_t evBegin = nullptr, evEnd = nullptr;
  hipEvent_t evCross[kMaxSlots] = {};
  int numSlots = 0;                  // slots with a stream: 0 .. numSlots-1
  int framesInFlight = 2;            // PT_OPT_FRAMES_IN_FLIGHT
  int concurrency = 0;               // streams of slots 0 .. framesInFlight-1 the device runs side by side (0: not measured yet)
  int lastSlot = 0;                  // slot of the most recent frame (what readback and pt_device_* show)
  int presentLag = 0;                // PT_OPT_PRESENT_LAG
  int shownSlot = -1;                // with presentLag: the slot pt_present last declared complete (-1: the most recent frame)
  bool renderedSincePresent = false; // a frame was queued since the last pt_present
  bool slotDrawn[kMaxSlots] = {};    // the slot holds a frame drawn since the slots were (re)configured

  // scene
  bool sceneValid = false;
  bool sceneHuge = false; // a sphere with a non-finite or > 1e9 centre/radius: rays take the exact routine
  int numSpheres = 0, numEllipsoids = 0, numMaterials = 0, numLights = 0, numVolumes = 0;
  int numEmSpheres = 0, numEmEllipsoids = 0;
  // The scene (image + plain sphere array) lives once on the host and once per frame slot on the device: pt_update_scene
  // only rebuilds the host copy, and a slot's device copy is refreshed on the slot's own stream in front of the next frame
  // drawn there (from a pinned staging block per slot), so an update between two frames does not wait for the frames
  // in flight (the reference's loop calls updateScene every frame, src/main_vulkan.cpp:352-360).
  std::vector<char> hostScene;               // image, then (256-byte aligned) the float4 sphere array
  size_t sceneBytes = 0, spheresOffset = 0;
  unsigned long long sceneVersion = 0, slotVersion[kMaxSlots] = {};
  DevBuf dScene[kMaxSlots];
  void *hStage[kMaxSlots] = {};              // pinned host memory, hStageCap bytes
  size_t hStageCap[kMaxSlots] = {};
  hipEvent_t evCopied[kMaxSlots] = {};       // recorded behind the copy out of hStage[k]
  std::vector<PtGPUEllipsoid> ellSeen;       // ellipsoid records of the previous update and what was derived from them
  std::vector<float> ellDerived;             // (16 floats each), reused for records that did not change
  DevBuf dVoxels, dBricks; // dBricks: volume 0 re-tiled into 8x8x8 bricks
  int brickRes[3] = {0, 0, 0};               // resolution dBricks was built for (0: not built)
  bool volumeBricks = true;                  // PT_OPT_VOLUME_BRICKS
  int volumeFilter = 0;                      // PT_OPT_VOLUME_FILTER
  DevBuf dAccum;                             // float4 sum per pixel (pt_render_samples)
  DevBuf dPlanes;                            // pt_render_samples, several samples per launch: one float4 plane per sample
  int sampleBatch = 0;                       // PT_OPT_SAMPLE_BATCH: samples per launch at most (0: as many as fit kPlaneBudget, 1: one)
  int batchSamples = 1;                      // samples the next launch draws (set by pt_render_samples)
  DevBuf dWaveLog;                           // pt_wave_timeline: 4 words per wave
  bool wantWaveLog = false;
  int accumCount = 0;                        // samples in dAccum
  int sampleIndex = 0;                       // sample drawn by the next launch
  bool accumulate = false;
  int imageFloat4s = 0, offRadius = 0, offEll = 0, offEllPair = 0, offMat = 0, offLight = 0, offEmS = 0, offEmE = 0, offMatIdx = 0, matIdxEllBase = 0;
  bool voxelsUploaded = false;
  size_t voxelBytes = 0;
  PtGPUVolumetricData volume0{};

  // output
  DevBuf dRgba8[kMaxSlots], dF32[kMaxSlots], dScratch[kMaxSlots]; // per slot; dScratch: see kScratchBytes
  DevBuf dTileOrder[kMaxSlots];           // 3 x numTiles tile indices (one list per cost class)
  int orderParity[kMaxSlots] = {};        // which of the slot's two sets of class counters the next ordered frame uses
  bool orderClean[kMaxSlots] = {};        // that set was cleared by the drawing-order pass of the slot's previous frame
  int tileOrder = 2;             // PT_OPT_TILE_ORDER: 0 off, 1 on, 2 decided per frame
  float ellCenterMax = 0.0f, ellScale = 0.0f; // see FrameParams
  bool sphereFilter = false;     // the sphere pairs of the image carry the filter constant (FrameParams::sphereFilter)
  bool sceneBounces = false;     // some object's material sends the path on (mirror or transparent)
  // Tuning switches for measurements (pt_set_option(100, bits); 0 = production, not part of the interface):
  //   bit 0       no wave priorities     bit 1  no fan-out of lighting rays     bit 4  no pooled end-of-frame tracing
  //   bit 2       discriminants in the dense sphere loop, no filter; bit 3: the filter whatever its margin costs
  //               (both read by pt_update_scene)
  //   bits 5-7    unit-direction tolerance x 2^n        bits 8-15   sparse-mode lane threshold + 1
  //   bit 24      the tolerance also for directions shorter than 1 (as before the loop took those)
  //   bits 16-20  waves per workgroup, bits 21-23 workgroups per CU (0: chosen from the scene size)
  //   bit 25      frames without solids on the general kernel, not the volume kernel
  //   bit 26      (instrumented phase kernel) per-phase and per-ticket clock stamps
  //   bits 27-29  n > 0: launch 1/2^(n-1) of the resident workgroups whatever is in flight
  //   bit 30      a lane whose lighting rays are all with helpers does not cast its next path ray in the same pass
  int tune = 0;
  void *extRgba8[kMaxSlots] = {}, *extF32[kMaxSlots] = {}; // caller-owned output per slot (pt_set_output: slot 0, pt_set_slot)
  bool floatOutput = true;
  bool honourMaxDepth = false; // PT_OPT_HONOUR_MAX_DEPTH
  int objectSource = 0;

  // row bands
  int bandRows = 0, bandIndex = 0, bandCount = 1;
  int localRows = 0;

  // band gather (pt_gather_bands): per slot, rank 0's staging for every rank's compact frame; one assembled frame
  DevBuf dGathered[kMaxSlots];
  DevBuf dAssembled[kMaxSlots];
  int assembledElem[kMaxSlots] = {};

  // last launch
  int lastBlocks = 0, lastThreads = 0, lastLds = 0;
};

// RCCL, bound at run time.
struct PtComm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

namespace {

// dScratch: class counts (two sets of 3 x 4 B, used alternately) at +16 and +32, counters (PT_CNT_COUNT x 8 B) at +64, tile-ticket counters at +512
static_assert(64 + PT_CNT_COUNT * sizeof(uint64_t) <= 512, "the work counters must end before the ticket counters");
constexpr size_t kPlaneBudget = (size_t)16 << 30; // pt_render_samples: device memory the sample planes of one launch may take
constexpr size_t kScratchBytes = 512 + (size_t)ptdev::kTicketGroups * ptdev::kTicketStride * sizeof(unsigned int);

int compute_local_rows(int height, int bandRows, int bandIndex, int bandCount) {
  if (bandCount <= 1) return height;
  int rows = 0;
  int nBands = (height + bandRows - 1) / bandRows;
  for (int b = bandIndex; b < nBands; b += bandCount) {
    int y0 = b * bandRows;
    rows += std::min(bandRows, height - y0);
  }
  return rows;
}

int ensure_outputs(PtRenderer *r, int slot) {
  size_t px = (size_t)r->width * (size_t)std::max(r->localRows, 1);
  const void *ext8 = r->extRgba8[slot], *extF = r->extF32[slot];
  if (!ext8) PT_HIP_CHECK(r->dRgba8[slot].reserve(px * 4));
  if (r->floatOutput && !extF && !ext8) PT_HIP_CHECK(r->dF32[slot].reserve(px * 16));
  PT_HIP_CHECK(r->dScratch[slot].reserve(kScratchBytes));
  return PT_OK;
}

int sync_all(PtRenderer *r) {
  PT_HIP_CHECK(hipSetDevice(r->device));
  for (int k = 0; k < r->numSlots; ++k) PT_HIP_CHECK(hipStreamSynchronize(r->stream[k]));
  return PT_OK;
}

// Streams (and the events that tie them together for the timers) for slots 0 .. n-1.
int ensure_slots(PtRenderer *r, int n) {
  for (int k = r->numSlots; k < n; ++k) {
    PT_HIP_CHECK(hipStreamCreateWithFlags(&r->ownStream[k], hipStreamNonBlocking));
    if (!r->stream[k]) r->stream[k] = r->ownStream[k];
    PT_HIP_CHECK(hipEventCreateWithFlags(&r->evCross[k], hipEventDisableTiming));
    r->numSlots = k + 1;
  }
  return PT_OK;
}

// Plain frames may go round the slots unless something ties them to one target: a caller's
// stream or output buffers given for some slots only, or the running sum of pt_render_samples.
bool pipelined(const PtRenderer *r) {
  if (r->framesInFlight < 2 || r->accumulate || r->numSlots < r->framesInFlight) return false;
  for (int k = 1; k < r->framesInFlight; ++k) {
    if ((r->stream[k] == r->ownStream[k]) != (r->stream[0] == r->ownStream[0])) return false;
    if ((r->extRgba8[k] == nullptr) != (r->extRgba8[0] == nullptr)) return false;
  }
  return true;
}

// How many of the slots' streams run side by side?  Streams share hardware queues (GPU_MAX_HW_QUEUES, 4 unless
// the environment says otherwise, and other users of the device in the process take their part), and two streams on
// one queue run one after the other.  One 1 ms spin per stream, all released together: the time they take all
// together tells.  Costs a device synchronisation and two to ten milliseconds, once per change of the slots.
int measure_concurrency(PtRenderer *r) {
  const int k = r->framesInFlight;
  r->concurrency = 1;
  if (k < 2) return PT_OK;
  int rc = sync_all(r);
  if (rc != PT_OK) return rc;
  hipEvent_t ev[2] = {nullptr, nullptr};
  PT_HIP_CHECK(hipEventCreate(&ev[0]));
  hipError_t e = hipEventCreate(&ev[1]);
  if (e != hipSuccess) { (void)hipEventDestroy(ev[0]); return fail(PT_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e)); }
  const float spinMs = 1.0f;
  auto run = [&]() -> hipError_t {
    hipError_t h;
    if ((h = hipEventRecord(ev[0], r->stream[0])) != hipSuccess) return h;
    for (int s = 1; s < k; ++s) if ((h = hipStreamWaitEvent(r->stream[s], ev[0], 0)) != hipSuccess) return h;
    for (int s = 0; s < k; ++s) hipLaunchKernelGGL(ptdev::pt_spin_kernel, dim3(1), dim3(64), 0, r->stream[s], (unsigned long long)(spinMs * 1e5f));
    if ((h = hipGetLastError()) != hipSuccess) return h;
    for (int s = 1; s < k; ++s) {
      if ((h = hipEventRecord(r->evCross[s], r->stream[s])) != hipSuccess) return h;
      if ((h = hipStreamWaitEvent(r->stream[0], r->evCross[s], 0)) != hipSuccess) return h;
    }
    if ((h = hipEventRecord(ev[1], r->stream[0])) != hipSuccess) return h;
    return hipEventSynchronize(ev[1]);
  };
  e = run();          // the first launch on a stream includes its set-up
  if (e == hipSuccess) e = run();
  float ms = 0.0f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, ev[0], ev[1]);
  (void)hipEventDestroy(ev[0]);
  (void)hipEventDestroy(ev[1]);
  if (e != hipSuccess) return fail(PT_ERR_HIP, std::string("measuring the streams' concurrency: ") + hipGetErrorString(e));
  // k spins in ms: ms / spinMs of them came one after the other
  const int serial = std::max(1, (int)(ms / spinMs + 0.7f));
  r->concurrency = std::max(1, std::min(k, (k + serial - 1) / serial));
  return PT_OK;
}

void *slot_rgba8(const PtRenderer *r, int slot) { return r->extRgba8[slot] ? r->extRgba8[slot] : r->dRgba8[slot].p; }
void *slot_f32(const PtRenderer *r, int slot) {
  if (r->extRgba8[slot]) return r->extF32[slot];
  return r->floatOutput ? r->dF32[slot].p : nullptr;
}

// pt_render_static: the frame's scratch (ticket counters, class counts, tile-order lists) in a
// file-scope device array instead of buffers the library allocates.  g_staticScratch is the host-side
// stand-in of the symbol, not a device address: host code uses the per-device address that pt_create
// fetched with hipGetSymbolAddress (every GPU holds its own copy of the array).
constexpr int kStaticMaxTiles = 1 << 18; // 8x8-pixel tiles: frames up to 16.7 M pixels (4K has 129 600)
constexpr size_t kStaticOrderOffset = 8192; // the counters block (kScratchBytes <= 8192) comes first
constexpr size_t kStaticScratchBytes = kStaticOrderOffset + (size_t)3 * kStaticMaxTiles * sizeof(int);
static_assert(kScratchBytes <= kStaticOrderOffset, "counter block outgrew its place in the static scratch");
static __device__ __align__(256) unsigned char g_staticScratch[kStaticScratchBytes];
constexpr int kMaxDevices = 64;
void *g_staticScratchPtr[kMaxDevices] = {}; // device address of g_staticScratch on device d (0: not fetched yet)

// Where a frame goes when the caller fixes everything (pt_render_static); nullptr: a frame slot of the renderer.
struct FrameTarget {
  hipStream_t stream;
  void *rgba8, *f32;
  char *scratch;  // kScratchBytes, zeroed by the launch
  int *tileOrder; // room for 3 x kStaticMaxTiles entries
};

// Room for the current scene in slot's device copy and staging block.  Growing the device copy waits for the frames in
// flight on that slot (they read it), growing the staging block for the copy out of it.
int reserve_scene(PtRenderer *r, int slot) {
  if (r->dScene[slot].cap < r->sceneBytes) {
    if (r->stream[slot]) PT_HIP_CHECK(hipStreamSynchronize(r->stream[slot]));
    PT_HIP_CHECK(r->dScene[slot].reserve(r->sceneBytes + r->sceneBytes / 2));
    r->slotVersion[slot] = 0;
  }
  if (r->hStageCap[slot] < r->sceneBytes) {
    if (r->evCopied[slot]) PT_HIP_CHECK(hipEventSynchronize(r->evCopied[slot]));
    if (r->hStage[slot]) (void)hipHostFree(r->hStage[slot]);
    r->hStage[slot] = nullptr;
    r->hStageCap[slot] = 0;
    const size_t want = std::max<size_t>(r->sceneBytes + r->sceneBytes / 2, 4096);
    PT_HIP_CHECK(hipHostMalloc(&r->hStage[slot], want, hipHostMallocDefault));
    r->hStageCap[slot] = want;
  }
  if (!r->evCopied[slot]) PT_HIP_CHECK(hipEventCreateWithFlags(&r->evCopied[slot], hipEventDisableTiming));
  return PT_OK;
}

// The slot's device copy of the scene, brought up to date on `stream` in front of the frame about to be drawn there.
int refresh_scene(PtRenderer *r, int slot, hipStream_t stream, bool may_allocate) {
  if (r->slotVersion[slot] == r->sceneVersion) return PT_OK;
  if (r->dScene[slot].cap < r->sceneBytes || r->hStageCap[slot] < r->sceneBytes || !r->evCopied[slot]) {
    if (!may_allocate)
      return fail(PT_ERR_STATE, "pt_render_static: the scene copy of frame slot 0 has no room for this scene (it is made when pt_update_scene runs)");
    int rc = reserve_scene(r, slot);
    if (rc != PT_OK) return rc;
  }
  // the staging block is free again once the previous copy out of it has run: that copy was queued a whole round of
  // slots ago, so this does not wait in a steady loop
  PT_HIP_CHECK(hipEventSynchronize(r->evCopied[slot]));
  std::memcpy(r->hStage[slot], r->hostScene.data(), r->sceneBytes);
  PT_HIP_CHECK(hipMemcpyAsync(r->dScene[slot].p, r->hStage[slot], r->sceneBytes, hipMemcpyHostToDevice, stream));
  PT_HIP_CHECK(hipEventRecord(r->evCopied[slot], stream));
  r->slotVersion[slot] = r->sceneVersion;
  return PT_OK;
}

int launch_frame(PtRenderer *r, const PtPushConstants *pc, int mode, const FrameTarget *fixed = nullptr) { // mode: the kernel's MODE (0 plain, 1 counting, 2 phase cycles)
  if (!r || !pc) return fail(PT_ERR_INVALID, "pt_render: null argument");
  if (!r->sceneValid) return fail(PT_ERR_STATE, "pt_render: pt_update_scene has not been called");
  if (pc->numSpheres < 0 || pc->numSpheres > r->numSpheres || pc->numEllipsoids < 0 ||
      pc->numEllipsoids > r->numEllipsoids || pc->numLights < 0 || pc->numLights > r->numLights ||
      pc->numVolumes < 0 || pc->numVolumes > r->numVolumes)
    return fail(PT_ERR_INVALID, "pt_render: push-constant object counts exceed the uploaded scene");
  PT_HIP_CHECK(hipSetDevice(r->device));
  const int slot = fixed ? 0 : ((mode == 0 && pipelined(r)) ? (r->lastSlot + 1) % r->framesInFlight : 0);
  const hipStream_t stream = fixed ? fixed->stream : r->stream[slot];
  if (!fixed) {
    int rc = ensure_outputs(r, slot);
    if (rc != PT_OK) return rc;
  }
  char *const scratch = fixed ? fixed->scratch : (char *)r->dScratch[slot].p;

  ptdev::FrameParams P;
  std::memset(&P, 0, sizeof P);
  // createCameraRay's pixel-invariant part (reference shaders/raytrace.comp:123-131)
  H3 eye{pc->cameraPos[0], pc->cameraPos[1], pc->cameraPos[2]};
  H3 target{2.0f, 1.5f, 0.0f};
  H3 forward = hnormalize({target.x - eye.x, target.y - eye.y, target.z - eye.z});
  H3 right = hnormalize(hcross(forward, {0.0f, 1.0f, 0.0f}));
  H3 up = hnormalize(hcross(right, forward));
  P.cam[0] = eye.x; P.cam[1] = eye.y; P.cam[2] = eye.z;
  P.fwd[0] = forward.x; P.fwd[1] = forward.y; P.fwd[2] = forward.z;
  P.right[0] = right.x; P.right[1] = right.y; P.right[2] = right.z;
  P.up[0] = up.x; P.up[1] = up.y; P.up[2] = up.z;
  P.fov = std::tan((60.0f * 0.017453292519943295f) / 2.0f);
  P.aspect = (float)r->width / (float)r->height;
  P.time = pc->time;
  for (int i = 0; i < 3; ++i) { P.bgTop[i] = pc->bgColorTop[i]; P.bgBottom[i] = pc->bgColorBottom[i]; }
  P.width = r->width;
  P.height = r->height;
  P.localRows = r->localRows;
  P.bandRows = r->bandCount > 1 ? r->bandRows : std::max(r->height, 1);
  P.bandIndex = r->bandCount > 1 ? r->bandIndex : 0;
  P.bandCount = std::max(r->bandCount, 1);
  P.numSpheres = pc->numSpheres;
  P.numEllipsoids = pc->numEllipsoids;
  P.numLights = pc->numLights;
  P.numVolumes = pc->numVolumes;
  P.numEmSpheres = r->numEmSpheres;
  P.numEmEllipsoids = r->numEmEllipsoids;
  // the reference shader ignores maxDepth and always runs 10 bounces (reference shaders/raytrace.comp:603)
  P.maxBounces = (r->honourMaxDepth && pc->maxDepth > 0) ? pc->maxDepth : 10;
  P.tilesX = (r->width + ptdev::kTileDim - 1) / ptdev::kTileDim;
  int tilesY = (r->localRows + ptdev::kTileDim - 1) / ptdev::kTileDim;
  P.numTiles = P.tilesX * tilesY;
  if (pc->numVolumes > 0) {
    const PtGPUVolumetricData &v = r->volume0;
    for (int i = 0; i < 3; ++i) {
      P.volPos[i] = v.position[i];
      P.v0[i] = v.v0[i];
      P.v1[i] = v.v1[i];
      P.boxMin[i] = v.position[i] + v.v0[i]; // reference shaders/raytrace.comp:549-550
      P.boxMax[i] = v.position[i] + v.v1[i];
      P.extent[i] = v.v1[i] - v.v0[i];       // reference shaders/raytrace.comp:503
      P.rcpExtent[i] = (float)(1.0 / (double)P.extent[i]);
    }
    P.exactExtent = 1;
    for (int i = 0; i < 3; ++i) {
      int e = 0;
      const float m = std::frexp(P.extent[i], &e); // a power of two with a representable reciprocal: mantissa 1/2, exponent in range
      if (!(m == 0.5f) || e < -100 || e > 100) P.exactExtent = 0;
    }
    P.res[0] = v.resolution_x; P.res[1] = v.resolution_y; P.res[2] = v.resolution_z;
    P.volMaterial = v.materialIndex;
  }
  {
    int rcScene = refresh_scene(r, slot, stream, /*may_allocate=*/!fixed);
    if (rcScene != PT_OK) return rcScene;
  }
  P.spheres = (const float4 *)((const char *)r->dScene[slot].p + r->spheresOffset);
  P.image = (const float4 *)r->dScene[slot].p;
  P.imageFloat4s = r->imageFloat4s;
  P.offRadius = r->offRadius;
  P.offEll = r->offEll;
  P.offEllPair = r->offEllPair;
  P.offMat = r->offMat;
  P.offLight = r->offLight;
  P.offEmS = r->offEmS;
  P.offEmE = r->offEmE;
  P.offMatIdx = r->offMatIdx;
  P.matIdxEllBase = r->matIdxEllBase;
  P.voxels = (const uint8_t *)r->dVoxels.p;
  if (r->volumeBricks && pc->numVolumes > 0 && r->brickRes[0] > 0) {
    P.voxels = (const uint8_t *)r->dBricks.p;
    P.brickX = (r->brickRes[0] + 7) / 8;
    P.brickY = (r->brickRes[1] + 7) / 8;
  }
  P.outRgba8 = (uint32_t *)(fixed ? fixed->rgba8 : slot_rgba8(r, slot));
  P.outF32 = (float4 *)(fixed ? fixed->f32 : slot_f32(r, slot));
  P.tileCounter = (unsigned int *)(scratch + 512);
  P.counters = (unsigned long long *)(scratch + 64);
  P.waveLog = (mode == 2 && r->wantWaveLog) ? (unsigned long long *)r->dWaveLog.p : nullptr;

  if (P.numTiles <= 0) return PT_OK;
  // A frame without solids: the camera rays, their march through the volume, the background.  That needs neither
  // the state machine nor the tickets of the general kernel (tune bit 25 keeps it on the general kernel; the
  // instrumented modes always are).
  if (mode == 0 && pc->numSpheres == 0 && pc->numEllipsoids == 0 && pc->numVolumes > 0 && !(r->tune & (1 << 25))) {
    P.volumeFilter = r->volumeFilter;
    P.sampleIndex = r->sampleIndex;
    P.sampleCount = 1;
    const size_t framePixels = (size_t)r->width * (size_t)std::max(r->localRows, 1);
    const bool planes = r->accumulate && !fixed && r->batchSamples > 1;
    if (r->accumulate && !fixed) {
      PT_HIP_CHECK(r->dAccum.reserve(framePixels * sizeof(float4)));
      P.accum = (float4 *)r->dAccum.p;
      P.accumCount = r->accumCount + 1;
    }
    if (planes) { // as in the general kernel below: one plane per sample of the launch, summed afterwards
      PT_HIP_CHECK(r->dPlanes.reserve(framePixels * sizeof(float4) * (size_t)r->batchSamples));
      P.accum = (float4 *)r->dPlanes.p;
      P.accumCount = 0;
      P.sampleCount = r->batchSamples;
      P.planeStride = (int)framePixels;
    }
    const int blocks = (int)(((long long)P.numTiles * P.sampleCount + 3) / 4);
    // the common volume (bricked copy, extents that are powers of two, box starting at v0 = 0) has variants compiled
    // for it alone, one per filter
    const bool commonVolume = P.brickX > 0 && P.exactExtent != 0 && P.v0[0] == 0.0f && P.v0[1] == 0.0f && P.v0[2] == 0.0f;
    const int kind = commonVolume ? (P.volumeFilter == 0 ? 1 : 2) : 0;
    // Pixels that can see the box: the bounding rectangle of its eight corners on the image plane, two pixels of margin
    // (the shader's slab test differs from the exact one by rounding errors, a millionth of that).  Only when every
    // corner lies well in front of the eye and everything is finite; otherwise the whole frame.
    int rect[4] = {INT_MIN / 2, INT_MIN / 2, INT_MAX / 2, INT_MAX / 2};
    {
      const double fwd[3] = {P.fwd[0], P.fwd[1], P.fwd[2]}, rgt[3] = {P.right[0], P.right[1], P.right[2]}, upv[3] = {P.up[0], P.up[1], P.up[2]};
      double lo[2] = {1e300, 1e300}, hi[2] = {-1e300, -1e300};
      bool ok = true;
      for (int c = 0; c < 8 && ok; ++c) {
        double v[3], len2 = 0.0;
        for (int a = 0; a < 3; ++a) {
          v[a] = (double)(((c >> a) & 1) ? P.boxMax[a] : P.boxMin[a]) - (double)P.cam[a];
          len2 += v[a] * v[a];
        }
        const double za = v[0] * fwd[0] + v[1] * fwd[1] + v[2] * fwd[2];
        const double xb = v[0] * rgt[0] + v[1] * rgt[1] + v[2] * rgt[2], yc = v[0] * upv[0] + v[1] * upv[1] + v[2] * upv[2];
        if (!(za > 1e-3 * std::sqrt(len2)) || !std::isfinite(len2)) { ok = false; break; }
        const double fx = (xb / za / ((double)P.aspect * (double)P.fov) + 0.5) * (double)P.width;
        const double fy = (1.0 - (yc / za / (double)P.fov + 0.5)) * (double)P.height;
        if (!std::isfinite(fx) || !std::isfinite(fy) || std::fabs(fx) > 1e8 || std::fabs(fy) > 1e8) { ok = false; break; }
        lo[0] = std::min(lo[0], fx); hi[0] = std::max(hi[0], fx);
        lo[1] = std::min(lo[1], fy); hi[1] = std::max(hi[1], fy);
      }
      if (ok) {
        rect[0] = (int)std::floor(lo[0]) - 2; rect[1] = (int)std::floor(lo[1]) - 2;
        rect[2] = (int)std::ceil(hi[0]) + 2;  rect[3] = (int)std::ceil(hi[1]) + 2;
      }
    }
#define PT_LAUNCH_V(KIND_, SAMPLES_) hipLaunchKernelGGL((ptdev::pt_volume_frame_kernel<KIND_, SAMPLES_>), dim3(blocks), dim3(256), 0, stream, P, rect[0], rect[1], rect[2], rect[3])
    if (planes) { if (kind == 1) PT_LAUNCH_V(1, true); else if (kind == 2) PT_LAUNCH_V(2, true); else PT_LAUNCH_V(0, true); }
    else { if (kind == 1) PT_LAUNCH_V(1, false); else if (kind == 2) PT_LAUNCH_V(2, false); else PT_LAUNCH_V(0, false); }
#undef PT_LAUNCH_V
    PT_HIP_CHECK(hipGetLastError());
    if (planes) {
      hipLaunchKernelGGL(ptdev::pt_sum_samples_kernel, dim3((unsigned int)((framePixels + 255) / 256)), dim3(256), 0, stream,
                         (const float4 *)r->dPlanes.p, (int)framePixels, r->batchSamples, (float4 *)r->dAccum.p, r->accumCount,
                         P.outRgba8, P.outF32);
    }
    PT_HIP_CHECK(hipGetLastError());
    if (!fixed) { r->lastSlot = slot; r->slotDrawn[slot] = true; r->renderedSincePresent = true; }
    r->lastBlocks = blocks;
    r->lastThreads = 256;
    r->lastLds = 1024;
    return PT_OK;
  }
  // The ordering pass pays off when pixels differ a lot in the number of rays
  // they cast (bounces) and looking at one pixel per tile is cheap next to the frame.
  const bool orderTiles =
      r->tileOrder == 1 ||
      (r->tileOrder == 2 && r->sceneBounces && P.maxBounces > 1 &&
       (long long)(pc->numSpheres + pc->numEllipsoids) * P.numTiles <= 50000000ll);
  if (fixed && P.numTiles > kStaticMaxTiles)
    return fail(PT_ERR_INVALID, "pt_render_static: the frame has " + std::to_string(P.numTiles) + " tiles of 8x8 pixels, the static scratch holds " +
                                    std::to_string(kStaticMaxTiles));
  if (orderTiles) {
    if (fixed) {
      P.tileOrder = fixed->tileOrder;
    } else {
      PT_HIP_CHECK(r->dTileOrder[slot].reserve((size_t)P.numTiles * 3 * sizeof(int)));
      P.tileOrder = (int *)r->dTileOrder[slot].p;
    }
    P.classCount = (unsigned int *)(scratch + 16 + 16 * (fixed ? 0 : r->orderParity[slot])); // two sets, see the memset below
  }

  // Launch geometry: persistent workgroups, 16 waves per CU.  Objects are
  // staged in LDS when they fit (<= 80 KiB: two 512-thread groups per CU;
  // <= 160 KiB: one 1024-thread group); otherwise read from global memory.
  // LDS per workgroup: the object image (if staged) + 1 KiB of queue/result scratch per wave.
  size_t objBytes = (size_t)r->imageFloat4s * sizeof(float4);
  bool useLds = objBytes + 16 * 1024 + 2048 <= 160 * 1024; // + per-wave scratch + the copy of the frame constants + the b/255 table
  if (r->objectSource == 1 && !useLds)
    return fail(PT_ERR_INVALID, "pt_render: scene does not fit the 160 KiB LDS (PT_OPT_OBJECT_SOURCE=1)");
  if (r->objectSource == 2) useLds = false;
  int threads, blocksPerCu;
  if (!useLds) { objBytes = 0; threads = 256; blocksPerCu = 4; }
  else if (objBytes <= 32 * 1024) { threads = 256; blocksPerCu = 4; }   // 4 x (32 + 4) KiB
  else if (objBytes <= 64 * 1024) { threads = 512; blocksPerCu = 2; }   // 2 x (64 + 8) KiB
  else { threads = 1024; blocksPerCu = 1; }                             // 1 x (<=144 + 16) KiB
  if ((r->tune >> 16) & 0xff) { threads = 64 * ((r->tune >> 16) & 0x1f); blocksPerCu = std::max(1, (r->tune >> 21) & 7); }
  size_t ldsBytes = objBytes + (size_t)(threads / 64) * 1024 + ((sizeof(ptdev::FrameParams) + 15) & ~(size_t)15) + 1024; // + the kernel's copy of the frame constants + its b/255 table
  // Pooled tracing at the end of the frame pays when one pass over the spheres is long.
  {
    const size_t coopBytes = 64 + (size_t)threads * (6 * sizeof(float) + 2 * sizeof(unsigned long long)) + // result slots: one set per pass parity
                             (size_t)(threads / 4) * (7 * sizeof(float) + sizeof(unsigned long long)); // pooled rays + exact-routine rays
    const bool fits = ldsBytes + coopBytes <= 160 * 1024 / (size_t)blocksPerCu;
    P.coop = (pc->numSpheres >= 256 && fits && !(r->tune & 16)) ? 1 : 0;
    if (P.coop) ldsBytes += coopBytes;
    // Fan-out of lighting rays at the end of the frame: where a hit casts several of them (and the pooled mode is off)
    const size_t fanBytes = (size_t)(threads / 64) * ptdev::kFanBytes;
    const bool fanFits = ldsBytes + fanBytes <= 160 * 1024 / (size_t)blocksPerCu;
    P.fanOut = (!P.coop && pc->numLights + r->numEmSpheres + r->numEmEllipsoids >= 2 && fanFits && !(r->tune & 2)) ? 1 : 0;
    if (P.fanOut) ldsBytes += fanBytes;
  }
  P.sceneHuge = r->sceneHuge ? 1 : 0;
  P.ellCenterMax = r->ellCenterMax;
  P.ellScale = r->ellScale;
  P.volumeFilter = r->volumeFilter;
  P.sampleIndex = r->sampleIndex;
  P.sampleCount = 1;
  const size_t framePixels = (size_t)r->width * (size_t)std::max(r->localRows, 1);
  const bool planes = r->accumulate && !fixed && mode == 0 && r->batchSamples > 1;
  if (r->accumulate && !fixed) {
    PT_HIP_CHECK(r->dAccum.reserve(framePixels * sizeof(float4)));
    P.accum = (float4 *)r->dAccum.p;
    P.accumCount = r->accumCount + 1;
  }
  if (planes) { // the launch draws batchSamples samples into planes; pt_sum_samples_kernel adds them to dAccum below
    PT_HIP_CHECK(r->dPlanes.reserve(framePixels * sizeof(float4) * (size_t)r->batchSamples));
    P.accum = (float4 *)r->dPlanes.p;
    P.accumCount = 0;
    P.sampleCount = r->batchSamples;
    P.planeStride = (int)framePixels;
  }
  P.tune = r->tune;
  // Directions longer than 1 by more than this take the exact routine (too many spheres would be candidates);
  // measured optimum on C3 (4096 spheres): 0.25, broad.  Shorter directions have fewer candidates and all take the loop.
  P.unitTol = 0.25f * (float)(1 << ((r->tune >> 5) & 7));
  P.sphereFilter = r->sphereFilter ? 1 : 0;
  if (P.sphereFilter) P.unitTol = std::min(P.unitTol, 0.75f); // the filter's error bound is derived for |d|^2 <= 1.75
  P.unitTolLow = (r->tune & (1 << 24)) ? P.unitTol : 1.0f;
  {
    // Break-even between one dense pass (about 7.5 instructions per sphere with the discriminant loop, 5.8 with
    // the filter, whatever the number of rays) and the exact routine (about 0.31 per sphere + 150, per ray).
    const double ns = (double)pc->numSpheres;
    const int t = (int)((r->sphereFilter ? 5.8 : 7.5) * ns / (0.31 * ns + 150.0));
    P.sparseLanes = ((r->tune >> 8) & 0xff) ? ((r->tune >> 8) & 0xff) - 1 : std::min(t, 32);
  }
  {
    // Refilling idle lanes costs about 190 vector instructions per pass whatever their number; an idle lane costs its
    // 1/64 share of the pass over the objects (about 600 + 5.8 per sphere + 9 per ellipsoid).  Measured on the 1080p /
    // 64-object frame the best threshold is near 16 idle lanes (-7 %), on the 4096-sphere frame 1.
    const double passCost = 600.0 + 5.8 * pc->numSpheres + 9.0 * pc->numEllipsoids;
    P.refillMin = std::max(1, std::min(24, (int)(16000.0 / passCost)));
  }
  long long wavesNeeded = ((long long)P.numTiles * P.sampleCount * ptdev::kTilePixels + 63) / 64;
  long long blocksNeeded = (wavesNeeded * 64 + threads - 1) / threads;
  int blocks = (int)std::min<long long>((long long)r->computeUnits * blocksPerCu, std::max<long long>(blocksNeeded, 1));
  // Frames in flight share the chip.  A frame with few tiles per wave (one rank's share of a split) is nearly all
  // end-of-frame work when it takes every compute unit: its lanes empty out as paths end and nothing refills them.
  // With k frames running side by side such a frame takes only a part of the resident workgroups, so that its waves draw
  // several tiles each and stay full until the tickets run out, while the other frames fill the rest of the chip.
  // Measured on one rank's 1/8 share (kernel ms per frame, whole launch -> this rule): 4096 spheres at 4K, 8 in
  // flight 2.30 -> 2.15, 4 in flight 2.45 -> 2.38 (2 in flight: 2.71 -> 2.88, so the pooled kernel keeps the whole
  // launch below 4); 64 objects at 1080p, 8 in flight 0.087 -> 0.079; volume only, 8 in flight 0.036 -> 0.021.
  // Tune bits 27-29 = n > 0 force a launch of 1/2^(n-1) of the workgroups instead.
  if (mode == 0 && pipelined(r) && !fixed) {
    if (r->concurrency == 0) {
      int rcM = measure_concurrency(r);
      if (rcM != PT_OK) return rcM;
    }
    const long long full = (long long)r->computeUnits * blocksPerCu;
    const int k = r->concurrency; // frames that really run side by side
    const int part = P.coop ? (k >= 4 ? k : 1) : std::max(1, k / 2);
    const long long perBlock = (long long)(P.coop ? 32 : 16) * (threads / 64); // more tiles per wave than this gain nothing
    blocks = (int)std::min<long long>(blocks, std::max((full + part - 1) / part, (P.numTiles + perBlock - 1) / perBlock));
  }
  if ((r->tune >> 27) & 7) blocks = std::max(1, (r->computeUnits * blocksPerCu) >> (((r->tune >> 27) & 7) - 1));
  blocks = (int)std::min<long long>(blocks, std::max<long long>(blocksNeeded, 1));

  // The counter block is cleared in front of the frame -- by a memset, or, where the drawing-order pass runs anyway,
  // by that pass: it clears the tickets and work counters of its own frame and the class counters of the next ordered
  // frame of the slot (two sets of class counters alternate, at +16 and +32), which leaves one launch less per frame.
  const bool selfClear = P.tileOrder != nullptr && !fixed;
  if (!(selfClear && r->orderClean[slot])) PT_HIP_CHECK(hipMemsetAsync(scratch, 0, kScratchBytes, stream));
  if (!fixed) r->orderClean[slot] = false; // until this frame's pass is queued
  unsigned int *const clearWords = selfClear ? (unsigned int *)(scratch + 64) : nullptr;
  const int clearCount = (int)((kScratchBytes - 64) / sizeof(unsigned int));
  unsigned int *const clearNextClass = selfClear ? (unsigned int *)(scratch + 16 + 16 * (r->orderParity[slot] ^ 1)) : nullptr;
  if (P.tileOrder) {
    const size_t stage = ((size_t)pc->numSpheres + 4 * (size_t)pc->numEllipsoids) * sizeof(float4);
    const bool staged = stage <= 48 * 1024;
    // a frame drawn alone: sixteen-wave workgroups, four lanes per tile; frames in flight: single waves (see the kernel)
    if (mode == 0 && pipelined(r) && !fixed) {
      const dim3 grid((P.numTiles + 63) / 64), block(64);
      if (staged) hipLaunchKernelGGL((ptdev::pt_tile_order_kernel<true, 1, 64>), grid, block, stage, stream, P, clearWords, clearCount, clearNextClass);
      else hipLaunchKernelGGL((ptdev::pt_tile_order_kernel<false, 1, 64>), grid, block, 0, stream, P, clearWords, clearCount, clearNextClass);
    } else {
      const dim3 grid((P.numTiles + 255) / 256), block(1024);
      if (staged) hipLaunchKernelGGL((ptdev::pt_tile_order_kernel<true, 4, 256>), grid, block, stage, stream, P, clearWords, clearCount, clearNextClass);
      else hipLaunchKernelGGL((ptdev::pt_tile_order_kernel<false, 4, 256>), grid, block, 0, stream, P, clearWords, clearCount, clearNextClass);
    }
    PT_HIP_CHECK(hipGetLastError());
  }
  if (P.waveLog && (size_t)blocks * (size_t)(threads / 64) * 32 > r->dWaveLog.cap)
    return fail(PT_ERR_STATE, "pt_wave_timeline: the launch has more waves than the log holds");
  dim3 grid(blocks), block(threads);
#define PT_LAUNCH_K(MODE_, LDS_, POOLED_, VOLUME_, SAMPLES_)                                                           \
  {                                                                                                                    \
    if (LDS_)                                                                                                          \
      PT_HIP_CHECK(hipFuncSetAttribute((const void *)ptdev::pt_render_kernel<MODE_, LDS_, POOLED_, VOLUME_, SAMPLES_>, \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));                    \
    hipLaunchKernelGGL((ptdev::pt_render_kernel<MODE_, LDS_, POOLED_, VOLUME_, SAMPLES_>), grid, block, ldsBytes,      \
                       stream, P);                                                                                     \
  }
#define PT_LAUNCH(MODE_, VOLUME_, SAMPLES_)                                                                            \
  if (useLds) {                                                                                                        \
    if (P.coop) PT_LAUNCH_K(MODE_, true, true, VOLUME_, SAMPLES_) else PT_LAUNCH_K(MODE_, true, false, VOLUME_, SAMPLES_) \
  } else {                                                                                                             \
    if (P.coop) PT_LAUNCH_K(MODE_, false, true, VOLUME_, SAMPLES_) else PT_LAUNCH_K(MODE_, false, false, VOLUME_, SAMPLES_) \
  }
  // the production kernel exists with and without the volume march, and for launches of several samples; the
  // instrumented ones always carry the march
  if (mode == 1) { PT_LAUNCH(1, true, false) } else if (mode == 2) { PT_LAUNCH(2, true, false) }
  else if (planes) { if (pc->numVolumes > 0) { PT_LAUNCH(0, true, true) } else { PT_LAUNCH(0, false, true) } }
  else if (pc->numVolumes > 0) { PT_LAUNCH(0, true, false) } else { PT_LAUNCH(0, false, false) }
#undef PT_LAUNCH
#undef PT_LAUNCH_K
  PT_HIP_CHECK(hipGetLastError());
  if (planes) {
    hipLaunchKernelGGL(ptdev::pt_sum_samples_kernel, dim3((unsigned int)((framePixels + 255) / 256)), dim3(256), 0, stream,
                       (const float4 *)r->dPlanes.p, (int)framePixels, r->batchSamples, (float4 *)r->dAccum.p, r->accumCount,
                       P.outRgba8, P.outF32);
    PT_HIP_CHECK(hipGetLastError());
  }
  if (!fixed) { r->lastSlot = slot; r->slotDrawn[slot] = true; r->renderedSincePresent = true; }
  if (selfClear) { r->orderParity[slot] ^= 1; r->orderClean[slot] = true; }
  r->lastBlocks = blocks;
  r->lastThreads = threads;
  r->lastLds = (int)ldsBytes;
  return PT_OK;
}

struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

// Opens librccl.so.1 (the copy a host program such as PyTorch has already loaded, if any: same SONAME) and binds
// the few entry points the gather needs.
int load_rccl() {
  if (g_rccl.handle) return PT_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) return fail(PT_ERR_STATE, std::string("RCCL is not available: ") + dlerror());
  RcclApi a;
  a.handle = h;
#define PT_BIND(field, name)                                                     \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));                 \
  if (!a.field) return fail(PT_ERR_STATE, std::string("RCCL lacks ") + name);
  PT_BIND(GetUniqueId, "ncclGetUniqueId")
  PT_BIND(CommInitRank, "ncclCommInitRank")
  PT_BIND(CommDestroy, "ncclCommDestroy")
  PT_BIND(GroupStart, "ncclGroupStart")
  PT_BIND(GroupEnd, "ncclGroupEnd")
  PT_BIND(Send, "ncclSend")
  PT_BIND(Recv, "ncclRecv")
  PT_BIND(GetErrorString, "ncclGetErrorString")
#undef PT_BIND
  g_rccl = a;
  return PT_OK;
}

#define PT_NCCL_CHECK(expr)                                                                            \
  do {                                                                                                 \
    ncclResult_t n_ = (expr);                                                                          \
    if (n_ != ncclSuccess) return fail(PT_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(n_)); \
  } while (0)

} // namespace

extern "C" {

int pt_comm_unique_id(void *id_out, size_t id_bytes) {
  if (!id_out || id_bytes < sizeof(ncclUniqueId)) return fail(PT_ERR_INVALID, "pt_comm_unique_id: need a buffer of PT_COMM_ID_BYTES");
  int rc = load_rccl();
  if (rc != PT_OK) return rc;
  ncclUniqueId id;
  PT_NCCL_CHECK(g_rccl.GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof id);
  return PT_OK;
}

int pt_comm_create(const void *id, size_t id_bytes, int rank, int world, int device, PtComm **out) {
  if (!out) return fail(PT_ERR_INVALID, "pt_comm_create: out is null");
  *out = nullptr;
  if (!id || id_bytes < sizeof(ncclUniqueId)) return fail(PT_ERR_INVALID, "pt_comm_create: need the PT_COMM_ID_BYTES of pt_comm_unique_id");
  if (world < 1 || rank < 0 || rank >= world) return fail(PT_ERR_INVALID, "pt_comm_create: need 0 <= rank < world");
  int rc = load_rccl();
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(hipSetDevice(device));
  PtComm *c = new (std::nothrow) PtComm();
  if (!c) return fail(PT_ERR_NOMEM, "pt_comm_create: out of host memory");
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof uid);
  ncclResult_t n = g_rccl.CommInitRank(&c->comm, world, uid, rank);
  if (n != ncclSuccess) {
    delete c;
    return fail(PT_ERR_HIP, std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(n));
  }
  *out = c;
  return PT_OK;
}

void pt_comm_destroy(PtComm *c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  delete c;
}

int pt_gather_bands(PtRenderer *r, PtComm *c, int elem_bytes, void *dst_dev) {
  if (!r || !c) return fail(PT_ERR_INVALID, "pt_gather_bands: null argument");
  if (elem_bytes != 4 && elem_bytes != 16) return fail(PT_ERR_INVALID, "pt_gather_bands: elem_bytes is 4 (rgba8) or 16 (float4)");
  if (c->device != r->device) return fail(PT_ERR_INVALID, "pt_gather_bands: communicator and renderer are on different devices");
  if (r->bandCount != c->world || r->bandIndex != c->rank)
    return fail(PT_ERR_STATE, "pt_gather_bands: pt_set_row_bands(rows, rank, world) must match the communicator");
  PT_HIP_CHECK(hipSetDevice(r->device));
  const int slot = r->lastSlot;
  const hipStream_t stream = r->stream[slot];
  const void *local = elem_bytes == 4 ? slot_rgba8(r, slot) : slot_f32(r, slot);
  if (!local) return fail(PT_ERR_STATE, "pt_gather_bands: no frame of that kind (float output is off)");
  const int bandRows = r->bandCount > 1 ? r->bandRows : std::max(r->height, 1);
  const int strideRows = compute_local_rows(r->height, bandRows, 0, r->bandCount); // rank 0 has the most rows
  const size_t rowBytes = (size_t)r->width * (size_t)elem_bytes;
  const size_t partBytes = (size_t)strideRows * rowBytes;
  if (c->rank != 0) {
    // every peer has its own xGMI link to rank 0; the compact frame goes in one message
    PT_NCCL_CHECK(g_rccl.Send(local, (size_t)r->localRows * rowBytes, ncclUint8, 0, c->comm, stream));
    return PT_OK;
  }
  PT_HIP_CHECK(r->dGathered[slot].reserve(partBytes * (size_t)c->world));
  char *const gathered = (char *)r->dGathered[slot].p;
  void *dst = dst_dev;
  r->assembledElem[slot] = 0; // a frame assembled into the caller's buffer leaves none here to read back
  if (!dst) {
    PT_HIP_CHECK(r->dAssembled[slot].reserve((size_t)r->height * rowBytes));
    dst = r->dAssembled[slot].p;
    r->assembledElem[slot] = elem_bytes;
  }
  PT_HIP_CHECK(hipMemcpyAsync(gathered, local, (size_t)r->localRows * rowBytes, hipMemcpyDeviceToDevice, stream));
  if (c->world > 1) {
    PT_NCCL_CHECK(g_rccl.GroupStart());
    for (int p = 1; p < c->world; ++p) {
      const size_t rows = (size_t)compute_local_rows(r->height, bandRows, p, c->world);
      ncclResult_t n = g_rccl.Recv(gathered + (size_t)p * partBytes, rows * rowBytes, ncclUint8, p, c->comm, stream);
      if (n != ncclSuccess) {
        (void)g_rccl.GroupEnd();
        return fail(PT_ERR_HIP, std::string("ncclRecv: ") + g_rccl.GetErrorString(n));
      }
    }
    PT_NCCL_CHECK(g_rccl.GroupEnd());
  }
  return pt_assemble_bands(r, gathered, dst, elem_bytes, bandRows, c->world, strideRows);
}

void *pt_device_assembled(PtRenderer *r) { return (r && r->assembledElem[r->lastSlot]) ? r->dAssembled[r->lastSlot].p : nullptr; }

int pt_readback_assembled(PtRenderer *r, void *dst, size_t dst_bytes) {
  if (!r || !dst) return fail(PT_ERR_INVALID, "pt_readback_assembled: null argument");
  const int slot = r->lastSlot;
  if (!r->dAssembled[slot].p || !r->assembledElem[slot]) return fail(PT_ERR_STATE, "pt_readback_assembled: no assembled frame (pt_gather_bands on rank 0 with dst_dev = NULL makes one)");
  const size_t bytes = (size_t)r->width * r->height * (size_t)r->assembledElem[slot];
  if (r->dAssembled[slot].cap < bytes) return fail(PT_ERR_STATE, "pt_readback_assembled: the assembled frame is smaller than the frame asked for");
  if (dst_bytes < bytes) return fail(PT_ERR_INVALID, "pt_readback_assembled: destination too small");
  int rc = sync_all(r);
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(hipMemcpy(dst, r->dAssembled[slot].p, bytes, hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_create(int width, int height, int device, PtRenderer **out) {
  if (!out) return fail(PT_ERR_INVALID, "pt_create: out is null");
  *out = nullptr;
  if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 30))
    return fail(PT_ERR_INVALID, "pt_create: bad frame size");
  if (!shader_paths_sorted()) return fail(PT_ERR_STATE, "pt_create: the surface-shader registry is not in sorted path order");
  int nDev = 0;
  PT_HIP_CHECK(hipGetDeviceCount(&nDev));
  if (device < 0 || device >= nDev) return fail(PT_ERR_INVALID, "pt_create: no such HIP device");
  PT_HIP_CHECK(hipSetDevice(device));
  PtRenderer *r = new (std::nothrow) PtRenderer();
  if (!r) return fail(PT_ERR_NOMEM, "pt_create: out of host memory");
  r->device = device;
  r->width = width;
  r->height = height;
  r->localRows = height;
  hipDeviceProp_t prop;
  hipError_t e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) { delete r; return fail(PT_ERR_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e)); }
  r->computeUnits = prop.multiProcessorCount;
  std::snprintf(r->deviceName, sizeof r->deviceName, "%s (%s)", prop.name, prop.gcnArchName);
  int rc = ensure_slots(r, r->framesInFlight);
  if (rc != PT_OK) { pt_destroy(r); return rc; }
  if (device < kMaxDevices && !g_staticScratchPtr[device]) {
    // once per device, with that device current: the address of this GPU's copy of the static scratch
    void *sp = nullptr;
    e = hipGetSymbolAddress(&sp, HIP_SYMBOL(g_staticScratch));
    if (e != hipSuccess || !sp) {
      pt_destroy(r);
      return fail(PT_ERR_HIP, std::string("hipGetSymbolAddress(static scratch): ") + hipGetErrorString(e));
    }
    g_staticScratchPtr[device] = sp;
  }
  for (hipEvent_t *ev : {&r->evBegin, &r->evEnd}) {
    e = hipEventCreate(ev);
    if (e != hipSuccess) {
      pt_destroy(r);
      return fail(PT_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
    }
  }
  rc = ensure_outputs(r, 0);
  if (rc != PT_OK) { pt_destroy(r); return rc; }
  *out = r;
  return PT_OK;
}

void pt_destroy(PtRenderer *r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  for (int k = 0; k < r->numSlots; ++k)
    if (r->stream[k]) (void)hipStreamSynchronize(r->stream[k]);
  DevBuf *bufs[] = {&r->dVoxels, &r->dBricks, &r->dAccum, &r->dPlanes, &r->dWaveLog};
  for (DevBuf *b : bufs) b->release();
  for (int k = 0; k < PtRenderer::kMaxSlots; ++k) {
    r->dRgba8[k].release(); r->dF32[k].release(); r->dScratch[k].release(); r->dTileOrder[k].release();
    r->dGathered[k].release(); r->dAssembled[k].release(); r->dScene[k].release();
    if (r->hStage[k]) (void)hipHostFree(r->hStage[k]);
    if (r->evCopied[k]) (void)hipEventDestroy(r->evCopied[k]);
    if (r->evCross[k]) (void)hipEventDestroy(r->evCross[k]);
    if (r->ownStream[k]) (void)hipStreamDestroy(r->ownStream[k]);
  }
  if (r->evBegin) (void)hipEventDestroy(r->evBegin);
  if (r->evEnd) (void)hipEventDestroy(r->evEnd);
  delete r;
}

int pt_update_scene(PtRenderer *r, const PtGPUSphere *spheres, int num_spheres, const PtGPUEllipsoid *ellipsoids,
                    int num_ellipsoids, const PtGPUMaterial *materials, int num_materials, const PtGPULight *lights,
                    int num_lights, const PtGPUVolumetricData *volumes, int num_volumes, const uint8_t *voxels,
                    size_t voxel_bytes) {
  if (!r) return fail(PT_ERR_INVALID, "pt_update_scene: renderer is null");
  if (num_spheres < 0 || num_ellipsoids < 0 || num_materials < 0 || num_lights < 0 || num_volumes < 0)
    return fail(PT_ERR_INVALID, "pt_update_scene: negative count");
  if (num_spheres > (1 << 24)) return fail(PT_ERR_INVALID, "pt_update_scene: more than 2^24 spheres");
  if ((num_spheres && !spheres) || (num_ellipsoids && !ellipsoids) || (num_materials && !materials) ||
      (num_lights && !lights) || (num_volumes && !volumes))
    return fail(PT_ERR_INVALID, "pt_update_scene: null array with a non-zero count");
  for (int i = 0; i < num_spheres; ++i)
    if (spheres[i].materialIndex < 0 || spheres[i].materialIndex >= num_materials)
      return fail(PT_ERR_INVALID, "pt_update_scene: sphere " + std::to_string(i) + " has material index " +
                                      std::to_string(spheres[i].materialIndex) + " outside [0, " +
                                      std::to_string(num_materials) + ")");
  for (int i = 0; i < num_ellipsoids; ++i)
    if (ellipsoids[i].materialIndex < 0 || ellipsoids[i].materialIndex >= num_materials)
      return fail(PT_ERR_INVALID, "pt_update_scene: ellipsoid " + std::to_string(i) + " has a material index out of range");
  if (num_volumes > 0) {
    const PtGPUVolumetricData &v = volumes[0];
    if (v.materialIndex < 0 || v.materialIndex >= num_materials)
      return fail(PT_ERR_INVALID, "pt_update_scene: volume 0 has a material index out of range");
    if (v.resolution_x <= 0 || v.resolution_y <= 0 || v.resolution_z <= 0 ||
        (long long)v.resolution_x * v.resolution_y * v.resolution_z > 0x7fffffffll)
      return fail(PT_ERR_INVALID, "pt_update_scene: volume 0 resolution must be positive with fewer than 2^31 voxels");
  }
  PT_HIP_CHECK(hipSetDevice(r->device));
  // Nothing here waits for the frames in flight unless the voxels change (below): the records go into the host copy
  // of the scene, and every frame slot refreshes its device copy on its own stream before its next frame.

  // AoS records -> one scene image (see FrameParams in pt_device.hpp) + the plain sphere array.
  if (num_materials > 65536) return fail(PT_ERR_INVALID, "pt_update_scene: more than 65536 materials");
  // Spheres: the reference repacks to {center, radius, materialIndex}
  // (reference src/VulkanRenderer.cpp:803-822); colour is dropped.
  std::vector<float4> hs((size_t)num_spheres);
  bool sceneHuge = false;
  for (int i = 0; i < num_spheres; ++i) {
    hs[i] = make_float4(spheres[i].center[0], spheres[i].center[1], spheres[i].center[2], spheres[i].radius);
    for (float v : {hs[i].x, hs[i].y, hs[i].z, hs[i].w})
      if (!(std::fabs(v) <= 1.0e9f)) sceneHuge = true;
  }
  // The dense loop's filter (FrameParams::sphereFilter): k = |c|^2 - r^2 - 2^-17 (|c|^2 + r^2) - 1e-30, rounded down.
  // Its margin widens every sphere by about 2^-17 (|o|^2 + |c|^2) in r^2; it is used when that is small
  // next to r^2 for the scene's spheres on average (taking |o| ~ |c|), else the loop computes discriminants.
  std::vector<float> hk((size_t)num_spheres);
  bool sphereFilter = false;
  if (!sceneHuge && num_spheres > 0 && !(r->tune & 4)) {
    double widen = 0.0;
    for (int i = 0; i < num_spheres; ++i) {
      const double c2 = (double)hs[i].x * hs[i].x + (double)hs[i].y * hs[i].y + (double)hs[i].z * hs[i].z;
      const double r2 = (double)hs[i].w * hs[i].w;
      const double margin = 0x1p-17 * (c2 + r2) + 1.0e-30;
      hk[i] = std::nextafterf((float)(c2 - r2 - margin), -INFINITY);
      widen += std::min(r2 > 0.0 ? 0x1p-17 * (2.0 * c2 + r2) / r2 : 4.0, 4.0);
    }
    sphereFilter = widen / num_spheres <= 0.5 || (r->tune & 8);
  }
  std::vector<float4> img;
  // The intersection loop reads spheres two at a time:
  //   (cx0, cx1, cy0, cy1) (cz0, cz1, w0, w1),  w = radius or the filter's k
  // padded with copies of the last sphere to whole groups of eight (the copies
  // can never win: their index is >= the sphere count).
  const int nGroups = (num_spheres + 7) / 8;
  for (int p = 0; p < nGroups * 4; ++p) {
    const int i0 = std::min(2 * p, num_spheres - 1), i1 = std::min(2 * p + 1, num_spheres - 1);
    const float4 &s0 = hs[i0];
    const float4 &s1 = hs[i1];
    img.push_back(make_float4(s0.x, s1.x, s0.y, s1.y));
    img.push_back(make_float4(s0.z, s1.z, sphereFilter ? hk[i0] : s0.w, sphereFilter ? hk[i1] : s1.w));
  }
  // radius per sphere, for the evaluation of the candidates
  const int offRadius = (int)img.size();
  for (int i = 0; i < num_spheres; i += 4) {
    auto rad = [&](int j) { return j < num_spheres ? hs[j].w : 0.0f; };
    img.push_back(make_float4(rad(i), rad(i + 1), rad(i + 2), rad(i + 3)));
  }
  // Ellipsoids: centre + qw, 1/radii + max radius, -q.xyz, -2 q.xyz.  1/radii is
  // the same IEEE division hitEllipsoid does per test (reference
  // shaders/raytrace.comp:178); negation and doubling are exact.
  const int offEll = (int)img.size();
  for (int i = 0; i < num_ellipsoids; ++i) {
    const PtGPUEllipsoid &e = ellipsoids[i];
    float maxR = std::max(std::max(e.radii[0], e.radii[1]), e.radii[2]); // raytrace.comp:467
    img.push_back(make_float4(e.center[0], e.center[1], e.center[2], e.rotation[3]));
    img.push_back(make_float4(1.0f / e.radii[0], 1.0f / e.radii[1], 1.0f / e.radii[2], maxR));
    img.push_back(make_float4(-e.rotation[0], -e.rotation[1], -e.rotation[2], 0.0f));
    img.push_back(make_float4(-2.0f * e.rotation[0], -2.0f * e.rotation[1], -2.0f * e.rotation[2], 0.0f));
  }
  // Per pair of ellipsoids, what the packed candidate loop reads (see the kernel): centre, the two constants of the
  // bounding-sphere test, the matrix M = diag(1/radii) R(conjugate q) and the two constants of its margin, every
  // field as (first, second).
  //   R = I + 2 qw K + 2 K^2, K = cross-product matrix of -q.xyz: the map rotateByQuaternion applies
  //   (raytrace.comp:99-115), for whatever q is stored (unit or not); 1/radii as the shader's float division.
  //   mu: an upper bound of |M|_2 (sqrt of the largest absolute row sum of M^T M);
  //   g = max(1/radii) (1 + 2|q||qw| + 2|q|^2) / mu: how large the intermediates of the shader's own evaluation
  //   (and with them its rounding errors) can get against |M|.
  //   Rounding analysis (u = 2^-24, Y = mu |o - c|, Z = mu |d|): the shader's float discriminant is within
  //   u Z^2 ((112 g + 16.1) Y^2 + 28 g + 6.1) of the exact one, the loop's within u Z^2 (72.4 Y^2 + 20.1);
  //   k1 and k2 are the sums with a tenth added, rounded up.
  //   Before that test the loop runs a cheaper one on the bounding sphere (12 packed instructions instead of 38), and
  //   only pairs of ellipsoids that some ray of the wave passes go on to the matrix test.  With sigma the smallest
  //   singular value of M (every point of the ellipsoid lies within Rb = 1/sigma of the centre), x = o - c and
  //   Delta the distance of the ray's line from the centre: the exact discriminant is D = |z|^2 (1 - delta^2) with
  //   delta >= sigma Delta the distance of the line from the origin in the scaled frame and |z| >= sigma |d|, so a
  //   non-negative float discriminant in the shader (D >= -|d|^2 (k1 |x|^2 + k2), k1 and k2 as above, which cover
  //   more than the shader's own rounding) needs Delta^2 <= Rb^2 (1 + Rb^2 (k1 |x|^2 + k2)), i.e.
  //     V = (x.d)^2 - |d|^2 (|x|^2 (1 - B1) - B2) >= 0,   B1 = Rb^4 k1, B2 = Rb^2 (1 + Rb^2 k2).
  //   Evaluated in float32 (x rounded, two three-term dot products, three fused steps, one product) V is off by at
  //   most u (19.4 |x|^2 |d|^2 + 6.2 |d|^2 B2); the constants handed to the loop, c1 = 1 - B1 - 24 u (rounded
  //   down) and c2 = B2 (1 + 8 u) + 1e-30 (rounded up), raise V by more than that.
  struct EllFilter { float c[3]; float m[9]; float k1, k2; float c1, nc2; };
  auto ellipsoid_filter = [](const PtGPUEllipsoid &e) {
    EllFilter f{};
    for (int k = 0; k < 3; ++k) f.c[k] = e.center[k];
    const double q[3] = {-(double)e.rotation[0], -(double)e.rotation[1], -(double)e.rotation[2]};
    const double qw = (double)e.rotation[3];
    const double K[3][3] = {{0.0, -q[2], q[1]}, {q[2], 0.0, -q[0]}, {-q[1], q[0], 0.0}};
    double M[3][3], sigma = 0.0;
    bool ok = true;
    for (int i = 0; i < 3; ++i) {
      const double inv = (double)(1.0f / e.radii[i]);
      sigma = std::max(sigma, std::fabs(inv));
      for (int j = 0; j < 3; ++j) {
        double kk = 0.0;
        for (int l = 0; l < 3; ++l) kk += K[i][l] * K[l][j];
        M[i][j] = inv * ((i == j ? 1.0 : 0.0) + 2.0 * qw * K[i][j] + 2.0 * kk);
        if (!std::isfinite(M[i][j])) ok = false;
      }
    }
    double mu2 = 0.0;
    for (int i = 0; i < 3; ++i) {
      double row = 0.0;
      for (int j = 0; j < 3; ++j) {
        double mtm = 0.0;
        for (int l = 0; l < 3; ++l) mtm += M[l][i] * M[l][j];
        row += std::fabs(mtm);
      }
      mu2 = std::max(mu2, row);
    }
    const double mu = std::sqrt(mu2) * (1.0 + 1e-12);
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    const double G = 1.0 + 2.0 * qn * std::fabs(qw) + 2.0 * qn * qn;
    const double g = sigma * G / mu;
    const double u = 0x1p-24;
    const double k1 = (124.0 * g + 98.0) * u * mu * mu * mu * mu, k2 = (31.0 * g + 29.0) * u * mu * mu;
    f.c1 = 0.0f;        // bounding-sphere test: always passed unless the constants below can be trusted
    f.nc2 = -INFINITY;
    if (ok && std::isfinite(k1) && std::isfinite(k2) && mu > 0.0 && k1 < 1.0e30 && k2 < 1.0e30) {
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) f.m[3 * i + j] = (float)M[i][j];
      f.k1 = std::nextafterf((float)k1, INFINITY);
      f.k2 = std::nextafterf((float)k2, INFINITY);
      // smallest eigenvalue of M^T M (cyclic Jacobi rotations on the symmetric 3 x 3 matrix)
      double A[3][3];
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
          A[i][j] = 0.0;
          for (int l = 0; l < 3; ++l) A[i][j] += M[l][i] * M[l][j];
        }
      for (int sweep = 0; sweep < 30; ++sweep) {
        const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
        if (off <= 1e-300 || off <= 1e-17 * (std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]))) break;
        for (int pI = 0; pI < 2; ++pI)
          for (int qI = pI + 1; qI < 3; ++qI) {
            if (A[pI][qI] == 0.0) continue;
            const double theta = (A[qI][qI] - A[pI][pI]) / (2.0 * A[pI][qI]);
            const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
            const double cs = 1.0 / std::sqrt(tt * tt + 1.0), sn = tt * cs;
            for (int l = 0; l < 3; ++l) { // A <- A J
              const double alp = A[l][pI], alq = A[l][qI];
              A[l][pI] = cs * alp - sn * alq;
              A[l][qI] = sn * alp + cs * alq;
            }
            for (int l = 0; l < 3; ++l) { // A <- J^T A
              const double apl = A[pI][l], aql = A[qI][l];
              A[pI][l] = cs * apl - sn * aql;
              A[qI][l] = sn * apl + cs * aql;
            }
          }
      }
      const double lmin = std::min(A[0][0], std::min(A[1][1], A[2][2])) * (1.0 - 1e-9);
      if (std::isfinite(lmin) && lmin > 0.0) {
        const double Rb2 = 1.0 / lmin; // Rb^2
        const double B1 = Rb2 * Rb2 * (double)f.k1, B2 = Rb2 * (1.0 + Rb2 * (double)f.k2);
        const double c1 = 1.0 - B1 - 24.0 * u, c2 = B2 * (1.0 + 8.0 * u) + 1.0e-30;
        if (std::isfinite(c1) && std::isfinite(c2) && c2 < 1.0e30 && c1 > -1.0e30) {
          f.c1 = std::nextafterf((float)c1, -INFINITY);
          f.nc2 = -std::nextafterf((float)c2, INFINITY);
        }
      }
    } else { // nothing to go by: always a candidate (M = 0 gives D' = 0, the margin is infinite)
      f.k1 = 0.0f;
      f.k2 = INFINITY;
    }
    return f;
  };
  // (a caller that moves a few ellipsoids per frame pays for those only: what was derived from a record is kept
  // and reused while the record's bytes stay the same)
  static_assert(sizeof(EllFilter) == 16 * sizeof(float), "EllFilter is kept as 16 floats");
  std::vector<float> derived((size_t)num_ellipsoids * 16);
  for (int i = 0; i < num_ellipsoids; ++i) {
    EllFilter f;
    if ((size_t)i < r->ellSeen.size() && std::memcmp(&r->ellSeen[i], &ellipsoids[i], sizeof(PtGPUEllipsoid)) == 0)
      std::memcpy(&f, &r->ellDerived[(size_t)i * 16], sizeof f);
    else
      f = ellipsoid_filter(ellipsoids[i]);
    std::memcpy(&derived[(size_t)i * 16], &f, sizeof f);
  }
  r->ellSeen.assign(ellipsoids, ellipsoids + num_ellipsoids);
  r->ellDerived.swap(derived);
  auto filter_of = [&](int i) { EllFilter f; std::memcpy(&f, &r->ellDerived[(size_t)i * 16], sizeof f); return f; };
  const int offEllPair = (int)img.size();
  for (int p = 0; 2 * p < num_ellipsoids; ++p) {
    const EllFilter a = filter_of(2 * p);
    const EllFilter b = filter_of(std::min(2 * p + 1, num_ellipsoids - 1));
    img.push_back(make_float4(a.c[0], b.c[0], a.c[1], b.c[1]));
    img.push_back(make_float4(a.c[2], b.c[2], a.c1, b.c1));
    img.push_back(make_float4(a.nc2, b.nc2, a.m[0], b.m[0]));
    img.push_back(make_float4(a.m[1], b.m[1], a.m[2], b.m[2]));
    img.push_back(make_float4(a.m[3], b.m[3], a.m[4], b.m[4]));
    img.push_back(make_float4(a.m[5], b.m[5], a.m[6], b.m[6]));
    img.push_back(make_float4(a.m[7], b.m[7], a.m[8], b.m[8]));
    img.push_back(make_float4(a.k1, b.k1, a.k2, b.k2));
  }
  const int offMat = (int)img.size();
  for (int i = 0; i < num_materials; ++i) {
    const float *f = materials[i].colorAndAmbient;
    for (int k = 0; k < 5; ++k) img.push_back(make_float4(f[4 * k], f[4 * k + 1], f[4 * k + 2], f[4 * k + 3]));
  }
  // Lights: vec4(position, intensity) (reference src/VulkanRenderer.cpp:876-883).
  const int offLight = (int)img.size();
  for (int i = 0; i < num_lights; ++i)
    img.push_back(make_float4(lights[i].position[0], lights[i].position[1], lights[i].position[2], lights[i].intensity));
  // Emissive objects, in object order: what calculateLighting's scan over all
  // objects and their materials finds (reference shaders/raytrace.comp:367-480).
  auto bits = [](int v) { float f; std::memcpy(&f, &v, 4); return f; };
  const int offEmS = (int)img.size();
  int nEmS = 0;
  for (int i = 0; i < num_spheres; ++i) {
    const PtGPUMaterial &m = materials[spheres[i].materialIndex];
    if (!(m.transparencyEmissive[1] > 0.0f)) continue;
    img.push_back(hs[i]);
    img.push_back(make_float4(m.colorAndAmbient[0], m.colorAndAmbient[1], m.colorAndAmbient[2], m.transparencyEmissive[1]));
    img.push_back(make_float4(bits((int)m.transparencyEmissive[2]), bits(i), 0.0f, 0.0f));
    ++nEmS;
  }
  const int offEmE = (int)img.size();
  int nEmE = 0;
  for (int i = 0; i < num_ellipsoids; ++i) {
    const PtGPUEllipsoid &e = ellipsoids[i];
    const PtGPUMaterial &m = materials[e.materialIndex];
    if (!(m.transparencyEmissive[1] > 0.0f)) continue;
    float maxR = std::max(std::max(e.radii[0], e.radii[1]), e.radii[2]);
    img.push_back(make_float4(e.center[0], e.center[1], e.center[2], maxR));
    img.push_back(make_float4(m.colorAndAmbient[0], m.colorAndAmbient[1], m.colorAndAmbient[2], m.transparencyEmissive[1]));
    img.push_back(make_float4(bits(i), 0.0f, 0.0f, 0.0f));
    ++nEmE;
  }
  // uint16 material index per sphere, then per ellipsoid
  const int offMatIdx = (int)img.size();
  {
    std::vector<uint16_t> mi((size_t)num_spheres + num_ellipsoids);
    for (int i = 0; i < num_spheres; ++i) mi[i] = (uint16_t)spheres[i].materialIndex;
    for (int i = 0; i < num_ellipsoids; ++i) mi[num_spheres + i] = (uint16_t)ellipsoids[i].materialIndex;
    const size_t n4 = (mi.size() * 2 + 15) / 16;
    img.resize(img.size() + n4, make_float4(0, 0, 0, 0));
    if (!mi.empty()) std::memcpy(&img[offMatIdx], mi.data(), mi.size() * 2);
  }

  // The host copy of the scene: the image, then the plain sphere array.
  {
    const size_t imageBytes = img.size() * sizeof(float4);
    const size_t spheresOffset = (imageBytes + 255) & ~(size_t)255;
    const size_t total = spheresOffset + std::max<size_t>(hs.size(), 1) * sizeof(float4);
    r->hostScene.assign(total, 0);
    if (imageBytes) std::memcpy(r->hostScene.data(), img.data(), imageBytes);
    if (!hs.empty()) std::memcpy(r->hostScene.data() + spheresOffset, hs.data(), hs.size() * sizeof(float4));
    r->sceneBytes = total;
    r->spheresOffset = spheresOffset;
    r->sceneVersion++;
    // room in every slot that exists (so that pt_render_static, which may not allocate, finds it)
    for (int k = 0; k < r->numSlots; ++k) {
      int rcRoom = reserve_scene(r, k);
      if (rcRoom != PT_OK) return rcRoom;
    }
  }

  if (num_volumes > 0) {
    r->volume0 = volumes[0];
    size_t need = (size_t)volumes[0].resolution_x * volumes[0].resolution_y * volumes[0].resolution_z;
    const PtGPUVolumetricData &v = volumes[0];
    const bool newVoxels = voxels && voxel_bytes > 0 && !r->voxelsUploaded;
    const bool grow = !newVoxels && (!r->voxelsUploaded || r->voxelBytes < need);
    const bool rebrick = newVoxels || grow || r->brickRes[0] != v.resolution_x || r->brickRes[1] != v.resolution_y ||
                         r->brickRes[2] != v.resolution_z;
    if (rebrick) {
      // The voxel buffers change under the frames in flight: wait for them (first upload, or a new resolution).
      int rcSync = sync_all(r);
      if (rcSync != PT_OK) return rcSync;
    }
    if (newVoxels) {
      // Uploaded once (reference src/VulkanRenderer.cpp:961-987).  All of the caller's bytes are kept: the
      // reference keeps the file's bytes and indexes them with whatever resolution the current scene
      // states.  Voxels the file did not supply read as 0.
      const size_t cap = std::max(need, voxel_bytes);
      PT_HIP_CHECK(r->dVoxels.reserve(cap + 16));
      PT_HIP_CHECK(hipMemsetAsync(r->dVoxels.p, 0, cap + 16, r->stream[0]));
      PT_HIP_CHECK(hipMemcpyAsync(r->dVoxels.p, voxels, voxel_bytes, hipMemcpyHostToDevice, r->stream[0]));
      r->voxelsUploaded = true;
      r->voxelBytes = cap;
      r->brickRes[0] = 0; // rebuild the bricked copy
    } else if (grow) {
      // A volume without data marches through zeros; a volume whose resolution outgrows what was
      // uploaded keeps the uploaded bytes (re-indexed, as in the reference) and reads zeros beyond them.
      DevBuf bigger;
      PT_HIP_CHECK(bigger.reserve(need + 16));
      PT_HIP_CHECK(hipMemsetAsync(bigger.p, 0, need + 16, r->stream[0]));
      if (r->voxelsUploaded && r->dVoxels.p && r->voxelBytes > 0)
        PT_HIP_CHECK(hipMemcpyAsync(bigger.p, r->dVoxels.p, r->voxelBytes, hipMemcpyDeviceToDevice, r->stream[0]));
      PT_HIP_CHECK(hipStreamSynchronize(r->stream[0]));
      r->dVoxels.release();
      r->dVoxels = bigger;
      r->voxelBytes = need;
      r->brickRes[0] = 0;
    }
    // Bricked copy for the march (the linear copy stays: the reference keeps the file's bytes and
    // indexes them with whatever resolution the current scene states).
    if (r->brickRes[0] != v.resolution_x || r->brickRes[1] != v.resolution_y || r->brickRes[2] != v.resolution_z) {
      const int nbx = (v.resolution_x + 7) / 8, nby = (v.resolution_y + 7) / 8, nbz = (v.resolution_z + 7) / 8;
      const long long brickBytes = (long long)nbx * nby * nbz * 512;
      if (brickBytes <= 0x7fffffffll && (long long)nbx * nby < 32768) { // the march indexes with 32-bit integers and unsigned 24-bit factors (512 nbx nby)
        PT_HIP_CHECK(r->dBricks.reserve((size_t)brickBytes + 16)); // the trilinear filter reads cells two bytes at a time
        const int grid = (int)std::min<long long>((brickBytes / 4 + 255) / 256, 65536);
        hipLaunchKernelGGL(ptdev::pt_brick_volume_kernel, dim3(grid), dim3(256), 0, r->stream[0], (const uint8_t *)r->dVoxels.p,
                           (uint8_t *)r->dBricks.p, v.resolution_x, v.resolution_y, v.resolution_z, nbx, nby, nbz);
        PT_HIP_CHECK(hipGetLastError());
        r->brickRes[0] = v.resolution_x; r->brickRes[1] = v.resolution_y; r->brickRes[2] = v.resolution_z;
      } else {
        r->brickRes[0] = r->brickRes[1] = r->brickRes[2] = 0;
      }
    }
    // The caller's voxel bytes die at return and the other slots' streams must see the new volume: finish first.
    if (rebrick) PT_HIP_CHECK(hipStreamSynchronize(r->stream[0]));
  }

  r->sceneHuge = sceneHuge;
  {
    // Magnitude bounds for the deferred ellipsoid path.  |rotate(v)| <= |v| (1 + 2|q||qw| + 2|q|^2).
    float cMax = 0.0f, scale = 0.0f;
    bool ok = true;
    for (int i = 0; i < num_ellipsoids; ++i) {
      const PtGPUEllipsoid &e = ellipsoids[i];
      float qn = 0.0f;
      for (int k = 0; k < 3; ++k) {
        const float c = std::fabs(e.center[k]), rad = std::fabs(e.radii[k]);
        if (!(c <= 1.0e9f) || !(rad >= 1.0e-9f && rad <= 1.0e9f)) ok = false;
        cMax = std::max(cMax, c);
        qn += e.rotation[k] * e.rotation[k];
      }
      const float qw = std::fabs(e.rotation[3]);
      if (!(qn <= 1.0e6f) || !(qw <= 1.0e3f)) { ok = false; continue; }
      const float grow = 1.0f + 2.0f * std::sqrt(qn) * qw + 2.0f * qn;
      for (int k = 0; k < 3; ++k) scale = std::max(scale, grow / std::fabs(e.radii[k]));
    }
    r->ellCenterMax = ok ? cMax : 0.0f;
    r->ellScale = ok ? 2.0f * scale : INFINITY; // 2: slack for the three-term sums
  }
  r->sceneBounces = false;
  auto bounces = [&](int m) { return materials[m].diffuseSpecularShiny[3] > 0.01f || materials[m].transparencyEmissive[0] > 0.01f; };
  for (int i = 0; i < num_spheres && !r->sceneBounces; ++i) r->sceneBounces = bounces(spheres[i].materialIndex);
  for (int i = 0; i < num_ellipsoids && !r->sceneBounces; ++i) r->sceneBounces = bounces(ellipsoids[i].materialIndex);
  r->numSpheres = num_spheres;
  r->numEllipsoids = num_ellipsoids;
  r->numMaterials = num_materials;
  r->numLights = num_lights;
  r->numVolumes = num_volumes;
  r->numEmSpheres = nEmS;
  r->numEmEllipsoids = nEmE;
  r->imageFloat4s = (int)img.size();
  r->offRadius = offRadius;
  r->sphereFilter = sphereFilter;
  r->offEll = offEll;
  r->offEllPair = offEllPair;
  r->offMat = offMat;
  r->offLight = offLight;
  r->offEmS = offEmS;
  r->offEmE = offEmE;
  r->offMatIdx = offMatIdx;
  r->matIdxEllBase = num_spheres;
  r->sceneValid = true;
  return PT_OK;
}

int pt_reset_voxels(PtRenderer *r) {
  if (!r) return fail(PT_ERR_INVALID, "pt_reset_voxels: renderer is null");
  r->voxelsUploaded = false;
  return PT_OK;
}

int pt_render(PtRenderer *r, const PtPushConstants *pc) { return launch_frame(r, pc, 0); }

int pt_render_static(PtRenderer *r, const PtPushConstants *pc, void *rgba8_dev, void *f32_dev, void *hip_stream) {
  if (!r || !pc || !rgba8_dev) return fail(PT_ERR_INVALID, "pt_render_static: renderer, push constants and the rgba8 frame are required");
  if (r->device >= kMaxDevices || !g_staticScratchPtr[r->device])
    return fail(PT_ERR_STATE, "pt_render_static: no static scratch on this device");
  if (r->accumulate) return fail(PT_ERR_STATE, "pt_render_static: not inside pt_render_samples");
  char *const base = (char *)g_staticScratchPtr[r->device];
  FrameTarget t;
  t.stream = hip_stream ? (hipStream_t)hip_stream : r->stream[0]; // slot 0's stream: what pt_present waits for
  t.rgba8 = rgba8_dev;
  t.f32 = f32_dev;
  t.scratch = base;
  t.tileOrder = (int *)(base + kStaticOrderOffset);
  return launch_frame(r, pc, 0, &t);
}

int pt_render_samples(PtRenderer *r, const PtPushConstants *pc, int first_sample, int num_samples, int keep) {
  if (!r || !pc) return fail(PT_ERR_INVALID, "pt_render_samples: null argument");
  if (first_sample < 0 || num_samples < 1) return fail(PT_ERR_INVALID, "pt_render_samples: first_sample >= 0 and num_samples >= 1 required");
  if (!keep) r->accumCount = 0;
  // Samples per launch.  One launch per sample ends every sample on its slowest pixels with most of the chip idle;
  // a launch that draws several samples hands out (tile, sample) tickets and ends once.  Each sample of the launch
  // needs a float4 plane of the frame (4K: 133 MB), summed up in sample order afterwards, so the result is the one the
  // launches one by one give.  (Frames without solids: the volume kernel's waves take (tile, sample) pairs the same way.)
  int batch = 1;
  if (r->sampleBatch != 1 && num_samples > 1) {
    const size_t framePixels = (size_t)r->width * (size_t)std::max(r->localRows, 1);
    size_t freeBytes = 0, totalBytes = 0;
    PT_HIP_CHECK(hipSetDevice(r->device));
    PT_HIP_CHECK(hipMemGetInfo(&freeBytes, &totalBytes));
    const size_t budget = std::min(kPlaneBudget, (freeBytes + r->dPlanes.cap) / 2); // never more than half of what is free
    size_t fit = budget / std::max<size_t>(framePixels * sizeof(float4), 1);
    fit = std::min(fit, (size_t)0x7fffffff / std::max<size_t>(framePixels, 1)); // plane offsets are 32-bit pixel indices
    if (r->sampleBatch > 1) fit = std::min(fit, (size_t)r->sampleBatch);
    batch = (int)std::max<size_t>(1, std::min<size_t>(fit, (size_t)num_samples));
  }
  r->accumulate = true;
  int rc = PT_OK;
  for (int s = 0; s < num_samples && rc == PT_OK;) {
    const int n = std::min(batch, num_samples - s);
    r->sampleIndex = first_sample + s;
    r->batchSamples = n;
    rc = launch_frame(r, pc, 0);
    if (rc == PT_OK) r->accumCount += n;
    s += n;
  }
  r->accumulate = false;
  r->sampleIndex = 0;
  r->batchSamples = 1;
  return rc;
}

int pt_render_counted(PtRenderer *r, const PtPushConstants *pc, uint64_t *counters) {
  if (!counters) return fail(PT_ERR_INVALID, "pt_render_counted: counters is null");
  int rc = launch_frame(r, pc, 1);
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(hipStreamSynchronize(r->stream[0]));
  PT_HIP_CHECK(hipMemcpy(counters, (char *)r->dScratch[0].p + 64, PT_CNT_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost));
  // the per-phase cycle sums come from a second frame drawn by the production code with timers only
  rc = launch_frame(r, pc, 2);
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(hipStreamSynchronize(r->stream[0]));
  uint64_t cyc[PT_CNT_COUNT];
  PT_HIP_CHECK(hipMemcpy(cyc, (char *)r->dScratch[0].p + 64, sizeof cyc, hipMemcpyDeviceToHost));
  for (int i = PT_CNT_CYC_REFILL; i < PT_CNT_COUNT; ++i) counters[i] = cyc[i];
  return PT_OK;
}

int pt_wave_timeline(PtRenderer *r, const PtPushConstants *pc, uint64_t *out, size_t capacity_waves, int *num_waves) {
  if (!r || !pc || !out || !num_waves) return fail(PT_ERR_INVALID, "pt_wave_timeline: null argument");
  // one record per wave of the launch: at most 7 workgroups per compute unit of at most 16 waves (the tuning switches
  // can ask for that many; production launches have 16 waves per compute unit)
  const size_t maxWaves = (size_t)r->computeUnits * 7 * 16;
  PT_HIP_CHECK(hipSetDevice(r->device));
  int rc = sync_all(r);
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(r->dWaveLog.reserve(maxWaves * 32));
  PT_HIP_CHECK(hipMemset(r->dWaveLog.p, 0, maxWaves * 32));
  r->wantWaveLog = true;
  rc = launch_frame(r, pc, 2);
  r->wantWaveLog = false;
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(hipStreamSynchronize(r->stream[0]));
  const size_t waves = (size_t)r->lastBlocks * (size_t)(r->lastThreads / 64);
  *num_waves = (int)waves;
  const size_t n = std::min(waves, capacity_waves);
  PT_HIP_CHECK(hipMemcpy(out, r->dWaveLog.p, n * 32, hipMemcpyDeviceToHost));
  return PT_OK;
}

// The slot whose frame readback and pt_device_* show: the most recent one, or with PT_OPT_PRESENT_LAG the one the last
// pt_present declared complete.
static int shown_slot(const PtRenderer *r) { return (r->presentLag && r->shownSlot >= 0) ? r->shownSlot : r->lastSlot; }

// Wait for what readback is about to copy: every frame in flight, or with PT_OPT_PRESENT_LAG the shown slot's frame only
// (the frame drawn after it keeps drawing).
static int sync_shown(PtRenderer *r) {
  if (!(r->presentLag && r->shownSlot >= 0)) return sync_all(r);
  PT_HIP_CHECK(hipSetDevice(r->device));
  PT_HIP_CHECK(hipStreamSynchronize(r->stream[r->shownSlot]));
  return PT_OK;
}

int pt_present(PtRenderer *r) {
  if (!r) return fail(PT_ERR_INVALID, "pt_present: renderer is null");
  const int n = r->framesInFlight;
  const int before = (r->lastSlot + n - 1) % std::max(n, 1);
  if (r->presentLag && pipelined(r) && r->renderedSincePresent && r->slotDrawn[before]) {
    // every frame but the one just queued; that one is shown by the next pt_present
    PT_HIP_CHECK(hipSetDevice(r->device));
    for (int k = 0; k < r->numSlots; ++k)
      if (k != r->lastSlot) PT_HIP_CHECK(hipStreamSynchronize(r->stream[k]));
    r->shownSlot = before;
    r->renderedSincePresent = false;
    return PT_OK;
  }
  r->shownSlot = -1;
  r->renderedSincePresent = false;
  return sync_all(r);
}

int pt_readback_rgba8(PtRenderer *r, uint8_t *dst, size_t dst_bytes) {
  if (!r || !dst) return fail(PT_ERR_INVALID, "pt_readback_rgba8: null argument");
  size_t bytes = (size_t)r->width * r->localRows * 4;
  if (dst_bytes < bytes) return fail(PT_ERR_INVALID, "pt_readback_rgba8: destination too small");
  int rc = sync_shown(r);
  if (rc != PT_OK) return rc;
  const void *src = slot_rgba8(r, shown_slot(r));
  PT_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_readback_f32(PtRenderer *r, float *dst, size_t dst_bytes) {
  if (!r || !dst) return fail(PT_ERR_INVALID, "pt_readback_f32: null argument");
  const void *src = slot_f32(r, shown_slot(r));
  if (!src) return fail(PT_ERR_STATE, "pt_readback_f32: float output is disabled");
  size_t bytes = (size_t)r->width * r->localRows * 16;
  if (dst_bytes < bytes) return fail(PT_ERR_INVALID, "pt_readback_f32: destination too small");
  int rc = sync_shown(r);
  if (rc != PT_OK) return rc;
  PT_HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_shader_count(void) { return kShaderCount; }

const char *pt_shader_path(int i) { return (i >= 0 && i < kShaderCount) ? kShaderPaths[i] : nullptr; }

int pt_shader_index(const char *path) {
  if (!path) return 0;
  for (int i = 0; i < kShaderCount; ++i)
    if (std::strcmp(path, kShaderPaths[i]) == 0) return i + 1;
  return 0;
}

const char *pt_last_error(void) { return g_last_error.c_str(); }

int pt_set_row_bands(PtRenderer *r, int band_rows, int band_index, int band_count) {
  if (!r) return fail(PT_ERR_INVALID, "pt_set_row_bands: renderer is null");
  if (band_count < 1 || band_index < 0 || band_index >= band_count || (band_count > 1 && band_rows < 1))
    return fail(PT_ERR_INVALID, "pt_set_row_bands: need band_rows >= 1 and 0 <= band_index < band_count");
  if (sync_all(r) != PT_OK) return PT_ERR_HIP; // the frame geometry changes under the frames in flight
  r->bandRows = band_rows;
  r->bandIndex = band_index;
  r->bandCount = band_count;
  r->localRows = compute_local_rows(r->height, band_rows, band_index, band_count);
  return PT_OK;
}

int pt_local_rows(PtRenderer *r) { return r ? r->localRows : 0; }

int pt_assemble_bands(PtRenderer *r, const void *src, void *dst, int elem_bytes, int band_rows, int band_count,
                      int stride_rows) {
  if (!r || !src || !dst) return fail(PT_ERR_INVALID, "pt_assemble_bands: null argument");
  if ((elem_bytes != 4 && elem_bytes != 16) || band_rows < 1 || band_count < 1 || stride_rows < 1)
    return fail(PT_ERR_INVALID, "pt_assemble_bands: bad geometry");
  int maxLocal = compute_local_rows(r->height, band_rows, 0, band_count);
  if (stride_rows < maxLocal) return fail(PT_ERR_INVALID, "pt_assemble_bands: stride_rows smaller than a rank's frame");
  PT_HIP_CHECK(hipSetDevice(r->device));
  dim3 block(256), grid((r->width + 255) / 256, std::min(r->height, 65535)); // the kernel strides over rows
  const hipStream_t astream = r->stream[r->lastSlot]; // behind the frame just drawn
  if (elem_bytes == 4)
    hipLaunchKernelGGL((ptdev::pt_assemble_bands_kernel<uint32_t>), grid, block, 0, astream, (const uint32_t *)src,
                       (uint32_t *)dst, r->width, r->height, band_rows, band_count, stride_rows);
  else
    hipLaunchKernelGGL((ptdev::pt_assemble_bands_kernel<float4>), grid, block, 0, astream, (const float4 *)src,
                       (float4 *)dst, r->width, r->height, band_rows, band_count, stride_rows);
  PT_HIP_CHECK(hipGetLastError());
  return PT_OK;
}

void *pt_device_rgba8(PtRenderer *r) { return r ? slot_rgba8(r, shown_slot(r)) : nullptr; }
void *pt_device_f32(PtRenderer *r) { return r ? slot_f32(r, shown_slot(r)) : nullptr; }
int pt_last_slot(PtRenderer *r) { return r ? r->lastSlot : 0; }

int pt_set_slot(PtRenderer *r, int slot, void *hip_stream, void *rgba8_dev, void *f32_dev) {
  if (!r) return fail(PT_ERR_INVALID, "pt_set_slot: renderer is null");
  if (slot < 0 || slot >= PtRenderer::kMaxSlots) return fail(PT_ERR_INVALID, "pt_set_slot: slot is 0 .. 7");
  if (!rgba8_dev && f32_dev) return fail(PT_ERR_INVALID, "pt_set_slot: a float frame needs an rgba8 frame too");
  int rc = sync_all(r);
  if (rc != PT_OK) return rc;
  if ((rc = ensure_slots(r, slot + 1)) != PT_OK) return rc;
  r->stream[slot] = hip_stream ? (hipStream_t)hip_stream : r->ownStream[slot];
  r->concurrency = 0;
  r->extRgba8[slot] = rgba8_dev;
  r->extF32[slot] = f32_dev;
  return PT_OK;
}

int pt_set_output(PtRenderer *r, void *rgba8_dev, void *f32_dev) {
  if (!r) return fail(PT_ERR_INVALID, "pt_set_output: renderer is null");
  if (!rgba8_dev && f32_dev) return fail(PT_ERR_INVALID, "pt_set_output: a float frame needs an rgba8 frame too");
  // a frame in flight keeps writing the buffers it was launched with, and readback names slot 0's current ones
  int rc = sync_all(r);
  if (rc != PT_OK) return rc;
  r->extRgba8[0] = rgba8_dev;
  r->extF32[0] = f32_dev;
  return PT_OK;
}

int pt_set_stream(PtRenderer *r, void *hip_stream) {
  if (!r) return fail(PT_ERR_INVALID, "pt_set_stream: renderer is null");
  int rc = sync_all(r);
  if (rc != PT_OK) return rc;
  r->stream[0] = hip_stream ? (hipStream_t)hip_stream : r->ownStream[0];
  r->concurrency = 0;
  return PT_OK;
}

int pt_stream_concurrency(PtRenderer *r, int *streams) {
  if (!r || !streams) return fail(PT_ERR_INVALID, "pt_stream_concurrency: null argument");
  PT_HIP_CHECK(hipSetDevice(r->device));
  if (r->concurrency == 0 || r->framesInFlight < 2) {
    int rc = measure_concurrency(r);
    if (rc != PT_OK) return rc;
  }
  *streams = r->concurrency;
  return PT_OK;
}

int pt_timer_begin(PtRenderer *r) {
  if (!r) return fail(PT_ERR_INVALID, "pt_timer_begin: renderer is null");
  // after everything queued so far on any slot; nothing queued later starts before it
  for (int k = 1; k < r->numSlots; ++k) {
    PT_HIP_CHECK(hipEventRecord(r->evCross[k], r->stream[k]));
    PT_HIP_CHECK(hipStreamWaitEvent(r->stream[0], r->evCross[k], 0));
  }
  PT_HIP_CHECK(hipEventRecord(r->evBegin, r->stream[0]));
  for (int k = 1; k < r->numSlots; ++k) PT_HIP_CHECK(hipStreamWaitEvent(r->stream[k], r->evBegin, 0));
  return PT_OK;
}
int pt_timer_end(PtRenderer *r) {
  if (!r) return fail(PT_ERR_INVALID, "pt_timer_end: renderer is null");
  for (int k = 1; k < r->numSlots; ++k) {
    PT_HIP_CHECK(hipEventRecord(r->evCross[k], r->stream[k]));
    PT_HIP_CHECK(hipStreamWaitEvent(r->stream[0], r->evCross[k], 0));
  }
  PT_HIP_CHECK(hipEventRecord(r->evEnd, r->stream[0]));
  return PT_OK;
}
int pt_timer_elapsed_ms(PtRenderer *r, float *ms) {
  if (!r || !ms) return fail(PT_ERR_INVALID, "pt_timer_elapsed_ms: null argument");
  PT_HIP_CHECK(hipEventSynchronize(r->evEnd));
  PT_HIP_CHECK(hipEventElapsedTime(ms, r->evBegin, r->evEnd));
  return PT_OK;
}

int pt_set_option(PtRenderer *r, int option, int value) {
  if (!r) return fail(PT_ERR_INVALID, "pt_set_option: renderer is null");
  switch (option) {
  case PT_OPT_FLOAT_OUTPUT: r->floatOutput = value != 0; return PT_OK;
  case PT_OPT_VOLUME_BRICKS: r->volumeBricks = value != 0; return PT_OK;
  case PT_OPT_FRAMES_IN_FLIGHT:
    if (value < 1 || value > PtRenderer::kMaxSlots) return fail(PT_ERR_INVALID, "pt_set_option: PT_OPT_FRAMES_IN_FLIGHT takes 1 .. 8");
    if (sync_all(r) != PT_OK) return PT_ERR_HIP;
    {
      int rcSlots = ensure_slots(r, value);
      if (rcSlots != PT_OK) return rcSlots;
    }
    r->framesInFlight = value;
    r->concurrency = 0;
    if (r->lastSlot >= value) r->lastSlot = 0;
    r->shownSlot = -1;
    for (bool &d : r->slotDrawn) d = false;
    return PT_OK;
  case PT_OPT_PRESENT_LAG:
    if (value != 0 && value != 1) return fail(PT_ERR_INVALID, "pt_set_option: PT_OPT_PRESENT_LAG takes 0 or 1");
    r->presentLag = value;
    r->shownSlot = -1;
    return PT_OK;
  case PT_OPT_SAMPLE_BATCH:
    if (value < 0) return fail(PT_ERR_INVALID, "pt_set_option: PT_OPT_SAMPLE_BATCH takes 0 (as many as fit), 1 (one launch per sample) or the largest number of samples per launch");
    r->sampleBatch = value;
    return PT_OK;
  case PT_OPT_VOLUME_FILTER:
    if (value != 0 && value != 1) return fail(PT_ERR_INVALID, "pt_set_option: PT_OPT_VOLUME_FILTER takes 0 or 1");
    r->volumeFilter = value;
    return PT_OK;
  case PT_OPT_TILE_ORDER:
    if (value < 0 || value > 2) return fail(PT_ERR_INVALID, "pt_set_option: PT_OPT_TILE_ORDER takes 0, 1 or 2");
    r->tileOrder = value;
    return PT_OK;
  case PT_OPT_HONOUR_MAX_DEPTH: r->honourMaxDepth = value != 0; return PT_OK;
  case 100: r->tune = value; return PT_OK; // tuning switches, see PtRenderer::tune
  case PT_OPT_OBJECT_SOURCE:
    if (value < 0 || value > 2) return fail(PT_ERR_INVALID, "pt_set_option: object source must be 0, 1 or 2");
    r->objectSource = value;
    return PT_OK;
  default: return fail(PT_ERR_INVALID, "pt_set_option: unknown option");
  }
}

int pt_device_info(PtRenderer *r, char *name, size_t name_bytes, int *compute_units) {
  if (!r) return fail(PT_ERR_INVALID, "pt_device_info: renderer is null");
  if (name && name_bytes) std::snprintf(name, name_bytes, "%s", r->deviceName);
  if (compute_units) *compute_units = r->computeUnits;
  return PT_OK;
}

int pt_last_launch(PtRenderer *r, int *blocks, int *threads, int *lds_bytes) {
  if (!r) return fail(PT_ERR_INVALID, "pt_last_launch: renderer is null");
  if (blocks) *blocks = r->lastBlocks;
  if (threads) *threads = r->lastThreads;
  if (lds_bytes) *lds_bytes = r->lastLds;
  return PT_OK;
}

} // extern "C"
