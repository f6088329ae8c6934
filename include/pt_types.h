/*
 * pt_types.h — plain-C byte layouts of the scene records that cross the
 * renderer boundary.
 *
 * These are the same records the reference hands to
 * VulkanRenderer::updateScene / VulkanRenderer::render, with glm::vec3/vec4
 * spelled out as floats so the header needs no C++ and no GLM:
 *
 *   PtGPUSphere          reference include/VulkanRenderer.h:16-21   (32 B)
 *   PtGPUEllipsoid       reference include/Ellipsoid.h:11-19        (64 B)
 *   PtGPUMaterial        reference include/VulkanRenderer.h:25-33   (80 B)
 *   PtGPULight           reference include/VulkanRenderer.h:35-40   (32 B)
 *   PtGPUVolumetricData  reference include/VulkanRenderer.h:42-52   (64 B)
 *   PtPushConstants      reference include/VulkanRenderer.h:55-71   (144 B)
 *
 * Both the HIP renderer (pt_hip.h) and the CPU oracle (oracle/pt_oracle.h)
 * take exactly these bytes, so a parity test feeds one buffer to both.
 */
#ifndef PT_TYPES_H
#define PT_TYPES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtGPUSphere {
  float center[3];
  float radius;
  float color[3];      /* uploaded by the reference, never read by shading */
  int32_t materialIndex;
} PtGPUSphere;

typedef struct PtGPUEllipsoid {
  float center[3];
  float padding1;
  float radii[3];
  int32_t materialIndex;
  float color[3];      /* never read by shading */
  float padding2;
  float rotation[4];   /* unit quaternion stored (x, y, z, w) */
} PtGPUEllipsoid;

typedef struct PtGPUMaterial {
  float colorAndAmbient[4];      /* color.xyz, ambient.w */
  float diffuseSpecularShiny[4]; /* diffuse.x, specular.y, shininess.z, reflectivity.w */
  float transparencyEmissive[4]; /* transparency.x, emissiveStrength.y, shaderFunctionIndex.z (as float), pad.w */
  float emissive[4];             /* never read by shading */
  float scatterAndAbsorption[4]; /* scatterColor.xyz, absorptionCoeff.w */
} PtGPUMaterial;

typedef struct PtGPULight {
  float position[3];
  float intensity;
  float color[3];      /* dropped by the reference before upload */
  float padding;
} PtGPULight;

typedef struct PtGPUVolumetricData {
  float position[3];
  float scale;         /* never read by shading */
  float v0[3];
  int32_t resolution_x;
  float v1[3];
  int32_t resolution_y;
  int32_t resolution_z;
  int32_t materialIndex;
  int32_t padding[2];
} PtGPUVolumetricData;

typedef struct PtPushConstants {
  float cameraMatrix[16]; /* offset 0   — ignored by the ray loop          */
  float cameraPos[3];     /* offset 64                                      */
  float time;             /* offset 76  — milliseconds                      */
  int32_t numSpheres;     /* offset 80                                      */
  int32_t numEllipsoids;  /* offset 84                                      */
  int32_t numLights;      /* offset 88                                      */
  int32_t numVolumes;     /* offset 92                                      */
  int32_t maxDepth;       /* offset 96  — see pt_hip.h: pt_render           */
  int32_t padding1;
  int32_t padding2;
  int32_t padding3;
  float bgColorTop[3];    /* offset 112                                     */
  float padding4;
  float bgColorBottom[3]; /* offset 128                                     */
  float padding5;
} PtPushConstants;

#if defined(__cplusplus)
static_assert(sizeof(PtGPUSphere) == 32, "GPUSphere is 32 bytes");
static_assert(sizeof(PtGPUEllipsoid) == 64, "GPUEllipsoid is 64 bytes");
static_assert(sizeof(PtGPUMaterial) == 80, "GPUMaterial is 80 bytes");
static_assert(sizeof(PtGPULight) == 32, "GPULight is 32 bytes");
static_assert(sizeof(PtGPUVolumetricData) == 64, "GPUVolumetricData is 64 bytes");
static_assert(sizeof(PtPushConstants) == 144, "PushConstants is 144 bytes");
static_assert(offsetof(PtPushConstants, cameraPos) == 64, "cameraPos at 64");
static_assert(offsetof(PtPushConstants, bgColorTop) == 112, "bgColorTop at 112");
static_assert(offsetof(PtPushConstants, bgColorBottom) == 128, "bgColorBottom at 128");
#else
_Static_assert(sizeof(PtGPUSphere) == 32, "GPUSphere is 32 bytes");
_Static_assert(sizeof(PtGPUEllipsoid) == 64, "GPUEllipsoid is 64 bytes");
_Static_assert(sizeof(PtGPUMaterial) == 80, "GPUMaterial is 80 bytes");
_Static_assert(sizeof(PtGPULight) == 32, "GPULight is 32 bytes");
_Static_assert(sizeof(PtGPUVolumetricData) == 64, "GPUVolumetricData is 64 bytes");
_Static_assert(sizeof(PtPushConstants) == 144, "PushConstants is 144 bytes");
#endif

/* Per-frame work counters, filled by the oracle and by the instrumented
 * (untimed) device kernel.  A "ray" is one closest-hit / any-hit query
 * against the whole object list (reference shaders/raytrace.comp:212-241). */
enum {
  PT_CNT_RAYS = 0,        /* all traceRay queries                         */
  PT_CNT_PATH_RAYS,       /* primary + bounce rays                        */
  PT_CNT_LIGHT_RAYS,      /* getShadowFactor queries                      */
  PT_CNT_EMISSIVE_RAYS,   /* emissive-object occlusion queries            */
  PT_CNT_SPHERE_TESTS,    /* ray-sphere tests, brute force (rays * Ns)    */
  PT_CNT_ELLIPSOID_TESTS, /* ray-ellipsoid tests, brute force (rays * Ne) */
  PT_CNT_VOXEL_SAMPLES,   /* volume march steps                           */
  PT_CNT_SHADED_HITS,     /* calculateLighting calls                      */
  PT_CNT_SPHERE_ROOTS,    /* sphere tests with discriminant >= 0 (sqrt taken) */
  PT_CNT_ELLIPSOID_ROOTS, /* ellipsoid tests with discriminant >= 0        */
  PT_CNT_WAVE_TRIPS,      /* device only: 8-sphere loop trips over all waves */
  PT_CNT_WAVE_SLOW_TRIPS, /* device only: trips that left the miss fast path */
  PT_CNT_EXACT_RAYS,      /* device only: rays routed to the exact (order-preserving) sphere routine */
  PT_CNT_CYC_REFILL,      /* device only: wave-cycles (s_memtime) spent assigning pixels      */
  PT_CNT_CYC_SPHERES,     /* device only: wave-cycles in the sphere loop and candidate queue    */
  PT_CNT_CYC_ELLIPSOIDS,  /* device only: wave-cycles in the ellipsoid loop                   */
  PT_CNT_CYC_SHADE,       /* device only: wave-cycles consuming results / shading / marching  */
  PT_CNT_CYC_EXACT,       /* device only: wave-cycles in the exact sphere routine             */
  PT_CNT_CYC_POOLED,      /* device only: wave-cycles from joining the pooled end-of-frame mode to exit */
  PT_CNT_POOLED_PASSES,   /* device only: passes made in pooled mode, summed over workgroups   */
  PT_CNT_POOLED_RAYS,     /* device only: rays traced through the pooled list                  */
  PT_CNT_CYC_TICKET,      /* device only: wave-cycles waiting for tile tickets (part of CYC_REFILL) */
  PT_CNT_TICKETS,         /* device only: tile tickets drawn                                   */
  PT_CNT_CYC_TAIL,        /* device only: wave-cycles from a wave's finding the tickets drained to its exit */
  PT_CNT_CYC_ALIVE,       /* device only: wave-cycles from start to exit, summed over the waves */
  PT_CNT_COUNT
};

#ifdef __cplusplus
}
#endif
#endif /* PT_TYPES_H */
