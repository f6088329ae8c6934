/*
 * pt_oracle.h — CPU restatement of the reference's per-pixel ray loop.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the HIP library,
 * the C++/Python host layers) may include, link or call this.  It exists so
 * tests, the smoke check and the bench's cpu_baseline leg have an
 * independent answer to compare the device kernels with.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, golden
 * images or known-answer vectors for this path, and its GLSL cannot execute
 * here (no glslc / Vulkan).  The restatement is pinned instead by
 *   - closed-form known answers (tests/test_oracle_known_answers.py),
 *   - vectors generated from the reference's own double-precision legacy
 *     classes that do compile here (oracle/ref_kat.cpp -> tests/golden/).
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include "../include/pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtoScene {
  const PtGPUSphere *spheres;
  const PtGPUEllipsoid *ellipsoids;
  const PtGPUMaterial *materials;
  const PtGPULight *lights;
  const PtGPUVolumetricData *volumes;
  const uint8_t *voxels; /* x fastest, then y, then z; may be NULL */
  uint64_t voxel_bytes;
} PtoScene;

/* Render rows [row_begin, row_end) of a width x height frame.
 *   max_bounces : bounce limit of traceIterative (the reference literal is 10)
 *   out_f32     : width*height*4 floats (clamped r, g, b, 1), may be NULL
 *   out_rgba8   : width*height*4 bytes, may be NULL
 *   counters    : PT_CNT_COUNT uint64 work counters, may be NULL
 *   nthreads    : rows are dealt round-robin to this many threads (>= 1)
 * Rows outside the range are left untouched.  Returns 0. */
int pto_render(const PtoScene *scene, const PtPushConstants *pc, int width,
               int height, int max_bounces, int row_begin, int row_end,
               float *out_f32, uint8_t *out_rgba8, uint64_t *counters,
               int nthreads);

/* Number of uint64 slots pto_render writes into `counters` (PT_CNT_COUNT of the build). */
int pto_counter_count(void);

/* Extensions without a counterpart in the reference (all zero = the reference). */
typedef struct PtoOptions {
  int volume_filter; /* 0 nearest neighbour (raytrace.comp:485-517), 1 trilinear */
  int first_sample;  /* index of the first sub-pixel sample; sample 0 is the reference's */
  int num_samples;   /* > 1: the frame is the mean of the clamped samples first..first+n-1 */
} PtoOptions;

int pto_render_ex(const PtoScene *scene, const PtPushConstants *pc, int width,
                  int height, int max_bounces, int row_begin, int row_end,
                  float *out_f32, uint8_t *out_rgba8, uint64_t *counters,
                  int nthreads, const PtoOptions *opt);

/* Sub-pixel offset of sample s of a pixel (sample 0: none). */
void pto_sample_offset(uint32_t pixel, uint32_t s, float *jx, float *jy);

/* Single-function probes used by the known-answer tests. */

/* hitSphere: returns 1 on hit and fills t, point[3], normal[3]. */
int pto_hit_sphere(const float origin[3], const float dir[3],
                   const float center[3], float radius, float t_min,
                   float t_max, float *t, float *point, float *normal);

/* hitEllipsoid, rotation = (x, y, z, w). */
int pto_hit_ellipsoid(const float origin[3], const float dir[3],
                      const float center[3], const float radii[3],
                      const float rotation[4], float t_min, float t_max,
                      float *t, float *point, float *normal);

/* rotateByQuaternion / inverseRotateByQuaternion. */
void pto_rotate(const float v[3], const float q[4], int inverse, float out[3]);

/* createCameraRay for pixel (px, py) of a width x height image. */
void pto_camera_ray(const float camera_pos[3], int width, int height, int px,
                    int py, float origin[3], float dir[3]);

/* rayMarchVolume on volume 0: returns the vec3 the shader returns. */
void pto_march_volume(const PtoScene *scene, const PtPushConstants *pc,
                      const float origin[3], const float dir[3],
                      float max_dist, float out[3], uint64_t *steps);

/* intersectBox: returns 1 when the ray meets the box and fills *t_near, *t_far. */
int pto_intersect_box(const float origin[3], const float dir[3], const float box_min[3],
                      const float box_max[3], float *t_near, float *t_far);

/* sampleVolume on volume 0 at a world position (volume_filter 0: nearest, as the shader). */
float pto_sample_volume(const PtoScene *scene, const PtPushConstants *pc, const float point[3],
                        int volume_filter);

/* surfaceShader_Rainbow / surfaceShader_Toon probes: index 1 / 2.
 * Returns color+emissive (what calculateLighting returns for the shader). */
void pto_surface_shader(const PtoScene *scene, const PtPushConstants *pc,
                        int shader_index, const float hit_point[3],
                        const float normal[3], const float view_dir[3],
                        int material_index, float out[3]);

#ifdef __cplusplus
}
#endif
#endif
