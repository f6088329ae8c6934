/*
 * pt_hip.h — C-ABI of the MI355X ray-loop renderer (libpt_hip.so).
 *
 * This is the boundary the reference draws with `class VulkanRenderer`
 * (reference include/VulkanRenderer.h:73-97); the reference has no FFI of its
 * own, so each entry point names the C++ member it stands in for.  Records are
 * passed as the byte layouts of pt_types.h.  Every function returns PT_OK (0)
 * or an error code, never throws, and leaves a message for pt_last_error().
 *
 * One PtRenderer = one GPU + one HIP stream.  A renderer is not thread-safe;
 * use one per device (one process per GPU for multi-GPU work).
 */
#ifndef PT_HIP_H
#define PT_HIP_H

#include "pt_types.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct PtRenderer PtRenderer;

enum {
  PT_OK = 0,
  PT_ERR_INVALID = 1, /* bad argument (null, negative count, index out of range) */
  PT_ERR_HIP = 2,     /* a HIP runtime call failed                              */
  PT_ERR_NOMEM = 3,   /* device or host allocation failed                       */
  PT_ERR_STATE = 4    /* call order (e.g. render before update_scene)           */
};

/* Options for pt_set_option. */
enum {
  PT_OPT_FLOAT_OUTPUT = 1, /* 1 (default): also keep the clamped float RGBA frame  */
  PT_OPT_OBJECT_SOURCE = 2, /* 0 auto, 1 objects staged in LDS, 2 objects read from
                               global memory (scenes too large for LDS use 2)      */
  PT_OPT_VOLUME_BRICKS = 4, /* 1 (default): the march reads a copy of volume 0 re-tiled
                               into 8x8x8-voxel bricks (same voxels, better locality);
                               0: it reads the linear x-fastest array as uploaded      */
  PT_OPT_VOLUME_FILTER = 5, /* 0 (default): nearest-neighbour voxel sampling as in the
                               reference (shaders/raytrace.comp:485-517); 1: trilinear
                               (a quality mode of this library, no counterpart there)  */
  PT_OPT_FRAMES_IN_FLIGHT = 6, /* n = 1 .. 8, default 2 (the reference's MAX_FRAMES_IN_FLIGHT,
                               src/VulkanRenderer.cpp:1080-1100): consecutive pt_render calls go
                               round n slots, each with its own stream and output frames, so a
                               frame starts while the previous ones finish; pt_present waits for
                               all, readback and pt_device_* show the most recent frame.  1: one
                               frame at a time.  More than 2 pays where a frame is short against
                               its longest chain of dependent rays (a rank's share of a frame
                               split over many GPUs).  A caller's stream or output buffers given
                               for slot 0 only imply 1.  With n >= 2 the caller is taken to keep n
                               frames queued: a frame with few tiles per wave is launched on a part
                               of the compute units only (see pt_stream_concurrency), which costs
                               latency if the frames are in fact drawn one by one.                 */
  PT_OPT_HONOUR_MAX_DEPTH = 7, /* 0 (default): every frame runs 10 bounces whatever
                               PushConstants.maxDepth says, as the reference shader does
                               (it ignores the field, shaders/raytrace.comp:603-605);
                               1: maxDepth > 0 is the bounce limit (an extension)       */
  PT_OPT_PRESENT_LAG = 8,   /* 0 (default): pt_present waits for every frame queued so far, the frame just
                               drawn included, as the reference's present() does (it drains the queue,
                               src/VulkanRenderer.cpp:1355-1356) -- a loop of render; present; draws one frame
                               at a time.  1 (needs PT_OPT_FRAMES_IN_FLIGHT >= 2): pt_present called after a
                               pt_render waits for the frames BEFORE that one only, and readback / pt_device_*
                               then show the frame before it: the same loop keeps the last frame drawing while
                               the host prepares the next, one frame later on the screen (0.41 instead of 0.59 ms
                               per 1080p frame of the 64-object scene).  A pt_present that no pt_render
                               preceded since the last one waits for everything and shows the last frame.   */
  PT_OPT_SAMPLE_BATCH = 9,  /* pt_render_samples: samples drawn per launch.  0 (default): as many as fit 16 GiB of
                               sample planes (one float4 frame per sample of the launch: 64 samples at 4K take
                               8.5 GB); 1: one launch per sample, no planes; n > 1: at most n.  The frames are the
                               same whatever the value (the planes are added up in sample order).              */
  PT_OPT_TILE_ORDER = 3     /* 1: a short pass before each frame sorts the 8x8 tiles so
                               that those whose pixels cast the longest chains of rays
                               are drawn first; 0: index order; 2 (default): sorted
                               when the scene has mirror or transparent objects and
                               the pass is cheap next to the frame.  The frame is the
                               same either way.                                      */
};

/* VulkanRenderer::initialize (reference include/VulkanRenderer.h:78,
 * src/VulkanRenderer.cpp:27-72).  No window: the frame stays in device memory
 * until pt_readback_*.  `device` is the HIP device ordinal. */
int pt_create(int width, int height, int device, PtRenderer **out);

/* VulkanRenderer::shutdown / destructor (reference include/VulkanRenderer.h:76,79).
 * NULL is allowed. */
void pt_destroy(PtRenderer *r);

/* VulkanRenderer::updateScene (reference include/VulkanRenderer.h:81-87,
 * src/VulkanRenderer.cpp:794-987).  Copies everything it needs before it
 * returns; device buffers grow on demand (the reference's are fixed at 100
 * spheres / ellipsoids / materials, 32 lights / volumes, 50 MB of voxels).
 * Voxels (x fastest, then y, then z) are uploaded on the first call that
 * passes any and ignored afterwards, as in the reference
 * (src/VulkanRenderer.cpp:961-987); pt_reset_voxels() re-arms the upload.
 * Material indices of spheres, ellipsoids and volumes must lie in
 * [0, num_materials) — the reference reads out of bounds instead. */
int pt_update_scene(PtRenderer *r, const PtGPUSphere *spheres, int num_spheres,
                    const PtGPUEllipsoid *ellipsoids, int num_ellipsoids,
                    const PtGPUMaterial *materials, int num_materials,
                    const PtGPULight *lights, int num_lights,
                    const PtGPUVolumetricData *volumes, int num_volumes,
                    const uint8_t *voxels, size_t voxel_bytes);

/* Forget the uploaded voxels so the next pt_update_scene uploads again.
 * No counterpart in the reference (it can never replace its volume). */
int pt_reset_voxels(PtRenderer *r);

/* VulkanRenderer::render (reference include/VulkanRenderer.h:89,
 * src/VulkanRenderer.cpp:1080-1145): enqueue one frame on the renderer's
 * stream and return.  The object/light/volume counts are taken from `pc` like
 * the shader does and must not exceed what pt_update_scene received.
 * `pc->maxDepth` is ignored and the frame runs 10 bounces, exactly as the
 * reference shader does (shaders/raytrace.comp:603-605; the reference demo
 * passes 5 and gets 10).  With PT_OPT_HONOUR_MAX_DEPTH = 1 a positive
 * maxDepth becomes the bounce limit. */
int pt_render(PtRenderer *r, const PtPushConstants *pc);

/* VulkanRenderer::present (reference include/VulkanRenderer.h:90): there is no
 * swapchain; this waits until the frames enqueued so far are complete (with
 * PT_OPT_PRESENT_LAG = 1: all but the one just enqueued). */
int pt_present(PtRenderer *r);

/* The reference reserves a "readback buffer for PPM export" it never creates
 * (reference include/VulkanRenderer.h:177-179); these are that readback.
 * They wait for the stream, then copy the frame (or, with row bands set, this
 * renderer's compact band-major rows) to host memory.
 *   rgba8: width*rows*4 bytes, R,G,B,255 (the shader's rgba8 image, row 0 = top)
 *   f32  : width*rows*4 floats, the clamped colour the shader hands to imageStore */
int pt_readback_rgba8(PtRenderer *r, uint8_t *dst, size_t dst_bytes);
int pt_readback_f32(PtRenderer *r, float *dst, size_t dst_bytes);

/* VulkanRenderer::getShaderPathToIndexMap (reference include/VulkanRenderer.h:95-97;
 * filled by ShaderCompiler, reference src/ShaderCompiler.cpp:131-185).  The
 * surface shaders are compiled-in device functions; the table maps the path
 * strings the reference would have scanned to the same indices:
 *   "shaders/surface_shaders/rainbow.glsl" -> 1, ".../toon.glsl" -> 2.
 * pt_shader_index returns 0 (built-in Phong) for an unknown path, which is
 * what Material::toGPU does (reference src/SceneWrappers.cpp:13-23). */
int pt_shader_count(void);
const char *pt_shader_path(int i); /* i in [0, pt_shader_count()) ; index = i+1 */
int pt_shader_index(const char *path);

/* Message of the calling thread's most recent failure ("" if none). */
const char *pt_last_error(void);

/* ---- Entry points without a counterpart in the reference ---- */

/* Multi-GPU image split: this renderer draws only bands band_index,
 * band_index + band_count, ... of `band_rows` rows each, into a compact
 * band-major frame (its bands stacked in order).  band_count = 1 restores the
 * whole frame.  pt_local_rows() tells how many rows the compact frame has. */
int pt_set_row_bands(PtRenderer *r, int band_rows, int band_index, int band_count);
int pt_local_rows(PtRenderer *r);

/* Rank-0 side of the gather: `src` holds the compact rgba8 (elem_bytes = 4) or
 * float (elem_bytes = 16) frames of ranks 0..band_count-1 back to back, each
 * padded to `stride_rows` rows; writes the assembled height x width frame to
 * `dst`.  All pointers are device pointers; runs on the renderer's stream. */
int pt_assemble_bands(PtRenderer *r, const void *src, void *dst, int elem_bytes,
                      int band_rows, int band_count, int stride_rows);

/* The gather itself over RCCL, without any framework in between (one process per
 * GPU, scene replicated, interleaved row bands, one exchange at the end of the frame).  RCCL is opened at run
 * time (librccl.so.1); programs that never call these do not need it.
 *   pt_comm_unique_id   rank 0 makes the PT_COMM_ID_BYTES that identify the communicator and hands them to the
 *                       other ranks by whatever means the program has (a file, MPI, torch.distributed, a socket);
 *   pt_comm_create      every rank, collectively (ncclCommInitRank on `device`);
 *   pt_gather_bands     after pt_render, on every rank, in the same order on all ranks: ranks 1 .. world-1 send
 *                       their compact band-major frame (elem_bytes 4: the rgba8 frame, 16: the float frame) to
 *                       rank 0, each over its own xGMI link; rank 0 posts the matching receives as one group
 *                       (ncclGroupStart / ncclRecv x (world-1) / ncclGroupEnd), copies its own bands beside them
 *                       and de-interleaves everything into `dst_dev` (height x width elements; NULL: a frame
 *                       the renderer owns, see pt_device_assembled / pt_readback_assembled).  Everything is
 *                       enqueued on the stream of the frame just drawn, so frames in flight overlap their
 *                       gathers; pt_present waits for all of it.  pt_set_row_bands(rows, rank, world) must have
 *                       been called with the communicator's rank and world.  world = 1 is allowed (a copy). */
#define PT_COMM_ID_BYTES 128
typedef struct PtComm PtComm;
int pt_comm_unique_id(void *id_out, size_t id_bytes);
int pt_comm_create(const void *id, size_t id_bytes, int rank, int world, int device, PtComm **out);
void pt_comm_destroy(PtComm *c);
int pt_gather_bands(PtRenderer *r, PtComm *c, int elem_bytes, void *dst_dev);
void *pt_device_assembled(PtRenderer *r);
int pt_readback_assembled(PtRenderer *r, void *dst, size_t dst_bytes);

/* Run the instrumented kernel for one frame (same image, untimed) and return
 * the PT_CNT_* work counters. */
int pt_render_counted(PtRenderer *r, const PtPushConstants *pc, uint64_t *counters);

/* How many of the PT_OPT_FRAMES_IN_FLIGHT slots' streams the device really runs side by side (streams share the
 * process's hardware queues: GPU_MAX_HW_QUEUES, 4 by default, less what other libraries in the process took).
 * The library measures it once per change of the slots (a device synchronisation and a few milliseconds, at the
 * first frame after the change or at this call) and sizes the launches of frames in flight by it: a frame with few
 * tiles per wave takes 1/n of the compute units so that the n frames run side by side with full waves instead of
 * one after the other with emptying ones.  No counterpart in the reference (one queue, MAX_FRAMES_IN_FLIGHT = 2,
 * reference include/VulkanRenderer.h:126). */
int pt_stream_concurrency(PtRenderer *r, int *streams);

/* Profiling aid: draws one frame with the kernel that stamps the clock three times per wave and
 * returns, for wave i of the launch, out[4i .. 4i+3] = the 100 MHz clock (s_memrealtime) at its start, when it found the tile tickets
 * drained, at its exit, and in the fourth word the number of passes it made (low 32 bits) and how many of them it had
 * made when it found the tickets drained (high 32 bits).  *num_waves = waves of the launch. */
int pt_wave_timeline(PtRenderer *r, const PtPushConstants *pc, uint64_t *out, size_t capacity_waves, int *num_waves);

/* pt_render without any allocation.  No counterpart in the reference; it is for
 * callers that cannot call hipMalloc where they render: code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation.  Draws the
 * frame of `pc` into the caller's device buffers (rgba8_dev: width*rows*4 bytes,
 * required; f32_dev: width*rows*16 bytes or NULL) on the caller's stream (NULL:
 * the renderer's own), with the uploaded scene of `r`.  The frame's scratch
 * (tile-ticket counters, cost-class counts, tile drawing-order lists) lives in a
 * static device array of 3 MiB + 8 KiB that every GPU the library runs on holds
 * until the process ends; a frame of more than 2^18 8x8-pixel tiles (16.7 M pixels)
 * is rejected with PT_ERR_INVALID.  The call allocates nothing, waits for nothing
 * and does not touch the renderer's frame slots (pt_readback_* and pt_device_*
 * keep showing the last pt_render frame).  All renderers on one GPU share that
 * GPU's array: calls that overlap on one GPU (two streams, two renderers, two
 * threads) race; order them on one stream or with events.  The output is the
 * same as pt_render's. */
int pt_render_static(PtRenderer *r, const PtPushConstants *pc, void *rgba8_dev, void *f32_dev, void *hip_stream);

/* Progressive accumulation (no counterpart in the reference, which draws one
 * sample per pixel at the pixel corner and never calls its random(), reference
 * shaders/raytrace.comp:117-119, 692-700).  Draws sub-pixel samples
 * first_sample .. first_sample + num_samples - 1 and leaves the mean of the
 * clamped samples in the output frames.  A call draws several samples per
 * launch (PT_OPT_SAMPLE_BATCH): the launch hands out (tile, sample) pairs, so the chip idles at the end of the call instead of at the end of
 * every sample; each sample's clamped colours go to a plane of their own and
 * are added up in sample order, which gives the sum launches one by one give.  Sample 0 is the
 * reference's sample; sample s > 0 is offset inside the pixel by a
 * counter-based hash of (pixel, s).  keep != 0 adds to the sum of the previous
 * call instead of starting a new one.  pt_render is (0, 1, 0). */
int pt_render_samples(PtRenderer *r, const PtPushConstants *pc, int first_sample, int num_samples, int keep);

/* Device pointers of the current output frames (f32 may be NULL). */
void *pt_device_rgba8(PtRenderer *r);
void *pt_device_f32(PtRenderer *r);
/* Render into caller-owned device memory instead (NULL restores the internal
 * buffers); sizes are width*pt_local_rows()*4 bytes / *16 bytes. */
int pt_set_output(PtRenderer *r, void *rgba8_dev, void *f32_dev);

/* Use a caller-owned hipStream_t (passed as void*); NULL restores the
 * renderer's own stream. */
int pt_set_stream(PtRenderer *r, void *hip_stream);

/* Both at once, per frame slot (0 .. 7): stream (NULL: the renderer's own) and
 * output buffers (NULL: internal) of the frames drawn in that slot.  With
 * PT_OPT_FRAMES_IN_FLIGHT = n >= 2, consecutive pt_render calls go round slots
 * 0 .. n-1 when these are set up alike (all internal, or all with the caller's
 * stream and buffers); pt_set_stream / pt_set_output configure slot 0 only and so
 * keep every frame in slot 0.  pt_last_slot tells which slot the most recent frame
 * went to; pt_assemble_bands runs on that slot's stream. */
int pt_set_slot(PtRenderer *r, int slot, void *hip_stream, void *rgba8_dev, void *f32_dev);
int pt_last_slot(PtRenderer *r);

/* hipEvent timing on the renderer's stream: begin/end bracket any number of
 * pt_render calls; elapsed waits for the end event. */
int pt_timer_begin(PtRenderer *r);
int pt_timer_end(PtRenderer *r);
int pt_timer_elapsed_ms(PtRenderer *r, float *ms);

int pt_set_option(PtRenderer *r, int option, int value);

/* Name of the device the renderer runs on, number of compute units, and the
 * launch geometry of the last frame (for logs). */
int pt_device_info(PtRenderer *r, char *name, size_t name_bytes, int *compute_units);
int pt_last_launch(PtRenderer *r, int *blocks, int *threads, int *lds_bytes);

#ifdef __cplusplus
}
#endif
#endif /* PT_HIP_H */
