"""Byte layouts of the scene records that cross the renderer boundary.

numpy mirrors of ``include/pt_types.h`` (which restates the reference's
``GPUSphere`` / ``GPUEllipsoid`` / ``GPUMaterial`` / ``GPULight`` /
``GPUVolumetricData`` / ``PushConstants``: reference include/VulkanRenderer.h:16-71,
include/Ellipsoid.h:11-19).  A structured array of one of these dtypes can be
handed to the C-ABI or to the oracle without conversion.
"""
from __future__ import annotations

import numpy as np

GPU_SPHERE = np.dtype(
    [("center", "<f4", 3), ("radius", "<f4"), ("color", "<f4", 3), ("materialIndex", "<i4")]
)
GPU_ELLIPSOID = np.dtype(
    [
        ("center", "<f4", 3),
        ("padding1", "<f4"),
        ("radii", "<f4", 3),
        ("materialIndex", "<i4"),
        ("color", "<f4", 3),
        ("padding2", "<f4"),
        ("rotation", "<f4", 4),  # (x, y, z, w)
    ]
)
GPU_MATERIAL = np.dtype(
    [
        ("colorAndAmbient", "<f4", 4),
        ("diffuseSpecularShiny", "<f4", 4),
        ("transparencyEmissive", "<f4", 4),
        ("emissive", "<f4", 4),
        ("scatterAndAbsorption", "<f4", 4),
    ]
)
GPU_LIGHT = np.dtype(
    [("position", "<f4", 3), ("intensity", "<f4"), ("color", "<f4", 3), ("padding", "<f4")]
)
GPU_VOLUME = np.dtype(
    [
        ("position", "<f4", 3),
        ("scale", "<f4"),
        ("v0", "<f4", 3),
        ("resolution_x", "<i4"),
        ("v1", "<f4", 3),
        ("resolution_y", "<i4"),
        ("resolution_z", "<i4"),
        ("materialIndex", "<i4"),
        ("padding", "<i4", 2),
    ]
)
PUSH_CONSTANTS = np.dtype(
    [
        ("cameraMatrix", "<f4", 16),
        ("cameraPos", "<f4", 3),
        ("time", "<f4"),
        ("numSpheres", "<i4"),
        ("numEllipsoids", "<i4"),
        ("numLights", "<i4"),
        ("numVolumes", "<i4"),
        ("maxDepth", "<i4"),
        ("padding1", "<i4"),
        ("padding2", "<i4"),
        ("padding3", "<i4"),
        ("bgColorTop", "<f4", 3),
        ("padding4", "<f4"),
        ("bgColorBottom", "<f4", 3),
        ("padding5", "<f4"),
    ]
)

assert GPU_SPHERE.itemsize == 32
assert GPU_ELLIPSOID.itemsize == 64
assert GPU_MATERIAL.itemsize == 80
assert GPU_LIGHT.itemsize == 32
assert GPU_VOLUME.itemsize == 64
assert PUSH_CONSTANTS.itemsize == 144

# Work-counter slots (include/pt_types.h, PT_CNT_*).
COUNTER_NAMES = (
    "rays",
    "path_rays",
    "light_rays",
    "emissive_rays",
    "sphere_tests",
    "ellipsoid_tests",
    "voxel_samples",
    "shaded_hits",
    "sphere_roots",
    "ellipsoid_roots",
    "wave_trips",       # device only
    "wave_slow_trips",  # device only
    "exact_rays",       # device only
    "cyc_refill",       # device only, wave-cycles
    "cyc_spheres",
    "cyc_ellipsoids",
    "cyc_shade",
    "cyc_exact",
    "cyc_pooled",       # wave-cycles between joining the pooled end-of-frame mode and exit
    "pooled_passes",    # passes in pooled mode, summed over workgroups
    "pooled_rays",      # rays traced through the pooled list
    "cyc_ticket",       # wave-cycles waiting for tile tickets (part of cyc_refill)
    "tickets",          # tile tickets drawn
    "cyc_tail",         # wave-cycles from a wave's finding the tickets drained to its exit
    "cyc_alive",        # wave-cycles from start to exit, summed over the waves
)
N_ALGORITHM_COUNTERS = 10  # the slots the oracle fills too

# Compiled-in surface shaders, keyed by the path string the reference's
# ShaderCompiler would have seen (reference src/ShaderCompiler.cpp:131-185:
# sorted paths get 1, 2, ...; 0 is the built-in Phong).
SHADER_RAINBOW = "shaders/surface_shaders/rainbow.glsl"
SHADER_TOON = "shaders/surface_shaders/toon.glsl"


# pt_set_option keys (include/pt_hip.h)
OPT_FLOAT_OUTPUT = 1
OPT_OBJECT_SOURCE = 2
OPT_TILE_ORDER = 3
OPT_VOLUME_BRICKS = 4
OPT_VOLUME_FILTER = 5
OPT_FRAMES_IN_FLIGHT = 6
OPT_HONOUR_MAX_DEPTH = 7
OPT_PRESENT_LAG = 8
OPT_SAMPLE_BATCH = 9
