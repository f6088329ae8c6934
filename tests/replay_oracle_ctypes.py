"""Replays, without a GPU, what the test process does with the oracle through ctypes in the GPU suite's first tests
(the orbit frames of the demo scene, with and without a volume one byte short, then the small synthetic volumes of the
extension tests), plus every call shape oracle_py offers (row ranges, no float frame, samples, trilinear, probes).
Run by tests/test_sanitizers.py under AddressSanitizer + UBSan: PT_ORACLE_LIB names the sanitizer build of the oracle
and the sanitizer runtime is preloaded."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import oracle_py  # noqa: E402
from path_tracer_amd import scenes  # noqa: E402

W, H, dt = 200, 150, 100.0
for k in range(3):
    sc = scenes.demo_scene(W, H, frame=k, with_volume=False, time_ms=np.float32(k * np.float32(dt)), max_depth=10)
    oracle_py.render(sc, nthreads=8, want_f32=False)
res = (50, 37, 44)
vox = scenes.synthetic_walnut(res).copy()
vox[-1] = 0
for k in range(2):
    sc = scenes.demo_scene(W, H, frame=k, with_volume=True, voxels=vox, volume_res=res,
                           time_ms=np.float32(k * np.float32(dt)), max_depth=10)
    _, _, cnt = oracle_py.render(sc, nthreads=8, want_f32=False)
    assert int(cnt[6]) > 0
# a voxel array shorter than the resolution says (the loader's short read): reads beyond it must give 0, not fault
sc = scenes.demo_scene(64, 48, frame=0, with_volume=True, voxels=vox[: len(vox) // 2], volume_res=res, max_depth=3)
oracle_py.render(sc, nthreads=3)
res2 = (40, 30, 36)
v2 = (scenes.synthetic_walnut(res2) // 8).astype(np.uint8)
sc = scenes.config4(96, 54, voxels=v2, resolution=res2)
for filt in (0, 1):
    oracle_py.render(sc, nthreads=5, volume_filter=filt)
    oracle_py.render(sc, rows=(10, 20), nthreads=16, want_f32=False, volume_filter=filt)
oracle_py.render(sc, nthreads=2, first_sample=0, num_samples=3)
oracle_py.render(sc, rows=(-5, 10 ** 6), nthreads=1)
oracle_py.render(sc, rows=(30, 20), nthreads=4)  # empty range
sc = scenes.config2(64, 36, n_spheres=9, n_ellipsoids=4)
oracle_py.render(sc, nthreads=64)  # more threads than rows
oracle_py.render(sc, max_bounces=10, nthreads=7, num_samples=2, first_sample=5)
oracle_py.hit_sphere((0, 0, 5), (0, 0, -1), (0, 0, 0), 1.0)
oracle_py.hit_ellipsoid((0, 0, 5), (0, 0, -1), (0, 0, 0), (1, 2, 3), (0, 0, 0, 1))
oracle_py.rotate((1, 2, 3), (0.1, 0.2, 0.3, 0.9), inverse=True)
oracle_py.camera_ray((0, 1.5, 6), 64, 48, 3, 4)
oracle_py.march_volume(scenes.config4(32, 18, voxels=v2, resolution=res2), (5, 1.5, 6), (-0.5, -0.05, -0.86))
oracle_py.surface_shader(sc, 1, (0, 0, 0), (0, 1, 0), (0, 0, 1), 0)
oracle_py.surface_shader(sc, 2, (0, 0, 0), (0, 1, 0), (0, 0, 1), 0)
oracle_py.sample_offset(7, 3)
print("replay ok")
