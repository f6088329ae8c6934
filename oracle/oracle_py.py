"""ctypes binding of the CPU oracle (oracle/libpt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()``
and by the ``cpu_baseline`` leg of bench.py — never by the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PT_ORACLE_LIB: another build of the same source (the sanitizer build of tests/test_sanitizers.py)
_LIB_PATH = os.environ.get("PT_ORACLE_LIB") or os.path.join(_HERE, "libpt_oracle.so")


class _PtoScene(C.Structure):
    _fields_ = [
        ("spheres", C.c_void_p),
        ("ellipsoids", C.c_void_p),
        ("materials", C.c_void_p),
        ("lights", C.c_void_p),
        ("volumes", C.c_void_p),
        ("voxels", C.c_void_p),
        ("voxel_bytes", C.c_uint64),
    ]


def build() -> str:
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.pto_render.restype = C.c_int
        L.pto_render.argtypes = [C.POINTER(_PtoScene), C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.pto_render_ex.restype = C.c_int
        L.pto_render_ex.argtypes = L.pto_render.argtypes + [C.POINTER(_PtoOptions)]
        L.pto_sample_offset.restype = None
        L.pto_sample_offset.argtypes = [C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        fp = C.POINTER(C.c_float)
        L.pto_hit_sphere.restype = C.c_int
        L.pto_hit_sphere.argtypes = [fp, fp, fp, C.c_float, C.c_float, C.c_float, fp, fp, fp]
        L.pto_hit_ellipsoid.restype = C.c_int
        L.pto_hit_ellipsoid.argtypes = [fp, fp, fp, fp, fp, C.c_float, C.c_float, fp, fp, fp]
        L.pto_rotate.restype = None
        L.pto_rotate.argtypes = [fp, fp, C.c_int, fp]
        L.pto_camera_ray.restype = None
        L.pto_camera_ray.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp]
        L.pto_march_volume.restype = None
        L.pto_march_volume.argtypes = [C.POINTER(_PtoScene), C.c_void_p, fp, fp, C.c_float, fp,
                                       C.POINTER(C.c_uint64)]
        L.pto_intersect_box.restype = C.c_int
        L.pto_intersect_box.argtypes = [fp, fp, fp, fp, fp, fp]
        L.pto_sample_volume.restype = C.c_float
        L.pto_sample_volume.argtypes = [C.POINTER(_PtoScene), C.c_void_p, fp, C.c_int]
        L.pto_counter_count.restype = C.c_int
        L.pto_counter_count.argtypes = []
        L.pto_surface_shader.restype = None
        L.pto_surface_shader.argtypes = [C.POINTER(_PtoScene), C.c_void_p, C.c_int, fp, fp, fp, C.c_int, fp]
        _lib = L
    return _lib


class _PtoOptions(C.Structure):
    _fields_ = [("volume_filter", C.c_int), ("first_sample", C.c_int), ("num_samples", C.c_int)]


def _f3(v):
    return (C.c_float * len(v))(*[float(x) for x in v])


def _ptr(a: Optional[np.ndarray]):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data


class _Held:
    """Keeps the arrays a _PtoScene points into alive."""

    def __init__(self, scene):
        self.arrays = [np.ascontiguousarray(scene.spheres), np.ascontiguousarray(scene.ellipsoids),
                       np.ascontiguousarray(scene.materials), np.ascontiguousarray(scene.lights),
                       np.ascontiguousarray(scene.volumes)]
        self.vox = None if scene.voxels is None else np.ascontiguousarray(scene.voxels, dtype=np.uint8)
        self.c = _PtoScene(_ptr(self.arrays[0]), _ptr(self.arrays[1]), _ptr(self.arrays[2]),
                           _ptr(self.arrays[3]), _ptr(self.arrays[4]), _ptr(self.vox),
                           0 if self.vox is None else self.vox.size)


def sample_offset(pixel: int, sample: int) -> Tuple[float, float]:
    jx, jy = C.c_float(), C.c_float()
    lib().pto_sample_offset(pixel, sample, C.byref(jx), C.byref(jy))
    return jx.value, jy.value


def render(scene, max_bounces: Optional[int] = None, rows: Optional[Tuple[int, int]] = None,
           nthreads: int = 0, want_f32: bool = True, volume_filter: int = 0, first_sample: int = 0,
           num_samples: int = 1):
    """Returns (rgba8[H,W,4] uint8, f32[H,W,4] float32 or None, counters uint64).

    volume_filter = 1 (trilinear) and num_samples > 1 (mean of jittered samples) are extensions
    without a counterpart in the reference; the defaults restate the reference."""
    L = lib()
    held = _Held(scene)
    pc = scene.push_constants()
    W, H = scene.width, scene.height
    mb = int(pc["maxDepth"]) if max_bounces is None else max_bounces
    r0, r1 = (0, H) if rows is None else rows
    rgba = np.zeros((H, W, 4), dtype=np.uint8)
    f32 = np.zeros((H, W, 4), dtype=np.float32) if want_f32 else None
    cnt = np.zeros(int(lib().pto_counter_count()), dtype=np.uint64)  # as many as the C side writes
    if nthreads <= 0:
        nthreads = min(os.cpu_count() or 1, 16)
    opt = _PtoOptions(volume_filter, first_sample, num_samples)
    rc = L.pto_render_ex(C.byref(held.c), pc.ctypes.data, W, H, mb, r0, r1,
                         None if f32 is None else f32.ctypes.data, rgba.ctypes.data, cnt.ctypes.data, nthreads,
                         C.byref(opt))
    assert rc == 0
    return rgba, f32, cnt


def hit_sphere(origin, direction, center, radius, tmin=0.001, tmax=1e10):
    t = C.c_float()
    p = (C.c_float * 3)()
    n = (C.c_float * 3)()
    hit = lib().pto_hit_sphere(_f3(origin), _f3(direction), _f3(center), radius, tmin, tmax,
                               C.byref(t), p, n)
    return bool(hit), t.value, np.array(p[:], np.float32), np.array(n[:], np.float32)


def hit_ellipsoid(origin, direction, center, radii, rotation, tmin=0.001, tmax=1e10):
    t = C.c_float()
    p = (C.c_float * 3)()
    n = (C.c_float * 3)()
    hit = lib().pto_hit_ellipsoid(_f3(origin), _f3(direction), _f3(center), _f3(radii), _f3(rotation),
                                  tmin, tmax, C.byref(t), p, n)
    return bool(hit), t.value, np.array(p[:], np.float32), np.array(n[:], np.float32)


def rotate(v, q_xyzw, inverse=False):
    out = (C.c_float * 3)()
    lib().pto_rotate(_f3(v), _f3(q_xyzw), 1 if inverse else 0, out)
    return np.array(out[:], np.float32)


def camera_ray(camera_pos, width, height, px, py):
    o = (C.c_float * 3)()
    d = (C.c_float * 3)()
    lib().pto_camera_ray(_f3(camera_pos), width, height, px, py, o, d)
    return np.array(o[:], np.float32), np.array(d[:], np.float32)


def march_volume(scene, origin, direction, max_dist=1e10):
    held = _Held(scene)
    pc = scene.push_constants()
    out = (C.c_float * 3)()
    steps = C.c_uint64()
    lib().pto_march_volume(C.byref(held.c), pc.ctypes.data, _f3(origin), _f3(direction), max_dist, out,
                           C.byref(steps))
    return np.array(out[:], np.float32), steps.value


def intersect_box(origin, direction, box_min, box_max):
    """intersectBox (raytrace.comp:519-532) -> (hit, tNear, tFar)."""
    tn, tf = C.c_float(), C.c_float()
    hit = lib().pto_intersect_box(_f3(origin), _f3(direction), _f3(box_min), _f3(box_max), C.byref(tn), C.byref(tf))
    return bool(hit), tn.value, tf.value


def sample_volume(scene, point, volume_filter: int = 0) -> float:
    """sampleVolume (raytrace.comp:485-517) of volume 0 at a world position."""
    held = _Held(scene)
    pc = scene.push_constants()
    return float(lib().pto_sample_volume(C.byref(held.c), pc.ctypes.data, _f3(point), volume_filter))


def surface_shader(scene, shader_index, hit_point, normal, view_dir, material_index):
    held = _Held(scene)
    pc = scene.push_constants()
    out = (C.c_float * 3)()
    lib().pto_surface_shader(C.byref(held.c), pc.ctypes.data, shader_index, _f3(hit_point), _f3(normal),
                             _f3(view_dir), material_index, out)
    return np.array(out[:], np.float32)
