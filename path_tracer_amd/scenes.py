"""Scene containers and the named benchmark / parity scenes.

The material helpers mirror the reference's named constructors
(reference include/SceneWrappers.h:34-97) and ``Material::toGPU``
(reference src/SceneWrappers.cpp:5-32); ``demo_scene`` restates the reference
demo (reference src/main_vulkan.cpp:136-223, 306-307, 340-354).

The synthetic configurations C1..C5 are deterministic: every random draw comes
from one PCG32 stream in the order the builder's code makes them, and every
stored value is float32.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import types as T

# ----------------------------------------------------------------------------
# PCG32 (XSH-RR 64/32), O'Neill's reference constants.
# ----------------------------------------------------------------------------
_M64 = (1 << 64) - 1


class PCG32:
    def __init__(self, seed: int, seq: int = 54):
        self.state = 0
        self.inc = ((seq << 1) | 1) & _M64
        self.next_u32()
        self.state = (self.state + (seed & _M64)) & _M64
        self.next_u32()

    def next_u32(self) -> int:
        old = self.state
        self.state = (old * 6364136223846793005 + self.inc) & _M64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, lo: float, hi: float) -> np.float32:
        """float32 in [lo, hi): lo + (hi-lo)*u with u = top 24 bits / 2^24, all float32."""
        u = np.float32(self.next_u32() >> 8) * np.float32(1.0 / 16777216.0)
        return np.float32(np.float32(lo) + np.float32(np.float32(hi) - np.float32(lo)) * u)

    def below(self, n: int) -> int:
        return self.next_u32() % n


# ----------------------------------------------------------------------------
# Materials
# ----------------------------------------------------------------------------
SHADER_INDEX: Dict[str, int] = {T.SHADER_RAINBOW: 1, T.SHADER_TOON: 2}


@dataclass
class Material:
    """Host-side material; field defaults follow reference include/SceneWrappers.h:19-31."""

    color: Sequence[float] = (1.0, 1.0, 1.0)
    ambient: float = 0.1
    diffuse: float = 0.5
    specular: float = 0.5
    shininess: float = 32.0
    reflectivity: float = 0.0
    transparency: float = 0.0
    emissive_strength: float = 0.0
    emissive: Sequence[float] = (0.0, 0.0, 0.0)
    scatter_color: Sequence[float] = (0.0, 0.0, 0.0)
    absorption_coeff: float = 0.0
    shader_path: str = ""

    @staticmethod
    def Diffuse(color, diffuse=0.7, ambient=0.1) -> "Material":
        return Material(color=color, ambient=ambient, diffuse=diffuse, specular=0.1, shininess=16.0)

    @staticmethod
    def Specular(color, specular, shininess, reflectivity=0.0) -> "Material":
        return Material(color=color, specular=specular, shininess=shininess, reflectivity=reflectivity)

    @staticmethod
    def Mirror(color=(0.9, 0.9, 0.9), reflectivity=0.7) -> "Material":
        return Material(color=color, diffuse=0.2, specular=0.8, shininess=128.0, reflectivity=reflectivity)

    @staticmethod
    def Volumetric(scatter_color, absorption_coeff) -> "Material":
        return Material(scatter_color=scatter_color, absorption_coeff=absorption_coeff)

    @staticmethod
    def Emissive(color, strength=1.0) -> "Material":
        return Material(color=color, ambient=0.0, diffuse=0.0, emissive_strength=strength)

    @staticmethod
    def CustomShader(color, shader_path) -> "Material":
        return Material(color=color, ambient=0.1, diffuse=0.7, specular=0.3, shininess=32.0,
                        shader_path=shader_path)

    def to_gpu(self, shader_index: Optional[Dict[str, int]] = None) -> np.ndarray:
        """reference src/SceneWrappers.cpp:5-32 (unknown shader path -> 0)."""
        table = SHADER_INDEX if shader_index is None else shader_index
        g = np.zeros((), dtype=T.GPU_MATERIAL)
        g["colorAndAmbient"] = (*self.color, self.ambient)
        g["diffuseSpecularShiny"] = (self.diffuse, self.specular, self.shininess, self.reflectivity)
        idx = table.get(self.shader_path, 0) if self.shader_path else 0
        g["transparencyEmissive"] = (self.transparency, self.emissive_strength, float(idx), 0.0)
        g["emissive"] = (*self.emissive, 0.0)
        g["scatterAndAbsorption"] = (*self.scatter_color, self.absorption_coeff)
        return g


# ----------------------------------------------------------------------------
# Scene container
# ----------------------------------------------------------------------------
@dataclass
class Scene:
    width: int
    height: int
    spheres: np.ndarray = field(default_factory=lambda: np.zeros(0, T.GPU_SPHERE))
    ellipsoids: np.ndarray = field(default_factory=lambda: np.zeros(0, T.GPU_ELLIPSOID))
    materials: np.ndarray = field(default_factory=lambda: np.zeros(0, T.GPU_MATERIAL))
    lights: np.ndarray = field(default_factory=lambda: np.zeros(0, T.GPU_LIGHT))
    volumes: np.ndarray = field(default_factory=lambda: np.zeros(0, T.GPU_VOLUME))
    voxels: Optional[np.ndarray] = None  # uint8, x fastest then y then z
    camera_pos: Sequence[float] = (0.0, 1.5, 6.0)
    time_ms: float = 0.0
    max_depth: int = 10
    bg_top: Sequence[float] = (0.4, 0.45, 1.0)
    bg_bottom: Sequence[float] = (1.0, 1.0, 1.0)
    name: str = ""

    def push_constants(self) -> np.ndarray:
        pc = np.zeros((), dtype=T.PUSH_CONSTANTS)
        pc["cameraPos"] = self.camera_pos
        pc["time"] = self.time_ms
        pc["numSpheres"] = len(self.spheres)
        pc["numEllipsoids"] = len(self.ellipsoids)
        pc["numLights"] = len(self.lights)
        pc["numVolumes"] = len(self.volumes)
        pc["maxDepth"] = self.max_depth
        pc["bgColorTop"] = self.bg_top
        pc["bgColorBottom"] = self.bg_bottom
        return pc


def _materials(mats: Sequence[Material]) -> np.ndarray:
    out = np.zeros(len(mats), dtype=T.GPU_MATERIAL)
    for i, m in enumerate(mats):
        out[i] = m.to_gpu()
    return out


def _sphere(center, radius, mat_index, color=(1.0, 1.0, 1.0)) -> np.ndarray:
    s = np.zeros((), dtype=T.GPU_SPHERE)
    s["center"] = center
    s["radius"] = radius
    s["color"] = color
    s["materialIndex"] = mat_index
    return s


def _ellipsoid(center, radii, mat_index, rotation=(0.0, 0.0, 0.0, 1.0), color=(1.0, 1.0, 1.0)) -> np.ndarray:
    e = np.zeros((), dtype=T.GPU_ELLIPSOID)
    e["center"] = center
    e["radii"] = radii
    e["materialIndex"] = mat_index
    e["color"] = color
    e["rotation"] = rotation
    return e


def _light(position, intensity, color=(1.0, 1.0, 1.0)) -> np.ndarray:
    l = np.zeros((), dtype=T.GPU_LIGHT)
    l["position"] = position
    l["intensity"] = intensity
    l["color"] = color
    return l


def _volume(position, v0, v1, resolution, mat_index, scale=0.001) -> np.ndarray:
    v = np.zeros((), dtype=T.GPU_VOLUME)
    v["position"] = position
    v["scale"] = scale
    v["v0"] = v0
    v["v1"] = v1
    v["resolution_x"], v["resolution_y"], v["resolution_z"] = resolution
    v["materialIndex"] = mat_index
    return v


def _stack(items: List[np.ndarray], dtype) -> np.ndarray:
    out = np.zeros(len(items), dtype=dtype)
    for i, it in enumerate(items):
        out[i] = it
    return out


DEMO_LIGHTS = (
    ((2.0, 2.0, 1.0), 0.5, (1.0, 0.9, 0.8)),
    ((-2.0, 1.0, 0.0), 0.5, (0.3, 0.5, 1.0)),
    ((0.0, -0.2, 0.5), 0.5, (1.0, 0.4, 0.4)),
)


def axis_angle_quat(axis, angle: float) -> np.ndarray:
    """Quaternion::fromAxisAngle in double, returned as float32 (x, y, z, w) —
    the ``quaternionToGlm`` repack (reference src/main_vulkan.cpp:79-81,
    src/Quaternion.cpp:13-23)."""
    ax = np.asarray(axis, dtype=np.float64)
    ax = ax / math.sqrt(float(ax @ ax))
    s, c = math.sin(angle * 0.5), math.cos(angle * 0.5)
    return np.array([ax[0] * s, ax[1] * s, ax[2] * s, c], dtype=np.float32)


# ----------------------------------------------------------------------------
# Synthetic "walnut": 400 x 296 x 352 uint8, x fastest.
# ----------------------------------------------------------------------------
WALNUT_RES = (400, 296, 352)


def _hash_u32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7FEB352D)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846CA68B)
    x ^= x >> np.uint32(16)
    return x


def synthetic_walnut(resolution=WALNUT_RES, seed: int = 0xC0FFEE04) -> np.ndarray:
    """Deterministic CT-like volume: a bumpy ellipsoidal shell, two lobed
    kernels separated by a septum gap, and low-amplitude hashed noise.
    Returned flat, uint8, x fastest then y then z (the ``.raw`` order,
    reference src/main_vulkan.cpp:61-66)."""
    rx, ry, rz = resolution
    out = np.empty((rz, ry, rx), dtype=np.uint8)
    xs = (np.arange(rx, dtype=np.float32) + np.float32(0.5)) / np.float32(rx) * 2 - 1
    ys = (np.arange(ry, dtype=np.float32) + np.float32(0.5)) / np.float32(ry) * 2 - 1
    X, Y = np.meshgrid(xs, ys)  # (ry, rx)
    lin = (np.arange(ry, dtype=np.uint32)[:, None] * np.uint32(rx) + np.arange(rx, dtype=np.uint32)[None, :])
    for z in range(rz):
        zf = np.float32((z + 0.5) / rz * 2 - 1)
        rr = np.sqrt((X / np.float32(0.86)) ** 2 + (Y / np.float32(0.80)) ** 2 + (zf / np.float32(0.90)) ** 2)
        ang = np.arctan2(Y, X)
        bump = np.float32(0.025) * np.sin(9 * ang + 5 * zf) * np.cos(7 * zf * np.float32(3.0) + 3 * ang)
        shell = (rr > 0.88 + bump) & (rr < 1.0 + bump)
        # kernels: two lobes left/right of the x = 0 septum, wrinkled surface
        wr = np.float32(0.06) * np.sin(14 * X + 3 * zf) * np.sin(12 * Y - 2 * zf) * np.sin(np.float32(11.0) * zf)
        lobe = (rr < 0.74 + wr) & (np.abs(X) > 0.035 + np.float32(0.02) * np.sin(8 * Y + 6 * zf))
        h = _hash_u32(lin + np.uint32((z * rx * ry + seed) & 0xFFFFFFFF))
        noise = (h & np.uint32(0xFF)).astype(np.float32) / np.float32(255.0)
        val = np.float32(6.0) * noise  # air
        val = np.where(lobe, np.float32(118.0) + np.float32(40.0) * noise, val)
        val = np.where(shell, np.float32(190.0) + np.float32(50.0) * noise, val)
        out[z] = np.clip(val, 0, 255).astype(np.uint8)
    return out.reshape(-1)


def load_walnut(raw_path: Optional[str] = None, resolution=WALNUT_RES) -> np.ndarray:
    """The real ``walnut.raw`` if a readable file of the right size is named
    (or found via ``PT_WALNUT_RAW``); otherwise the synthetic volume."""
    path = raw_path or os.environ.get("PT_WALNUT_RAW", "")
    n = resolution[0] * resolution[1] * resolution[2]
    if path and os.path.isfile(path) and os.path.getsize(path) >= n:
        return np.fromfile(path, dtype=np.uint8, count=n)
    return synthetic_walnut(resolution)


# ----------------------------------------------------------------------------
# Named scenes
# ----------------------------------------------------------------------------
def demo_materials() -> List[Material]:
    """Resolved order 0 red, 1 green-emissive, 2 blue-rainbow, 3 yellow, 4 mirror,
    5 walnut volumetric (first-seen over spheres -> ellipsoids -> volumes,
    reference src/main_vulkan.cpp:234, include/SceneWrappers.h:321-353)."""
    red = Material.Diffuse((0.8, 0.2, 0.2), 0.7, 0.1)
    red.specular, red.shininess = 0.3, 32.0
    green = Material.Emissive((0.2, 0.8, 0.2), 2.0)
    blue = Material.CustomShader((0.2, 0.2, 0.8), T.SHADER_RAINBOW)
    blue.diffuse, blue.ambient, blue.specular, blue.shininess, blue.emissive_strength = 0.7, 0.1, 0.5, 64.0, 1.0
    yellow = Material.Diffuse((0.8, 0.8, 0.2), 0.6, 0.1)
    yellow.specular, yellow.shininess = 0.1, 16.0
    mirror = Material.Mirror((0.9, 0.9, 0.9), 0.9)
    walnut = Material.Volumetric((0.8, 0.6, 0.4), 8.0)
    return [red, green, blue, yellow, mirror, walnut]


def demo_scene(width=800, height=600, frame: int = 0, with_volume: bool = True,
               voxels: Optional[np.ndarray] = None, volume_res=WALNUT_RES,
               time_ms: float = 0.0, max_depth: int = 10) -> Scene:
    """The reference demo at orbit frame ``frame`` (reference src/main_vulkan.cpp:136-223,
    340-354): eye = (2+3cos(k/180), 1.5, 6+3sin(k/180)); ellipsoids rotated about
    (2,1,0) by (k+1)/180 because theta is incremented before use."""
    mats = demo_materials()
    sph = [
        _sphere((0.0, 0.0, -1.0), 0.5, 0, (0.8, 0.2, 0.2)),
        _sphere((-1.0, 0.0, -1.0), 0.3, 1, (0.2, 0.8, 0.2)),
        _sphere((1.0, 0.0, -1.0), 0.3, 2, (0.2, 0.2, 0.8)),
        _sphere((0.0, -100.5, -1.0), 100.0, 3, (0.8, 0.8, 0.2)),
        _sphere((0.5, 0.3, -0.5), 0.2, 4, (0.9, 0.9, 0.9)),
        _sphere((2.0, 0.5, 1.5), 1.0, 3, (0.8, 0.8, 0.2)),
    ]
    theta = frame / 180.0
    q = axis_angle_quat((2.0, 1.0, 0.0), (frame + 1) / 180.0)
    ell = [
        _ellipsoid((-2.0, 0.8, -1.0), (0.5, 0.8, 0.3), 4, q, (0.8, 0.4, 0.8)),
        _ellipsoid((0.0, 1.2, -2.0), (0.6, 0.4, 0.4), 2, q, (0.4, 0.8, 0.8)),
    ]
    lights = [_light(*l) for l in DEMO_LIGHTS]
    sc = Scene(width=width, height=height, name="demo")
    sc.spheres = _stack(sph, T.GPU_SPHERE)
    sc.ellipsoids = _stack(ell, T.GPU_ELLIPSOID)
    sc.lights = _stack(lights, T.GPU_LIGHT)
    if with_volume:
        sc.materials = _materials(mats)
        sc.volumes = _stack([_volume((1.5, 1.0, -0.5), (0, 0, 0), (2, 2, 2), volume_res, 5)], T.GPU_VOLUME)
        sc.voxels = voxels if voxels is not None else load_walnut(resolution=volume_res)
    else:
        sc.materials = _materials(mats[:5])
    sc.camera_pos = (np.float32(2.0 + 3.0 * math.cos(theta)), 1.5, np.float32(6.0 + 3.0 * math.sin(theta)))
    sc.time_ms = time_ms
    sc.max_depth = max_depth
    return sc


def config1(width=640, height=480) -> Scene:
    """C1: single sphere + one point light, depth 1."""
    red = Material.Diffuse((0.8, 0.2, 0.2), 0.7, 0.1)
    red.specular, red.shininess = 0.3, 32.0
    sc = Scene(width=width, height=height, name="c1")
    sc.materials = _materials([red])
    sc.spheres = _stack([_sphere((2.0, 1.5, 0.0), 1.0, 0)], T.GPU_SPHERE)
    sc.lights = _stack([_light((4.0, 4.0, 4.0), 1.0)], T.GPU_LIGHT)
    sc.camera_pos = (0.0, 1.5, 6.0)
    sc.max_depth = 1
    return sc


def _c2_palette() -> List[Material]:
    yellow = Material.Diffuse((0.8, 0.8, 0.2), 0.6, 0.1)
    red = Material.Diffuse((0.8, 0.2, 0.2), 0.7, 0.1)
    red.specular, red.shininess = 0.3, 32.0
    blue = Material.Diffuse((0.3, 0.5, 0.8), 0.7, 0.1)
    m1 = Material.Mirror((0.9, 0.9, 0.9), 0.9)
    m2 = Material.Mirror((0.9, 0.7, 0.5), 0.7)
    glass = Material.Diffuse((0.6, 0.9, 0.7), 0.7, 0.1)
    glass.transparency = 0.5
    glow = Material.Emissive((0.2, 0.8, 0.2), 2.0)
    toon = Material.CustomShader((0.9, 0.4, 0.6), T.SHADER_TOON)
    return [yellow, red, blue, m1, m2, glass, glow, toon]


def _random_unit_quat(rng: PCG32) -> np.ndarray:
    """Uniform unit quaternion (Shoemake), computed in double, stored float32 (x,y,z,w)."""
    u1, u2, u3 = float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), float(rng.uniform(0, 1))
    a, b = math.sqrt(1.0 - u1), math.sqrt(u1)
    q = (a * math.sin(2 * math.pi * u2), a * math.cos(2 * math.pi * u2),
         b * math.sin(2 * math.pi * u3), b * math.cos(2 * math.pi * u3))
    return np.array(q, dtype=np.float32)


def config2(width=1920, height=1080, n_spheres=47, n_ellipsoids=16, seed=0xC0FFEE02,
            max_depth=8) -> Scene:
    """C2: ground sphere + 47 spheres + 16 ellipsoids, 8-material palette, demo lights.

    Draw order per sphere: cx, cy, cz, r, material; per ellipsoid: cx, cy, cz,
    rx, ry, rz, three quaternion uniforms, material.  Spheres first."""
    rng = PCG32(seed)
    pal = _c2_palette()
    sph = [_sphere((0.0, -100.5, -1.0), 100.0, 0)]
    for _ in range(n_spheres):
        c = (rng.uniform(-3, 7), rng.uniform(-0.2, 3), rng.uniform(-6, 3))
        r = rng.uniform(0.15, 0.5)
        sph.append(_sphere(c, r, rng.below(len(pal))))
    ell = []
    for _ in range(n_ellipsoids):
        c = (rng.uniform(-3, 7), rng.uniform(-0.2, 3), rng.uniform(-6, 3))
        rad = (rng.uniform(0.15, 0.6), rng.uniform(0.15, 0.6), rng.uniform(0.15, 0.6))
        q = _random_unit_quat(rng)
        ell.append(_ellipsoid(c, rad, rng.below(len(pal)), q))
    sc = Scene(width=width, height=height, name="c2")
    sc.materials = _materials(pal)
    sc.spheres = _stack(sph, T.GPU_SPHERE)
    sc.ellipsoids = _stack(ell, T.GPU_ELLIPSOID)
    sc.lights = _stack([_light(*l) for l in DEMO_LIGHTS], T.GPU_LIGHT)
    sc.camera_pos = (5.0, 1.5, 6.0)
    sc.max_depth = max_depth
    return sc


def config3(width=3840, height=2160, n_spheres=4095, seed=0xC0FFEE03, max_depth=8) -> Scene:
    """C3: ground sphere + 4095 random spheres; 16 materials = 8 diffuse, 6 mirror
    (reflectivity uniform 0.5-0.9), 2 transparent (0.5); objects pick a class with
    probability 0.5 / 0.4 / 0.1 and then a material of that class; one light.

    Draw order: the 16 materials first (diffuse: r,g,b; mirror: r,g,b,reflectivity;
    transparent: r,g,b), then per sphere cx, cy, cz, r, class draw, member draw."""
    rng = PCG32(seed)
    pal: List[Material] = []
    for _ in range(8):
        pal.append(Material.Diffuse((rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.9)), 0.7, 0.1))
    for _ in range(6):
        col = (rng.uniform(0.6, 0.95), rng.uniform(0.6, 0.95), rng.uniform(0.6, 0.95))
        pal.append(Material.Mirror(col, float(rng.uniform(0.5, 0.9))))
    for _ in range(2):
        m = Material.Diffuse((rng.uniform(0.5, 0.95), rng.uniform(0.5, 0.95), rng.uniform(0.5, 0.95)), 0.7, 0.1)
        m.transparency = 0.5
        pal.append(m)
    sph = np.zeros(n_spheres + 1, dtype=T.GPU_SPHERE)
    sph[0] = _sphere((0.0, -100.5, -1.0), 100.0, 0)
    for i in range(1, n_spheres + 1):
        c = (rng.uniform(-10, 14), rng.uniform(-0.3, 5), rng.uniform(-20, 3))
        r = rng.uniform(0.05, 0.25)
        u = float(rng.uniform(0, 1))
        if u < 0.5:
            mi = rng.below(8)
        elif u < 0.9:
            mi = 8 + rng.below(6)
        else:
            mi = 14 + rng.below(2)
        sph[i] = _sphere(c, r, mi)
    sc = Scene(width=width, height=height, name="c3")
    sc.materials = _materials(pal)
    sc.spheres = sph
    sc.lights = _stack([_light((2.0, 8.0, 4.0), 2.0)], T.GPU_LIGHT)
    sc.camera_pos = (5.0, 1.5, 6.0)
    sc.max_depth = max_depth
    return sc


def config4(width=1920, height=1080, voxels: Optional[np.ndarray] = None,
            resolution=WALNUT_RES, max_depth=8) -> Scene:
    """C4: no solids; the walnut volume at (1.5,1,-0.5), box (0,0,0)-(2,2,2),
    Material::Volumetric((0.8,0.6,0.4), 8.0)."""
    sc = Scene(width=width, height=height, name="c4")
    sc.materials = _materials([Material.Volumetric((0.8, 0.6, 0.4), 8.0)])
    sc.volumes = _stack([_volume((1.5, 1.0, -0.5), (0, 0, 0), (2, 2, 2), resolution, 0)], T.GPU_VOLUME)
    sc.voxels = voxels if voxels is not None else load_walnut(resolution=resolution)
    sc.camera_pos = (5.0, 1.5, 6.0)
    sc.max_depth = max_depth
    return sc


def config5(width=3840, height=2160, voxels: Optional[np.ndarray] = None,
            resolution=WALNUT_RES, max_depth=8) -> Scene:
    """C5: C2's objects plus the C4 volume at 4K (one sample per pixel here;
    accumulation over samples has no counterpart in the reference)."""
    sc = config2(width, height, max_depth=max_depth)
    mats = list(sc.materials)
    vol_mat = Material.Volumetric((0.8, 0.6, 0.4), 8.0).to_gpu()
    sc.materials = np.concatenate([sc.materials, np.array([vol_mat], dtype=T.GPU_MATERIAL)])
    sc.volumes = _stack([_volume((1.5, 1.0, -0.5), (0, 0, 0), (2, 2, 2), resolution, len(mats))], T.GPU_VOLUME)
    sc.voxels = voxels if voxels is not None else load_walnut(resolution=resolution)
    sc.name = "c5"
    return sc
