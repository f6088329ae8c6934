"""The C++ demo driver (examples/demo_orbit, the reference's main loop over the
C++ scene API and HipRenderer) against the oracle rendering of the same frames."""
import os
import subprocess

import numpy as np
import pytest

from path_tracer_amd import scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_ppm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"P6"
        w, h = (int(x) for x in f.readline().split())
        assert f.readline().strip() == b"255"
        return np.frombuffer(f.read(), np.uint8).reshape(h, w, 3)


def test_orbit_frames_match_the_oracle(oracle, tmp_path):
    exe = os.path.join(ROOT, "examples", "demo_orbit")
    assert os.path.exists(exe), "build examples/demo_orbit first (python __graft_entry__.py)"
    W, H, dt = 200, 150, 100.0
    subprocess.run([exe, "3", str(W), str(H), str(tmp_path), str(dt)], cwd=str(tmp_path), check=True, timeout=120)
    for k in range(3):
        img = _read_ppm(tmp_path / f"frame_{k:04d}.ppm")
        # no volume files in the working directory: the demo renders without the volume;
        # the reference shader always runs 10 bounces (referenceBounceLimit)
        sc = scenes.demo_scene(W, H, frame=k, with_volume=False, time_ms=np.float32(k * np.float32(dt)), max_depth=10)
        ref, _, _ = oracle.render(sc, nthreads=8, want_f32=False)
        d = np.abs(img.astype(np.int16) - ref[..., :3].astype(np.int16)).max(axis=-1)
        frac = float((d <= 1).mean())
        print(f"[demo frame {k}] within 1 LSB: {frac:.6f}, identical: {float((d == 0).mean()):.6f}")
        assert frac >= 0.999
