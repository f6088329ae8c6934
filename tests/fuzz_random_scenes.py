"""Long differential run, by hand: python tests/fuzz_random_scenes.py [first_seed] [count] [volume]
(with a third argument every scene is a frame without solids)
Random scenes of tests/test_gpu_random_scenes.py (all three generators) against the oracle, and three samples per pixel
drawn in one launch against the same samples drawn launch by launch; prints the seeds that differ."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from path_tracer_amd import HipRenderer, types as T
from oracle import oracle_py
import test_gpu_random_scenes as G

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
only_volume = len(sys.argv) > 3
oracle_py.lib()
orc = oracle_py
bad = []
for seed in range(first, first + count):
    sc = (G.random_volume_frame(seed) if only_volume or seed % 10 == 5 else
          G.random_hall_of_mirrors(seed) if seed % 10 == 0 else G.random_scene(seed))
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setScene(sc, honour_max_depth=True); pc = sc.push_constants()
    r.render(pc); r.present()
    g8, gf = r.readbackRGBA8(), r.readbackF32()
    cnt = r.renderCounted(pc)
    # three samples per pixel: all in one launch against one launch per sample, bit for bit
    frames = []
    for batch in (0, 1):
        r.setOption(T.OPT_SAMPLE_BATCH, batch)
        r.renderSamples(pc, 0, 3); r.present()
        frames.append((r.readbackRGBA8().copy(), r.readbackF32().copy()))
    r.shutdown()
    if not (np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1].view(np.uint32), frames[1][1].view(np.uint32))):
        bad.append(seed); print("DIFF (samples per launch)", seed, flush=True)
    r8, rf, rc = orc.render(sc, nthreads=16)
    same_cnt = np.array_equal(cnt[:T.N_ALGORITHM_COUNTERS], rc[:T.N_ALGORITHM_COUNTERS])
    d8 = np.abs(g8.astype(np.int16) - r8.astype(np.int16)).max(axis=-1)
    okf = ((np.abs(gf - rf) <= 1e-4 * np.maximum(np.abs(rf), 1e-2)) | (np.isnan(gf) & np.isnan(rf))).all(axis=-1)
    if not same_cnt or (d8 <= 1).mean() < 0.999 or okf.mean() < 0.999:
        bad.append(seed); print("DIFF", seed, same_cnt, float((d8 <= 1).mean()), float(okf.mean()), flush=True)
    if (seed - first) % 100 == 99: print("done", seed - first + 1, "bad", len(bad), flush=True)
print("seeds that differ:", bad)
sys.exit(1 if bad else 0)
