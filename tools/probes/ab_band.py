"""A/B of two builds of the library on one rank's 1/8 share of C2 and C3 (latency of a frame drawn alone and
time per frame with 8 in flight): python tools/probes/ab_band.py <a.so> <b.so> [rounds]"""
import os, subprocess, sys, json
import numpy as np
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer
out = {}
for name, mk, n in (('c2', scenes.config2, 48), ('c3', scenes.config3, 8)):
    sc = mk(); pc = sc.push_constants()
    for world in (1, 8):
        for fif in (1, 8):
            r = HipRenderer(); assert r.initialize(sc.width, sc.height)
            r.setOption(1, 0); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
            if world > 1: r.setRowBands(16, 0, world)
            res = []
            for rep in range(4):
                r.timerBegin()
                for i in range(n): r.render(pc)
                r.timerEnd(); res.append(r.timerElapsedMs() / n)
            out[f'{name}/w{world}/f{fif}'] = float(np.median(res[1:])); r.shutdown()
print(json.dumps(out))
'''
a, b = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
acc = {a: [], b: []}
env = dict(os.environ, GPU_MAX_HW_QUEUES='8')
for _ in range(rounds):
    for lib in (a, b):
        p = subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(lib)], capture_output=True, text=True, timeout=300, env=env)
        if p.returncode != 0:
            print(p.stderr[-2000:]); sys.exit(1)
        acc[lib].append(json.loads(p.stdout.strip().split('\n')[-1]))
for k in acc[a][0]:
    va = np.median([x[k] for x in acc[a]]); vb = np.median([x[k] for x in acc[b]])
    print(f'{k}: A {va:.4f}  B {vb:.4f}  B/A {vb / va:.3f}')
