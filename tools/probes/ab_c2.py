"""A/B of two builds on the 1080p / 64-object frame only, long runs: python tools/probes/ab_c2.py <a.so> <b.so> [rounds]
Each round runs both libraries in a process of its own, alternating; 6 x 400 frames per mode and process."""
import os, subprocess, sys, json
import numpy as np

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer
out = {}
sc = scenes.config2(); pc = sc.push_constants()
for fif in (1, 2):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
    res = []
    for rep in range(7):
        r.timerBegin()
        for i in range(400): r.render(pc)
        r.timerEnd(); res.append(r.timerElapsedMs() / 400)
    out[f'c2/{fif}'] = float(np.median(res[1:])); r.shutdown()
print(json.dumps(out))
'''

def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    acc = {a: [], b: []}
    for _ in range(rounds):
        for lib in (a, b):
            p = subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                print(p.stderr[-2000:]); sys.exit(1)
            acc[lib].append(json.loads(p.stdout.strip().split('\n')[-1]))
    for k in acc[a][0].keys():
        xa = [x[k] for x in acc[a]]; xb = [x[k] for x in acc[b]]
        print(f'{k}: A {np.median(xa):.4f} ({min(xa):.4f}-{max(xa):.4f})  B {np.median(xb):.4f} ({min(xb):.4f}-{max(xb):.4f})  B/A {np.median(xb) / np.median(xa):.3f}')

main()
