"""A/B of two builds of the library on one box: python tools/probes/ab_lib.py <a.so> <b.so> [rounds]
Each round runs both libraries in a process of its own, alternating; prints median kernel ms per workload
one frame at a time and with two frames in flight."""
import os, subprocess, sys, json
import numpy as np

CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer
out = {}
for name, mk, n in (('c2', scenes.config2, 40), ('c3', scenes.config3, 4), ('c4', scenes.config4, 40)):
    sc = mk(); pc = sc.push_constants()
    for fif in (1, 2):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
        res = []
        for rep in range(4):
            r.timerBegin()
            for i in range(n): r.render(pc)
            r.timerEnd(); res.append(r.timerElapsedMs() / n)
        out[f'{name}/{fif}'] = float(np.median(res[1:])); r.shutdown()
print(json.dumps(out))
'''

def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    acc = {a: [], b: []}
    for _ in range(rounds):
        for lib in (a, b):
            p = subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
            if p.returncode != 0:
                print(p.stderr[-2000:]); sys.exit(1)
            acc[lib].append(json.loads(p.stdout.strip().split('\n')[-1]))
    keys = acc[a][0].keys()
    for k in keys:
        va = np.median([x[k] for x in acc[a]]); vb = np.median([x[k] for x in acc[b]])
        print(f'{k}: A {va:.4f}  B {vb:.4f}  B/A {vb / va:.3f}')

main()
