"""python tools/probes/ab_c4.py <lib.so> [...]: the volume frame (C4) per library build, kernel ms one at a time / two in flight, nearest and trilinear."""
import os, subprocess, sys, json
CHILD = r'''
import sys, json, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import _lib
_lib.LIB_PATH = sys.argv[1]
from path_tracer_amd import scenes, HipRenderer
sc = scenes.config4(); pc = sc.push_constants(); out = {}
for filt in (0, 1):
    for fif in (1, 2):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setOption(5, filt); r.setScene(sc, honour_max_depth=True)
        res = []
        for rep in range(5):
            r.timerBegin()
            for i in range(40): r.render(pc)
            r.timerEnd(); res.append(r.timerElapsedMs() / 40)
        out[f'filter{filt}/{fif}'] = round(float(np.median(res[1:])), 4); r.shutdown()
print(json.dumps(out))
'''
for rnd in range(2):
    for lib in sys.argv[1:]:
        p = subprocess.run([sys.executable, '-c', CHILD, os.path.abspath(lib)], capture_output=True, text=True, timeout=300)
        print(lib, p.stdout.strip().split('\n')[-1] if p.returncode == 0 else p.stderr[-1500:], flush=True)
