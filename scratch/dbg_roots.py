import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, render_scene
from oracle import oracle_py as O
for n in (3,4,5,6,7,63,64,65,999,1000,1001,1002,1003):
    sc = scenes.config3(128,72,n_spheres=n)
    g = render_scene(sc, counted=True)
    r = O.render(sc, nthreads=16)
    same = np.array_equal(g[1].view(np.uint32), r[1].view(np.uint32))
    print(n+1, 'gpu roots', int(g[2][8]), 'oracle', int(r[2][8]), 'diff', int(r[2][8])-int(g[2][8]), 'rays', int(g[2][0]), 'bitexact', same, 'trips', int(g[2][10]), 'slow', int(g[2][11]))
