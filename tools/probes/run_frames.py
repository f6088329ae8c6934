"""python tools/probes/run_frames.py <workload> <tune> [frames] [lib.so]: draw frames one at a time (for rocprofv3 passes of one variant)."""
import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
if len(sys.argv) > 4:
    from path_tracer_amd import _lib
    _lib.LIB_PATH = __import__('os').path.abspath(sys.argv[4])
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; tune = int(sys.argv[2], 0); n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
def _c4_away():  # the C4 volume with the camera moved so that no ray meets the box: the frame's cost without the march
    sc = scenes.config4(); sc.camera_pos = (2.0, 1.5, -8.0); return sc
sc = {'c4tri': scenes.config4, 'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4, 'c4away': _c4_away,
      'empty': lambda: scenes.Scene(width=1920, height=1080)}[name](); pc = sc.push_constants()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
r.setOption(1, 0); r.setOption(6, 1); r.setOption(100, tune)
if name == 'c4tri': r.setOption(5, 1)  # the trilinear filter
r.setScene(sc, honour_max_depth=True)
for i in range(n):
    r.render(pc); r.present()
print(name, hex(tune), r.lastLaunch())
