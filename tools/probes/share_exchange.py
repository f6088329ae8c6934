"""python tools/probes/share_exchange.py <workload> [world=8] [frames]: what rank 0 of a <world>-way band split does per frame
besides drawing its share, rehearsed on one GPU with 8 frames in flight: the de-interleaving pass over the whole gathered frame
(pt_assemble_bands, on the stream of the frame just drawn).  The peers' sends cannot be rehearsed on one GPU; their size is printed."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer, bands
name = sys.argv[1]; world = int(sys.argv[2]) if len(sys.argv) > 2 else 8; n = int(sys.argv[3]) if len(sys.argv) > 3 else 32
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name](); pc = sc.push_constants()
stride = bands.stride_rows(sc.height, 16, world)
gathered = torch.zeros((world, stride, sc.width, 4), dtype=torch.uint8, device='cuda:0')
frame = torch.zeros((sc.height, sc.width, 4), dtype=torch.uint8, device='cuda:0')
res = {}
for with_assemble in (False, True, False, True):
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1, 0); r.setOption(6, 8); r.setScene(sc, honour_max_depth=True); r.setRowBands(16, 0, world)
    ms = []
    for rep in range(4):
        r.timerBegin()
        for i in range(n):
            r.render(pc)
            if with_assemble: r.assembleBands(gathered.data_ptr(), frame.data_ptr(), 4, 16, world, stride)
        r.timerEnd(); ms.append(r.timerElapsedMs() / n)
    res.setdefault(with_assemble, []).append(float(np.median(ms[1:]))); r.shutdown()
a, b = min(res[False]), min(res[True])
part = stride * sc.width * 4 / 1e6
print(f"{name} rank 0 of {world}, 8 frames in flight: share alone {a:.4f} ms/frame, with the de-interleave of the whole frame {b:.4f} "
      f"(+{b - a:.4f} ms); each peer sends {part:.2f} MB per frame over its own link ({part / 50e3 * 1e3:.3f} ms at 50 GB/s)")
