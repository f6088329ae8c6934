"""python tools/probes/torch_effect.py <workload> <0|1|2> [tune]: one rank's 1/8 share with 8 frames in flight in a process that has
not imported torch (0), has imported it (1), has made a CUDA tensor with it (2), has made an RCCL communicator of one rank through the library (3)."""
import sys, numpy as np
sys.path.insert(0, '.')
mode = int(sys.argv[2])
if mode >= 1: import torch
if mode >= 2: x = torch.zeros(16, device='cuda:0')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name](); pc = sc.push_constants()
r = HipRenderer(); assert r.initialize(sc.width, sc.height)
if mode == 3: comm = r.commCreate(HipRenderer.commUniqueId(), 0, 1, 0)  # an RCCL communicator in the process
r.setOption(1, 0); r.setOption(6, 8); r.setOption(100, int(sys.argv[3], 0) if len(sys.argv) > 3 else 0); r.setScene(sc, honour_max_depth=True); r.setRowBands(16, 0, 8)
ms = []
for rep in range(4):
    r.timerBegin()
    for i in range(32): r.render(pc)
    r.timerEnd(); ms.append(r.timerElapsedMs() / 32)
print(name, 'torch mode', mode, ' '.join('%.4f' % m for m in ms), r.lastLaunch())
