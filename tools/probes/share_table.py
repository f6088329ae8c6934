"""python tools/probes/share_table.py <workload> [frames]: kernel ms per frame of rank 0's share of a 1-, 2-, 4- and 8-way band
split with 1, 2, 4 and 8 frames in flight, in a process that holds an RCCL communicator made BEFORE the renderer's slots
(what a rank of bench.py --gpus N is); run with GPU_MAX_HW_QUEUES=16, as bench.py sets it when the frame is split."""
import os, sys, numpy as np
sys.path.insert(0, '.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sc = {'c3': scenes.config3, 'c2': scenes.config2, 'c4': scenes.config4}[name](); pc = sc.push_constants()
boot = HipRenderer(); assert boot.initialize(64, 64)
comm = boot.commCreate(HipRenderer.commUniqueId(), 0, 1, 0)  # stays for the life of the process
print(f'{name}: GPU_MAX_HW_QUEUES={os.environ.get("GPU_MAX_HW_QUEUES")}, communicator first; ms per frame (streams side by side)')
for fif in (1, 2, 4, 8):
    row = []
    for world in (1, 2, 4, 8):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1, 0); r.setOption(6, fif); r.setScene(sc, honour_max_depth=True)
        if world > 1: r.setRowBands(16, 0, world)
        res = []
        for rep in range(4):
            r.timerBegin()
            for i in range(max(n, 2 * fif)): r.render(pc)
            r.timerEnd(); res.append(r.timerElapsedMs() / max(n, 2 * fif))
        row.append(f'{np.median(res[1:]):.3f} ({r.streamConcurrency() if fif > 1 else 1})'); r.shutdown()
    print(f' {fif} in flight | ' + ' | '.join(row), flush=True)
boot.commDestroy(comm); boot.shutdown()
