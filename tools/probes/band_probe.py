import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1] if len(sys.argv)>1 else 'c3'
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
pc = sc.push_constants()
for world in (1,2,4,8):
    times=[]
    for rank in range(world):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1,0); r.setScene(sc, honour_max_depth=True)
        if world>1: r.setRowBands(16, rank, world)
        ms=[]
        for i in range(5):
            r.timerBegin(); r.render(pc); r.timerEnd(); ms.append(r.timerElapsedMs())
        times.append(float(np.median(ms[1:]))); r.shutdown()
    print(f"{name} world {world}: per-rank kernel ms {' '.join('%.3f'%t for t in times)}  max {max(times):.3f}  ideal {times[0] if world==1 else base/world:.3f}")
    if world==1: base=times[0]
