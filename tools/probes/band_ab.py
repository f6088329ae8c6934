import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
name = sys.argv[1]; opt, vals = sys.argv[2].split('='); opt=int(opt); vals=[int(v) for v in vals.split(',')]
sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
pc = sc.push_constants()
for world in (1,2,4,8):
    out=[]
    for v in vals:
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1,0); r.setOption(opt,v); r.setScene(sc, honour_max_depth=True)
        if world>1: r.setRowBands(16, 0, world)
        ms=[]
        for i in range(6):
            r.timerBegin(); r.render(pc); r.timerEnd(); ms.append(r.timerElapsedMs())
        out.append('%d: %.3f'%(v,float(np.median(ms[1:])))); r.shutdown()
    print(name,'world',world,'opt',opt,' | '.join(out))
