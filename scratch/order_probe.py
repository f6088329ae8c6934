import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
for name in sys.argv[1:] or ['c2','c3','c4']:
    sc = {'c3':scenes.config3,'c2':scenes.config2,'c4':scenes.config4}[name]()
    frames = {}
    for order in (0,1):
        r = HipRenderer(); assert r.initialize(sc.width, sc.height)
        r.setOption(1,0); r.setOption(3,order); r.setScene(sc); pc = sc.push_constants()
        ms=[]
        for i in range(5):
            r.timerBegin(); r.render(pc); r.timerEnd(); ms.append(r.timerElapsedMs())
        frames[order] = r.readbackRGBA8().copy()
        print(name, 'order', order, 'ms', ' '.join('%.3f'%m for m in ms))
        r.shutdown()
    print(name, 'frames equal', np.array_equal(frames[0], frames[1]))
