import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
from path_tracer_amd import types as T
for depth in (1,2,4,8):
    sc = scenes.config2(max_depth=depth)
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1,0); r.setScene(sc, honour_max_depth=True); pc = sc.push_constants()
    r.timerBegin(); cnt = r.renderCounted(pc); r.timerEnd(); cms = r.timerElapsedMs()
    d=dict(zip(T.COUNTER_NAMES,[int(x) for x in cnt]))
    best=1e9
    for i in range(4):
        r.timerBegin(); r.render(pc); r.timerEnd(); best=min(best,r.timerElapsedMs())
    tot=sum(int(cnt[i]) for i in range(13,18))
    passes = d['wave_trips']/6
    print(f"depth {depth}: rays {d['rays']/1e6:.2f}M rays/px {d['rays']/(sc.width*sc.height):.2f} passes/wave {passes/4096:.1f} ms {best:.3f} counted-ms {cms:.3f} wave-cycles/wave {tot/4096/1e6:.2f}M -> alive frac at 2.25GHz {tot/4096/(cms*1e-3*2.25e9):.2f} cycles/pass {tot/passes:.0f}")
    r.shutdown()
