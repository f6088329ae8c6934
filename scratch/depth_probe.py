import sys, numpy as np
sys.path.insert(0,'.')
from path_tracer_amd import scenes, HipRenderer
from path_tracer_amd import types as T
for depth in (1,2,3,4,8):
    sc = scenes.config3(max_depth=depth)
    r = HipRenderer(); assert r.initialize(sc.width, sc.height)
    r.setOption(1,0); r.setScene(sc); pc = sc.push_constants()
    cnt = r.renderCounted(pc)
    d=dict(zip(T.COUNTER_NAMES,[int(x) for x in cnt]))
    best=1e9
    for i in range(3):
        r.timerBegin(); r.render(pc); r.timerEnd(); best=min(best,r.timerElapsedMs())
    fast_ms = d['wave_trips']*135/(1024*2.4e9)*1e3
    print(f"depth {depth}: exact {d['exact_rays']} rays {d['rays']/1e6:.2f}M trips {d['wave_trips']/1e6:.1f}M slow {d['wave_slow_trips']/1e6:.1f}M ({d['wave_slow_trips']/d['wave_trips']:.3f}) roots {d['sphere_roots']/1e6:.1f}M roots/ray {d['sphere_roots']/d['rays']:.2f} ms {best:.2f} fast-only est {fast_ms:.2f} extra {best-fast_ms:.2f} ms -> {(best-fast_ms)*1e-3*1024*2.4e9/max(d['wave_slow_trips'],1):.0f} SIMD-cycles per slow trip")
    r.shutdown()
