"""Rehearsal on one GPU of the multi-GPU side measurements' code path under the nccl backend (world size 1):
band split + per-slot torch streams + gather + assemble for plain frames (8 in flight) and for accumulated samples;
the gathered frames must equal the frames drawn without a split."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
import bench
from path_tracer_amd import scenes
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda:0"))
ok = True
for name, mk in (("c3", lambda: scenes.config3(480, 270, n_spheres=600)), ("c4", lambda: scenes.config4(480, 270)),
                 ("c5", lambda: scenes.config5(480, 270))):
    sc = mk()
    rr = bench.Runner(sc, 0, 1, 0, torch, dist, "nccl", split=True, frames_in_flight=8 if name != "c5" else 1)
    solo = bench.Runner(sc, 0, 1, 0, frames_in_flight=1)
    if name == "c5":
        rr.step(samples=6); solo.r.renderSamples(solo.pc, 0, 6)
    else:
        for _ in range(11): rr.step()
        solo.step()
    rr.sync(); solo.sync()
    same = np.array_equal(rr.final_frame(), solo.final_frame())
    print(name, "gathered frame equals the plain frame:", same)
    ok = ok and same
    rr.close(); solo.close()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
