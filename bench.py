#!/usr/bin/env python3
"""bench.py — throughput of the per-pixel ray loop on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one frame of the workload.  The default workload is C2, the
configuration the headline metric is quoted on (1920x1080, 64 mixed
spheres/ellipsoids, depth 8): value = Mrays/s over all N GPUs, a ray being one
query of the whole object list (primary, bounce, light-shadow and
emissive-occlusion rays).  With N > 1 the frame is split into interleaved
16-row bands, one set per rank, and gathered to rank 0 over RCCL inside the
timed region (strong scaling: the total work is fixed).  The gather is the
library's own (pt_gather_bands: grouped ncclSend / ncclRecv behind the C-ABI);
torch.distributed only carries the rendezvous, the barriers and the reduction
of the timings (gloo, on the CPU).

The same JSON line carries
  roofline      algorithmic FLOP (17 per sphere test, 91 per ellipsoid test,
                brute force) / HIP-event kernel time, against the 157.3 TFLOP/s
                vector-FP32 peak — the bound of this kernel is the VALU, not
                HBM or MFMA;
  cpu_baseline  the CPU oracle (a restatement of the reference shader; the
                reference itself has no CPU path and cannot run without
                Vulkan) timed on this host's cores on the same frame;
  workloads     C3 (4K, 4096 spheres) and C4 (walnut volume) frame times.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# A split frame is drawn with eight frame slots, one stream each, and the slots only run side by side when each stream
# has a hardware queue of its own.  HIP gives a process four unless told otherwise, and an RCCL communicator (or torch's
# CUDA context) in the process takes its part of them: measured on one rank's 1/8 share of the 4K / 4096-sphere frame
# with a communicator in the process, 8 queues -> 4 streams side by side (2.72 ms per frame), 16 queues -> 8 (2.08 ms).
# HIP reads the variable when the runtime starts: set it before anything loads HIP.  The GPU boxes export it as 4, so a
# default would never apply: a split run raises whatever it finds below 16 (--hw-queues N overrides, 0 leaves it alone).
def _hw_queues_wanted():
    if "--hw-queues" in sys.argv[:-1]:
        return int(sys.argv[sys.argv.index("--hw-queues") + 1])
    return 16 if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--force-split" in sys.argv) else 0


if _hw_queues_wanted() > 0:
    try:
        _have = int(os.environ.get("GPU_MAX_HW_QUEUES", "0"))
    except ValueError:
        _have = 0
    if _have < _hw_queues_wanted() or "--hw-queues" in sys.argv[:-1]:
        os.environ["GPU_MAX_HW_QUEUES"] = str(min(_hw_queues_wanted(), 32))

import numpy as np  # noqa: E402

VALU_FP32_PEAK_TFLOPS = 157.3  # MI355X vector FP32, datasheet
HBM_PEAK_GBS = 8000.0
FLOP_PER_SPHERE_TEST = 17
FLOP_PER_ELLIPSOID_TEST = 91
BYTES_PER_VOXEL_SAMPLE = 4
BAND_ROWS = 16


def build_scene(name: str, scale: float = 1.0):
    from path_tracer_amd import scenes

    def dims(w, h):
        return max(16, int(round(w * scale))), max(16, int(round(h * scale)))

    if name == "c2":
        return scenes.config2(*dims(1920, 1080))
    if name == "c3":
        return scenes.config3(*dims(3840, 2160))
    if name == "c4":
        return scenes.config4(*dims(1920, 1080))
    if name == "c5":
        return scenes.config5(*dims(3840, 2160))
    if name == "c1":
        return scenes.config1(*dims(640, 480))
    raise SystemExit(f"unknown workload {name}")


class Runner:
    """One renderer on this rank's GPU, optionally band-split with a gather to rank 0.

    backend "rccl" (default): pt_gather_bands — the library sends / receives the compact frames over
    RCCL on the frame's own stream and de-interleaves on rank 0; no torch tensor is involved.
    backend "nccl": the compact frame is rendered straight into a torch CUDA tensor on a torch
    stream, gathered with torch.distributed (RCCL) and de-interleaved by the library's kernel.
    backend "gloo" (rehearsal on a box with fewer GPUs than ranks): the compact frame is read back
    and gathered through host memory.
    """

    def __init__(self, scene, rank: int, world: int, device: int, torch_mod=None, dist=None, backend="nccl",
                 split: bool = None, frames_in_flight: int = 2):
        from path_tracer_amd import HipRenderer
        from path_tracer_amd.distributed import BandGather

        self.scene, self.rank, self.world = scene, rank, world
        self.torch, self.dist, self.backend = torch_mod, dist, backend
        self.r = HipRenderer()
        if not self.r.initialize(scene.width, scene.height, device):
            raise SystemExit("renderer initialisation failed")
        self.r.setOption(1, 0)  # PT_OPT_FLOAT_OUTPUT off: the benchmark keeps the rgba8 frame only
        self.set_frames_in_flight(frames_in_flight)
        self.r.setScene(scene, honour_max_depth=True)
        self.pc = scene.push_constants()
        self.bgs, self.frames, self.tstreams = [], [], []
        self.comm = None
        self.split = (world > 1) if split is None else split
        if self.split and backend == "rccl":
            uid = [HipRenderer.commUniqueId() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            self.comm = self.r.commCreate(uid[0], rank, world, device)
            self.r.setRowBands(BAND_ROWS, rank, world)
        elif self.split:
            t = torch_mod
            self.r.setRowBands(BAND_ROWS, rank, world)
            dev = f"cuda:{device}" if backend == "nccl" else "cpu"
            # one set of buffers per frame slot, so that a frame can be drawn while the previous one is gathered
            for slot in range(max(frames_in_flight, 2) if backend == "nccl" else 1):
                bg = BandGather(t, dist, scene.width, scene.height, rank, world, dev, band_rows=BAND_ROWS)
                assert bg.local_rows == self.r.localRows()
                self.bgs.append(bg)
                if backend == "nccl":
                    # streams of our own: the handle of torch's default stream is 0, which the library reads as
                    # "the renderer's own stream", and render, gather and assemble of a frame must stay in order
                    ts = t.cuda.Stream(device=dev)
                    self.tstreams.append(ts)
                    self.r.setSlot(slot, ts.cuda_stream, bg.local.data_ptr(), None)
                    self.frames.append(t.zeros((scene.height, scene.width, 4), dtype=t.uint8, device=dev)
                                       if rank == 0 else None)
            self.last = 0
        self.bg = self.bgs[0] if self.bgs else None

    def set_frames_in_flight(self, n: int):
        self.frames_in_flight = n
        self.r.setOption(6, n)

    def _reduce(self, values, op="sum"):
        if not self.split:
            return values
        t = self.torch
        dev = self.bg.local.device if self.bg is not None else "cpu"
        x = t.tensor(values, dtype=t.float64, device=dev)
        self.dist.all_reduce(x, op=self.dist.ReduceOp.SUM if op == "sum" else self.dist.ReduceOp.MAX)
        return x.cpu().numpy()

    def counters(self):
        return self._reduce(self.r.renderCounted(self.pc).astype(np.float64))

    def step(self, samples: int = 0):
        """One frame: a plain render, or `samples` accumulated sub-pixel samples (gathered once, after the last)."""
        if samples:
            self.r.renderSamples(self.pc, 0, samples)
        else:
            self.r.render(self.pc)
        if self.split:
            if self.backend == "rccl":
                self.r.gatherBands(self.comm, 4)  # on the stream of the frame just drawn
            elif self.backend == "nccl":
                slot = self.last = self.r.lastSlot()
                bg = self.bgs[slot]
                with self.torch.cuda.stream(self.tstreams[slot]):
                    bg.gather()
                if self.rank == 0:  # runs on the stream of the slot just drawn, behind its gather
                    self.r.assembleBands(bg.gathered.data_ptr(), self.frames[slot].data_ptr(), 4, BAND_ROWS, self.world,
                                         bg.stride_rows)
            else:
                part = self.r.readbackRGBA8()
                self.bg.local[:part.shape[0]] = self.torch.from_numpy(part)
                self.bg.gather()
                if self.rank == 0:
                    self.host_frame = self.bg.assemble_host()

    def sync(self):
        self.r.present()
        if self.split and self.backend == "nccl":
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.split:
            self.dist.barrier()

    def timed(self, steps: int, warmup: int):
        """Returns (wall seconds for `steps` frames, max over ranks; HIP-event ms of this rank's region)."""
        for _ in range(warmup):
            self.step()
        self.sync()
        self.barrier()
        self.sync()
        t0 = time.perf_counter()
        self.r.timerBegin()
        for _ in range(steps):
            self.step()
        self.r.timerEnd()
        self.sync()
        self.barrier()
        self.sync()
        dt = time.perf_counter() - t0
        ev_ms = self.r.timerElapsedMs()
        dt = float(self._reduce([dt], "max")[0])
        return dt, ev_ms

    def final_frame(self):
        """Rank 0: the assembled (or whole) rgba8 frame as numpy."""
        if not self.split:
            return self.r.readbackRGBA8()
        if self.rank != 0:
            return None
        if self.backend == "rccl":
            return self.r.readbackAssembled(4)
        return self.frames[self.last].cpu().numpy() if self.backend == "nccl" else self.host_frame

    def close(self):
        self.r.present()
        if self.comm is not None:
            self.r.commDestroy(self.comm)
            self.comm = None
        self.r.shutdown()


def measured_traffic(workload: str):
    """Memory-side bytes per launch of the timed kernel for this workload: FETCH_SIZE and WRITE_SIZE of the rocprofv3
    PMC passes that tools/round_profile.sh ran on THIS build (they need a pass each and cannot be read from inside the
    process); profiles/traffic.json records the hash of the sources it was measured on, and a file from any other
    build is ignored (None).  FETCH_SIZE is doubled, as on gfx950 it tallies 128-byte requests at 64 bytes."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        import __graft_entry__ as g

        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_hash") != g.source_hash():
            return None
        t = doc.get(workload)
        return None if t is None else float(2.0 * t["fetch_bytes"] + t["write_bytes"])
    except (OSError, ValueError, KeyError, ImportError):
        return None


def measured_cache(workload: str):
    """L2 counters per launch of the timed kernel from the same profiling run (or None): requests, hits, misses of
    the L2 (TCC) and the read requests the L1s (TCP) sent to it."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        import __graft_entry__ as g

        with open(path) as f:
            doc = json.load(f)
        if doc.get("source_hash") != g.source_hash():
            return None
        return doc.get(workload, {}).get("l2")
    except (OSError, ValueError, KeyError, ImportError):
        return None


def roofline_for(cnt, kernel_ms: float, traffic=None, cache=None):
    flop = cnt[4] * FLOP_PER_SPHERE_TEST + cnt[5] * FLOP_PER_ELLIPSOID_TEST
    vox_bytes = cnt[6] * BYTES_PER_VOXEL_SAMPLE
    if flop > 0:
        achieved = flop / (kernel_ms * 1e-3) / 1e12
        return {"bound": "valu_fp32", "achieved": round(achieved, 3), "peak": VALU_FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / VALU_FP32_PEAK_TFLOPS, 4), "traffic": traffic,
                "kernel": "pt_render_kernel", "kernel_ms": round(kernel_ms, 4),
                "flop_per_launch": float(flop),
                "note": "17 flop per sphere test + 91 per ellipsoid test, brute force (rays x objects); "
                        "shading, sqrt/pow and hit-path work not counted"}
    achieved = vox_bytes / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "valu_issue", "priced_against": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "kernel": "pt_volume_frame_kernel",
           "kernel_ms": round(kernel_ms, 4), "bytes_per_launch": float(vox_bytes),
           "note": "achieved / peak / frac price the algorithmic bytes (4 per nearest-neighbour voxel sample) against the 8 TB/s of "
                   "HBM, as the contract asks; what bounds the kernel is vector-instruction issue (about 25 instructions per "
                   "sample-wave, SQ counters in profiles/), not a memory level: HBM delivers hbm_GBs (FETCH_SIZE x 2 + WRITE_SIZE: "
                   "the 41.7 MB volume about twice, once per group of L2s, plus the frame), the L1s ask the L2 for l2_requested_GBs"}
    if traffic is not None:
        out["hbm_GBs"] = round(traffic / (kernel_ms * 1e-3) / 1e9, 1)
    if cache:
        out["l2"] = dict(cache)
        if cache.get("tcp_read_req"):
            # every L1 miss asks the L2 for one 128-byte line (64-byte requests count as one too: an upper bound)
            out["l2_requested_GBs"] = round(cache["tcp_read_req"] * 128.0 / (kernel_ms * 1e-3) / 1e9, 1)
        if cache.get("req"):
            out["l2_hit_rate"] = round(cache.get("hit", 0.0) / cache["req"], 4)
    return out


def cpu_baseline(scene, budget_s: float = 20.0):
    """The oracle on this host's cores, on a row-subsampled frame sized for ~budget_s.  A one-GPU box is a slice of a
    larger machine: os.cpu_count() shows the whole machine, the slice's share is what OMP_NUM_THREADS states (16 on
    the GPU boxes), and the oracle uses that many threads (all cores where no share is stated)."""
    from oracle import oracle_py

    host_cpus = os.cpu_count() or 1
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    threads = min(host_cpus, share) if share > 0 else host_cpus
    # probe 1/16 of the rows to size the sample
    H = scene.height
    probe_rows = max(H // 16, 1)
    y0 = (H - probe_rows) // 2
    t0 = time.perf_counter()
    _, _, cnt = oracle_py.render(scene, rows=(y0, y0 + probe_rows), nthreads=threads, want_f32=False)
    dt = time.perf_counter() - t0
    est_full = dt * H / probe_rows
    if est_full <= budget_s:
        t0 = time.perf_counter()
        _, _, cnt = oracle_py.render(scene, nthreads=threads, want_f32=False)
        dt = time.perf_counter() - t0
        sample = f"full {scene.width}x{scene.height} frame, 1 frame"
    else:
        sample = f"rows {y0}..{y0 + probe_rows} of {H} ({scene.width} px wide), 1 pass"
    return {"value": round(float(cnt[0]) / dt / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": sample, "seconds": round(dt, 3), "host_cpu_count": host_cpus,
            "cpu_share": share if share > 0 else None,
            "note": "CPU oracle: float32 restatement of the reference shader (the reference has no CPU renderer)"}


def dynamic_loop(device: int, frames: int = 120):
    """The reference's main loop on C2's objects at 1080p (reference src/main_vulkan.cpp:340-376): per frame the camera
    moves along the orbit, every ellipsoid gets the rotation about (2,1,0) by theta_(k+1), updateScene uploads the
    records and render draws.  Timed against the same camera path drawn WITHOUT the per-frame updateScene (ellipsoids
    as the last update left them: the same amount of work per frame up to the ellipsoids' orientation), in two modes:
    frames queued as the library takes them (two in flight; present() once at the end) and present() after every
    frame as the reference's loop does.  The records and push constants of every frame are built before the clock
    starts: what is timed is updateScene + render (+ present)."""
    import math

    from path_tracer_amd import HipRenderer, scenes

    sc = scenes.config2()
    r = HipRenderer()
    if not r.initialize(sc.width, sc.height, device):
        raise SystemExit("renderer initialisation failed")
    r.setOption(1, 0)
    r.setScene(sc, honour_max_depth=True)
    pcs, ells = [], []
    for k in range(frames):
        theta = k / 180.0
        pc = sc.push_constants().copy()
        pc["cameraPos"] = (np.float32(2.0 + 3.0 * math.cos(theta)), 1.5, np.float32(6.0 + 3.0 * math.sin(theta)))
        pcs.append(pc)
        e = sc.ellipsoids.copy()
        e["rotation"] = scenes.axis_angle_quat((2.0, 1.0, 0.0), (k + 1) / 180.0)
        ells.append(e)

    def run(update: bool, present_each: bool, lag: int = 0):
        r.setOption(8, lag)  # PT_OPT_PRESENT_LAG
        for k in range(4):
            r.render(pcs[k])
        r.present()
        t0 = time.perf_counter()
        r.timerBegin()
        for k in range(frames):
            if update:
                r.updateScene(sc.spheres, ells[k], sc.materials, sc.lights, sc.volumes)
            r.render(pcs[k])
            if present_each:
                r.present()
        r.timerEnd()
        r.present()
        return (time.perf_counter() - t0) / frames * 1e3, r.timerElapsedMs() / frames

    out = {"frames": frames, "workload": "c2 objects, 1080p, depth 8, orbit camera + all 16 ellipsoids re-oriented per frame"}
    for mode, present_each in (("frames_in_flight_2", False), ("present_every_frame", True)):
        run(True, present_each)  # warm: staging blocks, first copies
        dyn_wall, dyn_ev = run(True, present_each)
        sta_wall, sta_ev = run(False, present_each)
        out[mode] = {"update_and_render_ms_per_frame": round(dyn_wall, 4), "render_only_ms_per_frame": round(sta_wall, 4),
                     "ratio": round(dyn_wall / sta_wall, 3), "update_and_render_event_ms": round(dyn_ev, 4),
                     "render_only_event_ms": round(sta_ev, 4)}
    # the same loop with PT_OPT_PRESENT_LAG = 1: present() waits for the frames before the one just queued
    run(True, True, 1)
    dyn_wall, dyn_ev = run(True, True, 1)
    sta_wall, sta_ev = run(False, True, 1)
    out["present_every_frame_lag_1"] = {
        "update_and_render_ms_per_frame": round(dyn_wall, 4), "render_only_ms_per_frame": round(sta_wall, 4),
        "ratio": round(dyn_wall / sta_wall, 3), "update_and_render_event_ms": round(dyn_ev, 4),
        "render_only_event_ms": round(sta_ev, 4),
        "note": "PT_OPT_PRESENT_LAG = 1 (an extension): present() after render() waits for the earlier frames only and "
                "shows the frame before the one just queued"}
    r.setOption(8, 0)
    r.shutdown()
    out["note"] = ("updateScene rebuilds the host copy of the scene and each frame slot refreshes its device copy on its own "
                   "stream (pinned staging, hipMemcpyAsync) in front of its next frame: no wait for the frames in flight")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 200 frames: with frames in flight the timed region holds one pipeline fill and one tail whatever K is
    # (0.50 ms per frame at K = 10, 0.46 at 20, 0.42 from 100 on), and a region of 9 ms is short for a wall clock
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c2", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--scale", type=float, default=1.0, help="resolution scale (testing only; 1.0 = the named config)")
    ap.add_argument("--no-extras", action="store_true", help="skip the C3/C4/C5 side measurements")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    ap.add_argument("--backend", default="rccl", choices=["rccl", "nccl", "gloo"],
                    help="gather for --gpus > 1: rccl = the library's pt_gather_bands (RCCL behind the C-ABI; torch.distributed "
                         "only for rendezvous and barriers, on gloo), nccl = torch.distributed.gather on CUDA tensors, "
                         "gloo = host-memory rehearsal")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank renders on GPU 0")
    ap.add_argument("--check", action="store_true", help="rank 0 verifies the final frame against a single-GPU render")
    ap.add_argument("--frames-in-flight", type=int, default=None, choices=range(1, 9),
                    help="frames queued at once, each on a stream of its own (default: 2 on one GPU, as the reference's "
                         "MAX_FRAMES_IN_FLIGHT; 8 when the frame is split over several GPUs, where a rank's share is short "
                         "against its longest chain of dependent rays); 1: one at a time")
    ap.add_argument("--hw-queues", type=int, default=None,
                    help="GPU_MAX_HW_QUEUES for this process (read before HIP starts; default: 16 when the frame is split, "
                         "else whatever the environment says)")
    ap.add_argument("--force-split", action="store_true",
                    help="testing: take the band-split + gather path even with one rank (exercises the RCCL plumbing on one GPU)")
    args = ap.parse_args()

    # Libraries (RCCL's version banner, for one) write to stdout; the JSON line must be the only thing there.
    sys.stdout.flush()
    stdout_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch_mod = dist = None
    if world > 1 or args.force_split:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        import torch
        import torch.distributed as dist_mod

        if args.same_device:
            local_rank = 0
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist_mod.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        else:  # rccl: the library talks to RCCL itself; the process group is the control plane only
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the ranks meet on the loopback interface
            dist_mod.init_process_group("gloo")
        torch_mod, dist = torch, dist_mod
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    n_gpus = world

    scene = build_scene(args.workload, args.scale)
    fif = args.frames_in_flight or (8 if (world > 1 or args.force_split) else 2)
    run = Runner(scene, rank, world, local_rank, torch_mod, dist, args.backend, split=(world > 1 or args.force_split),
                 frames_in_flight=fif)
    cnt = run.counters()
    # The headline: K frames in the library's production mode (two frames in flight like the reference's
    # MAX_FRAMES_IN_FLIGHT unless --frames-in-flight 1).
    dt, ev_ms = run.timed(args.steps, args.warmup)
    # The roofline leg: the same K frames one at a time, so that a launch's duration is that of a kernel running alone
    # (and is what a rocprofv3 trace of `--frames-in-flight 1` shows).
    if fif >= 2:
        run.set_frames_in_flight(1)
        dt1, ev1_ms = run.timed(args.steps, 1)
        run.set_frames_in_flight(fif)
    else:
        dt1, ev1_ms = dt, ev_ms
    # Steady state: with frames in flight the K timed frames hold one frame's start-up and tail, so ms_per_step falls
    # as K grows; the rate between frame 20 and frame 120 of one run is what a long sequence sees per frame.
    steady = None
    if fif >= 2 and args.scale == 1.0:
        d_a, _ = run.timed(20, 2)
        d_b, _ = run.timed(120, 2)
        steady = max(d_b - d_a, 0.0) / 100.0
    check = None
    if (args.check or world > 1) and (world > 1 or args.force_split):
        frame = run.final_frame()
        if rank == 0:
            solo = Runner(scene, 0, 1, local_rank, frames_in_flight=1)
            solo.step()
            solo.sync()
            check = bool(np.array_equal(frame, solo.final_frame()))
            solo.close()
    ms_per_step = dt / args.steps * 1e3
    kernel_ms = ev1_ms / args.steps
    dev_name, cus = run.r.deviceInfo()
    launch = run.r.lastLaunch()
    side_by_side = run.r.streamConcurrency() if fif >= 2 else 1  # what the device gives the slots' streams
    run.close()

    result = None
    if rank == 0:
        rays = float(cnt[0])
        if args.workload == "c4":
            metric, unit, value = "Mvoxel-samples/s (walnut volume march, 1080p)", "Msamples/s", cnt[6] / (dt / args.steps) / 1e6
        else:
            metric, unit, value = "Mrays/s (all bounces) at 1080p depth-8", "Mrays/s", rays / (dt / args.steps) / 1e6
        if args.workload == "c3":
            metric = "Mrays/s (all bounces) at 4K, 4096 spheres, depth-8"
        result = {
            "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "strong" if n_gpus > 1 else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {scene.width}x{scene.height}, {len(scene.spheres)} spheres, "
                                   f"{len(scene.ellipsoids)} ellipsoids, {len(scene.lights)} lights, "
                                   f"{len(scene.volumes)} volumes, depth {scene.max_depth}",
                       "rays_per_frame": rays, "split": f"interleaved {BAND_ROWS}-row bands, RCCL gather to rank 0"
                       if n_gpus > 1 else "single GPU", "device": dev_name, "compute_units": cus,
                       "launch": {"blocks": launch[0], "threads": launch[1], "lds_bytes": launch[2]}},
            "roofline": roofline_for(cnt / max(n_gpus, 1), kernel_ms,
                                     measured_traffic(args.workload) if args.scale == 1.0 else None,
                                     measured_cache(args.workload) if args.scale == 1.0 else None) if n_gpus == 1 else None,
        }
        result["frames_in_flight"] = fif
        result["streams_side_by_side"] = side_by_side
        result["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
        if n_gpus == 1:
            # the headline mode itself: the algorithmic FLOP of the K timed frames over the HIP-event time of the region
            hl = roofline_for(cnt, ev_ms / args.steps)
            hl.pop("traffic", None)
            hl["note"] = (f"headline mode, {fif} frames in flight: FLOP of the {args.steps} timed frames / HIP-event time of the "
                          "whole region (kernels of consecutive frames overlap, so this is not one launch's duration)")
            result["roofline_frames_in_flight"] = hl
        if steady is not None:
            unit_count = float(cnt[6]) if args.workload == "c4" else float(cnt[0])
            result["steady_state"] = {"ms_per_step": round(steady * 1e3, 4), "value": round(unit_count / steady / 1e6, 2),
                                      "unit": unit, "note": "(time of 120 frames - time of 20 frames) / 100, same mode as the headline"}
        if fif >= 2:
            unit_count = float(cnt[6]) if args.workload == "c4" else rays
            result["one_frame_at_a_time"] = {
                "ms_per_step": round(dt1 / args.steps * 1e3, 4), "value": round(unit_count / (dt1 / args.steps) / 1e6, 2),
                "unit": unit,
                "note": "same frames with PT_OPT_FRAMES_IN_FLIGHT = 1; the roofline object and the rocprofv3 summaries "
                        "describe this mode (a kernel running alone); the headline has frames_in_flight frames queued at "
                        "once (2 is the library's default, as the reference's MAX_FRAMES_IN_FLIGHT; 8 with a band split)"}
        if check is not None:
            result["config"]["gathered_frame_equals_single_gpu_frame"] = check
        if n_gpus > 1:
            result["roofline"] = roofline_for(cnt / n_gpus, kernel_ms)
            result["roofline"]["note"] += "; per-rank share, one frame at a time, event time includes the gather"

    # side measurements (single GPU only): C3 and C4 frame times
    if n_gpus == 1 and not args.no_extras and args.scale == 1.0:
        extras = {}
        for name, steps in (("c3", 3), ("c4", 10)):
            if name == args.workload:
                continue
            sc = build_scene(name)
            rr = Runner(sc, 0, 1, local_rank, frames_in_flight=1)
            c = rr.counters()
            d, e = rr.timed(steps, 1)
            rr.set_frames_in_flight(2)
            d2, _ = rr.timed(steps, 2)
            rr.close()
            extras[name] = {"ms_per_frame": round(d / steps * 1e3, 4), "ms_per_frame_2_in_flight": round(d2 / steps * 1e3, 4),
                            "Mrays_per_s": round(c[0] / (d / steps) / 1e6, 2),
                            "rays_per_frame": float(c[0]),
                            "roofline": roofline_for(c, e / steps, measured_traffic(name), measured_cache(name))}
            if name == "c4":
                extras[name]["Msamples_per_s"] = round(c[6] / (d / steps) / 1e6, 2)
                # the trilinear quality mode (no counterpart in the reference): 8 voxel bytes per sample
                rr = Runner(sc, 0, 1, local_rank, frames_in_flight=1)
                rr.r.setOption(5, 1)  # PT_OPT_VOLUME_FILTER
                ct = rr.counters()
                dt_, et_ = rr.timed(steps, 1)
                rr.close()
                extras[name]["trilinear"] = {
                    "ms_per_frame": round(dt_ / steps * 1e3, 4), "Msamples_per_s": round(ct[6] / (dt_ / steps) / 1e6, 2),
                    "requested_GBs": round(ct[6] * 8 / (et_ / steps * 1e-3) / 1e9, 1),
                    "note": "PT_OPT_VOLUME_FILTER = 1: 8 voxel bytes per sample (four 16-bit fetches, the two cells along x together), bricked volume"}
        if args.workload != "c5":
            # 64 samples per pixel at 4K (C2's objects + the volume), mean on the device: all samples in one launch
            # ((tile, sample) tickets, one plane per sample, summed in sample order), and one launch per sample beside it
            sc = build_scene("c5")
            rr = Runner(sc, 0, 1, local_rank)
            c = rr.counters()  # sample 0 (the reference's sample)
            spp = 64
            took = {}
            for batch in (0, 1):
                rr.r.setOption(9, batch)  # PT_OPT_SAMPLE_BATCH
                rr.r.renderSamples(rr.pc, 0, spp)  # untimed: the sample planes are allocated here
                rr.sync()
                t0 = time.perf_counter()
                rr.r.renderSamples(rr.pc, 0, spp)
                rr.sync()
                took[batch] = time.perf_counter() - t0
            rr.close()
            d = took[0]
            extras["c5"] = {"spp": spp, "ms_per_frame": round(d * 1e3, 3), "ms_per_sample": round(d / spp * 1e3, 4),
                            "ms_per_frame_one_launch_per_sample": round(took[1] * 1e3, 3),
                            "rays_sample0": float(c[0]), "Mrays_per_s": round(c[0] * spp / d / 1e6, 2),
                            "note": "sub-pixel accumulation has no counterpart in the reference (one corner sample per "
                                    "pixel); rays per sample taken from sample 0; PT_OPT_SAMPLE_BATCH = 0: the 64 samples "
                                    "are one launch, 8.5 GB of sample planes"}
        result["workloads"] = extras
        result["dynamic_loop"] = dynamic_loop(local_rank)

    # N > 1: the 4K / 4096-sphere frame (the configuration the multi-GPU metric is quoted on), the volume frame and the
    # 64-sample 4K frame beside the default workload, same band split and gather; every rank takes part
    if n_gpus > 1 and not args.no_extras and args.scale == 1.0:
        try:
            extras = {}
            for name, frames in (("c3", 16), ("c4", 32)):
                if name == args.workload:
                    continue
                rr = Runner(build_scene(name), rank, world, local_rank, torch_mod, dist, args.backend, frames_in_flight=fif)
                c = rr.counters()
                d, e = rr.timed(frames, fif)
                rr.close()
                extras[name] = {"ms_per_frame": round(d / frames * 1e3, 4), "Mrays_per_s": round(c[0] / (d / frames) / 1e6, 2),
                                "rays_per_frame": float(c[0]), "n_gpus": n_gpus, "frames": frames, "frames_in_flight": fif,
                                "roofline": roofline_for(c / n_gpus, e / frames)}
                if name == "c4":
                    extras[name]["Msamples_per_s"] = round(c[6] / (d / frames) / 1e6, 2)
            if args.workload != "c5":
                # every rank accumulates its bands over 64 samples, one gather after the last sample
                spp = 64
                rr = Runner(build_scene("c5"), rank, world, local_rank, torch_mod, dist, args.backend, frames_in_flight=1)
                c = rr.counters()
                rr.step(samples=spp)  # untimed: the sample planes are allocated here
                rr.sync(); rr.barrier(); rr.sync()
                t0 = time.perf_counter()
                rr.step(samples=spp)
                rr.sync(); rr.barrier(); rr.sync()
                d = float(rr._reduce([time.perf_counter() - t0], "max")[0])
                rr.close()
                extras["c5"] = {"spp": spp, "ms_per_frame": round(d * 1e3, 3), "ms_per_sample": round(d / spp * 1e3, 4),
                                "rays_sample0": float(c[0]), "Mrays_per_s": round(c[0] * spp / d / 1e6, 2), "n_gpus": n_gpus}
            if rank == 0:
                result["workloads"] = extras
        except Exception as exc:  # the same on every rank (a bug here must not cost the headline line)
            if rank == 0:
                result["workloads_error"] = repr(exc)

    if rank == 0 and n_gpus == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(scene)
    elif rank == 0:
        result["cpu_baseline"] = None

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(stdout_fd, 1)
    os.close(stdout_fd)
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
