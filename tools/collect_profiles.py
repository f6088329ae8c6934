#!/usr/bin/env python3
"""Copy the judged summaries of one tools/round_profile.sh run into profiles/.

    tools/collect_profiles.py gpurun_out/<tag> r1      # writes profiles/r1_*.{txt,json}

Outputs (all text, small):
  <round>_bench.json          the bench.py line of that run
  <round>_kernel_trace.txt    rocprofv3 --kernel-trace --stats of the same bench command: per-kernel calls / total / average
  <round>_pmc.txt             per-dispatch PMC sums for workloads c2, c3, c4 (SQ pass, FETCH_SIZE pass, WRITE_SIZE pass)
  <round>_two_in_flight_trace.txt  begin / end stamps of the timed kernel in the default mode (two frames in flight)
  <round>_traffic.json        FETCH_SIZE / WRITE_SIZE per launch of the timed kernel, in bytes (rocprofv3 reports KiB)
"""
import glob
import io
import json
import os
import sqlite3
import sys
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import pmc_summary  # noqa: E402


# the timed (uninstrumented) kernel of each workload: the general frame kernel, or the kernel for frames without solids
TIMED_KERNELS = ("pt_render_kernel<0,", "pt_volume_frame_kernel")


def kernel_counter(db, counter, kernels=TIMED_KERNELS):
    con = sqlite3.connect(db)
    acc, disp = 0.0, set()
    for k, did, cname, val in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        if any(name in k for name in kernels) and cname == counter:
            acc += val
            disp.add(did)
    return acc / max(len(disp), 1)


def two_in_flight(db, path):
    """Begin / end stamps of the timed kernel in the default mode (frames in flight): how the launches overlap."""
    con = sqlite3.connect(db)
    rows = list(con.execute("select queue_id, start, end from kernels where name like '%pt_render_kernel<0%' order by start"))
    if not rows:
        return
    t0 = rows[0][1]
    # the first run of alternating queues is the timed region of the headline (frames go round two slots)
    run = []
    for q, s, e in rows:
        if run and q == run[-1][0]:
            if len(run) >= 8:
                run.pop()  # that launch already belongs to what follows (frames drawn one at a time)
                break
            run = []
        run.append((q, s, e))
    both = 0
    for i in range(1, len(run)):
        both += max(0, min(run[i - 1][2], run[i][2]) - run[i][1])
    span = run[-1][2] - run[0][1]
    with open(path, "w") as f:
        f.write("# rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-extras (default: two frames in flight)\n")
        f.write("# pt_render_kernel<0,...> launches of the first region whose frames alternate between two hardware queues; microseconds from its first launch\n")
        f.write(f"# {len(run)} launches in {span / 1e3:.1f} us = {span / 1e3 / len(run):.1f} us per frame; two kernels run side by side "
                f"{both / span:.2f} of that time; a kernel takes {sum(e - s for _, s, e in run) / len(run) / 1e3:.1f} us from begin to end\n")
        f.write("# queue   begin     end  duration\n")
        for q, s, e in run:
            f.write(f"{q:7d} {(s - run[0][1]) / 1e3:7.1f} {(e - run[0][1]) / 1e3:7.1f} {(e - s) / 1e3:9.1f}\n")


def main():
    src, rnd = sys.argv[1], sys.argv[2]
    out = os.path.join(os.path.dirname(HERE), "profiles")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(src, "bench.json")) as f:
        line = [l for l in f.read().splitlines() if l.startswith("{")][-1]
    with open(os.path.join(out, f"{rnd}_bench.json"), "w") as f:
        f.write(line + "\n")
    buf = io.StringIO()
    with redirect_stdout(buf):
        sys.argv = ["pmc_summary", os.path.join(src, "trace_bench")]
        pmc_summary.main()
    with open(os.path.join(out, f"{rnd}_kernel_trace.txt"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-extras --frames-in-flight 1\n")
        f.write("# (default workload c2: 12 launches of pt_render_kernel<0,...> = 2 warm-up + 10 timed (c4: pt_volume_frame_kernel); <1,...> and <2,...> are the\n")
        f.write("#  instrumented, untimed counting and phase-timing launches)\n")
        f.write(buf.getvalue())
        for wl in ("c3", "c4"):
            d = os.path.join(src, f"trace_{wl}")
            if os.path.isdir(d):
                b2 = io.StringIO()
                with redirect_stdout(b2):
                    sys.argv = ["pmc_summary", d]
                    pmc_summary.main()
                f.write(f"# same, --workload {wl} --steps 3 --warmup 1\n" + b2.getvalue())
    tdb = glob.glob(os.path.join(src, "trace_two_in_flight", "*_results.db"))
    if tdb:
        two_in_flight(tdb[0], os.path.join(out, f"{rnd}_two_in_flight_trace.txt"))
    buf = io.StringIO()
    with redirect_stdout(buf):
        for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
            if os.path.isdir(d):
                sys.argv = ["pmc_summary", d]
                print(f"#### {os.path.basename(d)}")
                pmc_summary.main()
    text = "\n".join(l for l in buf.getvalue().splitlines()
                     if "rocclr" not in l and not (l.startswith("   ") and float(l.split()[-1]) < 1e4))
    with open(os.path.join(out, f"{rnd}_pmc.txt"), "w") as f:
        f.write("# one rocprofv3 --pmc pass per block; bench.py --workload <wl> --steps 3 --warmup 1 --no-extras --no-cpu --frames-in-flight 1\n")
        f.write("# FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; <1,...> and <2,...> are the instrumented (untimed) launch\n")
        f.write(text + "\n")
    traffic = {}
    for wl in ("c2", "c3", "c4"):
        fdb = glob.glob(os.path.join(src, f"pmc_fetch_{wl}", "*_results.db"))
        wdb = glob.glob(os.path.join(src, f"pmc_write_{wl}", "*_results.db"))
        if fdb and wdb:
            traffic[wl] = {"fetch_bytes": kernel_counter(fdb[0], "FETCH_SIZE") * 1024.0,
                           "write_bytes": kernel_counter(wdb[0], "WRITE_SIZE") * 1024.0}
            ldb = glob.glob(os.path.join(src, f"pmc_l2_{wl}", "*_results.db"))
            if ldb:
                traffic[wl]["l2"] = {"req": kernel_counter(ldb[0], "TCC_REQ_sum"), "hit": kernel_counter(ldb[0], "TCC_HIT_sum"),
                                     "miss": kernel_counter(ldb[0], "TCC_MISS_sum"),
                                     "tcp_read_req": kernel_counter(ldb[0], "TCP_TCC_READ_REQ_sum")}
    # the build these passes ran on: bench.py only quotes a traffic.json measured on the build it is running
    try:
        with open(os.path.join(src, "source_hash.txt")) as f:
            traffic["source_hash"] = f.read().strip()
    except OSError:
        traffic["source_hash"] = None
    traffic["unit"] = "bytes per launch of the timed kernel (pt_render_kernel<0,...>, or pt_volume_frame_kernel for c4); FETCH_SIZE and WRITE_SIZE as rocprofv3 reports them (KiB x 1024), uncorrected"
    for name in (f"{rnd}_traffic.json", "traffic.json"):  # traffic.json = the latest, read by bench.py
        with open(os.path.join(out, name), "w") as f:
            json.dump(traffic, f, indent=1)
            f.write("\n")
    print("wrote", sorted(os.listdir(out)))


if __name__ == "__main__":
    main()
