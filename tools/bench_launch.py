#!/usr/bin/env python3
"""tools/bench_launch.py -- how a bench.py process decides what it is (a rank, a plain single-GPU run, or the launcher of N ranks) and
how the CPU leg of a multi-rank job is handed from whoever timed it to rank 0. No GPU, no torch: bench.py imports these names and
tests/test_bench_launch.py exercises them on the CPU (BASELINE configs[4])."""
import json
import os
import sys
import time


def visible_gpus():
    """HIP devices this process can open, counted through the product library (hipGetDeviceCount): no torch import, no
    context created in a process that is about to exec. 0 when there is no device or no driver."""
    try:
        import pirip_amd
        return max(int(pirip_amd.device_count()), 0)
    except Exception:
        return 0


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_plan(gpus, exercise_gather, environ, visible):
    """What a bench.py process started with these arguments has to do before anything else:
      ("run", None)      -- it is a rank (torchrun environment present) or a plain single-GPU run: go on in this process
      ("spawn", None)    -- no torchrun environment, but N > 1 ranks (or the gather path at N = 1) are wanted: re-exec
                            under torch.distributed.run with N local ranks
      ("refuse", reason) -- the request cannot be met on this box
    `visible` is a callable so that the GPU count is only taken when it matters."""
    if gpus < 1:
        return "refuse", f"--gpus {gpus}: need at least 1"
    if "WORLD_SIZE" in environ:
        world = int(environ["WORLD_SIZE"])
        if world != gpus:
            return "refuse", f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks"
        return "run", None
    if gpus == 1 and not exercise_gather:
        return "run", None
    v = visible()
    if v < gpus:
        return "refuse", f"{gpus} GPUs requested, {v} visible"
    return "spawn", None


def launcher_argv(gpus, port, script, script_args, python=None):
    """The command line the driver itself uses for N > 1 (one rank per GPU over RCCL, rendezvous on 127.0.0.1)."""
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(script_args)


CPU_ENV = "PIRIP_BENCH_CPU_BASELINE"          # the CPU leg's JSON, handed from the self-launcher to rank 0


def cpu_flag_path(environ, ppid=None):
    """Where rank 0 of a launcher-started job tells the other local ranks that the CPU leg is over. All ranks of one
    torch.distributed.run agent share their parent process and the rendezvous port: that pair names the job."""
    ppid = os.getppid() if ppid is None else ppid
    return os.path.join(environ.get("TMPDIR", "/tmp"), f"pirip_bench_cpu_{ppid}_{environ.get('MASTER_PORT', '0')}.json")


def cpu_leg_plan(rank, world, environ, disabled):
    """The CPU leg ("the reference CPU fsk_demod timed on the node's own host cores ... in the same run") at EVERY N:
      ("skip", None)     -- switched off (--no-cpu-baseline, or a noisy batch whose buffer only exists on the device)
      ("env", text)      -- this job was started by bench.py's own launcher, which timed the oracle BEFORE it became
                            torch.distributed.run and left the JSON in the environment: rank 0 quotes it, nobody waits
      ("measure", path)  -- rank 0 of a job the driver launched itself: time the oracle now, before this process touches the
                            GPU, then create `path` (None at world 1: nobody is waiting)
      ("wait", path)     -- any other rank: do NOTHING (no synthesis, no torch import, no GPU) until `path` exists, so that
                            the host cores belong to the CPU leg while it runs"""
    if disabled:
        return "skip", None
    if CPU_ENV in environ:
        return ("env", environ[CPU_ENV]) if rank == 0 else ("skip", None)
    if rank == 0:
        return "measure", (cpu_flag_path(environ) if world > 1 else None)
    return "wait", cpu_flag_path(environ)


def wait_for_cpu_leg(path, limit_s):
    """Sleep until rank 0 has published the CPU leg (or the limit passes: a rank 0 that died must not hang the job here --
    the rendezvous that follows reports it)."""
    t0 = time.perf_counter()
    while not os.path.exists(path) and time.perf_counter() - t0 < limit_s:
        time.sleep(0.05)
    return time.perf_counter() - t0


def clear_stale_cpu_leg(path):
    """A flag file left behind by a crashed job with the same parent pid and port would let waiting ranks skip the wait: rank 0
    removes it before it starts the CPU leg (the other ranks only look once rank 0 has had that chance: they were started together)."""
    try:
        os.unlink(path)
    except OSError:
        pass


def publish_cpu_leg(path, res):
    tmp = path + ".tmp"
    with open(tmp, "w") as f:
        json.dump(res, f)
    os.replace(tmp, path)

