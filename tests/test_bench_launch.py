"""bench.py's own launcher (BASELINE configs[4]): `python bench.py --gpus N` must be runnable exactly like
`--gpus 1` -- started without a torchrun environment it starts its N ranks itself (torch.distributed.run on 127.0.0.1),
and refuses with a plain sentence when the box has fewer GPUs. CPU part: the decision and the command line. GPU part:
the self-launched N = 1 run with the RCCL gather exercised, parsed like the driver parses it."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bench():
    import importlib
    return importlib.import_module("bench")


def test_launch_plan_decisions():
    b = _bench()
    never = lambda: (_ for _ in ()).throw(AssertionError("GPU count must not be taken here"))  # noqa: E731
    # the driver's N = 1 form: stay in this process, do not even count GPUs
    assert b.launch_plan(1, False, {}, never) == ("run", None)
    # a rank started by torchrun (the driver's N > 1 form): stay, whatever N
    assert b.launch_plan(8, False, {"WORLD_SIZE": "8", "RANK": "3"}, never) == ("run", None)
    assert b.launch_plan(1, True, {"WORLD_SIZE": "1"}, never) == ("run", None)
    # no torchrun environment and N > 1: start the ranks ourselves, if the GPUs are there
    assert b.launch_plan(8, False, {}, lambda: 8) == ("spawn", None)
    assert b.launch_plan(2, False, {}, lambda: 8) == ("spawn", None)
    assert b.launch_plan(1, True, {}, lambda: 1) == ("spawn", None)
    what, why = b.launch_plan(2, False, {}, lambda: 1)
    assert what == "refuse" and why == "2 GPUs requested, 1 visible"
    what, why = b.launch_plan(8, False, {}, lambda: 0)
    assert what == "refuse" and why == "8 GPUs requested, 0 visible"
    # a launcher that started a different number of ranks than --gpus says is an error, not a silent relabel
    what, why = b.launch_plan(8, False, {"WORLD_SIZE": "4"}, never)
    assert what == "refuse" and "WORLD_SIZE=4" in why
    assert b.launch_plan(0, False, {}, never)[0] == "refuse"


def test_launcher_command_line_is_the_drivers():
    b = _bench()
    argv = b.launcher_argv(8, 29511, "/x/bench.py", ["--gpus", "8", "--steps", "20", "--warmup", "3"], python="/usr/bin/python3")
    assert argv == ["/usr/bin/python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                    "--master-addr", "127.0.0.1", "--master-port", "29511", "/x/bench.py",
                    "--gpus", "8", "--steps", "20", "--warmup", "3"]
    p = b.free_port()
    assert 1024 < p < 65536


def test_direct_invocation_without_gpus_refuses_in_words():
    """Here (no GPU): `python bench.py --gpus 2` must say how many GPUs it found, not print a launcher hint or a traceback."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import pirip_amd
    if pirip_amd.device_count() >= 2:
        pytest.skip("box has 2+ GPUs")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0
    assert "2 GPUs requested" in r.stderr and "visible" in r.stderr
    assert "Traceback" not in r.stderr and "torch.distributed.run" not in r.stderr


@pytest.mark.gpu
def test_self_launched_gather_run_prints_one_line():
    """The N > 1 code path end to end on one GPU: bench.py re-execs itself under torch.distributed.run (1 rank), RCCL
    world of 1, packed message written in place, asynchronous gather every step, the line parsed like the driver does."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--exercise-gather", "--streams", "96",
                        "--samples", "240000", "--steps", "2", "--warmup", "1", "--no-extra", "--no-cpu-baseline",
                        "--check-streams", "8"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "starting -m torch.distributed.run" in r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1
    assert out["rccl"]["world_size"] == 1 and out["rccl"]["backend"] == "nccl"
    assert len(out["per_gpu"]) == 1 and out["per_gpu"][0]["kernel_ms"] > 0
    assert out["gather_check"]["rank0_echo"] is True and out["gather_check"]["frames_per_rank"][0] > 0
    assert out["bit_errors_vs_cpu_ref"] == 0
    assert out["roofline"]["achieved"] > 0 and out["value"] > 0


@pytest.mark.gpu
def test_asking_for_more_gpus_than_visible_refuses():
    import pirip_amd
    n = pirip_amd.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode != 0 and f"{n} GPUs requested, {n - 1} visible" in r.stderr


@pytest.mark.gpu
def test_cpp_host_with_a_real_rccl_communicator_at_world_one(tmp_path):
    """The product's C++ multi-GPU host (pirip_amd/tools/mgpu_receiver.cpp: C-ABI demodulator writing the packed message in place,
    pirip_hip_rccl_init's file rendezvous, pirip_hip_gather_bits on its own stream) started by tools/launch_mgpu.sh on the one GPU
    there is: ncclCommInitRank with a real unique id, the gathered bits are the transmitted test frames."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PIRIP_RCCL_SESSION")}
    env["PIRIP_MGPU"] = "cpp"
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "launch_mgpu.sh"), "1", "--streams", "384", "--samples", "240000", "--steps", "3",
                        "--warmup", "1"], capture_output=True, text=True, env=env, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["streams_per_gpu"] == 384 and out["value"] > 0
    assert out["frames_gathered_per_step"] >= 384 * 199 and out["test_bits_checked"] > 5000 and out["bit_errors_vs_tx"] == 0
