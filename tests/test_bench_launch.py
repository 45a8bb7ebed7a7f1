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


def test_cpu_leg_plan_covers_every_world_size():
    """north_star: the CPU figure "in the same run" at 1, 2, 4 and 8 GPUs -- every rank-0 line must carry cpu_baseline."""
    b = _bench()
    # N = 1, plain run: measure, nobody to tell
    assert b.cpu_leg_plan(0, 1, {}, False) == ("measure", None)
    # ranks the driver started itself (torchrun environment, no hand-over variable): rank 0 measures and publishes, others wait
    env = {"WORLD_SIZE": "8", "MASTER_PORT": "29512"}
    how0, path0 = b.cpu_leg_plan(0, 8, env, False)
    how5, path5 = b.cpu_leg_plan(5, 8, env, False)
    assert (how0, how5) == ("measure", "wait") and path0 == path5 and "29512" in path0 and str(os.getppid()) in path0
    # ranks started by bench.py's own launcher: the JSON is in the environment, nobody measures or waits
    env[b.CPU_ENV] = '{"value": 1.0}'
    assert b.cpu_leg_plan(0, 8, env, False) == ("env", '{"value": 1.0}')
    assert b.cpu_leg_plan(3, 8, env, False) == ("skip", None)
    assert b.cpu_leg_plan(0, 8, {}, True) == ("skip", None)


def test_cpu_baseline_only_prints_the_object():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-seconds", "0.2", "--samples", "120000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    o = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert o["value"] > 0 and o["cores"] >= 1 and o["kind"] == "port" and o["single_core_value"] > 0


def test_cpu_leg_hand_over_between_two_ranks():
    """Two ranks as the driver's torchrun would start them (same parent, same rendezvous port, no hand-over variable): rank 0 times the
    oracle, rank 1 does nothing until rank 0 has published, then both go on. --cpu-leg-only stops them before the GPU part."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", _bench().CPU_ENV)}
    port = str(_bench().free_port())
    common = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--cpu-leg-only", "--cpu-seconds", "1", "--samples", "120000"]
    ps = [subprocess.Popen(common, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port))
          for r in (1, 0)]
    outs = []
    for p in ps:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-2000:]
        outs.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    r1, r0 = outs
    assert r0["how"] == "measure" and r0["cpu_baseline"]["value"] > 0 and r0["cpu_baseline"]["cores"] >= 1
    assert "rank 0" in r0["cpu_baseline"]["timed"]
    assert r1["how"] == "wait" and r1["cpu_baseline"] is None
    assert r1["waited_s"] >= 1.0            # at least the single-core + all-core legs' compute time
    assert not os.path.exists(_bench().cpu_flag_path(dict(env, MASTER_PORT=port)))


def test_launcher_times_the_cpu_leg_before_exec(monkeypatch, tmp_path):
    """The "spawn" branch: the CPU leg runs in a child BEFORE this process becomes torch.distributed.run, and travels in the environment."""
    b = _bench()
    seen = {}

    def fake_execve(path, argv, env):
        seen["argv"], seen["env"] = argv, env
        raise SystemExit(0)

    monkeypatch.setattr(b.os, "execve", fake_execve)
    monkeypatch.setattr(b, "visible_gpus", lambda: 2)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--cpu-seconds", "0.2", "--samples", "120000"])
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", b.CPU_ENV):
        monkeypatch.delenv(k, raising=False)
    with pytest.raises(SystemExit):
        b.main()
    assert "--nproc-per-node=2" in seen["argv"]
    o = json.loads(seen["env"][b.CPU_ENV])
    assert o["value"] > 0 and o["cores"] >= 1


@pytest.mark.gpu
def test_self_launched_gather_run_prints_one_line():
    """The N > 1 code path end to end on one GPU: bench.py re-execs itself under torch.distributed.run (1 rank), RCCL
    world of 1, packed message written in place, asynchronous gather every step, the line parsed like the driver does."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--exercise-gather", "--streams", "96",
                        "--samples", "240000", "--steps", "2", "--warmup", "1", "--no-extra", "--cpu-seconds", "1",
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
    assert out["roofline"]["achieved"] > 0 and out["value"] > 0 and 0 < out["roofline"]["frac"] < 1
    # the CPU leg of a launcher-started job: timed by the launcher before the ranks existed, quoted by rank 0
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1 and "launcher" in out["cpu_baseline"]["timed"]
    assert len(out["per_gpu"]) == out["n_gpus"]


@pytest.mark.gpu
def test_driver_launched_rank_carries_the_cpu_leg():
    """The driver's own N > 1 form (`python -m torch.distributed.run ... bench.py --gpus N`): no hand-over variable, rank 0 times
    the oracle itself before touching the GPU. Run at N = 1 under the launcher (the one GPU there is)."""
    b = _bench()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", b.CPU_ENV)}
    argv = b.launcher_argv(1, b.free_port(), os.path.join(ROOT, "bench.py"),
                           ["--gpus", "1", "--exercise-gather", "--streams", "96", "--samples", "240000", "--steps", "2", "--warmup", "1",
                            "--no-extra", "--cpu-seconds", "1", "--check-streams", "8"])
    r = subprocess.run(argv, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["rccl"]["world_size"] == 1 and out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] >= 1
    assert out["bit_errors_vs_cpu_ref"] == 0


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
