"""The C++ multi-GPU exchange (include/pirip_hip_rccl.h: pirip_hip_gather_bits, pirip_hip_gather_layout, the file rendezvous
of pirip_hip_rccl_init) at world size 2 and 3 on a box WITHOUT GPUs: the product's own source pirip_amd/csrc/rccl_gather.hip is
compiled as plain C++ against a test-only fake of the few RCCL / HIP entry points it calls (tests/fake_rccl/: ncclSend /
ncclRecv / ncclGroup* over named pipes between processes, hipMemcpyAsync = memcpy). Checked: every rank's message lands in its
own slot on rank 0; the message layout the C++ host writes (packed bits | pad | int32 frame counts, counts 4-byte aligned for
any stream count) is the one pirip_amd/shard.py parses; a stale unique-id file left by a crashed run -- or by another session --
is ignored. The real thing (RCCL over xGMI) is the driver's 8-GPU run; this pins the host logic around it."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    d = tmp_path_factory.mktemp("fake_nccl_build")
    out = str(d / "gather_world2")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "tests", "fake_rccl"), "-I", os.path.join(ROOT, "include"),
                           "-x", "c++", os.path.join(ROOT, "pirip_amd", "csrc", "rccl_gather.hip"),
                           os.path.join(ROOT, "tests", "fake_rccl", "fake_nccl.cpp"), os.path.join(ROOT, "tests", "cprog", "gather_world2.cpp"),
                           "-o", out])
    return out


@pytest.mark.parametrize("world,streams,maxf,fb", [(2, 3, 5, 7), (3, 4, 2, 13), (2, 1, 1, 1)])
def test_cpp_gather_slots_layout_and_rendezvous(exe, tmp_path, world, streams, maxf, fb):
    import torch
    from pirip_amd import shard
    idf = tmp_path / "rccl_id"
    # what a crashed earlier run leaves behind: right size, right magic, another session's tag -- and garbage at the temp name
    idf.write_bytes(b"PIRIPID1" + b"stale-session".ljust(56, b"\0") + bytes(128))
    env = dict(os.environ, FAKE_NCCL_DIR=str(tmp_path), PIRIP_RCCL_SESSION=f"test-{world}-{streams}")
    procs = []
    for r in list(range(1, world)) + [0]:                        # rank 0 last: the others meet the stale file first
        procs.append((r, subprocess.Popen([exe, str(r), str(world), str(idf), str(streams), str(maxf), str(fb)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.PIPE)))
    outs = {}
    for r, p in procs:
        o, e = p.communicate(timeout=60)
        assert p.returncode == 0, (r, e.decode())
        outs[r] = o
    nb, off, total = shard._payload_layout(streams, maxf, fb * 8)
    assert off % 4 == 0 and len(outs[0]) == world * total                          # the C layout is shard.py's
    got = np.frombuffer(outs[0], dtype=np.uint8).reshape(world, total)
    for r in range(world):
        payload = torch.from_numpy(got[r].copy())
        bits, nfr = shard.split_payload(payload, streams, maxf, fb * 8)               # the parser of the Python path reads the C++ message
        packed = shard.pack_bits(bits).numpy()
        s, f, b = np.meshgrid(np.arange(streams), np.arange(maxf), np.arange(fb), indexing="ij")
        assert np.array_equal(packed, ((17 * r + 5 * s + 3 * f + b + 101) % 256).astype(np.uint8))   # second round's message, rank r's slot
        assert list(nfr.numpy()) == [(r + 1) * 1000 + i + 1 for i in range(streams)]


def test_rendezvous_times_out_with_a_message_when_rank0_never_comes(exe, tmp_path):
    """A rank that only ever sees another session's file gives up with an explanation instead of joining the wrong job."""
    idf = tmp_path / "rccl_id"
    idf.write_bytes(b"PIRIPID1" + b"someone-else".ljust(56, b"\0") + bytes(128))
    env = dict(os.environ, FAKE_NCCL_DIR=str(tmp_path), PIRIP_RCCL_SESSION="mine", PIRIP_RCCL_TIMEOUT_S="1")
    p = subprocess.run([exe, "1", "2", str(idf), "1", "1", "1"], env=env, capture_output=True, timeout=90)
    assert p.returncode == 4 and b"session 'mine'" in p.stderr


def test_world_above_one_needs_a_session_tag_from_the_launcher(exe, tmp_path):
    """Without $PIRIP_RCCL_SESSION two runs from one shell cannot be told apart: refused with a sentence, on every rank."""
    env = {k: v for k, v in os.environ.items() if k != "PIRIP_RCCL_SESSION"}
    env["FAKE_NCCL_DIR"] = str(tmp_path)
    for rank in ("0", "1"):
        p = subprocess.run([exe, rank, "2", str(tmp_path / "rccl_id"), "1", "1", "1"], env=env, capture_output=True, timeout=90)
        assert p.returncode != 0 and b"PIRIP_RCCL_SESSION" in p.stderr
