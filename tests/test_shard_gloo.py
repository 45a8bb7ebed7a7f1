"""world_size-2 CPU test (gloo) of the N>1 path: stream sharding + the single gather of decoded
bits to rank 0. Each rank 'demodulates' its shard with the CPU oracle (this is a
test -- the product path has no CPU demodulator), rank 0 checks the gathered result against a
single-process run over all streams."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def test_shard_range_covers_everything_once():
    from pirip_amd.shard import shard_range
    for total in (0, 1, 7, 8, 9, 4096, 4099):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = shard_range(total, r, world)
                seen.extend(range(s, s + c))
            assert seen == list(range(total))
            sizes = [shard_range(total, r, world)[1] for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_unpack_roundtrip_msb_first():
    import torch
    from pirip_amd.shard import pack_bits, unpack_bits
    rng = np.random.default_rng(0)
    b = torch.from_numpy(rng.integers(0, 2, (3, 5, 50)).astype(np.uint8))
    p = pack_bits(b)
    assert p.shape == (3, 5, 7)
    ref = np.packbits(b.numpy(), axis=-1)          # numpy packs MSB first
    assert np.array_equal(p.numpy(), ref)
    assert torch.equal(unpack_bits(p, 50), b)


def _worker(rank, world, port, total, q):
    import torch
    import torch.distributed as dist
    from oracle import binding as ob
    import sigutil
    from pirip_amd.shard import (shard_range, pad_streams, assemble, make_payload, split_payload, gather_payload,
                                 alloc_payload, pack_bits, _payload_layout)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = sigutil.CFG1
    start, count = shard_range(total, rank, world)
    slots = pad_streams(count, total, world)
    nsamp, maxf = 6000, 9          # 3 slots x 9 frames x 7 bytes = 189: exercises the 4-byte alignment of the counts
    bits = torch.zeros((slots, maxf, 50), dtype=torch.uint8)
    nfr = torch.zeros(slots, dtype=torch.int32)
    for i in range(count):
        s = start + i
        u8, _ = sigutil.make_u8_stream(ob, c, 300, seed=s, offset=s % 24, random_bits=True)
        rx = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
        r = rx.demod(u8[:nsamp], ob.IN_CU8_FSKDEMOD, want_filt=False)
        bits[i, :r["nframes"]] = torch.from_numpy(r["bits"])
        nfr[i] = r["nframes"]
    dist.barrier()
    # bench.py's layout: the demodulator writes packed bits and frame counts straight into the message
    payload, packed_view, nfr_view = alloc_payload(slots, maxf, 50, "cpu")
    packed_view.copy_(pack_bits(bits)); nfr_view.copy_(nfr)
    assert torch.equal(payload, make_payload(bits, nfr))
    out, work = gather_payload(payload, dist, rank, world, 0, None, async_op=True)
    work.wait()
    if rank == 0:
        assert out[0].numel() == _payload_layout(slots, maxf, 50)[2] == 192 + 4 * slots
        parts = [split_payload(o, slots, maxf, 50) for o in out]
        got = assemble([p[0] for p in parts], [p[1] for p in parts], total, world)
        q.put([g.numpy().copy() for g in got])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    from oracle import binding as ob
    import sigutil
    total, world, port = 5, 2, 29517 + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    c = sigutil.CFG1
    assert len(got) == total
    for s in range(total):
        u8, _ = sigutil.make_u8_stream(ob, c, 300, seed=s, offset=s % 24, random_bits=True)
        rx = ob.OracleFsk(c["Fs"], c["Rs"], c["M"], P=c["P"], est_min=c["est_min"], est_max=c["est_max"])
        r = rx.demod(u8[:6000], ob.IN_CU8_FSKDEMOD, want_filt=False)
        assert np.array_equal(got[s], r["bits"]), s
