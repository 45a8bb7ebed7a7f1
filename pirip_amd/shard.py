"""Multi-GPU layout of the receive path: IQ streams are independent, so they shard
in contiguous blocks, one block per rank (one process per GPU), with no data-path collective.
The single exchange is a gather of the decoded bits to rank 0 -- torch.distributed backend
"nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests. Host logic only; no compute here."""
from typing import List, Optional, Tuple


def shard_range(total_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, start+count) of `total_streams` owned by `rank` (remainder spread
    over the first ranks, so sizes differ by at most one)."""
    if not (0 <= rank < world) or total_streams < 0:
        raise ValueError("bad rank/world/total")
    base, rem = divmod(total_streams, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def pad_streams(count: int, total_streams: int, world: int) -> int:
    """Per-rank stream slots so every rank gathers the same shape (torch gather needs equal sizes)."""
    return (total_streams + world - 1) // world if world > 0 else count


def pack_bits(bits):
    """[..., Nbits] one-bit-per-byte (0/1) -> [..., ceil(Nbits/8)] bytes, MSB first (codec2's
    freedv_pack order, the order `rpitx_fsk --packed` consumes: /root/reference/tx/rpitx_fsk.cpp:75-83).
    torch tensor in, torch tensor out, on the tensor's device."""
    import torch
    nbits = bits.shape[-1]
    nbytes = (nbits + 7) // 8
    pad = nbytes * 8 - nbits
    if pad:
        bits = torch.nn.functional.pad(bits, (0, pad))
    w = torch.tensor([128, 64, 32, 16, 8, 4, 2, 1], dtype=torch.uint8, device=bits.device)
    return (bits.reshape(*bits.shape[:-1], nbytes, 8) * w).sum(dim=-1, dtype=torch.uint8)


def unpack_bits(packed, nbits: int):
    import torch
    sh = torch.tensor([7, 6, 5, 4, 3, 2, 1, 0], dtype=torch.uint8, device=packed.device)
    b = (packed.unsqueeze(-1) >> sh) & 1
    return b.reshape(*packed.shape[:-1], packed.shape[-1] * 8)[..., :nbits]


def gather_bits(bits, nframes, dist=None, rank: int = 0, world: int = 1, dst: int = 0):
    """One gather of decoded bits (+ frame counts) to `dst`.
    bits: uint8 tensor [slots, max_frames, Nbits]; nframes: int32 tensor [slots].
    Returns (list_of_bits_per_rank, list_of_nframes_per_rank) on dst, (None, None) elsewhere.
    With world == 1 / dist None it is the identity."""
    if dist is None or world == 1:
        return [bits], [nframes]
    import torch
    gb = [torch.empty_like(bits) for _ in range(world)] if rank == dst else None
    gn = [torch.empty_like(nframes) for _ in range(world)] if rank == dst else None
    dist.gather(bits, gb, dst=dst)
    dist.gather(nframes, gn, dst=dst)
    return gb, gn


def _payload_layout(slots: int, max_frames: int, nbits: int):
    """(bytes of packed bits, offset of the int32 frame counts (4-byte aligned), total bytes)."""
    nbytes = (nbits + 7) // 8
    nb = slots * max_frames * nbytes
    off = (nb + 3) // 4 * 4
    return nb, off, off + 4 * slots


def alloc_payload(slots: int, max_frames: int, nbits: int, device):
    """One flat uint8 message per rank and step, laid out so the demodulator writes it in place
    (pirip_hip_set_bit_packing(h, 1): d_bits = packed.data_ptr(), d_nframes = nframes.data_ptr()) and the
    exchange needs no staging copy. Returns (payload, packed view [slots, max_frames, ceil(nbits/8)],
    nframes view int32 [slots])."""
    import torch
    nb, off, total = _payload_layout(slots, max_frames, nbits)
    payload = torch.zeros(total, dtype=torch.uint8, device=device)
    packed = payload[:nb].view(slots, max_frames, (nbits + 7) // 8)
    nframes = payload[off:off + 4 * slots].view(torch.int32)
    return payload, packed, nframes


def make_payload(bits, nframes):
    """Same message built from one-bit-per-byte output (host-side packing; the in-place path is
    alloc_payload): MSB-first packed bits followed by the int32 frame counts -- a single gather."""
    import torch
    slots, max_frames, nbits = bits.shape
    payload, packed, nfr = alloc_payload(slots, max_frames, nbits, bits.device)
    packed.copy_(pack_bits(bits))
    nfr.copy_(nframes)
    return payload


def split_payload(payload, slots: int, max_frames: int, nbits: int):
    """Inverse on the gathering rank: (bits [slots, max_frames, nbits], nframes [slots])."""
    import torch
    nb, off, total = _payload_layout(slots, max_frames, nbits)
    packed = payload[:nb].reshape(slots, max_frames, (nbits + 7) // 8)
    nframes = payload[off:off + 4 * slots].contiguous().view(torch.int32)
    return unpack_bits(packed, nbits), nframes


def gather_payload(payload, dist=None, rank: int = 0, world: int = 1, dst: int = 0, out=None, async_op: bool = False):
    """The single exchange of the path: gather every rank's payload to `dst` (RCCL on GPUs, gloo in the
    CPU tests). Returns (list_of_payloads or None, work_handle or None). With async_op the collective runs
    on the backend's own stream and overlaps the next step's kernel; wait() on the handle before reading."""
    if dist is None:
        return [payload], None
    import torch
    if rank == dst and out is None:
        out = [torch.empty_like(payload) for _ in range(world)]
    work = dist.gather(payload, out if rank == dst else None, dst=dst, async_op=async_op)
    return (out if rank == dst else None), (work if async_op else None)


def assemble(gb: List, gn: List, total_streams: int, world: int):
    """Rank-0 view after gather: per global stream s -> (bits[:nframes])."""
    out = []
    for r in range(world):
        start, count = shard_range(total_streams, r, world)
        for i in range(count):
            n = int(gn[r][i])
            out.append(gb[r][i, :n])
    return out
