"""Multi-GPU layer of the inference path: one process per GPU, weights replicated, utterances / text chunks are
the independent units (the reference loops over them sequentially: F/infer/utils_infer.py:441-482).

The only data that must cross GPUs is the reference-audio latents (cond mel [n_ref, 100] fp32 + reference token
ids), broadcast once per request from rank 0 over RCCL/xGMI, and -- when rank 0 must return the audio -- a gather of
the waveforms.  Nothing is exchanged inside the ODE loop, so there is no all-reduce anywhere on the path.
Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests."""
from __future__ import annotations

import torch
import torch.distributed as dist


def unit_cost(n_frames: int, dim: int = 1024, depth: int = 22, ff_mult: int = 2) -> float:
    """MACs of one DiT forward for a unit of n_frames: GEMMs are linear in N, attention quadratic."""
    per_tok = depth * (4 * dim * dim + 2 * ff_mult * dim * dim)
    return n_frames * (per_tok + depth * 2 * n_frames * dim)


def shard_units(frames: list[int], world_size: int, **arch) -> list[list[int]]:
    """Longest-processing-time-first dealing of unit indices to ranks (the inference analogue of the reference's
    frame-budget DynamicBatchSampler, F/model/dataset.py:178-237).  Deterministic, identical on every rank."""
    order = sorted(range(len(frames)), key=lambda i: (-unit_cost(frames[i], **arch), i))
    load = [0.0] * world_size
    out: list[list[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += unit_cost(frames[i], **arch)
    for lst in out:
        lst.sort()
    return out


def broadcast_ref_latents(cond_mel: torch.Tensor | None, ref_ids: torch.Tensor | None, device, src: int = 0,
                          max_frames: int = 4096, mel_dim: int = 100, max_ids: int = 1024):
    """Rank `src` owns the reference-audio latents; every rank returns (cond_mel [n, mel] fp32, ref_ids [k] int64).
    Two collectives: a 2-int header, then one packed fp32 payload (ids ride as floats: vocab < 2^24)."""
    rank = dist.get_rank() if dist.is_initialized() else 0
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return cond_mel.to(device), ref_ids.to(device)
    hdr = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == src:
        assert cond_mel.shape[0] <= max_frames and cond_mel.shape[1] == mel_dim and ref_ids.numel() <= max_ids
        hdr[0], hdr[1] = cond_mel.shape[0], ref_ids.numel()
    dist.broadcast(hdr, src=src)
    n, k = int(hdr[0]), int(hdr[1])
    buf = torch.empty(n * mel_dim + k, dtype=torch.float32, device=device)
    if rank == src:
        buf[: n * mel_dim] = cond_mel.to(device, torch.float32).reshape(-1)
        buf[n * mel_dim:] = ref_ids.to(device, torch.float32)
    dist.broadcast(buf, src=src)
    return buf[: n * mel_dim].view(n, mel_dim), buf[n * mel_dim:].round().to(torch.int64)


def gather_waves(wave: torch.Tensor, dst: int = 0):
    """Gathers one fp32 waveform per rank to `dst` (list in rank order there, None elsewhere); lengths may differ."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [wave]
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([wave.numel()], dtype=torch.int64, device=wave.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    nmax = int(max(int(s) for s in sizes))
    pad = torch.zeros(nmax, dtype=torch.float32, device=wave.device)
    pad[: wave.numel()] = wave.reshape(-1)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return [b[: int(s)] for b, s in zip(bufs, sizes)]
