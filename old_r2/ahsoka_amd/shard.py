"""Multi-GPU sharding of the phasing path: one process per GPU.

Chains are independent (each iteration of the chain loop of
``alignmentsToReadset`` -- src/alignmentstoreadset.cpp:75-490 -- reads only its
own alignments and allele paths), so the data path needs no collective: a
static partition of the chain list over the ranks, every rank phases its shard
on its own MI355X, and ONE gather of the per-chain result records to rank 0 at
the end (``torch.distributed`` -- backend "nccl" is RCCL over xGMI on ROCm, "gloo"
in the CPU tests).  Rank 0 puts the records back into the global chain order, so
the rendered output does not depend on the number of GPUs.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

from .capi import PhaseResult


def block_partition(n_items: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks (first, count); sizes differ by at most one."""
    base, extra = divmod(n_items, world)
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def lpt_partition(costs: Sequence[float], world: int) -> List[np.ndarray]:
    """Cost-weighted static partition (longest processing time first) for skewed
    chain sizes (BASELINE config 5): returns, per rank, the global chain indices
    it owns, each list ascending.  Deterministic: ties go to the lower rank."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    owner = np.empty(len(costs), dtype=np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += costs[i]
    return [np.flatnonzero(owner == r) for r in range(world)]


def chain_cost(n_reads: np.ndarray, n_bubbles: np.ndarray) -> np.ndarray:
    """Work model of one chain: the pair matrix dominates (R^2 per position
    word) plus the R*B match matrices."""
    r = n_reads.astype(np.float64)
    b = n_bubbles.astype(np.float64)
    return r * r * np.maximum(1.0, b / 64.0) + r * b


def pack_records(res: PhaseResult, max_chains: int, max_pos: int) -> np.ndarray:
    """Fixed-size int32 record per chain: [status, P, pos[max_pos], hap[ploidy*max_pos]]
    (vectorised: 10^4 chains per step must not cost Python-loop time)."""
    p = res.ploidy
    C = res.n_chains
    width = 2 + max_pos + p * max_pos
    rec = np.full((max_chains, width), -1, dtype=np.int32)
    if C == 0:
        return rec
    P = np.diff(res.pos_off).astype(np.int64)
    rec[:C, 0] = res.status
    rec[:C, 1] = P
    total = int(res.pos_off[-1])
    if total:
        row = np.repeat(np.arange(C), P)
        col = np.arange(total) - np.repeat(res.pos_off[:-1], P)
        rec[row, 2 + col] = res.pos
        base = p * np.repeat(res.pos_off[:-1], P)
        Prep = np.repeat(P, P)
        for i in range(p):
            rec[row, 2 + max_pos + i * max_pos + col] = res.hap_allele[base + i * Prep + col]
    return rec


def unpack_records(rec: np.ndarray, ploidy: int, max_pos: int):
    """-> (status, pos_off, pos, hap_allele) in the order of the rows of rec
    (rows with status -1 are padding and are dropped)."""
    rec = rec[rec[:, 0] >= 0]
    C = len(rec)
    status = rec[:, 0].astype(np.int32)
    P = rec[:, 1].astype(np.int64)
    pos_off = np.concatenate([[0], np.cumsum(P)]).astype(np.int64)
    total = int(pos_off[-1])
    if total == 0:
        return status, pos_off, np.zeros(0, np.int32), np.zeros(0, np.int32)
    row = np.repeat(np.arange(C), P)
    col = np.arange(total) - np.repeat(pos_off[:-1], P)
    pos = rec[row, 2 + col].astype(np.int32)
    hap = np.empty(ploidy * total, dtype=np.int32)
    base = ploidy * np.repeat(pos_off[:-1], P)
    Prep = np.repeat(P, P)
    for i in range(ploidy):
        hap[base + i * Prep + col] = rec[row, 2 + max_pos + i * max_pos + col]
    return status, pos_off, pos, hap


def gather_to_rank0(res: PhaseResult, owned: np.ndarray, n_total: int, max_pos: int, device=None, group=None,
                    max_chains: Optional[int] = None, unpack: bool = True):
    """The single collective of the path.  ``owned`` = global chain indices of this
    rank's shard, in the order the shard was phased.  Returns on rank 0
    (status, pos_off, pos, hap_allele) for all n_total chains in global chain
    order (``unpack=False``: the fixed-width GatheredRecords instead, which is all
    the hot loop needs), on other ranks None.  ``max_chains`` = largest shard (records are
    padded to it); a block partition knows it without communication, otherwise
    it is agreed on with one extra tiny all_gather."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    if max_chains is None:
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([len(owned)], dtype=torch.int64, device=dev), group=group)
        max_chains = int(max(int(c.item()) for c in counts))
    rec = pack_records(res, max_chains, max_pos)
    idx = np.full(max_chains, -1, dtype=np.int32)
    idx[: len(owned)] = owned
    payload = torch.from_numpy(np.concatenate([idx[:, None], rec], axis=1)).to(dev)
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    dist.gather(payload, bufs, dst=0, group=group)
    if rank != 0:
        return None
    allrec = torch.cat(bufs, dim=0).cpu().numpy()
    allrec = allrec[allrec[:, 0] >= 0]
    if not (np.diff(allrec[:, 0]) > 0).all():  # block partitions arrive in global order already
        allrec = allrec[np.argsort(allrec[:, 0], kind="stable")]
    assert len(allrec) == n_total and (allrec[:, 0] == np.arange(n_total)).all(), "shards do not tile the chain list"
    rec = GatheredRecords(allrec[:, 1:], res.ploidy, max_pos)
    return rec.unpack() if unpack else rec


class GatheredRecords:
    """All chains' result records on rank 0, in global chain order: one fixed-width
    int32 row per chain [status, P, pos[max_pos], hap[ploidy][max_pos]] (padding -1).
    ``unpack()`` gives the ragged (status, pos_off, pos, hap_allele) arrays of a
    PhaseResult; rendering and tests use it, the hot loop does not need to."""

    def __init__(self, rec: np.ndarray, ploidy: int, max_pos: int):
        self.rec, self.ploidy, self.max_pos = rec, ploidy, max_pos

    @property
    def status(self) -> np.ndarray:
        return self.rec[:, 0]

    def chain_haps(self, c: int) -> np.ndarray:
        P = int(self.rec[c, 1])
        o = 2 + self.max_pos
        return np.stack([self.rec[c, o + i * self.max_pos: o + i * self.max_pos + P] for i in range(self.ploidy)])

    def chain_pos(self, c: int) -> np.ndarray:
        return self.rec[c, 2:2 + int(self.rec[c, 1])]

    def unpack(self):
        return unpack_records(self.rec, self.ploidy, self.max_pos)


def gather_device_blocks(block, group=None):
    """The single collective of the path on device memory: every rank passes its
    packed result block (DeviceBatch.result_tensor(): [status | npos | pos | hap],
    same length on every rank for equal shard shapes); rank 0 gets the list of all
    ranks' blocks (still on its GPU), other ranks None.  With backend "nccl" this is
    one RCCL gather over xGMI and no host staging."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bufs = [torch.empty_like(block) for _ in range(world)] if rank == 0 else None
    dist.gather(block, bufs, dst=0, group=group)
    return bufs


def pack_block(res: PhaseResult, bubble_off: np.ndarray) -> np.ndarray:
    """Host mirror of the device result block (``ahsoka_batch_result_device``): int32
    ``[status[C] | npos[C] | pos[SB] | hap[ploidy*SB]]`` with SB = bubble_off[-1]; chain c
    owns pos[bubble_off[c]:][:npos] and hap[ploidy*bubble_off[c]:][:ploidy*npos] (ploidy rows
    of npos alleles), the rest is padding (-1 here, unspecified in the device block).  The layout depends only on the batch shape,
    so equal shards give equal block lengths on every rank."""
    p = res.ploidy
    C = res.n_chains
    SB = int(bubble_off[-1])
    blk = np.full(2 * C + SB + p * SB, -1, dtype=np.int32)
    npos = np.diff(res.pos_off).astype(np.int64)
    blk[:C] = res.status
    blk[C:2 * C] = npos
    total = int(res.pos_off[-1])
    if total:
        row = np.repeat(np.arange(C), npos)
        col = np.arange(total) - np.repeat(res.pos_off[:-1], npos)
        blk[2 * C + bubble_off[:-1][row] + col] = res.pos
        base_in = p * np.repeat(res.pos_off[:-1], npos)
        base_out = 2 * C + SB + p * bubble_off[:-1][row]
        Prep = np.repeat(npos, npos)
        for i in range(p):
            blk[base_out + i * Prep + col] = res.hap_allele[base_in + i * Prep + col]
    return blk


def decode_block(block: np.ndarray, bubble_off: np.ndarray, ploidy: int):
    """host view of one packed block -> (status, pos_off, pos, hap_allele) like PhaseResult"""
    C = len(bubble_off) - 1
    SB = int(bubble_off[-1])
    status = block[:C].astype(np.int32)
    npos = np.where(status == 0, block[C:2 * C], 0).astype(np.int64)
    pos_raw = block[2 * C:2 * C + SB]
    hap_raw = block[2 * C + SB:]
    pos_off = np.concatenate([[0], np.cumsum(npos)]).astype(np.int64)
    total = int(pos_off[-1])
    if total == 0:
        return status, pos_off, np.zeros(0, np.int32), np.zeros(0, np.int32)
    row = np.repeat(np.arange(C), npos)
    col = np.arange(total) - np.repeat(pos_off[:-1], npos)
    pos = pos_raw[bubble_off[:-1][row] + col].astype(np.int32)
    hap = np.empty(ploidy * total, dtype=np.int32)
    base_out = ploidy * np.repeat(pos_off[:-1], npos)
    base_in = ploidy * bubble_off[:-1][row]
    Prep = np.repeat(npos, npos)
    for i in range(ploidy):
        hap[base_out + i * Prep + col] = hap_raw[base_in + i * Prep + col]
    return status, pos_off, pos, hap


def plan_partition(n_bubbles: np.ndarray, n_reads: np.ndarray, world: int, equal_shapes: bool) -> Tuple[str, List[np.ndarray]]:
    """The static partition every rank computes for itself from the workload's shapes (no
    communication): contiguous blocks when all chains have the same shape, the cost-weighted
    LPT partition otherwise.  Returns (name, per rank the ascending global chain indices)."""
    n = len(n_bubbles)
    if equal_shapes or world == 1:
        return "block", [np.arange(f, f + c) for f, c in block_partition(n, world)]
    return "lpt", lpt_partition(chain_cost(np.asarray(n_reads), np.asarray(n_bubbles)), world)


def block_length(n_bubbles_of_shard: np.ndarray, ploidy: int) -> int:
    """int32 length of a shard's packed result block [status | npos | pos | hap]"""
    return 2 * len(n_bubbles_of_shard) + (1 + ploidy) * int(np.sum(n_bubbles_of_shard))


def decode_gathered(blocks: Sequence[np.ndarray], owned: Sequence[np.ndarray], n_bubbles: np.ndarray, ploidy: int):
    """Rank 0, after the gather: every rank's (padded) block back into ONE result in global
    chain order -> (status, pos_off, pos, hap_allele) like PhaseResult.  ``owned[r]`` are the
    global indices of rank r's chains in the order of its block; shards must tile the list."""
    n = len(n_bubbles)
    status = np.full(n, -1, dtype=np.int32)
    npos = np.zeros(n, dtype=np.int64)
    parts = []
    for r, blk in enumerate(blocks):
        nb = np.asarray(n_bubbles)[owned[r]]
        boff = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        st, pos_off, pos, hap = decode_block(np.asarray(blk)[: block_length(nb, ploidy)], boff, ploidy)
        status[owned[r]] = st
        npos[owned[r]] = np.diff(pos_off)
        parts.append((pos_off, pos, hap))
    if (status < 0).any():
        raise ValueError("the shards do not tile the chain list")
    pos_off = np.concatenate([[0], np.cumsum(npos)]).astype(np.int64)
    pos = np.zeros(int(pos_off[-1]), dtype=np.int32)
    hap = np.zeros(ploidy * int(pos_off[-1]), dtype=np.int32)
    for r, (po, ps, hp) in enumerate(parts):
        for j, c in enumerate(owned[r]):
            a, b = int(po[j]), int(po[j + 1])
            pos[pos_off[c]:pos_off[c + 1]] = ps[a:b]
            hap[ploidy * pos_off[c]:ploidy * pos_off[c + 1]] = hp[ploidy * a:ploidy * b]
    return status, pos_off, pos, hap
