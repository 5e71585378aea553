"""The N > 1 path on CPU: world_size 2, gloo.  Each rank takes its block of one
global synthetic workload (the generator is seeded per chain, so a rank builds
exactly its shard), 'phases' it -- here with the CPU oracle standing in for the
GPU, which is what tests may do -- and the single gather to rank 0 must
reproduce the one-process result chain for chain, in global order."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, cfg, q):
    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ahsoka_amd import capi, shard
    import oracle_binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total = cfg["n_chains"]
    first, count = shard.block_partition(total, world)[rank]
    mine = dict(cfg)
    mine["n_chains"] = count
    mine["first_chain"] = first
    b = capi.synth_batch(**mine)
    res = ob.phase_batch(b, capi.default_params(ploidy=cfg["ploidy"]))
    owned = np.arange(first, first + count)
    got = shard.gather_to_rank0(res, owned, total, cfg["n_bubbles"])
    if rank == 0:
        q.put(tuple(np.asarray(x) for x in got))
    dist.barrier()
    dist.destroy_process_group()


def _worker_blocks(rank, world, port, cfg, q):
    """the bench's N > 1 step: fixed-layout result blocks, one gather, decode on rank 0"""
    import torch
    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ahsoka_amd import capi, shard
    import oracle_binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    per = cfg["n_chains"] // world
    mine = dict(cfg)
    mine["n_chains"] = per
    mine["first_chain"] = rank * per
    b = capi.synth_batch(**mine)
    boff = b.arrays()["bubble_off"]
    res = ob.phase_batch(b, capi.default_params(ploidy=cfg["ploidy"]))
    blocks = shard.gather_device_blocks(torch.from_numpy(shard.pack_block(res, boff)))
    if rank == 0:
        out = []
        for r, blk in enumerate(blocks):
            c = dict(cfg)
            c["n_chains"] = per
            c["first_chain"] = r * per
            out.append(shard.decode_block(blk.numpy(), capi.synth_batch(**c).arrays()["bubble_off"], cfg["ploidy"]))
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _worker_lpt(rank, world, port, cfg, q):
    """the N > 1 step for skewed chains: every rank computes the same LPT partition from the
    workload's shapes, generates exactly its (non-contiguous) chains, phases them, and the
    padded packed blocks meet on rank 0 in ONE gather; rank 0 restores the global order"""
    import torch
    import torch.distributed as dist

    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ahsoka_amd import capi, shard
    import oracle_binding as ob

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nb, nr = capi.synth_shapes(**cfg)
    name, owned = shard.plan_partition(nb, nr, world, cfg["b_mode"] == 0)
    mine = owned[rank]
    b = capi.synth_select(mine, **cfg) if name == "lpt" else capi.synth_batch(**dict(cfg, n_chains=len(mine), first_chain=int(mine[0])))
    res = ob.phase_batch(b, capi.default_params(ploidy=cfg["ploidy"]))
    blk = shard.pack_block(res, b.arrays()["bubble_off"])
    maxlen = max(shard.block_length(nb[o], cfg["ploidy"]) for o in owned)
    pad = torch.full((maxlen,), -1, dtype=torch.int32)
    pad[: len(blk)] = torch.from_numpy(blk)
    bufs = shard.gather_device_blocks(pad)
    if rank == 0:
        q.put((name, [o.tolist() for o in owned], shard.decode_gathered([x.numpy() for x in bufs], owned, nb, cfg["ploidy"])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["lpt_c5", "strong_block"])
def test_two_ranks_partition_end_to_end(mode):
    """config 5's cost-weighted partition (non-contiguous ownership -> global order) and the
    strong-scaling split of a fixed workload with unequal blocks, world_size 2 over gloo"""
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ahsoka_amd import capi
    import oracle_binding as ob

    if mode == "lpt_c5":
        cfg = dict(capi.CONFIGS["c5"], n_chains=61, first_chain=0)
        cfg["seed"] = 0xA450CA05
    else:
        cfg = dict(n_chains=9, n_bubbles=6, n_reads=60, ploidy=2, b_mode=0, mean_span=7.0, seed=79, first_chain=0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + (7 if mode == "lpt_c5" else 0)
    procs = [ctx.Process(target=_worker_lpt, args=(r, 2, port, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    name, owned, (status, pos_off, pos, hap) = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = ob.phase_batch(capi.synth_batch(**cfg), capi.default_params())
    np.testing.assert_array_equal(status, ref.status)
    np.testing.assert_array_equal(pos_off, ref.pos_off)
    np.testing.assert_array_equal(pos, ref.pos)
    np.testing.assert_array_equal(hap, ref.hap_allele)
    if mode == "lpt_c5":
        assert name == "lpt" and sorted(owned[0] + owned[1]) == list(range(61))
        assert any(b != a + 1 for o in owned for a, b in zip(o, o[1:]))  # ownership is not a range
    else:
        assert name == "block" and len(owned[0]) == 5 and len(owned[1]) == 4  # unequal blocks, padded gather


def test_two_ranks_block_gather():
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ahsoka_amd import capi
    import oracle_binding as ob

    cfg = dict(n_chains=8, n_bubbles=6, n_reads=60, ploidy=2, seed=78)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_blocks, args=(r, 2, port, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = ob.phase_batch(capi.synth_batch(**cfg), capi.default_params())
    status = np.concatenate([x[0] for x in parts])
    pos = np.concatenate([x[2] for x in parts])
    hap = np.concatenate([x[3] for x in parts])
    np.testing.assert_array_equal(status, ref.status)
    np.testing.assert_array_equal(pos, ref.pos)
    np.testing.assert_array_equal(hap, ref.hap_allele)


@pytest.mark.parametrize("total", [7, 10])
def test_two_ranks_equal_one_process(total):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    from ahsoka_amd import capi
    import oracle_binding as ob

    cfg = dict(n_chains=total, n_bubbles=6, n_reads=60, ploidy=2, seed=77)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + total
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cfg, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, pos_off, pos, hap = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = ob.phase_batch(capi.synth_batch(**cfg), capi.default_params())
    np.testing.assert_array_equal(status, ref.status)
    np.testing.assert_array_equal(pos_off, ref.pos_off)
    np.testing.assert_array_equal(pos, ref.pos)
    np.testing.assert_array_equal(hap, ref.hap_allele)


def test_partitions():
    from ahsoka_amd import shard

    assert shard.block_partition(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert shard.block_partition(3, 8)[-1] == (3, 0)
    costs = [100, 1, 1, 1, 50, 50, 2, 2]
    parts = shard.lpt_partition(costs, 2)
    assert sorted(np.concatenate(parts).tolist()) == list(range(8))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 7
    # deterministic
    assert all((a == b).all() for a, b in zip(parts, shard.lpt_partition(costs, 2)))


def test_pack_unpack_roundtrip():
    from ahsoka_amd import capi, shard
    import oracle_binding as ob

    b = capi.synth_batch(n_chains=5, n_bubbles=9, n_reads=40, seed=5)
    r = ob.phase_batch(b)
    rec = shard.pack_records(r, 8, 9)
    assert rec.shape == (8, 2 + 9 + 2 * 9) and (rec[5:, 0] == -1).all()
    status, pos_off, pos, hap = shard.unpack_records(rec, 2, 9)
    np.testing.assert_array_equal(status, r.status)
    np.testing.assert_array_equal(pos_off, r.pos_off)
    np.testing.assert_array_equal(pos, r.pos)
    np.testing.assert_array_equal(hap, r.hap_allele)
