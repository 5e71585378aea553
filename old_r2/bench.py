#!/usr/bin/env python3
"""bench.py -- bubbles/s phased on the BASELINE.json headline workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (k_readset + k_phase, then the results to
host memory: the download at N = 1, the single gather of every rank's packed
result block to rank 0 plus its copy to the host for N > 1) over one synthetic
batch whose inputs are already resident in HBM.

Workload: BASELINE.json configs[2] -- 10 000 chains x 50 bubbles x 1000
reads/chain, ploidy 2.  For N > 1 the default is what BASELINE.json states:
that FIXED workload sharded over the N GPUs (strong scaling; `value` and
`"scaling": "strong"`), one process per GPU, static partition, no data-path
collective, one RCCL gather at the end of the step; the same run also measures
the weak form (10 000 chains per GPU) and reports it as a second figure
(`"weak"`).  `--scaling weak` makes the weak form the headline instead.
Chains of equal shape (configs 1-4) are split into contiguous blocks; the
power-law config 5 (`--config c5`) is split by the cost-weighted LPT partition
(shard.lpt_partition over shard.chain_cost), which every rank computes for
itself from the workload's shapes.

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      dominant kernel (k_phase): algorithmic bytes / its mean HIP-event
                duration on the launch stream, against 8 TB/s HBM
  cpu_baseline  the CPU oracle ("port": oracle/oracle.cpp, NOT the upstream
                binary, which cannot be built here) on a bounded sample of the
                same workload, timed on this box's host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


class Shard:
    """This rank's share of one global synthetic workload, resident in HBM."""

    def __init__(self, capi, shard, cfg, total_chains, rank, world, device_index, max_slots=0):
        self.capi, self.shardmod = capi, shard
        self.cfg = dict(cfg)
        self.cfg["n_chains"] = total_chains
        self.cfg["first_chain"] = 0
        self.total = total_chains
        self.ploidy = cfg["ploidy"]
        self.world, self.rank = world, rank
        # every rank derives the same partition from the workload's shapes: no communication
        self.nb, self.nr = capi.synth_shapes(**self.cfg)
        self.partition, self.owned = shard.plan_partition(self.nb, self.nr, world, cfg.get("b_mode", 0) == 0)
        mine = self.owned[rank]
        t0 = time.time()
        if self.partition == "block":
            c = dict(self.cfg)
            c["n_chains"] = len(mine)
            c["first_chain"] = int(mine[0]) if len(mine) else 0
            self.batch = capi.synth_batch(**c)
        else:
            self.batch = capi.synth_select(mine, **self.cfg)
        self.synth_s = time.time() - t0
        self.bubble_off = self.batch.arrays()["bubble_off"]
        self.prm = capi.default_params(ploidy=self.ploidy, device=device_index, max_slots=max_slots)
        t0 = time.time()
        self.dev = capi.DeviceBatch(self.batch, self.prm)
        self.upload_s = time.time() - t0
        # length of every rank's packed result block [status | npos | pos | hap]: 2C + (1+p) * sum B
        self.block_len = [shard.block_length(self.nb[o], self.ploidy) for o in self.owned]
        self.max_block = max(self.block_len)

    def free(self):
        self.dev.free()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", help="BASELINE config (c3 = headline); others for experiments")
    ap.add_argument("--chains", type=int, default=0, help="override the workload's chain count (experiments only)")
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="N > 1: strong = the fixed workload split over the ranks (default), weak = that workload per rank")
    ap.add_argument("--max-slots", type=int, default=0, help="ahsoka_phase_params.max_slots (experiments only)")
    ap.add_argument("--cpu-sample", type=int, default=24, help="chains timed on the CPU oracle (0 = skip)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    from ahsoka_amd import capi, shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the phasing path has no CPU fallback")
    # rehearsal aid (one-GPU box): AHSOKA_BENCH_BACKEND=gloo runs every rank on cuda:0 and
    # gathers over gloo; the driver's multi-GPU runs use the default, nccl = RCCL over xGMI
    backend = os.environ.get("AHSOKA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    cfg = dict(capi.CONFIGS[args.config])
    base_chains = args.chains if args.chains > 0 else cfg["n_chains"]
    scaling = args.scaling if args.scaling != "auto" else "strong"

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def run(total_chains, steps, warmup):
        """-> dict of measurements of `steps` timed steps on the workload of total_chains chains"""
        sh = Shard(capi, shard, cfg, total_chains, rank, world, local_rank, args.max_slots)
        pad = None
        if world > 1:
            pad = torch.full((sh.max_block,), -1, dtype=torch.int32, device=coll_device)

        def step():
            sh.dev.phase()
            if world == 1:
                return sh.dev.download(), None
            # the one collective of the path: every rank's packed result block (device memory,
            # layout fixed by the shard's shape, padded to the longest) to rank 0 -- one RCCL
            # gather over xGMI -- and from there to the host
            block = sh.dev.result_tensor()
            pad[: block.numel()].copy_(block if backend == "nccl" else block.cpu())
            bufs = shard.gather_device_blocks(pad)
            host = torch.stack(bufs).cpu().numpy() if rank == 0 else None
            return None, host

        for _ in range(warmup):
            step()
        k_phase_ms, k_readset_ms = [], []
        fence()
        t0 = time.perf_counter()
        res = host = None
        for _ in range(steps):
            res, host = step()
            kt = sh.dev.kernel_times_ms()
            k_phase_ms.append(kt["k_phase"])
            k_readset_ms.append(kt["k_readset"])
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        out = {"elapsed": elapsed, "steps": steps, "k_phase_ms": float(np.mean(k_phase_ms)),
               "k_readset_ms": float(np.mean(k_readset_ms)), "shard": sh}
        if rank == 0:
            if world == 1:
                out["phased"] = res.phased_bubbles(sh.bubble_off)
                out["P_total"] = int(res.pos_off[-1])
                out["n_phased"] = int((res.status == capi.ST_PHASED).sum())
                out["verified"] = None
                own = res
            else:
                # outside the timed region: decode every rank's block into the global chain
                # order and check rank 0's own shard against the library's download path
                status, pos_off, pos, hap = shard.decode_gathered(host, sh.owned, sh.nb, sh.ploidy)
                out["phased"] = int(sh.nb[status == capi.ST_PHASED].sum())
                out["P_total"] = int(pos_off[-1])
                out["n_phased"] = int((status == capi.ST_PHASED).sum())
                own = sh.dev.download()
                mine = sh.owned[0]
                ok = bool((status[mine] == own.status).all() and (np.diff(pos_off)[mine] == np.diff(own.pos_off)).all())
                for j in (0, len(mine) // 2, len(mine) - 1):  # spot-check alleles of three of rank 0's chains
                    if len(mine) == 0:
                        break
                    c = int(mine[j])
                    ok = ok and bool((hap[sh.ploidy * pos_off[c]:sh.ploidy * pos_off[c + 1]]
                                      == own.hap_allele[sh.ploidy * own.pos_off[j]:sh.ploidy * own.pos_off[j + 1]]).all())
                out["verified"] = ok
                assert out["verified"], "gathered block of rank 0 differs from its download"
            # algorithmic bytes of one launch on rank 0: inputs once + outputs once
            Pt = int(own.pos_off[-1])
            out["alg_bytes"] = sh.batch.algorithmic_input_bytes(sh.ploidy) + 4 * (
                Pt + sh.ploidy * Pt + int((own.status == capi.ST_PHASED).sum()))
        return out

    total_main = base_chains * (world if scaling == "weak" else 1)
    m = run(total_main, args.steps, args.warmup)
    second = None
    if world > 1 and args.scaling == "auto":
        m["shard"].free()
        second = run(base_chains * world, max(1, args.steps // 2), min(1, args.warmup))

    if rank == 0:
        sh = m["shard"]
        ploidy = sh.ploidy
        ms_per_step = m["elapsed"] * 1e3 / m["steps"]
        value = m["phased"] / (m["elapsed"] / m["steps"])
        alg_bytes = m["alg_bytes"]
        kp = m["k_phase_ms"]
        achieved = alg_bytes / (kp * 1e-3) / 1e9
        traffic = pmc_traffic(args.config, len(sh.owned[0]))
        stream = stream_copy_gbps(torch, device)
        shape = (f"{cfg['n_bubbles']} bubbles x {cfg['n_reads']} reads/chain" if cfg.get("b_mode", 0) == 0
                 else "power-law bubble counts (1..500), ~40x read depth")
        out = {
            "metric": "bubbles/s phased (ploidy 2, 1000 reads/chain)",
            "value": value,
            "unit": "bubbles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[{int(args.config[1:]) - 1}]: {sh.total} chains x {shape}, ploidy {ploidy}"
                            + (f", sharded over {world} GPUs" if world > 1 else ""),
                "config_name": args.config,
                "chains_total": sh.total,
                "chains_rank0": len(sh.owned[0]),
                "parallelism": f"chain-shard x{world} ({sh.partition} partition), one gather",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_phase",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": alg_bytes,
                "kernel_ms": kp,
                "k_readset_ms": m["k_readset_ms"],
                # the measured traffic (committed PMC summary of this workload) over this run's kernel
                # time, against the nominal peak and against a streaming copy measured in this job
                "traffic_rate_GBps": (traffic / (kp * 1e-3) / 1e9) if traffic else None,
                "stream_copy_GBps": stream,
            },
            "phased_bubbles_per_step": m["phased"],
            "phased_chains": m["n_phased"],
            "gather_verified": m["verified"],
            "upload_s": sh.upload_s,
            "synth_s": sh.synth_s,
            "backend": capi.backend_info(),
        }
        if second is not None:
            out["weak"] = {
                "value": second["phased"] / (second["elapsed"] / second["steps"]),
                "unit": "bubbles/s",
                "ms_per_step": second["elapsed"] * 1e3 / second["steps"],
                "steps": second["steps"],
                "chains_per_gpu": base_chains,
                "phased_bubbles_per_step": second["phased"],
                "k_phase_ms_rank0": second["k_phase_ms"],
            }
        if world == 1 and args.cpu_sample > 0:
            c1 = dict(cfg)
            out["cpu_baseline"] = cpu_baseline(c1, sh.prm, args.cpu_sample)
        print(json.dumps(out), flush=True)
    (second or m)["shard"].free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def stream_copy_gbps(torch, device):
    """STREAM-style device copy: read + write bytes per second of a plain
    1 GiB -> 1 GiB copy on this GPU, measured in the same job; the yardstick for what the
    HBM can deliver to a perfectly streaming kernel."""
    n = 1 << 28  # int32 elements = 1 GiB
    a = torch.empty(n, dtype=torch.int32, device=device).fill_(1)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    del a, b
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def pmc_traffic(config, chains):
    """HBM bytes of one k_phase launch from the committed rocprofv3 PMC summary
    (tools/pmc.sh, separate FETCH_SIZE / WRITE_SIZE passes) of this very workload;
    None if the summary is for another workload.  (2*FETCH_SIZE + WRITE_SIZE)*1024:
    the gfx950 half-count of FETCH_SIZE was calibrated on scattered 4- and 16-byte
    reads as well (profiles/r01_fetch_calibration.json): 64 B tallied per request."""
    import glob
    import re

    def order(path):  # r<round>_v<version>_pmc_hbm.json, numerically
        m = re.search(r"r(\d+)_v(\d+)_pmc_hbm", os.path.basename(path))
        return (int(m.group(1)), int(m.group(2))) if m else (-1, -1)

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")), key=order)
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("config") == config and d.get("chains") == chains:
            return d["kernels"]["k_phase"]["hbm_bytes_fetch_x2"]
    return None


def cpu_baseline(cfg, prm, n_sample):
    """The CPU oracle timed on this box's host cores on the first n_sample chains
    of the same workload.  This is the checker used as a yardstick, never part
    of the measured GPU path."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from ahsoka_amd import capi

    c = dict(cfg)
    c["n_chains"] = n_sample
    c["first_chain"] = 0
    b = capi.synth_batch(**c)
    cores = max(1, min(n_sample, os.cpu_count() or 1))
    t0 = time.perf_counter()
    r = ob.phase_batch(b, prm, n_threads=cores)
    dt = time.perf_counter() - t0
    bubbles = r.phased_bubbles(b.arrays()["bubble_off"])
    # the reference runs one thread (src/polyassembly.cpp:167-172): the same port on one core,
    # on a smaller sample
    n1 = max(1, min(4, n_sample))
    c1 = dict(c)
    c1["n_chains"] = n1
    b1 = capi.synth_batch(**c1)
    t1 = time.perf_counter()
    r1 = ob.phase_batch(b1, prm, n_threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": bubbles / dt,
        "unit": "bubbles/s",
        "cores": cores,
        "value_1thread": r1.phased_bubbles(b1.arrays()["bubble_off"]) / dt1,
        "sample_1thread": f"first {n1} chains, 1 thread, {dt1:.2f} s",
        "kind": "port",
        "sample": f"first {n_sample} chains of the same workload ({bubbles} bubbles) in {dt:.2f} s wall, "
                  f"{cores} threads; CPU re-statement (oracle/oracle.cpp), not the upstream binary",
    }


if __name__ == "__main__":
    main()
