#!/usr/bin/env python3
"""bench.py -- bubbles/s phased on the BASELINE.json headline workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path (k_readset + k_phase, then the result
download and, for N > 1, the single gather to rank 0) over one synthetic batch
whose inputs are already resident in HBM.  Workload at every N: BASELINE.json
configs[2] -- 10k chains x 50 bubbles x 1000 reads/chain, ploidy 2 -- PER GPU
(weak scaling: rank r generates chains [r*10k, (r+1)*10k) of one global
deterministic workload; no data-path collective).

Prints ONE JSON line on rank 0 (contract in the task description), including
  roofline      dominant kernel (k_phase): algorithmic bytes / its mean HIP-event
                duration on the launch stream, against 8 TB/s HBM
  cpu_baseline  the CPU oracle ("port": oracle/oracle.cpp, NOT the upstream
                binary, which cannot be built here) on a bounded sample of the
                same workload, timed on this box's host cores.
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="c3", help="BASELINE config (c3 = headline); others for experiments")
    ap.add_argument("--chains", type=int, default=0, help="override chains per GPU (experiments only)")
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
    per_gpu = args.chains if args.chains > 0 else cfg["n_chains"]
    cfg["n_chains"] = per_gpu
    cfg["first_chain"] = rank * per_gpu
    ploidy = cfg["ploidy"]
    t_gen = time.time()
    batch = capi.synth_batch(**cfg)
    t_gen = time.time() - t_gen
    arr = batch.arrays()
    bubble_off = arr["bubble_off"]
    max_pos = int(np.diff(bubble_off).max())
    prm = capi.default_params(ploidy=ploidy, device=local_rank)

    # host buffers -> HBM once; the timed region starts with inputs resident
    t_up = time.time()
    dev = capi.DeviceBatch(batch, prm)
    t_up = time.time() - t_up
    owned = np.arange(rank * per_gpu, (rank + 1) * per_gpu)

    def step():
        dev.phase()
        res = dev.download()
        gathered = None
        if world > 1:
            # the one collective of the path: every rank's packed result block
            # (device memory, fixed layout) to rank 0 -- RCCL gather over xGMI
            # (.clone(): the collective then works on memory torch itself allocated; the
            # library's block lives in the system HIP runtime's allocation table)
            block = dev.result_tensor()
            gathered = shard.gather_device_blocks(block.clone() if backend == "nccl" else block.cpu())
        return res, gathered

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    k_phase_ms, k_readset_ms = [], []
    fence()
    t0 = time.perf_counter()
    res = None
    for _ in range(args.steps):
        res, gathered = step()
        kt = dev.kernel_times_ms()
        k_phase_ms.append(kt["k_phase"])
        k_readset_ms.append(kt["k_readset"])
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    phased = res.phased_bubbles(bubble_off)
    if world > 1:
        t = torch.tensor([phased], dtype=torch.int64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        total_phased = int(t.item())
    else:
        total_phased = phased
    ms_per_step = elapsed * 1e3 / args.steps
    value = total_phased / (elapsed / args.steps)

    if rank == 0:
        # algorithmic bytes of one launch: inputs once + outputs once
        P_total = int(res.pos_off[-1])
        n_phased = int((res.status == capi.ST_PHASED).sum())
        alg_bytes = batch.algorithmic_input_bytes(ploidy) + 4 * (P_total + ploidy * P_total + n_phased)
        kp = float(np.mean(k_phase_ms))
        achieved = alg_bytes / (kp * 1e-3) / 1e9
        traffic = pmc_traffic(args.config, per_gpu)
        stream = stream_copy_gbps(torch, device)
        out = {
            "metric": "bubbles/s phased (ploidy 2, 1000 reads/chain)",
            "value": value,
            "unit": "bubbles/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int64",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[2]: {per_gpu} chains x {cfg['n_bubbles']} bubbles x "
                            f"{cfg['n_reads']} reads/chain, ploidy {ploidy}, per GPU",
                "config_name": args.config,
                "chains_per_gpu": per_gpu,
                "parallelism": f"chain-shard x{world}, one gather",
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
                "k_readset_ms": float(np.mean(k_readset_ms)),
                # the measured traffic (committed PMC summary of this workload) over this run's kernel
                # time, against the nominal peak and against a streaming copy measured in this job
                "traffic_rate_GBps": (traffic / (kp * 1e-3) / 1e9) if traffic else None,
                "stream_copy_GBps": stream,
            },
            "phased_bubbles_per_step": total_phased,
            "phased_chains_rank0": n_phased,
            "upload_s": t_up,
            "synth_s": t_gen,
            "backend": capi.backend_info(),
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, prm, args.cpu_sample)
        print(json.dumps(out), flush=True)
    dev.free()
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
