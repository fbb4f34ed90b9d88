#!/usr/bin/env python3
"""bench.py — headline benchmark of the ICP pair-registration hot path on MI355X.

Metric (BASELINE.json): point-pairs/sec per ICP iteration on a 1M <-> 1M cloud pair.
One "step" = one complete pair registration from clouds already resident in HBM:
target index build (the analogue of PCL's per-align kd-tree build) + source load +
`iterations` fixed ICP iterations (NN search, gate, 17 sums, host Umeyama/SVD, transform).
value = N_src * iterations * steps / wall time of the timed region (max over ranks).

N = 1 : config.workload "icp_pair_1Mx1M_30it" (the configuration the metric is quoted on).
N > 1 : BASELINE configs[3] — the same single pair, source sharded in N contiguous blocks,
        target replicated, one RCCL all-reduce of 17 doubles per iteration ("strong" scaling).

Launch for N > 1 (the driver does this):
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
      --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", default=None, choices=["50k", "N300", "N1M"],
                    help="points per cloud (default: N1M for the pair workload -- the configuration the metric is quoted on -- and N300 for the chain, configs[4])")
    ap.add_argument("--iterations", type=int, default=30)
    ap.add_argument("--max-dist", type=float, default=0.05)
    ap.add_argument("--pipeline", type=int, default=2,
                    help="0 staged kernels, 1 fused kernel + host 3x3 solve per iteration, 2 fused kernel + device-resident loop")
    ap.add_argument("--shard-block", type=int, default=256, help="N > 1: source points per block, blocks dealt round-robin to the ranks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the warm-up and the timed headline steps (no pipeline-1, from-identity, events-in-all-steps, host-buffer or "
                         "reference-mode legs, no CPU baseline): what rocprofv3 is pointed at, so that its per-kernel means are the headline's")
    ap.add_argument("--force-dist", action="store_true",
                    help="dev: take the multi-rank code path (process group, native RCCL transport, all-reduce per iteration) "
                         "even with one rank, e.g. under `torch.distributed.run --nproc-per-node 1`")
    ap.add_argument("--no-events", action="store_true", help="dev: run without the per-kernel HIP events (no roofline numbers)")
    ap.add_argument("--event-every", type=int, default=10,
                    help="the per-launch HIP events of the roofline leg are recorded in every E-th step of the timed region (two records "
                         "between every two dependent launches cost 0.22 ms of a 3.4 ms step: in all steps they would lower `value` by 6 %%); 1: in all")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: if the native RCCL transport (rsreg_comm_init) cannot be set up, carry the 17 sums over torch.distributed "
                         "instead of exiting with an error (the line then says so in config.sharding)")
    ap.add_argument("--cpu-iterations", type=int, default=10)
    ap.add_argument("--devices-per-node", type=int, default=0,
                    help="dev: rank r runs on device r %% D (0: one device per rank, the driver's launch).  With more ranks than devices "
                         "the process group is gloo (RCCL refuses two ranks on one device) and the 17 sums travel over torch.distributed: "
                         "a rehearsal of the N > 1 code path on a one-GPU box, never a scaling number")
    ap.add_argument("--workload", default="pair", choices=["pair", "chain"],
                    help="pair: ONE 1M pair, source sharded over the ranks (configs[1]/[3], the headline line); "
                         "chain: 2 x N frames of 300k points as consecutive pairs, one pair per GPU at a time (configs[4])")
    ap.add_argument("--frames", type=int, default=0,
                    help="chain: frames of the chain (0: 2 x N, about two pairs per GPU whatever N -- weak scaling; 16: configs[4] as written, "
                         "its 15 pairs dealt to the ranks)")
    ap.add_argument("--in-flight", type=int, default=0,
                    help="chain: pairs registered SIDE BY SIDE on every GPU, each on a context of its own driven by its own thread "
                         "(rsreg_amd/chain.py: ChainRegistrar; the frames are read where they lie, through raw device pointers).  "
                         "0: the rank's pairs one after the other on one context, through cloud handles (rounds 2-5)")
    a = ap.parse_args()
    if a.size is None and a.workload == "pair":
        a.size = "N1M"
    if a.headline_only:
        a.no_cpu_baseline = True
    return a


def run_chain(a, rank, world, local_rank, dist, coll_dev="cuda"):
    """BASELINE configs[4]: a chain of frames as independent consecutive pairs (k-1, k), dealt round-robin
    to the ranks, pair transforms gathered and composed on the host (rsreg_amd/chain.py).  Weak scaling:
    2 x N frames (16 at N = 8), i.e. about two pairs per GPU whatever N.  No collective on the data path."""
    import ctypes as C

    import torch

    from rsreg_amd import api, chain, lib, synth

    size = a.size or "N300"     # the config's frame size unless another one is asked for
    n_frames = a.frames if a.frames >= 2 else 2 * world
    mine = chain.pair_assignment(n_frames, rank, world)
    ctx = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream, profiling=not a.no_events)
    need = sorted({k for k in mine} | {k - 1 for k in mine})
    host = {k: synth.render_frame(k, size, "bench") for k in need}
    dev = {k: api.DeviceCloud(host[k], ctx) for k in need}      # frames resident in HBM before the clock starts
    perturb = synth.small_transform(0.5, (0.005, -0.004, 0.005))   # the guess a pose prior would give: truth off by 0.5 deg / 8 mm
    guess = {k: np.ascontiguousarray((perturb @ synth.ground_truth(k, k - 1, "bench")).astype(np.float32).T) for k in mine}
    L = lib.lib()
    prm = api.icp_params(max_iterations=a.iterations, criteria_mode=1, pipeline_mode=a.pipeline, max_correspondence_distance=a.max_dist)
    res = lib.IcpResult()
    gi = lib.GridInfo()
    n_pts = len(next(iter(host.values()))) if host else 0

    def allgather(buf):
        if dist is None:
            return [buf]
        t = torch.from_numpy(buf).to(coll_dev)
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return [o.cpu().numpy() for o in outs]

    stats = {"ms_nn": 0.0, "launch": 0, "ms_build": 0.0}
    registrar = shared = None
    if a.in_flight >= 1:
        registrar = chain.ChainRegistrar(local_rank, a.in_flight, params=prm)
        ctx.synchronize()
        shared = {k: chain.ChainRegistrar.share(dev[k]) for k in need}     # settled: any stream may read them from here on
        stats_lock = __import__("threading").Lock()

        def collect_pair(k, r, c):
            g = lib.GridInfo()
            L.rsreg_icp_grid_info(c.h, C.byref(g))
            ht = lib.HostTiming()
            L.rsreg_ctx_host_timing(c.h, C.byref(ht))
            with stats_lock:
                stats["ms_nn"] += r.ms_nn
                stats["launch"] += r.n_nn_launches
                stats["ms_build"] += g.ms_build
                stats["ms_enqueue"] = stats.get("ms_enqueue", 0.0) + ht.loop_enqueue
                stats["pairs"] = stats.get("pairs", 0) + 1

    def step(collect=False):
        if registrar is not None:
            local = registrar.register(shared, mine, guess, collect=collect_pair if collect else None)
            return chain.compose_chain(chain.gather_pairs(local, n_frames, allgather), n_frames), local
        local = {}
        for k in mine:
            lib.check(L.rsreg_icp_set_target_cloud(ctx.h, dev[k - 1].h, a.max_dist), ctx.h)
            lib.check(L.rsreg_icp_set_source_cloud(ctx.h, dev[k].h), ctx.h)
            lib.check(L.rsreg_icp_align(ctx.h, guess[k].ctypes.data, C.byref(prm), C.byref(res), None, 0), ctx.h)
            local[k] = api._rowmajor(res.transform)
            if collect:
                stats["ms_nn"] += res.ms_nn
                stats["launch"] += res.n_nn_launches
                L.rsreg_icp_grid_info(ctx.h, C.byref(gi))
                stats["ms_build"] += gi.ms_build
        return chain.compose_chain(chain.gather_pairs(local, n_frames, allgather), n_frames), local

    def set_profiling(on):
        for c in ([ctx] if registrar is None else registrar.contexts):
            c.set_profiling(on)

    def sync():
        if registrar is not None:
            for c in registrar.contexts:
                c.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    every = max(1, a.event_every)
    set_profiling(False)
    t0 = time.perf_counter()
    for k in range(a.steps):
        timed = (not a.no_events) and k % every == 0     # (this step carries the per-launch HIP events: bench.py --event-every)
        if timed:
            set_profiling(True)
        poses, local = step(collect=timed)
        if timed:
            set_profiling(False)
    sync()
    elapsed = time.perf_counter() - t0
    L.rsreg_icp_grid_info((registrar.contexts[0] if registrar is not None else ctx).h, C.byref(gi))
    # the same step with records the handles have never measured under the SOURCE frames (re-uploaded with the clock stopped): in the
    # timed region above every handle keeps the bounding box an earlier load measured, which a stream of new frames has for its
    # targets (last pair's sources) but never for its sources
    fresh_s, fresh_n = 0.0, (min(3, a.steps) if registrar is None else 0)   # (raw device pointers carry no measured box: every step is "fresh")
    seq_diff = None
    if registrar is not None:
        # every pair once more, alone and through the cloud handles (the path of rounds 2-5): the bits must be the same
        keep = registrar
        registrar = None
        _, seq_local = step()
        registrar = keep
        seq_diff = float(max(np.abs(np.asarray(local[k], np.float64) - np.asarray(seq_local[k], np.float64)).max() for k in mine)) if mine else 0.0
    for _ in range(fresh_n):
        for k in mine:
            dev[k].upload(host[k])
        sync()
        tf = time.perf_counter()
        step()
        sync()
        fresh_s += time.perf_counter() - tf
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank != 0:
        return None
    n_pairs = n_frames - 1
    pairs = float(n_pts) * a.iterations * n_pairs * a.steps
    n_unique, n_distinct = int(gi.n_unique_points), int(gi.n_source_distinct) or n_pts
    alg_bytes = 32 * n_distinct + 16 * n_unique
    avg_ms = stats["ms_nn"] / max(stats["launch"], 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    gt = [np.linalg.inv(synth.frame_pose(0, "bench")) @ synth.frame_pose(k, "bench") for k in range(n_frames)]
    out = {
        "metric": "point-pairs/sec per ICP iteration", "value": pairs / elapsed, "unit": "point-pairs/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "icp_chain_%dx%s_%dit_consecutive_pairs" % (n_frames, size, a.iterations), "n_frames": n_frames,
            "points_per_frame": n_pts, "pairs": n_pairs, "in_flight": a.in_flight, "iterations": a.iterations, "max_corr_dist": a.max_dist, "criteria": "fixed",
            "sharding": "pairs (k-1, k) dealt round-robin to %d ranks, no collective on the data path; 4x4s gathered and composed on the host" % world,
            "step": "every pair: index build + source load + %d iterations from frames resident in HBM; gather + composition of the chain" % a.iterations +
                    ("; %d pairs side by side per GPU, a context and a thread each (chain.ChainRegistrar)" % a.in_flight if a.in_flight >= 1 else ""),
            "deviation": "consecutive pairs instead of the reference's frame-to-model chain (incremental_icp.hpp:51-66 is sequential)",
        },
        "roofline": {"bound": "hbm", "kernel": "k_icp_fused_dense", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_source": None, "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_ms": avg_ms, "launches": stats["launch"]},
        "chain_pose_error_vs_ground_truth_frobenius_max": float(max(np.linalg.norm(poses[k] - gt[k]) for k in range(n_frames))),
        "ms_per_step_fresh_source_frames": (fresh_s / fresh_n * 1e3) if fresh_n else None,
        "ms_per_pair": elapsed / a.steps * 1e3 / max(1, len(mine)),
        "pairs_on_rank0": len(mine),
        # in flight: the per-launch HIP events of the roofline leg time launches that SHARE the GPU with other alignments' kernels
        "roofline_note": ("launch durations measured with %d alignments in flight: they overlap, their sum exceeds the wall time" % a.in_flight)
                         if a.in_flight > 1 else None,
        "in_flight_vs_sequential_max_abs_diff": seq_diff,
        # what queueing one alignment's device loop (2 launches per iteration) took its host thread, mean over the instrumented steps
        "host_ms_to_queue_one_alignments_loop": (stats["ms_enqueue"] / stats["pairs"]) if stats.get("pairs") else None,
    }
    if not a.no_cpu_baseline:
        import oracle  # cpu_baseline leg: the oracle as the timed CPU port, never the product

        oracle.build()
        k = mine[0]
        o = oracle.IcpOracle()
        p = oracle.IcpParams.default()
        p.max_iterations, p.criteria_mode, p.max_correspondence_distance, p.num_threads = a.cpu_iterations, 1, a.max_dist, 1
        tc = time.perf_counter()
        o.set_target(host[k - 1].points, dedup=True)
        o.set_source(host[k].points)
        r = o.align(guess[k].T, p)
        cpu_s = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": float(n_pts) * a.cpu_iterations / cpu_s, "unit": "point-pairs/s", "cores": 1, "kind": "port",
                               "sample": "pair (%d, %d) of the chain, kd-tree build + %d of %d iterations, 1 thread" % (k - 1, k, a.cpu_iterations, a.iterations),
                               "seconds": cpu_s, "host_cpus": os.cpu_count()}
        prm2 = api.icp_params(max_iterations=a.cpu_iterations, criteria_mode=1, pipeline_mode=a.pipeline, max_correspondence_distance=a.max_dist)
        lib.check(L.rsreg_icp_set_target_cloud(ctx.h, dev[k - 1].h, a.max_dist), ctx.h)
        lib.check(L.rsreg_icp_set_source_cloud(ctx.h, dev[k].h), ctx.h)
        lib.check(L.rsreg_icp_align(ctx.h, guess[k].ctypes.data, C.byref(prm2), C.byref(res), None, 0), ctx.h)
        out["transform_error_vs_cpu_frobenius"] = float(np.linalg.norm(api._rowmajor(res.transform) - r.T))
    return out


def main():
    a = parse()
    import torch

    import rsreg_amd
    from rsreg_amd import api, lib, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (a.gpus, a.gpus))
        a.gpus = world
    if not os.path.exists(lib.SO_PATH):
        lib.build()
    if not torch.cuda.is_available() or api.device_count() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    n_dev = torch.cuda.device_count()
    per_node = a.devices_per_node if a.devices_per_node > 0 else max(n_dev, 1)
    shared_devices = world > per_node          # several ranks on one device: a rehearsal (--devices-per-node), gloo
    if a.devices_per_node == 0 and local_rank >= n_dev:
        raise SystemExit("rank %d has no device of its own (%d visible); --devices-per-node D rehearses N > D ranks on D devices" % (local_rank, n_dev))
    device = local_rank % per_node
    local_rank = device                        # (what every Context below is created on)
    torch.cuda.set_device(device)
    dist = None
    multi = world > 1 or a.force_dist
    if multi:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")

        if shared_devices:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    coll_dev = "cpu" if shared_devices else "cuda"   # where the tensors of the bench's own collectives live

    if a.workload == "chain":
        out = run_chain(a, rank, world, local_rank, dist, coll_dev)
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- synthetic D435i-like pair, identical on every rank
    tgt = synth.render_frame(0, a.size, "bench")
    src = synth.render_frame(1, a.size, "bench")
    guess = synth.small_transform(1.0, (0.008, -0.004, 0.006)).astype(np.float32)
    n_src_total, n_tgt = len(src), len(tgt)
    from rsreg_amd import sharded

    mine = sharded.shard_blocks(n_src_total, rank, world, a.shard_block)   # all points when world == 1
    d_tgt = torch.from_numpy(tgt.points.view(np.uint8).reshape(-1)).cuda()
    d_src = torch.from_numpy(np.ascontiguousarray(src.points[mine]).view(np.uint8).reshape(-1)).cuda()
    n_src = len(mine)
    stride = tgt.points.dtype.itemsize

    ctx = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream, profiling=not a.no_events)
    transport = "none"
    if multi:
        # native transport: RCCL all-reduce of the 17 sums on the ctx stream inside rsreg_icp_align.
        # If the communicator cannot be set up, all ranks agree to fall back to the step-wise
        # driver whose all-reduce is torch.distributed (also RCCL on ROCm).
        ok = 1
        try:
            if shared_devices:
                raise RuntimeError("%d ranks on %d device(s): RCCL refuses two ranks on one device" % (world, per_node))
            uid = torch.zeros(lib.UNIQUE_ID_BYTES, dtype=torch.uint8, device=coll_dev)
            if rank == 0:
                uid = torch.frombuffer(bytearray(api.comm_unique_id()), dtype=torch.uint8).to(coll_dev)
            dist.broadcast(uid, 0)
            ctx.comm_init(bytes(uid.cpu().numpy().tobytes()), rank, world)
        except Exception as e:  # noqa: BLE001
            print("[bench] rank %d: native RCCL transport unavailable (%s)" % (rank, e), file=sys.stderr)
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        transport = "rccl-native" if int(flag.item()) == 1 else ("torch-distributed (gloo: %d ranks on %d device(s), a rehearsal)" % (world, per_node)
                                                                   if shared_devices else "torch-distributed")
        if transport != "rccl-native" and ok:
            lib.lib().rsreg_comm_destroy(ctx.h)
        if transport != "rccl-native" and not a.allow_fallback:
            # the line the driver records must be the transport BASELINE configs[3] names; a silent change of transport
            # would be measured as if it were RCCL over xGMI
            if rank == 0:
                print("[bench] --gpus %d: the native RCCL transport is not available on every rank; re-run with --allow-fallback to "
                      "measure the torch.distributed transport instead" % world, file=sys.stderr)
            dist.barrier()
            dist.destroy_process_group()
            raise SystemExit(3)

    import ctypes as C

    L = lib.lib()
    prm = api.icp_params(max_iterations=a.iterations, criteria_mode=1, pipeline_mode=a.pipeline,
                         max_correspondence_distance=a.max_dist)
    g = np.ascontiguousarray(guess.T).copy()
    res = lib.IcpResult()
    gi = lib.GridInfo()

    if transport.startswith("torch-distributed"):
        stepper = api.IterativeClosestPoint(ctx)
        stepper.params = prm
        stepper._quiet_search = True

        def allreduce(v):
            t = torch.from_numpy(v).to(coll_dev)
            dist.all_reduce(t)
            return t.cpu().numpy()

        def step():
            nonlocal res
            stepper.setInputTargetDevice(d_tgt.data_ptr(), n_tgt, stride)
            stepper.setInputSourceDevice(d_src.data_ptr(), n_src, stride)
            res = sharded.run_sharded_icp(stepper, allreduce, guess)
    else:
        def step():
            # source, then target: the reference's order (incremental_icp.hpp:57-58)
            lib.check(L.rsreg_icp_set_source_device(ctx.h, d_src.data_ptr(), n_src, stride, 0), ctx.h)
            lib.check(L.rsreg_icp_set_target_device(ctx.h, d_tgt.data_ptr(), n_tgt, stride, 0, a.max_dist), ctx.h)
            lib.check(L.rsreg_icp_align(ctx.h, g.ctypes.data, C.byref(prm), C.byref(res), None, 0), ctx.h)

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    ms_nn = ms_red = ms_build = ms_allreduce = 0.0
    n_launch = n_event_steps = 0
    every = max(1, a.event_every)
    ctx.set_profiling(False)
    t0 = time.perf_counter()
    for k in range(a.steps):
        timed = (not a.no_events) and k % every == 0     # (this step carries the per-launch HIP events)
        if timed:
            ctx.set_profiling(True)
        step()
        if timed:
            ms_nn += res.ms_nn
            ms_red += res.ms_reduce + res.ms_transform
            ms_allreduce += res.ms_allreduce
            n_launch += res.n_nn_launches
            L.rsreg_icp_grid_info(ctx.h, C.byref(gi))
            ms_build += gi.ms_build
            n_event_steps += 1
            ctx.set_profiling(False)
    sync()
    elapsed = time.perf_counter() - t0
    if not n_event_steps:
        L.rsreg_icp_grid_info(ctx.h, C.byref(gi))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    T_gpu = api._rowmajor(res.transform)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    pairs = float(n_src_total) * a.iterations * a.steps
    value = pairs / elapsed
    # roofline of the dominant kernel (fused search+sums in pipeline 1, plain search in pipeline 0;
    # the _dense variants run when the target index is the dense cell-start table):
    # algorithmic bytes per launch, DESIGN.md §5: fused 32*N_src' + 16*N_tgt', staged 24*N_src' + 16*N_tgt'
    # (per rank; N' = distinct points actually resident -- exact copies are merged on load)
    n_unique = int(gi.n_unique_points)
    n_distinct = int(gi.n_source_distinct) or n_src
    kern = ("k_icp_fused" if a.pipeline >= 1 else "k_nn_search") + ("_dense" if gi.index_kind == 1 else "")
    alg_bytes = (32 if a.pipeline >= 1 else 24) * n_distinct + 16 * n_unique
    avg_ms = ms_nn / max(n_launch, 1)
    achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside
    # this process, so the number comes from the committed rocprofv3 passes of this same
    # command (profiles/*_traffic.json, collected per MI355X_MICROARCH.md's HBM section)
    traffic = None
    traffic_source = None
    try:
        tj = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
        if tj:
            t = json.load(open(os.path.join(ROOT, "profiles", tj[-1])))
            w = t.get("workload", {})
            if (w.get("size") == a.size and w.get("iterations") == a.iterations and w.get("max_dist") == a.max_dist
                    and w.get("pipeline") == ("fused" if a.pipeline >= 1 else "staged") and world == 1 and kern in t["kernels"]):
                traffic = t["kernels"][kern]["traffic_bytes_corrected"]
                traffic_source = "profiles/%s (static: rocprofv3 --pmc passes of this command, not measured in this run)" % tj[-1]
    except Exception:  # noqa: BLE001
        traffic = None
    out = {
        "metric": "point-pairs/sec per ICP iteration",
        "value": value,
        "unit": "point-pairs/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "icp_pair_%sx%s_%dit" % (a.size, a.size, a.iterations),
            "n_src": n_src_total, "n_tgt": n_tgt, "iterations": a.iterations, "max_corr_dist": a.max_dist,
            "criteria": "fixed",
            "pipeline": ["staged", "fused, 3x3 solve on the host each iteration", "fused, device-resident loop (3x3 solve on the GPU)"][a.pipeline],
            "sharding": "source blocks of %d points dealt round-robin to %d ranks, all-reduce of 17 f64 per iteration (%s)"
                        % (a.shard_block, world, transport) if world > 1 else "none",
            "step": "grid build + source load + %d iterations, inputs resident in HBM" % a.iterations,
        },
        "roofline": {
            "bound": "hbm", "kernel": kern,
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms,
            "launches": n_launch,
            "events": "HIP events around every search launch of %d of the %d timed steps (every %d-th; their two records between "
                      "every two dependent launches cost 0.22 ms of a step, so the other steps run without them)" % (n_event_steps, a.steps, every),
        },
        "breakdown_ms_per_step": {
            "grid_build": ms_build / max(n_event_steps, 1), "nn_kernel": ms_nn / max(n_event_steps, 1), "reduce_transform": ms_red / max(n_event_steps, 1),
            # N > 1, native transport: the ncclAllReduce of the 17 sums between HIP events on the ctx stream, all iterations of a step
            # (inside reduce_transform); None where nothing was all-reduced by the library (one rank, or the torch.distributed transport)
            "allreduce": (ms_allreduce / max(n_event_steps, 1)) if (multi and transport == "rccl-native") else None,
            "iteration_rate_pairs_per_s": float(n_src_total) * a.iterations * max(n_event_steps, 1) / max((ms_nn + ms_red) * 1e-3, 1e-12),
            "of": "the %d steps that carry the events" % n_event_steps,
        },
        "grid": {"kind": "dense cell-start table" if gi.index_kind == 1 else "brick hash", "cell_size": float(gi.cell_size),
                 "n_cells": int(gi.n_cells), "n_unique_points": n_unique, "n_source_distinct": n_distinct,
                 "index_bytes": int(gi.index_bytes)},
    }

    # the roofline that binds (profiles/*_issue.json: SQ counters of this command + the VALU issue microbenchmark, both
    # collected on MI355X; PMC counters cannot be read from inside this process)
    try:
        ij = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_issue.json"))
        if ij and world == 1 and a.size == "N1M" and a.pipeline >= 1:
            t = json.load(open(os.path.join(ROOT, "profiles", ij[-1])))
            bound_us = t["valu_insts"] * t["ns_per_valu_inst_per_simd"] / t["simds"] * 1e-3
            out["roofline_issue"] = {
                "bound": "valu-issue", "kernel": kern, "valu_insts": t["valu_insts"], "salu_insts": t["salu_insts"], "vmem_insts": t["vmem_insts"],
                "cycles_per_inst": t["cycles_per_inst_at_2p4ghz"], "ns_per_inst_per_simd": t["ns_per_valu_inst_per_simd"],
                "cycles_per_inst_source": t["cycles_per_inst_source"], "simds": t["simds"], "bound_us": bound_us,
                "avg_launch_us": avg_ms * 1e3, "frac": bound_us / (avg_ms * 1e3) if avg_ms > 0 else None,
                "lane_utilisation": t.get("lane_utilisation"),
                # the other issue-side resource, and the one the launch time actually follows (profiles/r03_what_bounds_the_kernel.txt):
                # vector memory wave-instructions through the CU's address unit
                "vmem_cycles_per_inst": t.get("vmem_cycles_per_inst"), "vmem_bound_us": t.get("vmem_bound_us"),
                "vmem_frac": (t["vmem_bound_us"] / (avg_ms * 1e3)) if t.get("vmem_bound_us") and avg_ms > 0 else None,
                "vmem_cycles_source": t.get("vmem_cycles_source"),
                "binding": "neither alone: +49 % VALU per trip costs +5 %, +100 % vector memory instructions per trip +38 % "
                           "(profiles/r03_what_bounds_the_kernel.txt): the launch follows the vector memory instruction count and the "
                           "dependent L2 round trip behind every trip",
                "source": "profiles/%s (static: rocprofv3 --pmc SQ_* passes of this command + tools/microbench/valu_issue.hip; the launch "
                          "time is this run's)" % ij[-1],
            }
            # where the launch's time goes while the chip is full, and how much of it is the emptying tail
            # (profiles/*_wave_timeline.json: start / end stamp of every wave of a scheduled launch, tools/wave_timeline.py)
            wj = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_wave_timeline.json"))
            if wj and t.get("vmem_bound_us"):
                w = json.load(open(os.path.join(ROOT, "profiles", wj[-1])))
                # the address units' cycles are spent where the wave time is: the share of the wave time that falls
                # into the period with every wave slot taken, over that period
                out["roofline_issue"]["vmem_frac_while_full"] = t["vmem_bound_us"] * w["slots"] / w["sum_wave_time_us"]
                # (above 1: the kernel's own instruction mix issues faster than the candidate-scoring sequence the
                # microbenchmark prices -- `bound_us` is an upper bound of the VALU time, not the VALU time)
                out["roofline_issue"]["valu_frac_while_full_upper_bound"] = bound_us * w["slots"] / w["sum_wave_time_us"]
                out["roofline_issue"]["tail_fraction"] = 1.0 - w["sum_wave_time_us"] / (w["slots"] * w["span_us"])
                out["roofline_issue"]["wave_timeline"] = {"slots": w["slots"], "span_us": w["span_us"], "sum_wave_time_us": w["sum_wave_time_us"],
                                                          "slots_full_until_us": w.get("full_until_us"), "waves": w.get("waves"),
                                                          "source": "profiles/%s (static)" % wj[-1]}
                # the numbers that bind first, the VALU figure (an upper bound of the VALU time over the WHOLE launch: see
                # valu_frac_while_full_upper_bound) under a name that says so
                ri = out["roofline_issue"]
                # The denominator that has a meaning for this kernel (no single-resource roofline binds a search of
                # dependent gathers): a SCHEDULE bound -- the full phase at the address units' floor (every vector-memory wave-
                # instruction at its 16-clock minimum over 256 CUs: vmem_bound_us) plus one light wave's chain of dependent round
                # trips on an EMPTY machine as the last job (12 us: the whole 50 k launch is 31 us, profiles/r05_launch_times.txt).
                light_wave_us = 12.0
                ri["schedule_bound_us"] = t["vmem_bound_us"] + light_wave_us
                ri["frac_of_schedule_bound"] = ri["schedule_bound_us"] / (avg_ms * 1e3) if avg_ms > 0 else None
                ri["schedule_bound_terms"] = {"full_phase_at_address_unit_floor_us": t["vmem_bound_us"], "last_light_wave_alone_us": light_wave_us,
                                              "measured_full_phase_us": w["sum_wave_time_us"] / w["slots"],
                                              "note": "frac_of_schedule_bound = schedule_bound_us / avg_launch_us: 1.0 would be a launch whose full phase "
                                                      "runs at the vector-memory issue floor and whose tail is one light wave"}
                ri["valu_bound_over_launch_upper_bound"] = ri.pop("frac")
                ri["bound"] = "vector memory issue while the wave slots are full, then an emptying tail (the VALU figure is an upper bound)"
                lead = ("bound", "kernel", "schedule_bound_us", "frac_of_schedule_bound", "vmem_frac_while_full", "tail_fraction", "avg_launch_us",
                        "vmem_bound_us", "vmem_frac")
                out["roofline_issue"] = {**{k: ri[k] for k in lead if k in ri}, **{k: v for k, v in ri.items() if k not in lead}}
            out["roofline"]["binding"] = "not HBM (the working set lives in the Infinity Cache): see roofline_issue (the contract's roofline stays the HBM one)"
    except Exception:  # noqa: BLE001
        pass
    # the index build as a rate: SURVEY.md §8d prices it at 116 algorithmic bytes per target point
    if ms_build > 0:
        gb = 116.0 * n_tgt / (ms_build / max(n_event_steps, 1) * 1e-3) / 1e9
        out["grid_build"] = {"ms": ms_build / max(n_event_steps, 1), "algorithmic_bytes": 116 * n_tgt, "achieved": gb, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                             "frac": gb / HBM_PEAK_GBS}
    if world == 1 and a.pipeline == 2 and not a.headline_only:
        # the north star's default pipeline next to the headline: fused kernel, 136 bytes to the host and a host 3x3
        # solve per iteration (pipeline 1), without the HIP events of the roofline leg
        ctx1 = api.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream, profiling=False)
        prm1 = api.icp_params(max_iterations=a.iterations, criteria_mode=1, pipeline_mode=1, max_correspondence_distance=a.max_dist)
        res1 = lib.IcpResult()

        def step1():
            lib.check(L.rsreg_icp_set_source_device(ctx1.h, d_src.data_ptr(), n_src, stride, 0), ctx1.h)
            lib.check(L.rsreg_icp_set_target_device(ctx1.h, d_tgt.data_ptr(), n_tgt, stride, 0, a.max_dist), ctx1.h)
            lib.check(L.rsreg_icp_align(ctx1.h, g.ctypes.data, C.byref(prm1), C.byref(res1), None, 0), ctx1.h)

        for _ in range(2):
            step1()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps1 = max(3, min(a.steps, 10))
        for _ in range(reps1):
            step1()
        torch.cuda.synchronize()
        out["pipeline1_ms_per_step"] = (time.perf_counter() - t1) / reps1 * 1e3
        out["pipeline1_same_transform"] = bool((api._rowmajor(res1.transform) == T_gpu).all())
        # does ICP move the pair at all?  The headline's guess is two thirds of the way there (guess 1.0 degree / (8, -4, 6) mm against the
        # preset's motion of 1.5 degrees / (12, -6, 9) mm: a third of the motion off), so its 30 iterations run mostly in the settled,
        # seeded regime.  From the identity (the whole motion off) the same 30 iterations must close most of the gap -- and that
        # alignment is TIMED here too, the same step as the headline's (source load + index build + 30 iterations), no events
        gt_ = synth.ground_truth(1, 0, "bench")
        eye = np.ascontiguousarray(np.eye(4, dtype=np.float32))
        prm2_ = api.icp_params(max_iterations=a.iterations, criteria_mode=1, pipeline_mode=2, max_correspondence_distance=a.max_dist)

        def step_eye():
            lib.check(L.rsreg_icp_set_source_device(ctx1.h, d_src.data_ptr(), n_src, stride, 0), ctx1.h)
            lib.check(L.rsreg_icp_set_target_device(ctx1.h, d_tgt.data_ptr(), n_tgt, stride, 0, a.max_dist), ctx1.h)
            lib.check(L.rsreg_icp_align(ctx1.h, eye.ctypes.data, C.byref(prm2_), C.byref(res1), None, 0), ctx1.h)

        for _ in range(2):
            step_eye()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps1):
            step_eye()
        torch.cuda.synchronize()
        eye_ms = (time.perf_counter() - t1) / reps1 * 1e3
        out["convergence_from_identity"] = {
            "initial_error_frobenius": float(np.linalg.norm(np.eye(4) - gt_)),
            "final_error_frobenius": float(np.linalg.norm(api._rowmajor(res1.transform) - gt_)), "iterations": int(res1.iterations),
            "ms_per_step": eye_ms, "point_pairs_per_s": float(n_src_total) * a.iterations / (eye_ms * 1e-3), "steps_timed": reps1,
            "note": "same pair, gate and step as the headline, guess = identity (the frame-to-frame motion of the bench preset is "
                    "1.5 degrees + 16 mm; the headline's guess is a third of that motion off)"}
        # the headline's own step with the per-launch HIP events in EVERY step (rounds 1-3 measured `value` that way): the like-for-like
        # column for comparisons across rounds
        if not a.no_events and every > 1:
            ctx.set_profiling(True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps1):
                step()
            torch.cuda.synchronize()
            ev_ms = (time.perf_counter() - t1) / reps1 * 1e3
            ctx.set_profiling(False)
            out["value_with_events_in_all_steps"] = {"value": float(n_src_total) * a.iterations / (ev_ms * 1e-3), "ms_per_step": ev_ms, "steps_timed": reps1,
                                                     "note": "`value` is measured with the events in every %d-th step only" % every}

    if not a.no_cpu_baseline:
        import oracle  # cpu_baseline leg: the oracle as the timed CPU port, never the product

        oracle.build()
        cores = 1
        o = oracle.IcpOracle()
        p = oracle.IcpParams.default()
        p.max_iterations, p.criteria_mode, p.max_correspondence_distance = a.cpu_iterations, 1, a.max_dist
        p.num_threads = cores
        tc = time.perf_counter()
        o.set_target(tgt.points, dedup=True)
        o.set_source(src.points)
        r = o.align(guess, p)
        cpu_s = time.perf_counter() - tc
        cpu_value = float(n_src_total) * a.cpu_iterations / cpu_s
        out["cpu_baseline"] = {
            "value": cpu_value, "unit": "point-pairs/s", "cores": cores, "kind": "port",
            "sample": "same %s pair, kd-tree build + %d of %d iterations, 1 thread (PCL's ICP is single-threaded); "
                      "the reference's PCL build is not available here" % (a.size, a.cpu_iterations, a.iterations),
            "seconds": cpu_s, "host_cpus": os.cpu_count(),
        }
        # same-iteration-count transform agreement (north-star bar 1e-4 Frobenius), untimed
        if world == 1:
            prm2 = api.icp_params(max_iterations=a.cpu_iterations, criteria_mode=1, pipeline_mode=a.pipeline,
                                  max_correspondence_distance=a.max_dist)
            res2 = lib.IcpResult()
            lib.check(L.rsreg_icp_align(ctx.h, g.ctypes.data, C.byref(prm2), C.byref(res2), None, 0), ctx.h)
            out["transform_error_vs_cpu_frobenius"] = float(np.linalg.norm(api._rowmajor(res2.transform) - r.T))
        out["speedup_vs_cpu_port"] = value / cpu_value
        # the same sample with the queries of an iteration spread over the host's cores (OpenMP over queries;
        # the tree build stays serial, as in FLANN) -- what a multi-threaded PCL-like port would reach
        threads = max(1, min(os.cpu_count() or 1, 64))
        if threads > 1:
            p.num_threads = threads
            tc = time.perf_counter()
            o.set_target(tgt.points, dedup=True)
            o.set_source(src.points)
            o.align(guess, p)
            omp_s = time.perf_counter() - tc
            out["cpu_baseline_all_cores"] = {"value": float(n_src_total) * a.cpu_iterations / omp_s, "unit": "point-pairs/s",
                                             "cores": threads, "kind": "port", "seconds": omp_s}
    if world == 1 and not a.headline_only:
        # PCIe-inclusive rate (never `value`): the same pair handed over as HOST buffers, as the
        # reference's call surface does (clouds in host memory in, 4x4 out)
        th = time.perf_counter()
        reps = 3
        for _ in range(reps):
            lib.check(L.rsreg_icp_set_target(ctx.h, tgt.points.ctypes.data, n_tgt, stride, 0, a.max_dist), ctx.h)
            lib.check(L.rsreg_icp_set_source(ctx.h, src.points.ctypes.data, n_src, stride, 0), ctx.h)
            lib.check(L.rsreg_icp_align(ctx.h, g.ctypes.data, C.byref(prm), C.byref(res), None, 0), ctx.h)
        hs = (time.perf_counter() - th) / reps
        out["host_buffers"] = {"ms_per_pair": hs * 1e3, "point_pairs_per_s": float(n_src_total) * a.iterations / hs,
                               "note": "host AoS clouds packed + copied over PCIe inside the call; not used as value"}
    if world == 1 and not a.no_cpu_baseline:
        # the reference's own parameters (one iteration, 1 cm gate: SURVEY.md App. A.4) on a 1M pair inside
        # that gate, per pair: index build + source load + align + aligned cloud.  Host clouds in and out
        # (what the PCL call surface does) against clouds resident in HBM (rsreg_cloud_*: the frame loop of
        # the schemes, nothing crosses PCIe).  Secondary line, never `value`.
        tp, sp = synth.render_frame(0, a.size, "parity"), synth.render_frame(1, a.size, "parity")
        ref = api.IterativeClosestPoint(ctx)
        ref.params = api.icp_params(reference=True)

        def pair(t, s_):
            ref.setInputSource(s_)
            ref.setInputTarget(t)
            return ref.align()

        modes = {}
        for name, (t, s_) in (("host_clouds", (tp, sp)), ("device_clouds", (api.DeviceCloud(tp, ctx), api.DeviceCloud(sp, ctx)))):
            pair(t, s_)
            ctx.synchronize()
            reps = 5
            acc = 0.0
            for _ in range(reps):
                if name == "device_clouds":
                    # fresh records under the handles (outside the clock): a handle keeps the bounding box an earlier load
                    # measured, and a pair of frames that have never been seen has none
                    t.upload(tp)
                    s_.upload(sp)
                    ctx.synchronize()
                t0 = time.perf_counter()
                out_cloud = pair(t, s_)
                ctx.synchronize()
                acc += time.perf_counter() - t0
            modes[name] = acc / reps * 1e3
            modes[name + "_T"] = ref.getFinalTransformation()
            if name == "device_clouds":   # the same pair again and again: both boxes known (a chain's frame k: source, then target)
                t0 = time.perf_counter()
                for _ in range(reps):
                    pair(t, s_)
                ctx.synchronize()
                modes["device_clouds_boxes_known"] = (time.perf_counter() - t0) / reps * 1e3
        # (the timed pairs above run without the per-launch HIP events; nine more pairs with them for the launch's own duration:
        # the median -- one alignment in nine times its first launch, unscheduled, and the eight that follow run under the
        # schedule built from it)
        ctx.set_profiling(True)
        first_us = []
        for _ in range(9):
            pair(t, s_)
            ctx.synchronize()
            if ref.result.ms_nn > 0:
                first_us.append(float(ref.result.ms_nn) / max(int(ref.result.n_nn_launches), 1) * 1e3)
        ctx.set_profiling(False)
        # the literal call surface from C++ (tools/cpp/pair_time.cpp, host mode: rsreg_icp_set_source / _set_target / rsreg_icp_align with
        # aligned_out on host clouds, no Python between the calls), with the library's own account of the host's time
        cpp = None
        try:
            import subprocess
            import tempfile

            exe = os.path.join(ROOT, "tools", "_build", "pair_time")
            pkg = os.path.dirname(lib.SO_PATH)
            os.makedirs(os.path.dirname(exe), exist_ok=True)
            src_cpp = os.path.join(ROOT, "tools", "cpp", "pair_time.cpp")
            if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src_cpp), os.path.getmtime(os.path.join(ROOT, "include", "rsreg.h"))):
                subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), src_cpp, "-o", exe, "-L", pkg, "-lrsreg",
                                "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"], check=True, capture_output=True)
            with tempfile.TemporaryDirectory() as d:
                pt, ps = os.path.join(d, "t.f32"), os.path.join(d, "s.f32")
                tp.points.tofile(pt)
                sp.points.tofile(ps)
                r = subprocess.run([exe, pt, ps, str(len(tp.points)), str(len(sp.points)), "host"], capture_output=True, text=True, timeout=300, check=True)
            cpp = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        except Exception as e:  # noqa: BLE001  (no g++ on the box, ...: the line says so instead of failing the bench)
            cpp = {"error": str(e)[:200]}
        # the CPU side of THIS mode: the port's kd-tree build + the one iteration the reference's parameters run, 1 thread
        o_ref = oracle.IcpOracle()
        tcr = time.perf_counter()
        o_ref.set_target(tp.points, dedup=True)   # (as in the headline's baseline: 10^5 copies of the point (0, 0, 0) make one kd-tree leaf, 15 s per pair)
        o_ref.set_source(sp.points)
        r_ref = o_ref.align(None, oracle.IcpParams.reference())
        cpu_ref_s = time.perf_counter() - tcr
        out["reference_mode"] = {
            "workload": "icp_pair_%sx%s reference parameters (100 max iterations, 1 cm gate, eps 1 / 1000 -> 1 iteration)" % (a.size, a.size),
            "iterations": int(ref.result.iterations), "n_correspondences": int(ref.result.n_correspondences),
            "ms_per_pair_host_clouds": modes["host_clouds"], "ms_per_pair_device_clouds": modes["device_clouds"],
            "ms_per_pair_device_clouds_boxes_known": modes["device_clouds_boxes_known"],
            "point_pairs_per_s_device_clouds": float(len(sp)) * ref.result.iterations / (modes["device_clouds"] * 1e-3),
            "same_transform": bool((modes["host_clouds_T"] == modes["device_clouds_T"]).all()),
            # what the reference's one iteration per align runs: the first, unseeded search launch (median and slowest of nine pairs)
            "first_launch_us": float(np.median(first_us)) if first_us else None,
            "first_launch_us_max": float(max(first_us)) if first_us else None,
            "ms_per_pair_host_clouds_cpp": cpp.get("ms_per_pair") if cpp else None,
            "host_clouds_cpp": cpp,
            "cpu_baseline": {"ms_per_pair": cpu_ref_s * 1e3, "value": float(len(sp)) * int(r_ref.iterations) / cpu_ref_s, "unit": "point-pairs/s", "cores": 1,
                             "kind": "port", "sample": "the same pair: kd-tree build (exact copies of a target point dropped) + %d iteration(s), 1 thread" % int(r_ref.iterations),
                             "transform_error_vs_cpu_frobenius": float(np.linalg.norm(modes["device_clouds_T"] - r_ref.T))},
        }
    gt = synth.ground_truth(1, 0, "bench")
    out["transform_error_vs_ground_truth_frobenius"] = float(np.linalg.norm(T_gpu - gt))
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
