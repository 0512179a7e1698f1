#!/usr/bin/env python
"""bench.py — residue frames/s of the TIMED forward pass on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over the whole per-GPU job: BASELINE.json configs[1],
"TIMED 21x21x21x6, 100k synthetic frames, 1x MI355X" — 100 000 fp32 frames already resident in HBM
(generated on the device, 22.2 GB), through th_predict_device (include/timed_hip.h), probabilities
left in HBM.  With N > 1 every rank runs the same job on its own GPU (weak scaling, frames are
independent) and each step ends with the one real exchange of the path: an RCCL gather of the
[100k, n_classes] fp32 shards to rank 0 (th_comm_gather_rows).  No PyTorch: the barrier / max-reduce
of the driver contract and the RCCL id broadcast run over the product's own TCP rendezvous
(timed_hip/rendezvous.py) on the RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT the launcher sets.

The JSON line carries, besides the contract fields:
  roofline      the dominant kernel (largest share of device time), timed live with HIP events on the
                model's stream inside the timed region: algorithmic FLOPs per launch / mean launch
                duration vs the dense fp32 MFMA peak (157.3 TFLOP/s).  The path is compute-bound
                (~2800 FLOP/B), so the binding roofline is "mfma"; the HBM view of the
                whole job is reported beside it in "hbm".
  cpu_baseline  the NumPy/BLAS oracle (a port, not TensorFlow) timed on this host's cores on a bounded
                sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "timed-design_amd"))

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA = vector peak
PEAK_HBM_GBS = 8000.0           # HBM3E spec


def keras_cpu_baseline(cfg, weights, topology: str, budget_s: float = 15.0):
    """The reference's own CPU path — tf.keras Model.predict (reference predict.py:142, TensorFlow 2.13 on the host cores) —
    on the same topology, when TensorFlow is importable on the measuring host (BASELINE.md B1').  It is not
    in this image, so this normally returns None and the NumPy/BLAS port below is reported instead."""
    try:
        import tensorflow as tf      # noqa: F401
    except Exception:
        return None
    from timed_hip import synth
    try:
        tf.config.set_visible_devices([], "GPU")
        model = tf.keras.Model.from_config(cfg["config"])
        for layer in model.layers:
            if weights.get(layer.name):
                layer.set_weights([np.asarray(a) for a in weights[layer.name]])
        n, dt = 64, 0.0
        model.predict(synth.synthetic_frames(8, seed=999), batch_size=500, verbose=0)
        while True:
            frames = synth.synthetic_frames(n, seed=1000)
            t0 = time.perf_counter()
            model.predict(frames, batch_size=500, verbose=0)          # batch size of scripts/run_benchmark_models.sh:1-6
            dt = time.perf_counter() - t0
            if dt >= 10.0 or n >= 16384:
                break
            n = int(min(16384, max(2 * n, n * budget_s / max(dt, 1e-3))))
        return dict(value=n / dt, unit="frames/s", cores=os.cpu_count(), kind="reference",
                    sample=f"{n} synthetic frames of {topology} through tf.keras Model.predict(batch_size=500) of TensorFlow "
                           f"{tf.__version__} on the host CPUs, {dt:.1f} s wall")
    except Exception as e:
        print(f"[bench] TensorFlow is importable but the Keras baseline failed ({e!r}); using the NumPy port", file=sys.stderr)
        return None


def cpu_baseline(cfg, weights, topology: str, budget_s: float = 15.0, min_s: float = 10.0):
    """Time the oracle (NumPy + multithreaded BLAS) on a bounded sample of the same workload — or, when TensorFlow is
    importable on this host, the reference's own Keras CPU path."""
    ref = keras_cpu_baseline(cfg, weights, topology, budget_s)
    if ref is not None:
        return ref
    from oracle import cnn_oracle
    from timed_hip import _lib, synth
    # the container may expose every host core but enforce a CPU quota (cgroup cpu.max): a BLAS pool wider than the
    # quota is throttled in 100 ms periods, so the pool is limited to the CPUs this process can really use
    usable = _lib.load().th_host_cpus()
    limiter = None
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=usable, user_api="blas")
    except Exception:
        pass
    # the convolutions (the FLOP sink) run on the blocked direct convolution of oracle/conv3d_omp.c with OpenMP over the usable
    # cores; everything else is the NumPy oracle.  Without the built library: NumPy im2col + BLAS sgemm.
    c_conv = cnn_oracle.use_c_conv(usable)
    cnn_oracle.forward(cfg, weights, synth.synthetic_frames(1, seed=999))  # warm up the thread pools
    # grow the sample until it holds >= 10 s of CPU work (small batches run far below the large-batch rate,
    # so a single calibration point would under-size it); the last, largest run is the one reported
    n, dt = 32, 0.0
    while True:
        frames = synth.synthetic_frames(n, seed=1000)
        t0 = time.perf_counter()
        cnn_oracle.forward(cfg, weights, frames)
        dt = time.perf_counter() - t0
        if dt >= min_s or n >= 16384:
            break
        n = int(min(16384, max(2 * n, n * budget_s / max(dt, 1e-3))))
    cnn_oracle.use_numpy_conv()
    threads = usable if c_conv else 1
    if not c_conv:
        try:
            from threadpoolctl import threadpool_info
            threads = max([p.get("num_threads", 1) for p in threadpool_info() if p.get("user_api") == "blas"] or [1])
        except Exception:
            pass
    if limiter is not None:
        limiter.restore_original_limits()
    how = (f"oracle/cnn_oracle.py with its convolutions on oracle/conv3d_omp.c (blocked direct conv, OpenMP x{threads}, AVX clones)"
           if c_conv else f"oracle/cnn_oracle.py (NumPy im2col + BLAS sgemm on {threads} threads)")
    return dict(value=n / dt, unit="frames/s", cores=threads, kind="port",
                sample=f"{n} synthetic frames of {topology} through {how}; {os.cpu_count()} host cores visible, {usable} usable under "
                       f"the container's CPU quota; fp32, {dt:.1f} s wall")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=100_000, help="frames per GPU per step (BASELINE config: 100k)")
    ap.add_argument("--topology", default="timed", choices=["timed", "timed_rotamer", "densecpd", "prodconn"])
    ap.add_argument("--chunk", type=int, default=4096, help="frames per launch (activation arenas are sized for this many frames)")
    ap.add_argument("--allow-host-exchange", action="store_true",
                    help="development only: ranks that SHARE a GPU exchange their rows over TCP instead of RCCL (the line says so); "
                         "without it a --gpus N run that is not N RCCL ranks with a verified gather exits with an error")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP events (roofline omitted)")
    ap.add_argument("--no-extras", action="store_true", help="skip the e2e / sampler / other-topology legs (N=1 only)")
    ap.add_argument("--e2e-frames", type=int, default=100000, help="frames in the synthetic frame packs of the predict.py leg")
    ap.add_argument("--e2e-hdf5-frames", type=int, default=40000, help="frames in the synthetic gzip .hdf5 of the predict.py leg")
    ap.add_argument("--e2e-rotamer-frames", type=int, default=125000,
                    help="uint8 frames in the predict.py --predict_rotamers leg (config 4's per-GPU share: 1 M / 8)")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc child passes (roofline.traffic stays null)")
    ap.add_argument("--other-frames", type=int, default=100000, help="frames for the densecpd / timed_rotamer legs (BASELINE config 3: 100 k frames)")
    args = ap.parse_args()

    # stdout carries exactly one JSON line: gloo / RCCL banners written to fd 1 by native code go to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world

    from timed_hip import _lib, engine, synth
    import ctypes as C

    lib = _lib.load()
    ndev = _lib.device_count()
    if ndev < 1:
        sys.exit("no HIP device visible; bench.py measures the GPU path only")
    device = local_rank % ndev

    rdzv = None
    if world > 1:
        from timed_hip.rendezvous import HostRendezvous
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        rdzv = HostRendezvous(rank, world)

    def barrier():
        if rdzv is not None:
            rdzv.barrier()

    cfg, weights = synth.TOPOLOGIES[args.topology]()
    model = engine.HipFrameModel.from_keras(cfg, weights, device=device, name=args.topology)
    model.set_chunk(args.chunk)
    D, H, W, Cc = model.input_shape
    n = args.frames
    frame_floats = D * H * W * Cc
    d_frames = engine.DeviceBuffer(n * frame_floats * 4, device)
    d_probs = engine.DeviceBuffer(n * model.n_classes * 4, device)
    _lib.check(lib.th_dev_synth_frames(device, C.c_void_p(d_frames.ptr), n, D, Cc, 200, 1234 + rank))

    # multi-GPU exchange: RCCL gather of the probability shards to rank 0
    comm = None
    d_gather = None
    host_gather = None
    exchange = "rccl gather to rank 0 (grouped send/recv over xGMI)" if world > 1 else "none"
    if world > 1:
        from timed_hip import distributed as td
        why = ""
        try:
            comm = td.RcclGather.from_environment(rank, world, device, rendezvous=rdzv)   # raises on EVERY rank if any fails
            if rank == 0:
                d_gather = engine.DeviceBuffer(world * n * model.n_classes * 4, device)
        except RuntimeError as e:
            why = str(e)
            if ndev >= world or not args.allow_host_exchange:
                # a --gpus N line must be N RCCL ranks, one per GPU, with the gather verified: a host exchange measured as if it
                # were the xGMI gather is refused (ranks sharing a GPU: --allow-host-exchange, development only)
                sys.exit(f"[bench] rank {rank}: RCCL communicator over {world} ranks could not be created ({ndev} device(s) visible): {why}")
            # ranks share a GPU (fewer devices than ranks): RCCL cannot span them; measure with the exchange done on the
            # host (device->host copy + TCP gather) and say so in the JSON line
            exchange = f"HOST FALLBACK (TCP gather of downloaded rows): {world} ranks on {ndev} device(s): " + why
            if rank == 0:
                print("[bench] " + exchange, file=sys.stderr)
            host_gather = rdzv

    def step():
        model.predict_device(d_frames.ptr, n, d_probs.ptr)
        if comm is not None:
            comm.gather_rows_device(d_probs.ptr, [n] * world, model.n_classes, 0, d_gather.ptr if d_gather else 0)
        elif host_gather is not None:
            host_gather.allgather(d_probs.download((n, model.n_classes), np.float32).tobytes())

    # warm-up steps run with every plan step bracketed by HIP events (the per-layer table); the timed steps
    # bracket only the dominant kernel (2 events per chunk), which is what `roofline` is computed from
    if not args.no_profile:
        model.profile(1)
    for _ in range(args.warmup):
        step()
    _lib.check(lib.th_dev_sync(device))
    warm_steps = [s for s in model.steps() if s["launches"]] if not args.no_profile else []
    if not args.no_profile:
        model.profile(2)
    barrier()
    _lib.check(lib.th_dev_sync(device))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.th_dev_sync(device))
    barrier()
    elapsed = time.perf_counter() - t0
    if rdzv is not None:
        elapsed = rdzv.all_max_float(elapsed)

    # sanity: EVERY row the timed work produced is a probability vector (all n rows are brought back: 8 MB at 20 classes)
    out_rows = d_probs.download((n, model.n_classes), np.float32)
    row_sums = out_rows.sum(1, dtype=np.float64)
    bad = int(np.count_nonzero(~np.isfinite(out_rows).all(1) | (np.abs(row_sums - 1.0) > 1e-4) | (out_rows < 0).any(1)))
    assert bad == 0, f"bench output is not a probability matrix: {bad} of {n} rows are non-finite, negative or do not sum to 1"
    del out_rows

    # N > 1: the gathered matrix on rank 0 must hold rank r's rows in block r, bit for bit.  Every rank ships the head
    # and tail rows of its own shard over gloo (host) and rank 0 compares them with the same rows of the RCCL result.
    gather_verified = None
    if world > 1 and comm is not None:
        k = min(n, 64)
        mine = np.concatenate([d_probs.download((k, model.n_classes), np.float32),
                               d_probs.download((k, model.n_classes), np.float32, offset=(n - k) * model.n_classes * 4)])
        parts = [np.frombuffer(b, dtype=np.float32).reshape(mine.shape) for b in rdzv.allgather(mine.tobytes())]
        if rank == 0:
            gather_verified = True
            for r in range(world):
                base = r * n * model.n_classes * 4
                got = np.concatenate([d_gather.download((k, model.n_classes), np.float32, offset=base),
                                      d_gather.download((k, model.n_classes), np.float32, offset=base + (n - k) * model.n_classes * 4)])
                if not np.array_equal(got, parts[r]):
                    gather_verified = False
            if not gather_verified:
                sys.exit("[bench] RCCL gather delivered rows that differ from the ranks' own shards")
            # distinct seeds per rank: two blocks holding the same rows would mean a misrouted transfer
            assert not np.array_equal(parts[0], parts[1]), "ranks produced identical shards"
        # every rank's own count of the RCCL transfers it issued: per step the root posts world - 1 ncclRecv, a peer one ncclSend
        st = comm.stats()
        per_rank = rdzv.allgather_ints([st["sends"], st["recvs"]])
        if rank == 0:
            calls = args.steps + args.warmup
            want = [[0, (world - 1) * calls]] + [[calls, 0]] * (world - 1)
            if per_rank != want:
                sys.exit(f"[bench] RCCL transfer counts {per_rank} differ from the {want} that {calls} gathers over {world} ranks issue")
    if world > 1 and rank == 0 and not args.allow_host_exchange and not (comm is not None and gather_verified):
        sys.exit(f"[bench] --gpus {world}: not {world} RCCL ranks with a verified gather (rccl_ranks={world if comm else 0}, "
                 f"gather_verified={gather_verified})")

    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_legs
        import bench_line
        cost = model.cost()
        kernel_flops = sum(s["flops"] for s in model.steps())
        total_frames = n * world * args.steps
        fps = total_frames / elapsed
        algo_bytes = frame_floats * 4 + 4 * model.n_classes
        line = {
            "metric": "residue_frames_per_sec", "value": fps, "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": bench_line.workload_string(model.name, (D, H, W, Cc), n, model.n_classes),
                       "frames_per_gpu": n, "chunk": args.chunk, "parallelism": f"frame-shard x{world}",
                       "exchange": exchange, "rccl_ranks": world if comm is not None else 0, "gather_verified": gather_verified,
                       "rows_verified": n,
                       "algo_mflop_per_frame": cost["algo_flops"] / 1e6, "kernel_mflop_per_frame": kernel_flops / 1e6,
                       "exec_mflop_per_frame": cost["exec_flops"] / 1e6, "winograd_layers": sum("k_wino_gemm" in s["label"] for s in model.steps()),
                       "split_gemm_layers": sum("bf16x3" in s["label"] for s in model.steps()),      # (steps, the first layer included)
                       "arithmetic": "bf16 MFMA on operands split exactly into 3 bf16 pieces (6 products, fp32 accumulate) in every 3x3x3 layer of "
                                     "TIMED: first layer, fused 10^3 layer, 5^3 Winograd GEMMs; fp32-input MFMA elsewhere (tests hold the same 5e-6 bound)",
                       "emulated_fp32": True,      # fp32 results from bf16-pipe products on exactly split operands (5 of the plan's steps)
                       "knobs": model.knobs(), "guard": {k: (round(v, 9) if isinstance(v, float) else v) for k, v in model.guard().items() if k != "note"},
                       "device": f"{model.device_arch} {model.device_cus} CUs"},
            # FLOPs the kernels really compute per frame (the SURVEY §8d direct-form count, except that layers on the Cook-Toom /
            # Winograd path count their own, fewer multiply-adds) priced against the fp32-MFMA peak; the direct-form equivalent —
            # what a direct convolution would have to sustain for the same frames/s — is reported beside it and may exceed the peak
            # schema 5: `model_tflops` has its round 1-3 meaning again — SURVEY §8d's direct-form FLOPs x frames/s, which
            # minimal-filtering layers do not execute (it may exceed the fp32 peak); what the kernels compute is in the two keys
            # after it: their own fp32-equivalent multiply-adds, and the share of the wall time the matrix pipes would need for
            # them at their dense peaks (fp32-input MFMA 157.3 TFLOP/s; bf16x3-split GEMMs: 6 products on the 2500 TFLOP/s bf16 pipe)
            "schema": 5,
            "model_tflops": fps / world * cost["algo_flops"] / 1e12,
            "model_kernel_tflops_fp32_equiv": fps / world * kernel_flops / 1e12,
            "model_pipe_time_frac": bench_legs.pipe_time_frac(model.steps(), fps / world),
            "hbm": {"algo_bytes_per_frame": algo_bytes, "achieved_GBps": fps / world * algo_bytes / 1e9,
                    "frac": fps / world * algo_bytes / 1e9 / PEAK_HBM_GBS},
        }
        if not args.no_profile:
            dom = max((s for s in model.steps() if s["launches"]), key=lambda s: s["ms"])   # timed region: dominant step only
            avg_ms = dom["ms"] / dom["launches"]
            frames_per_launch = n * args.steps / dom["launches"]
            pipe, peak, pflops = bench_legs.step_pipe(dom)       # a bf16x3-split step is priced on the bf16 pipe (6 products)
            achieved = pflops * frames_per_launch / (avg_ms * 1e-3) / 1e12
            line["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "pipe": pipe,
                                "frac": achieved / peak, "traffic": None,
                                "kernel": dom["label"], "avg_launch_ms": avg_ms, "launches": dom["launches"],
                                "frames_per_launch": frames_per_launch, "algorithmic_flops_per_launch": pflops * frames_per_launch,
                                "measured": "HIP events around every launch of this kernel inside the timed region",
                                "exec_tflops": dom["exec_flops"] * frames_per_launch / (avg_ms * 1e-3) / 1e12}
            if warm_steps:
                tot_ms = sum(s["ms"] for s in warm_steps)
                wdom = [s for s in warm_steps if s["label"] == dom["label"]]
                if wdom:
                    line["roofline"]["share_of_device_time"] = wdom[0]["ms"] / tot_ms
                line["kernels_from"] = f"{args.warmup} warm-up step(s), every plan step bracketed by HIP events"
                line["kernels"] = [{"label": s["label"], "ms_total": round(s["ms"], 3), "launches": s["launches"],
                                    "tflops_algo": (s["flops"] * n * args.warmup / (s["ms"] * 1e-3) / 1e12) if s["ms"] else 0.0,
                                    "pipe": bench_legs.step_pipe(s)[0] if s["flops"] else None,
                                    "frac_of_pipe_peak": (bench_legs.step_pipe(s)[2] * n * args.warmup / (s["ms"] * 1e-3) / 1e12 /
                                                          bench_legs.step_pipe(s)[1]) if s["ms"] and s["flops"] else None}
                                   for s in warm_steps]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(cfg, weights, f"{model.name}-synth")
        if world == 1 and not args.no_extras:
            # secondary legs, outside the timed region (tools/bench_legs.py): each is reported, none feeds `value`
            model.profile(0)
            t_legs = time.perf_counter()
            others = [t for t in ("densecpd", "timed_rotamer", "prodconn") if t != args.topology]
            # HBM bytes of every kernel of every topology, measured now (two rocprofv3 --pmc child passes on this build)
            pmc = bench_legs.pmc_traffic_inrun([args.topology] + others, args.chunk) if not args.no_pmc else {"error": "--no-pmc"}
            line["pmc"] = {k: v for k, v in pmc.items() if k in ("error", "source")}
            mine = pmc.get(args.topology, {})
            if "roofline" in line and mine:
                rl = line["roofline"]
                rl["traffic"] = mine.get("steps", {}).get(rl["kernel"])
                rl["traffic_frames"] = args.chunk
                rl["traffic_source"] = pmc.get("source")
                dom_step = next(s for s in model.steps() if s["label"] == rl["kernel"])
                rl["algorithmic_bytes"] = dom_step["bytes"] * args.chunk
                line["hbm"]["model_traffic_per_chunk"] = mine.get("model")
                line["hbm"]["model_algorithmic_bytes_per_chunk"] = algo_bytes * args.chunk
                for k in line.get("kernels", []):
                    k["traffic_per_chunk"] = mine.get("steps", {}).get(k["label"])
            e2e = bench_legs.host_resident(model, d_frames.ptr, n, fps)
            e2e.update(bench_legs.predict_py_e2e(cfg, weights, n_pack=args.e2e_frames, n_hdf5=args.e2e_hdf5_frames,
                                                 n_rotamer=args.e2e_rotamer_frames))
            line["e2e"] = e2e
            line["sampler"] = bench_legs.sampler_config5(device)
            base = None if args.no_cpu_baseline else (lambda c, w, t: cpu_baseline(c, w, t, budget_s=8.0, min_s=5.0))
            line["other_configs"] = [bench_legs.topology_rate(t, device, d_frames.ptr, min(n, args.other_frames), args.chunk,
                                                              traffic=pmc.get(t), cpu_baseline=base) for t in others]
            # the same plans with every layer on the fp32-input matrix pipe (TH_WINO_SPLIT=0 TH_FIRST_SPLIT=0: exact fp32 products instead
            # of the bf16x3 split): same frames, the rate and the logits against the default plan's
            line["other_configs"] += [bench_legs.topology_rate(t, device, d_frames.ptr, min(n, args.other_frames), args.chunk,
                                                               env={"TH_WINO_SPLIT": "0", "TH_FIRST_SPLIT": "0"})
                                      for t in (args.topology, "timed_rotamer") if t in ("timed", "timed_rotamer")]
            line["extras_wall_s"] = time.perf_counter() - t_legs
        # the full record (per-kernel tables of every topology, all e2e/sampler legs) goes to a side file and stderr;
        # stdout gets ONE compact line (tools/bench_line.py, < 4 KB) — the driver only reads the tail of stdout
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_line
        detail = os.environ.get("TH_BENCH_DETAIL", os.path.join(ROOT, "bench_detail.json"))
        try:
            with open(detail, "w") as f:
                json.dump(line, f, indent=1)
            scratch = os.path.join(ROOT, "gpurun_out")
            if os.path.isdir(scratch):
                with open(os.path.join(scratch, "bench_detail.json"), "w") as f:
                    json.dump(line, f, indent=1)
        except OSError as e:
            print(f"[bench] could not write {detail}: {e}", file=sys.stderr)
            detail = None
        print("[bench] full record:\n" + json.dumps(line, indent=1), file=sys.stderr)
        sys.stderr.flush()
        sys.stdout.flush()
        os.write(json_fd, (bench_line.dumps(line, os.path.basename(detail) if detail else None) + "\n").encode())
    if comm is not None:
        comm.close()
    if rdzv is not None:
        rdzv.barrier()
        rdzv.close()


if __name__ == "__main__":
    main()
