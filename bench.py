"""Headline benchmark of the MI355X-native video-query hot path.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Primary line (BASELINE.json metric, first half): clips/sec of TSN feature extraction on configs[1] -- BN-Inception
RGB stream, 224x224x3, T = 3 segments, B = 32 clips per GPU (96 crops per step), uint8 crops already resident
in HBM, random-init weights, fp32 arithmetic on the fp32 matrix cores.  A step = one full forward of the batch
(preprocess, 69 conv+BN+ReLU, 13 pools, global pool, segment consensus) and, for N > 1, the RCCL all-gather of
the per-GPU [32,1024] fp64 feature blocks.  Weak scaling: every rank extracts its own 32 clips.

Secondary object "similarity" (second half of the metric): queries/sec of the weighted similarity scan + score
over a 1M-clip x (2 streams x 5 splits) x 1024 fp32 database (configs[3], 40.96 GB; row-sharded over the ranks,
score slices all-gathered), HBM-bound.

Both carry a `roofline` (HIP-event time of the dominant kernel inside the timed region vs the gfx950 peak) and a
`cpu_baseline` (the oracle timed on the host cores, rank 0, N = 1 only, bounded sample).

Timed region (configs[1]): K forwards as the PRODUCT runs them -- ``vq_tsn_forward`` with its default of two sub-batches on two HIP
streams (VQ_TSN_SPLIT=2: one sub-batch's launch ramps and tails overlap the other's steady state; same bits as one stream); no launch
of that region carries an event.  `single_stream` = the same K steps, same bracketing, all on ONE stream (the timed mode of rounds
1-4; first-class, never shed from the line): there a launch's duration is the kernel alone on the chip, so every per-kernel figure of
the line comes from THAT region -- sample_every(K) of its steps (>= 3 at the driver's K = 20) run with a start / stop event on every
launch (the dispatch packets' own timestamps), `roofline.kernel_fields_from` says so and `roofline.profiled_steps` counts them.
TSN roofline: `achieved` / `frac` = ALGORITHMIC direct-convolution FLOPs of a step (2 x MACs of Appendix A x crops =
390.06 GFLOP) / the whole timed step (`ms_per_step`, product mode) / 157.3 TFLOP/s.  Beside it, from the one-stream region's sampled
steps: `kernel_frac` = the same FLOPs / the convolution launches' own time (exceeds the pipe's rate because the Winograd form skips
20/36 of the multiplies) and `matrix_pipe_frac` = MFMA work the matrix pipe really EXECUTED (Winograd layers: 16 multiplies per 2x2
tile, K and tile padding of every kernel included) / that time.  Two streams overlap, so the product-mode step can be SHORTER than the
sum of its kernels' one-stream durations (`conv_ms_per_step`): profiles/r06_two_queue_trace.txt is the kernel trace that shows it.
`roofline.traffic` is HBM bytes PER LAUNCH (like `avg_launch_ms`); `traffic_per_step` = x launches_per_step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import video_query_algorithms_amd as vqa
from video_query_algorithms_amd._lib import call
from video_query_algorithms_amd.shard import all_gather_rows, shard_range
from video_query_algorithms_amd.tsn import bn_inception, net as tsn_net

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0             # HBM3E spec (6.29 TB/s measured achievable per the same guide)
B_CLIPS, T_SEG, CH = 32, 3, 3     # configs[1]
MIN_SAMPLED = 4                   # steps of the one-stream region that carry per-launch events (~0.1 ms each: 37 launches x ~3 us of signals)


def sample_every(steps):
    """Every n-th step of the one-stream region is sampled: at least min(steps, MIN_SAMPLED) of them (K = 20 -> 4, K = 50 -> 5)."""
    return max(1, steps // MIN_SAMPLED)


SIM_N, SIM_S, SIM_E, SIM_D = 1_000_000, 2, 5, 1024   # configs[3]
TILED_GROUP = 4                   # csrc/vq_sim.hip: kTiledGroup


class _DevArray:
    """Zero-copy view of library-owned device memory as a torch tensor (for the RCCL all-gather)."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def dev_tensor(ptr, shape, typestr, device):
    return torch.as_tensor(_DevArray(ptr, shape, typestr), device=device)


def cdiv(a, b):
    return (a + b - 1) // b


def host_cores():
    """CPU cores this process may actually use (affinity mask, capped by the cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def tsn_roofline(model, n_crops, steps, every):
    """Roofline object of a timed region from the per-launch begin/end timestamps the library kept (mean over the profiled
    steps): executed and algorithmic MFMA FLOP/s over the convolution launches, by kernel family."""
    names, kinds, ms_layers, fl = model.layer_times()    # mean over the profiled steps of the timed region
    ms_layers = ms_layers.astype(np.float64)
    conv = np.array([k == "conv" for k in kinds])
    item_of_layer, n_items = model.launch_items()
    # a launch counts as a convolution launch if it carries a convolution; the pooling layer that rides in a Winograd
    # launch is then convolution-kernel time too (as rocprofv3 sees it: one kernel)
    conv_items = {int(item_of_layer[i]) for i in np.flatnonzero(conv)}
    in_conv_launch = np.array([int(it) in conv_items for it in item_of_layer])
    conv_ms = float(ms_layers[in_conv_launch].sum())
    conv_flops = float(fl[conv].sum())
    # MFMA work actually executed: the direct kernels run K padded to 32 (the 7x7 stem in space-to-depth form: 147 -> 192);
    # the Winograd kernel runs 16 multiplies per 2x2 output tile instead of 36 (tiles padded to whole 2x2 blocks)
    tiles = model.layer_tiles(n_crops)
    issued = np.zeros(len(kinds))
    for i, op in enumerate(model.plan.ops):
        if op.kind != "conv":
            continue
        t = model.plan.tensors[op.segments[0].dst if op.segments else op.dst]
        if tiles[i, 3] == 2:
            issued[i] = 2.0 * n_crops * ((t.h + 1) // 2) * ((t.w + 1) // 2) * 16 * op.cin * op.cout
        else:
            issued[i] = 2.0 * n_crops * t.h * t.w * op.cout * model.conv_kp[i]       # K as packed (padded to 32)
    wino_items = {int(item_of_layer[i]) for i in np.flatnonzero(conv & (tiles[:, 3] == 2))}
    wino = np.array([int(it) in wino_items for it in item_of_layer])                 # incl. the pooling that rides along
    direct = in_conv_launch & ~wino
    conv_launches = len(conv_items)

    def family(mask, kernel):
        ms = float(ms_layers[mask].sum())
        return {"kernel": kernel, "layers": int((mask & conv).sum()), "launches": len({int(item_of_layer[i]) for i in np.flatnonzero(mask)}),
                "ms_per_step": ms, "executed_tflops": float(issued[mask].sum()) / ms / 1e9,
                "matrix_pipe_frac": float(issued[mask].sum()) / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                "algorithmic_tflops": float(fl[mask].sum()) / ms / 1e9, "kernel_frac": float(fl[mask].sum()) / ms / 1e9 / PEAK_FP32_MFMA_TFLOPS}
    fam = {"direct": family(direct, "conv_igemm_pipe_kernel / conv_igemm_kernel (implicit GEMM: 1x1, stride-2 and stem layers)"),
           "winograd": family(wino, "wino_f2x2_3x3_kernel (F(2x2,3x3): the 3x3 stride-1 layers; a launch carries the sibling arms "
                                    "of an inception module and its pooling layer)")}
    executed = float(issued.sum())
    # `achieved` / `frac` (algorithmic FLOPs over the whole timed step) are filled in by the caller, who owns the clock
    roof = {"bound": "mfma",
            "kernel": "the %d convolution launches of a step (%d layers): %d direct implicit-GEMM launches + %d Winograd F(2x2,3x3) launches; "
                      "fp32 v_mfma_f32_32x32x2" % (conv_launches, int(conv.sum()), fam["direct"]["launches"], fam["winograd"]["launches"]),
            "achieved": None, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
            "kernel_tflops": conv_flops / conv_ms / 1e9, "kernel_frac": conv_flops / conv_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS,
            "matrix_pipe_tflops": executed / conv_ms / 1e9, "matrix_pipe_frac": executed / conv_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS,
            "launches_per_step": conv_launches, "all_launches_per_step": int(n_items), "avg_launch_ms": conv_ms / conv_launches,
            "conv_ms_per_step": conv_ms, "other_kernels_ms_per_step": float(ms_layers[~in_conv_launch].sum()),
            # the pooling layers that ride in Winograd launches take part of those launches' time (their share by workgroup
            # count); the stem pools inside the loaders of the 1x1 GEMMs behind them cannot be separated and stay in
            "pooling_share_of_conv_launches_ms": float(ms_layers[in_conv_launch & ~conv].sum()),
            "flops_per_step": conv_flops, "executed_flops_per_step": executed, "profiled_steps": cdiv(steps, every),
            "families": fam,
            "note": "achieved / frac = ALGORITHMIC direct-convolution FLOPs (SURVEY 8(d): 2 x MACs of Appendix A x crops) per second of the "
                    "whole timed step.  kernel_* prices the same FLOPs against the convolution launches' own time, matrix_pipe_* the MFMA "
                    "FLOPs the pipe EXECUTED (K / tile padding included; Winograd layers issue 16 of the 36 direct-form multiplies) against "
                    "it.  Kernel times: the launches' own begin/end timestamps on every %d-th step of the timed region, which runs on one "
                    "stream (a launch shared by sibling layers is split between them by matrix work)" % every}
    return roof


def bench_tsn(args, rank, world, device, stream):
    if args.profile_only:
        os.environ["VQ_TSN_SPLIT"] = "1"             # read when the network handle is created
    g = bn_inception.bn_inception(CH)
    weights = tsn_net.synthetic_weights(g, seed=2)
    n_crops = B_CLIPS * T_SEG
    # the tilings are timed in THIS process (behind the autotuner's own warm-up), not taken from a table another process left behind
    model = tsn_net.TsnNet(g, weights, max_crops=n_crops, device=device.index, tune_cache=args.tune_cache)
    model.set_stream(stream.cuda_stream)
    if args.tiles and os.path.exists(args.tiles):          # tiling table of an earlier run: skip the autotune launches
        with open(args.tiles) as f:
            model.set_layer_tiles(n_crops, np.array(json.load(f)["tiles"], dtype=np.int32))
    gen = torch.Generator(device=device).manual_seed(1 + rank)
    crops = torch.randint(0, 256, (n_crops, 224, 224, CH), dtype=torch.uint8, device=device, generator=gen)
    feat_ptr, _ = model.feat_devptr()
    feat = dev_tensor(feat_ptr, (B_CLIPS, model.feature_dim), "<f8", device)
    import torch.distributed as dist

    gather_events = []

    def step(timed=False):
        model.forward_device(crops.data_ptr(), n_crops, T_SEG, tsn_net.RGB_MEAN)
        if world > 1:
            if timed:                                        # the collective's own time: events on the stream it is ordered on
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            all_gather_rows(feat, world * B_CLIPS)           # RCCL over xGMI: per-GPU feature blocks
            if timed:
                e1.record(stream)
                gather_events.append((e0, e1))

    every = sample_every(args.steps)
    depth = min(cdiv(args.steps, every), 1024)
    with torch.cuda.stream(stream):
        model.set_profile(1)                         # one sampled forward: tunes the tilings of the whole batch on one stream
        step()
        model.set_profile(0)
        for _ in range(max(args.warmup, 2)):         # the product's mode: tunes the sub-batch size
            step()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        # The timed region: the product's own forward (two sub-batches on two HIP streams), no events on any launch.
        # --profile-only: ALL steps on one stream, every `every`-th with start/stop events on every launch (no host sync).
        if args.profile_only:
            model.set_profile(depth, every=every, split_between=False)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(timed=True)
        torch.cuda.synchronize(device)
        dt_rank = time.perf_counter() - t0                   # this rank's own K steps (before the closing barrier)
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
    rank_ms = dt_rank / args.steps * 1e3
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else None
    feats = feat.clone()
    # The same K steps once more, ALL on one stream, `every`-th step with events on every launch: the timed mode of rounds 1-4,
    # reported beside `value` (never instead of it) -- and the region every per-kernel figure of the line comes from.
    single_ms = float("nan")
    with torch.cuda.stream(stream):
        if not args.profile_only:
            model.set_profile(depth, every=every, split_between=False)
            for _ in range(2):
                step()
            model.set_profile(depth, every=every, split_between=False)      # counters back to zero: the region's own samples only
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(0 if args.profile_only else args.steps):
            step()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        if not args.profile_only:
            single_ms = (time.perf_counter() - t1) / args.steps * 1e3
    roof = tsn_roofline(model, n_crops, args.steps, every)
    roof["kernel_fields_from"] = ("the timed region (--profile-only: every step on one stream)" if args.profile_only else
                                  "single_stream region") + ": %d sampled steps (every %d%s of %d)" % (
                                      roof["profiled_steps"], every, "th" if every > 3 else ("st", "nd", "rd")[every - 1], args.steps)
    roof["single_stream_ms_per_step"] = single_ms
    roof["rank_ms_per_step"] = rank_ms
    if gather_ms is not None:
        roof["all_gather_ms_per_step"] = gather_ms
    # Off-line PMC evidence for the same command, committed under profiles/ (NOT measured in this run): HBM bytes per conv
    # launch (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled as the microarchitecture guide
    # prescribes for gfx950) and the SQ matrix-pipe utilisation per kernel family.
    for name in ("r06_tsn_traffic.json", "r05_tsn_traffic.json", "r04_tsn_traffic.json", "r03_tsn_traffic.json", "r02_tsn_traffic.json", "r01_tsn_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as f:
                roof["traffic"] = json.load(f)["hbm_bytes_per_launch"]
            roof["traffic_per_step"] = roof["traffic"] * roof["launches_per_step"]
            roof["traffic_source"] = "profiles/%s: committed PMC passes of this command (tools/pmc_tsn.sh), not collected in this run" % name
            break
    for name in ("r06_mfma_util.json", "r05_mfma_util.json", "r04_mfma_util.json", "r03_mfma_util.json", "r02_mfma_util.json", "r01_mfma_util.json"):
        upath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(upath):
            with open(upath) as f:
                u = json.load(f)
            roof["pmc_matrix_pipe_utilisation"] = {k: v["matrix_pipe_utilisation"] for k, v in u.items() if isinstance(v, dict)}
            roof["pmc_source"] = "profiles/%s: SQ_INSTS_MFMA x 64 / SIMD-cycles per kernel family (tools/pmc_mfma.sh), committed, not collected in this run" % name
            break
    model.set_profile(0)
    if args.tiles and rank == 0 and not os.path.exists(args.tiles):
        os.makedirs(os.path.dirname(os.path.abspath(args.tiles)), exist_ok=True)
        with open(args.tiles, "w") as f:
            json.dump({"n_crops": n_crops, "tiles": model.layer_tiles(n_crops).tolist()}, f)
    return dt, roof, model, crops, feats


def bench_two_stream(args, device, stream, with_cpu):
    """configs[2]: TSN two-stream, T = 7 segments, B = 64 clips: RGB crops [448,224,224,3] and 5-frame flow stacks
    [448,224,224,10] (x/y interleaved, mean 128) resident in HBM, one forward per stream per step, both on this GPU one after
    the other.  value = clips/s for the PAIR of streams; each stream carries its own roofline from the launches' own timestamps.
    Algorithmic work: 7 x (4.063 + 4.614) = 60.74 GFLOP per clip, 3.887 TFLOP per step."""
    B2, T2 = 64, 7
    n_crops = B2 * T2
    steps = max(4, args.steps // 5)
    every = steps                                     # the first step of the region is the sampled one
    out = {"metric": "clips/sec TSN two-stream feature-extract (RGB + 5-frame flow stack)", "unit": "clips/s", "steps": steps,
           "config": {"workload": "configs[2]: TSN two-stream RGB + warped-optical-flow (5-frame stack = 10 channels), T=7 segments, "
                                  "B=64 clips: 448 + 448 uint8 crops of 224x224 resident in HBM, random-init weights",
                      "crops_per_step": 2 * n_crops},
           "dtype": "f32", "streams": {}}
    total_ms = 0.0
    flops = 0.0
    keep = {}
    for name, ch, seed, mean in (("rgb", 3, 2, tsn_net.RGB_MEAN), ("warped_optical_flow", 10, 5, tsn_net.FLOW_MEAN)):
        if args.profile_only:
            os.environ["VQ_TSN_SPLIT"] = "1"
        g = bn_inception.bn_inception(ch)
        weights = tsn_net.synthetic_weights(g, seed=seed)
        model = tsn_net.TsnNet(g, weights, max_crops=n_crops, device=device.index, tune_cache=args.tune_cache)
        model.set_stream(stream.cuda_stream)
        gen = torch.Generator(device=device).manual_seed(40 + ch)
        crops = torch.randint(0, 256, (n_crops, 224, 224, ch), dtype=torch.uint8, device=device, generator=gen)
        depth = cdiv(steps, every)
        with torch.cuda.stream(stream):
            model.set_profile(1)                        # tunes the 448-crop tilings (one stream) ...
            model.forward_device(crops.data_ptr(), n_crops, T2, mean)
            model.set_profile(0)                        # ... and the sub-batches' (the product's mode)
            for _ in range(2):
                model.forward_device(crops.data_ptr(), n_crops, T2, mean)
            torch.cuda.synchronize(device)
            # timed like configs[1]: the product's forward, the first of every `every` steps sampled on one stream
            model.set_profile(depth, every=every, split_between=not args.profile_only)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                model.forward_device(crops.data_ptr(), n_crops, T2, mean)
            torch.cuda.synchronize(device)
            ms = (time.perf_counter() - t0) / steps * 1e3
        roof = tsn_roofline(model, n_crops, steps, every)
        fl = model.flops_per_crop() * n_crops
        roof["achieved"] = fl / ms / 1e9
        roof["frac"] = roof["achieved"] / PEAK_FP32_MFMA_TFLOPS
        feat_ptr, _ = model.feat_devptr()
        keep[name] = (g, weights, crops[:T2].cpu().numpy(), dev_tensor(feat_ptr, (B2, model.feature_dim), "<f8", device)[:1].cpu().numpy(), mean)
        single_ms = float("nan")
        if not args.profile_only:                       # all steps on one stream (the timed mode of rounds 1-4), for comparison
            with torch.cuda.stream(stream):
                model.set_profile(depth, every=every, split_between=False)
                model.forward_device(crops.data_ptr(), n_crops, T2, mean)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(steps):
                    model.forward_device(crops.data_ptr(), n_crops, T2, mean)
                torch.cuda.synchronize(device)
                single_ms = (time.perf_counter() - t0) / steps * 1e3
        model.set_profile(0)
        out["streams"][name] = {"ms_per_step": ms, "clips_per_s": B2 / ms * 1e3, "algorithmic_gflop_per_clip": fl / B2 / 1e9,
                                "single_stream_ms_per_step": single_ms, "roofline": roof}
        total_ms += ms
        flops += fl
        model.close()
        del crops
    out["value"] = B2 / total_ms * 1e3
    out["ms_per_step"] = total_ms
    out["algorithmic_gflop_per_clip"] = flops / B2 / 1e9
    single = sum(v["single_stream_ms_per_step"] for v in out["streams"].values())
    if single == single:
        out["single_stream"] = {"value": B2 / single * 1e3, "unit": "clips/s", "ms_per_step": single}
    # one roofline for the pair: SURVEY 8(d) -- the algorithmic FLOPs of both forwards over the whole timed step; beside it the two
    # streams' convolution launches together (sampled steps, one stream)
    ex = sum(v["roofline"]["executed_flops_per_step"] for v in out["streams"].values())
    al = sum(v["roofline"]["flops_per_step"] for v in out["streams"].values())
    cm = sum(v["roofline"]["conv_ms_per_step"] for v in out["streams"].values())
    nl = sum(v["roofline"]["launches_per_step"] for v in out["streams"].values())
    out["roofline"] = {"bound": "mfma", "achieved": flops / total_ms / 1e9, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                       "frac": flops / total_ms / 1e9 / PEAK_FP32_MFMA_TFLOPS, "traffic": None,
                       "kernel_frac": al / cm / 1e9 / PEAK_FP32_MFMA_TFLOPS, "matrix_pipe_frac": ex / cm / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                       "conv_ms_per_step": cm, "launches_per_step": nl, "avg_launch_ms": cm / nl,
                       "kernel": "the convolution launches of both streams' forwards (per stream: streams.*.roofline)"}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import tsn_oracle as to
        threads = host_cores()
        t0 = time.perf_counter()
        errs = {}
        reps = 0
        while reps == 0 or time.perf_counter() - t0 < 8.0:       # a bounded sample: the same clip through both streams, repeatedly
            for name, (g, weights, x, got, mean) in keep.items():
                ps, _ = to.features(g.layers, "data", weights, x, mean, T2, dtype=np.float32, threads=threads)
                ref = ps.astype(np.float64).reshape(1, T2, -1).mean(axis=1)
                errs[name] = float(np.abs(got - ref).max() / np.abs(ref).max())
            reps += 1
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": reps / dt, "unit": "clips/s", "cores": threads, "kind": "port",
                               "sample": "%d x clip 0 of the batch through both streams (7 + 7 crops each time), oracle/tsn_oracle.py fp32 "
                                         "torch-CPU, %d threads, %.1f s" % (reps, threads, dt)}
        out["parity_vs_oracle_rel_err"] = errs
    return out


def cpu_baseline_tsn(crops_u8, weights_graph, seconds_target=12.0):
    """The oracle (fp32 torch-CPU evaluation of the layer list, all host threads) on a bounded sample: the algorithmic
    one-crop-per-snippet path, and -- for the record, SURVEY.md 8(d) -- the reference-faithful variant that forwards all
    10 over-sampled crops of a snippet and keeps crop 0 (calcSig_wOF.py:94-95)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tsn_oracle as to
    g, weights = weights_graph
    threads = host_cores()
    x = crops_u8[:T_SEG]
    t0 = time.perf_counter()
    to.features(g.layers, "data", weights, x, tsn_net.RGB_MEAN, T_SEG, dtype=np.float32, threads=threads)
    one = time.perf_counter() - t0                       # includes first-touch; used to size the sample
    n_clips = int(max(1, min(B_CLIPS, seconds_target / max(one, 1e-3))))
    x = crops_u8[:n_clips * T_SEG]
    t0 = time.perf_counter()
    reps = 0
    while reps == 0 or time.perf_counter() - t0 < 8.0:
        ps, _ = to.features(g.layers, "data", weights, x, tsn_net.RGB_MEAN, T_SEG, dtype=np.float32, threads=threads)
        reps += 1
    dt = time.perf_counter() - t0
    # 10-crop variant: the same clip's snippets, ten crops each (the nine discarded ones cost the same arithmetic)
    x10 = np.repeat(crops_u8[:T_SEG], 10, axis=0)
    t0 = time.perf_counter()
    to.features(g.layers, "data", weights, x10, tsn_net.RGB_MEAN, T_SEG * 10, dtype=np.float32, threads=threads)
    dt10 = time.perf_counter() - t0
    return {"value": n_clips * reps / dt, "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": "%d x %d clips (%d crops) of the same cfg2 batch, oracle/tsn_oracle.py fp32 torch-CPU, %d threads, %.1f s"
                      % (reps, n_clips, n_clips * T_SEG, threads, dt),
            "ten_crop_variant": {"value": 1.0 / dt10, "unit": "clips/s", "cores": threads,
                                 "sample": "1 clip = %d snippets x 10 over-sampled crops (what the reference forwards, keeping crop 0), %.1f s"
                                           % (T_SEG, dt10)}}, ps


def bench_sim(args, rank, world, device, stream):
    """configs[3]: 1 query x 1M clips x (2 streams x 5 splits) x 1024 fp32, row-sharded over the ranks through the product's own
    ShardedFeatureDB (N > 1) / FeatureDB (N = 1), tiled in place after loading.  A step = one scan launch per rank (dots, ensemble
    mean, weighted score) + the all-gather of the score slices into global order on every rank's device."""
    import torch.distributed as dist
    rehearse = world > 1 and dist.get_backend() != "nccl"
    sdb = None
    if world > 1:
        from video_query_algorithms_amd.sharded_db import ShardedFeatureDB
        sdb = ShardedFeatureDB.synthetic(SIM_N, SIM_S, SIM_E, SIM_D, seed=17, scales=(4.0, 1.0), device=device.index,
                                         stream=None if rehearse else stream)
        db, row0, rows = sdb.local, sdb.row0, sdb.local.n
        if rehearse:
            db.set_stream(stream.cuda_stream)
    else:
        row0, rows = 0, SIM_N
        db = vqa.FeatureDB.synthetic(rows, SIM_S, SIM_E, SIM_D, seed=17, scales=(4.0, 1.0), row0=row0, device=device.index)
        db.set_stream(stream.cuda_stream)
    front = sdb if sdb is not None else db
    w = [1.0, 1.5]
    tm = C.c_void_p()
    call("vq_timer_create", C.byref(tm))
    sptr = C.c_void_p(stream.cuda_stream)

    def timed_scans(n_steps, gather):
        kern, gath = 0.0, []
        t0 = time.perf_counter()
        for _ in range(n_steps):
            call("vq_timer_start", tm, sptr)
            front.scan(weights=w)                            # one launch per rank: dots, ensemble mean, weighted score
            call("vq_timer_stop", tm, sptr)
            if gather and sdb is not None:
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(stream)
                sdb.scores_tensor()                          # score slices -> global order on every rank (N x 8 B; RCCL)
                g1.record(stream)
                gath.append((g0, g1))
            ms = C.c_float()
            call("vq_timer_elapsed_ms", tm, C.byref(ms))
            kern += ms.value
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0, kern / n_steps, gath

    steps, warm = max(args.steps, 5), max(args.warmup, 2)
    with torch.cuda.stream(stream):
        t = front.set_query_from_row(12345)                  # the owner's GPU scales the row; the others receive 80 KB
        # for the record: the row-major block as loaded (what rounds 1-3 measured), then the SAME block tiled in place
        timed_scans(2, False)
        _, rows_ms, _ = timed_scans(3, False)
        torch.cuda.synchronize(device)
        tp = time.perf_counter()
        front.set_layout("tiled")
        torch.cuda.synchronize(device)
        prepare_ms = (time.perf_counter() - tp) * 1e3
        timed_scans(warm, True)
        if world > 1:
            dist.barrier()
        dt, kern_ms, sim_gather = timed_scans(steps, True)
        if world > 1:
            dist.barrier()
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in sim_gather])) if sim_gather and not rehearse else None
    nbytes = rows * SIM_S * SIM_E * SIM_D * 4 + rows * 8
    roof = {"bound": "hbm", "kernel": "scan_tiled_kernel<2,5,4,%d>" % TILED_GROUP, "achieved": nbytes / kern_ms / 1e6, "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": nbytes / kern_ms / 1e6 / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": kern_ms,
            "bytes_per_launch": nbytes, "score_all_gather_ms_per_query": gather_ms,
            "row_major": {"kernel": "scan_kernel<float,2,5,4>", "avg_launch_ms": rows_ms, "frac": nbytes / rows_ms / 1e6 / PEAK_HBM_GBS},
            "tile_in_place_ms": prepare_ms}
    # HBM bytes per launch from the PMC counters (collected off-line by tools/pmc_sim.sh on the full 1M-row launch and
    # committed; FETCH_SIZE doubled as the microarchitecture guide prescribes for gfx950); scaled to this rank's rows
    scan_traffic = None
    for name in ("r05_scan_traffic.json", "r04_scan_traffic.json", "r03_scan_traffic.json", "r01_scan_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(tpath):
            with open(tpath) as f:
                scan_traffic = json.load(f)
            roof["traffic"] = scan_traffic["hbm_bytes_per_launch"] * rows / SIM_N
            roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, per launch; committed, not collected in this run)" % name
            break
    # batched form (reported beside the single-query metric): 16 queries per pass over the database on the fp64 matrix cores
    # -- the pass stays HBM-bound, so its roofline is the same byte count over its own duration
    Q = 16
    tb = np.stack([front.set_query_from_row(12345 + 1000 * i) for i in range(Q)])
    wb = np.stack([[1.0, 1.5 + 0.05 * i] for i in range(Q)])
    front.set_query(t)                                      # leave the single-query state as the checks below expect it
    front.scan(weights=w)
    with torch.cuda.stream(stream):
        front.scan_batch(tb, wb, want=False)
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        tb0 = time.perf_counter()
        reps = max(3, steps // 4)
        ev_ms = 0.0
        for _ in range(reps):
            call("vq_timer_start", tm, sptr)
            front.scan_batch(tb, wb, want=False)
            call("vq_timer_stop", tm, sptr)
            ms = C.c_float()
            call("vq_timer_elapsed_ms", tm, C.byref(ms))
            ev_ms += ms.value
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        bdt = (time.perf_counter() - tb0) / reps
        ev_ms /= reps
    # one launch reads the database once and writes Q score vectors
    bbytes = rows * SIM_S * SIM_E * SIM_D * 4 + Q * rows * 8
    roof["batched"] = {"queries_per_pass": Q, "value": Q / bdt, "unit": "queries/s", "ms_per_pass": bdt * 1e3,
                       "kernel": "batch_fused_kernel<float,4,2,false,8,tiled>", "pass_ms_by_hip_events": ev_ms, "bytes_per_pass": bbytes,
                       "hbm_GBps": bbytes / ev_ms / 1e6, "hbm_frac": bbytes / ev_ms / 1e6 / PEAK_HBM_GBS,
                       "traffic": (scan_traffic or {}).get("batch_fused_kernel", {}).get("hbm_bytes_per_launch"),
                       "mfma_f64_tflops": 2.0 * Q * rows * SIM_S * SIM_E * SIM_D / ev_ms / 1e9, "mfma_f64_peak_tflops": 78.6,
                       "database_bytes_resident": rows * SIM_S * SIM_E * SIM_D * 4,
                       "note": "vq_db_scan_batch on the block tiled in place (no second copy): the database is read ONCE for 16 queries by a single "
                               "launch; value = wall clock over the passes incl. the 1.3 MB query upload of each; hbm_* from the HIP events around the pass"}
    call("vq_timer_destroy", tm)
    return dt, steps, roof, front, row0, rows


def cpu_baseline_sim(db, row0):
    """The oracle's dense restatement of ticket.py:120-180 (numpy fp64) on a bounded sample of the same synthetic rows:
    once on one thread (the reference itself is single-threaded by construction) and once on all host cores (row chunks
    on a thread pool; numpy releases the GIL inside einsum), as SURVEY.md 8(d) asks."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import sim_oracle as so
    from concurrent.futures import ThreadPoolExecutor
    n = 20000
    x = so.synth_features(17, row0, n, SIM_S, SIM_E, SIM_D, (4.0, 1.0))
    t = np.stack([[so.scale_feature(x[7, s, e].astype(np.float64)) for e in range(SIM_E)] for s in range(SIM_S)])

    def one_pass(rows):
        _, avg, _ = so.dense_similarities(rows, t)
        return so.dense_scores(avg, [1.0, 1.5])

    def timed(fn, budget):
        t0 = time.perf_counter()
        reps = 0
        while reps == 0 or time.perf_counter() - t0 < budget:
            fn()
            reps += 1
        return (time.perf_counter() - t0) / reps
    dt1 = timed(lambda: one_pass(x), 6.0)
    cores = host_cores()
    chunks = np.array_split(np.arange(n), cores * 2)
    with ThreadPoolExecutor(max_workers=cores) as pool:
        dtn = timed(lambda: list(pool.map(lambda idx: one_pass(x[idx[0]:idx[-1] + 1]), chunks)), 6.0)
    out = {"value": 1.0 / (dtn * SIM_N / n), "unit": "queries/s", "cores": cores, "kind": "port",
           "sample": "%d of the 1M rows (same generator), oracle/sim_oracle.py numpy fp64 einsum, %d row chunks on %d threads, %.3f s "
                     "per pass, scaled by 1M/%d" % (n, len(chunks), cores, dtn, n),
           "single_thread": {"value": 1.0 / (dt1 * SIM_N / n), "unit": "queries/s", "cores": 1,
                             "sample": "same sample on one thread, %.3f s per pass" % dt1}}
    # the UNMODIFIED reference cannot travel to the GPU box; its own rate was taken in the build container
    rpath = os.path.join(ROOT, "profiles", "r01_reference_cfg1_container.json")
    if os.path.exists(rpath):
        with open(rpath) as f:
            r = json.load(f)
        out["reference_unmodified_in_build_container"] = {
            "value": r.get("queries_per_s"), "unit": "queries/s", "cores": 1,
            "config": "BASELINE configs[0]: 10 000 clips x 2 streams x 3 splits x 1024 (NOT the 1M-row workload of this object)",
            "source": "profiles/r01_reference_cfg1_container.json (oracle/time_reference_cfg1.py: Ticket.compute_similarities + "
                      "compute_scores + select_clips_to_review of the reference, imported unmodified; 8-vCPU build container)"}
    return out


def bench_flow(device_index, with_cpu):
    """The step two in front of hot path A: TV-L1 flow of a batch of 340 x 256 grey frame pairs -- the
    size the TSN pipeline resizes to -- plain and camera-motion compensated ("warped"), through the Python handle (frames go
    in and 8-bit flow images come out over PCIe: 22 MB per batch, inside the timing).  Synthetic frames: a smooth texture that
    moves by a different sub-pixel translation per pair."""
    from video_query_algorithms_amd.tsn.flow import Tvl1Flow
    n, h, w = 64, 256, 340
    rng = np.random.default_rng(4)
    f0 = np.empty((n, h, w), np.uint8)
    f1 = np.empty((n, h, w), np.uint8)
    for k in range(n):
        t = np.random.default_rng(k % 8).random((h + 64, w + 64))
        for _ in range(6):
            t = (t + np.roll(t, 1, 0) + np.roll(t, -1, 0) + np.roll(t, 1, 1) + np.roll(t, -1, 1)) / 5.0
        t = (t - t.min()) / (t.max() - t.min()) * 255.0
        dx, dy = rng.uniform(-5, 5), rng.uniform(-3, 3)
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
        x, y = xs + 32 - dx, ys + 32 - dy
        x0, y0 = np.floor(x).astype(int), np.floor(y).astype(int)
        fx, fy = x - x0, y - y0
        moved = t[y0, x0] * (1 - fx) * (1 - fy) + t[y0, x0 + 1] * fx * (1 - fy) + t[y0 + 1, x0] * (1 - fx) * fy + t[y0 + 1, x0 + 1] * fx * fy
        f0[k], f1[k] = np.rint(t[32:32 + h, 32:32 + w]), np.rint(moved)
    m = Tvl1Flow(n, h, w, device=device_index)
    m.flow(f0, f1, fields=False)
    reps = 3
    inner_ms, launches = 0.0, 0
    t0 = time.perf_counter()
    for _ in range(reps):
        r = m.flow(f0, f1, fields=False, iterations=True)
        ms, nl = m.last_timing()                                  # HIP events around every inner loop, inside the library
        inner_ms += ms / reps
        launches = nl
    dt = (time.perf_counter() - t0) / reps
    its = r["iters"]                                              # [levels, warps, pairs], coarsest level first
    px = np.array([a * b for a, b in m.levels[::-1]], dtype=np.float64)
    pixel_iters = float((its.sum(axis=1) * px[:, None]).sum())
    # Algorithmic bytes: a block of 4 iterations reads u1, u2, p11..p22 and the four constant planes of its pixels and writes u1,
    # u2, p11..p22 (16 floats per pixel and block: the blocked form keeps the fields on chip in between); blocks run = ceil(iters / 4)
    # per (level, warp, pair).  The streaming two-launch form of rounds 1-2 moved 22 floats per pixel and ITERATION.
    pixel_blocks = float((np.ceil(its / 4.0).sum(axis=1) * px[:, None]).sum())
    abytes = pixel_blocks * 16 * 4
    gbs = abytes / inner_ms / 1e6
    m.warped(f0, f1)
    t0 = time.perf_counter()
    for _ in range(reps):
        wr = m.warped(f0, f1)
    dw = (time.perf_counter() - t0) / reps
    prof = None
    for name in ("r05_flow_summary.json", "r04_flow_summary.json", "r03_flow_summary.json"):
        ppath = os.path.join(ROOT, "profiles", name)
        if os.path.exists(ppath):
            with open(ppath) as f:
                prof = json.load(f)
            prof["file"] = name
            break
    # The blocked kernel is bound by vector-ALU issue, not by bandwidth (five correctly rounded divisions and two square roots per pixel and
    # iteration; the fields stay on chip between the 4 iterations of a launch).  achieved = wave instructions per second: the committed
    # PMC pass's SQ_INSTS_VALU per pixel-iteration (the instruction count of an iteration does not depend on the run) x this run's
    # pixel-iterations / this run's device time of the inner loops; peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction.
    peak_valu = 256 * 4 * 2.4e9 / 4 / 1e9
    per_pi = (prof or {}).get("valu_wave_insts_per_pixel_iteration")
    flow_roof = {"kernel": "tvl1_tile_kernel<512> (4 inner iterations per launch on tiles fitted to the level, resident in registers / LDS)",
                 "avg_launch_ms": inner_ms / max(launches, 1), "pixel_iterations_per_second": pixel_iters / inner_ms * 1e3,
                 "traffic": prof.get("hbm_bytes_per_batch") if prof else None, "hbm_GBps_algorithmic": gbs, "hbm_frac": gbs / PEAK_HBM_GBS,
                 "bytes_per_batch": abytes,
                 "pmc_source": ("profiles/%s (separate rocprofv3 --pmc passes of tools/flow_profile.py: SQ_INSTS_VALU / GRBM_GUI_ACTIVE, FETCH_SIZE x2, "
                                "WRITE_SIZE; committed, not collected in this run)" % prof["file"]) if prof else None}
    if per_pi:
        wips = per_pi * pixel_iters / inner_ms / 1e6                                  # G wave-instructions / s
        flow_roof.update({"bound": "valu", "achieved": wips, "peak": peak_valu, "unit": "G wave-inst/s", "frac": wips / peak_valu,
                          "valu_thread_insts_per_pixel_iteration": per_pi * 64, "pmc_valu_issue_utilisation": prof.get("valu_issue_utilisation")})
    else:                                                                              # no VALU pass committed yet: the byte figure only
        flow_roof.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS})
    out = {"metric": "frame pairs/sec TV-L1 flow (340x256, OpenCV default parameters)", "value": n / dt, "unit": "pairs/s",
           "batch_pairs": n, "ms_per_batch": dt * 1e3, "mean_inner_iterations_per_warp": float(its.mean()),
           "pixel_iterations_per_batch": pixel_iters, "iteration_launches_per_batch": launches,
           "inner_loops_device_ms_per_batch": inner_ms,
           "roofline": flow_roof,
           "warped": {"value": n / dw, "unit": "pairs/s", "ms_per_batch": dw * 1e3, "mean_corners": float(wr["matches"].mean()),
                      "mean_inliers": float(wr["inliers"].mean()),
                      "note": "first-pass flow + corners + RANSAC homography + second-pass flow on the compensated frame"},
           "parity": "unpinned (third-party binary absent from the reference); kernels vs oracle/tvl1_oracle.py and oracle/warp_oracle.py "
                     "in tests/test_flow_gpu.py, tests/test_warp_gpu.py"}
    # the opt-in form with the hardware's reciprocal / square root (VQ_FLOW_FAST=1: 1-ulp operations; the 8-bit flow images equal the
    # default's on 99.6 % of the pixels and differ freely where the flow is not determined -- outside the tested tolerances, hence opt-in)
    os.environ["VQ_FLOW_FAST"] = "1"
    try:
        mf = Tvl1Flow(n, h, w, device=device_index)
    finally:
        del os.environ["VQ_FLOW_FAST"]
    mf.flow(f0, f1, fields=False)
    t0 = time.perf_counter()
    for _ in range(reps):
        mf.flow(f0, f1, fields=False)
    dtf = (time.perf_counter() - t0) / reps
    mf.warped(f0, f1)
    t0 = time.perf_counter()
    for _ in range(reps):
        mf.warped(f0, f1)
    dwf = (time.perf_counter() - t0) / reps
    mf.close()
    out["fast_math"] = {"value": n / dtf, "unit": "pairs/s", "warped": n / dwf, "opt_in": "VQ_FLOW_FAST=1"}
    if with_cpu:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
        import tvl1_oracle
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 4.0 and k < n:
            tvl1_oracle.tvl1_flow(f0[k], f1[k])
            k += 1
        out["cpu_baseline"] = {"value": k / (time.perf_counter() - t0), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": "%d of the %d pairs through oracle/tvl1_oracle.py (numpy, one thread)" % (k, n)}
    m.close()
    return out


def bench_jpeg(device_index):
    """Frame ingest in front of hot path A: a batch of 340 x 256, 4:2:0, quality-95 JPEG frames (cv2.imwrite's defaults)
    decoded into device memory -- Huffman decoding on the library's host threads, IDCT / upsampling / colour on the GPU.
    Needs Pillow to MAKE the test files (and to time libjpeg-turbo beside it); absent -> the section is skipped."""
    try:
        import io
        from PIL import Image
    except ImportError:
        return None
    from video_query_algorithms_amd.tsn.jpeg import JpegDecoder
    n, h, w = 256, 256, 340
    rng = np.random.default_rng(9)
    ys, xs = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 100 * np.sin(xs / 7.0) + 20 * np.cos(ys / 3.0), 127 + 90 * np.cos(ys / 9.0) + 30 * np.sin(xs / 2.5),
                     127 + 80 * np.sin((xs + ys) / 11.0)], -1)
    files = []
    for _ in range(n):
        buf = io.BytesIO()
        Image.fromarray(np.clip(base + rng.normal(0, 12, base.shape), 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=95, subsampling=2)
        files.append(buf.getvalue())
    dec = JpegDecoder(n, h, w, device_index)
    dec.decode_to_device(files)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        dec.decode_to_device(files)
    dt = (time.perf_counter() - t0) / reps
    got = dec.decode(files[:4])
    same = all((got[i] == np.asarray(Image.open(io.BytesIO(files[i])).convert("RGB"))[:, :, ::-1]).all() for i in range(4))
    t0 = time.perf_counter()
    for f in files[:64]:
        np.asarray(Image.open(io.BytesIO(f)).convert("RGB"))
    dp = (time.perf_counter() - t0) / 64
    dec.close()
    # large batches (what the command line hands over: the files of 32 clips at once): entropy decoding on the device
    big = {}
    flow_blobs = []
    for k in range(16):
        buf = io.BytesIO()
        flow = 128 + 20 * np.sin(xs / (23.0 + k)) * np.cos(ys / 31.0) + rng.normal(0, 1.0, (h, w))
        Image.fromarray(np.clip(flow, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=95)
        flow_blobs.append(buf.getvalue())
    for label, blobs, color, nb in (("rgb_4096", files, True, 4096), ("grey_flow_8192", flow_blobs, False, 8192)):
        batch = [blobs[i % len(blobs)] for i in range(nb)]
        d2 = JpegDecoder(nb, h, w, device_index)
        d2.decode_to_device(batch, color=color)
        t0 = time.perf_counter()
        for _ in range(2):
            d2.decode_to_device(batch, color=color)
        bt = (time.perf_counter() - t0) / 2
        d2.close()
        big[label] = {"value": nb / bt, "unit": "frames/s", "batch_frames": nb, "ms_per_batch": bt * 1e3, "mean_file_kb": sum(map(len, batch)) / nb / 1024,
                      "entropy_decoding": "device (jpeg_entropy_idct_kernel: one lane per stream)"}
    # the command line's own batch: 32 clips at T = 25 = 800 RGB files (below the 2 048-stream threshold: host entropy decoding)
    # and 8 000 grey flow files (device entropy decoding), one call per stream as CaffeNet.crops_from_jpegs makes them
    cli = {}
    for label, blobs, color, nb in (("rgb", files, True, 800), ("flow", flow_blobs, False, 8000)):
        batch = [blobs[i % len(blobs)] for i in range(nb)]
        d3 = JpegDecoder(nb, h, w, device_index)
        d3.decode_to_device(batch, color=color)
        t0 = time.perf_counter()
        for _ in range(3):
            d3.decode_to_device(batch, color=color)
        cli[label] = (time.perf_counter() - t0) / 3
        d3.close()
    cli_frames = 8800
    return {"metric": "JPEG frames/sec decoded into device memory (340x256, quality 95: 4:2:0 RGB and grey flow frames), the batch the drop-in "
                      "command line hands over",
            "value": cli_frames / (cli["rgb"] + cli["flow"]), "unit": "frames/s", "batch_frames": cli_frames,
            "config": {"workload": "32 clips x T=25 as calcSig_wOF.py --device_jpeg reads them: one call of 800 RGB files (entropy decoding on host "
                                   "threads) + one call of 8 000 grey flow files (entropy decoding on the device)"},
            "ms_per_batch": {"rgb_800": cli["rgb"] * 1e3, "flow_8000": cli["flow"] * 1e3},
            "small_batch": {"value": n / dt, "unit": "frames/s", "batch_frames": n, "ms_per_batch": dt * 1e3, "mean_file_kb": sum(len(f) for f in files) / n / 1024,
                            "entropy_decoding": "host threads (batches below 2 048 streams)",
                            "note": "256 RGB frames in one call: the round-2 figure, bound by %d host threads" % min(16, os.cpu_count() or 1)},
            "large_batches": big,
            "host_threads": min(16, os.cpu_count() or 1), "bit_identical_to_libjpeg_turbo": bool(same),
            "cpu_baseline": {"value": 1.0 / dp, "unit": "frames/s", "cores": 1, "kind": "reference",
                             "sample": "64 of the files through Pillow's libjpeg-turbo (the library cv2.imread decodes with), one thread"}}


def bench_e2e_cli(device_index):
    """The drop-in command line end to end on ONE GPU: a JPEG frame tree (img_ / flow_x_ / flow_y_ files as build_wof_clips.py leaves
    them) -> decode -> resize + crop -> both TSN streams at the reference's default T = 25 -> consensus -> CSV tree.  256 clips of 30
    frames (340 x 256, quality 95, 4:2:0 / grey); per clip the command line reads 25 RGB and 250 flow files.  --device_jpeg: the files'
    bytes go to the library (entropy decoding on host threads for the 800 RGB files of a batch, on the device for its 8 000 flow
    files), nothing is decoded by a host image library.  Needs Pillow to MAKE the files; absent -> skipped."""
    try:
        import io
        from PIL import Image
    except ImportError:
        return None
    import contextlib
    import shutil
    import tempfile
    from video_query_algorithms_amd import calcSig_wOF
    from video_query_algorithms_amd.tsn.caffe_net import CaffeNet
    n_clips, n_frames, h, w = 256, 30, 256, 340
    rng = np.random.default_rng(21)
    ys, xs = np.mgrid[0:h, 0:w]
    rgb_blobs, grey_blobs = [], []
    for k in range(8):
        base = np.stack([127 + 90 * np.sin(xs / (7.0 + k) + k) + 20 * np.cos(ys / 3.0), 127 + 80 * np.cos(ys / 9.0) + 30 * np.sin(xs / 2.5),
                         127 + 70 * np.sin((xs + ys) / 11.0)], -1)
        buf = io.BytesIO()
        Image.fromarray(np.clip(base + rng.normal(0, 6, base.shape), 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=95, subsampling=2)
        rgb_blobs.append(buf.getvalue())
        buf = io.BytesIO()
        flow = 128 + 20 * np.sin(xs / (23.0 + k)) * np.cos(ys / 31.0) + rng.normal(0, 1.0, (h, w))          # smooth, like a flow image
        Image.fromarray(np.clip(flow, 0, 255).astype(np.uint8)).save(buf, "JPEG", quality=95)
        grey_blobs.append(buf.getvalue())
    root = tempfile.mkdtemp(prefix="vq_e2e_")
    try:
        for c in range(n_clips):
            d = os.path.join(root, "frames", "video", "clip_%04d" % (c + 1))
            os.makedirs(d)
            for i in range(1, n_frames + 1):
                for name, blob in (("img", rgb_blobs[(c + i) % 8]), ("flow_x", grey_blobs[(c + i) % 8]), ("flow_y", grey_blobs[(c + 3 * i) % 8])):
                    with open(os.path.join(d, "%s_%05d.jpg" % (name, i)), "wb") as f:
                        f.write(blob)

        def factory(proto, weights, dev, **kw):
            ch = 3 if proto == "rgb" else 10
            return CaffeNet(bn_inception.bn_inception(ch), "synthetic:%d" % int(weights.split("seed")[1].split(".")[0]), dev, **kw)
        os.makedirs(os.path.join(root, "frames32", "video"))
        for c in range(32):                                   # the first 32 clips once more as a tree of their own (links)
            os.symlink(os.path.join(root, "frames", "video", "clip_%04d" % (c + 1)), os.path.join(root, "frames32", "video", "clip_%04d" % (c + 1)))
        times = []
        # the first run builds the weight cache (its tilings come with the library); the 256-clip job is then timed three times (the same binary moves by
        # +-10 % from run to run on one box: thread scheduling around the first batch) and the MEDIAN is the value
        for rep, tree in enumerate(("frames", "frames", "frames", "frames", "frames32")):
            out_dir = os.path.join(root, "features%d" % rep)
            argv = [os.path.join(root, tree), "rgb", "rgb_seed2.caffemodel", "flow", "flow_seed5.caffemodel", "--outFeatures_dir", out_dir,
                    "--modelname", "UCF101_split1", "--num_worker", "16", "--gpus", str(device_index), "--device_jpeg"]
            if os.environ.get("VQ_BENCH_BATCH_CLIPS"):                  # A/B of the command line's --batch_clips (tools/e2e_ab.sh)
                argv += ["--batch_clips", os.environ["VQ_BENCH_BATCH_CLIPS"]]
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                rc = calcSig_wOF.main(argv, net_factory=factory)
            times.append(time.perf_counter() - t0)
            assert rc == 0
        csv = os.path.join(root, "features1", "video", "UCF101_split1", "rgb_global_pool_features.csv")
        rows = sum(1 for _ in open(csv)) - 1
        # the ensemble of calcSig_wOF_ensemble.sh:13-37 (three weight sets over the same frame tree) as ONE command: every frame read,
        # decoded and resized once, three RGB and three flow networks; against three runs of the command above
        ens_t = []
        for rep in range(2):
            argv = [os.path.join(root, "frames"), "rgb", "rgb_seed2.caffemodel", "flow", "flow_seed5.caffemodel", "--outFeatures_dir",
                    os.path.join(root, "ens%d" % rep), "--modelname", "UCF101_split1", "--num_worker", "16", "--gpus", str(device_index), "--device_jpeg",
                    "--ensemble", "UCF101_split2", "rgb_seed3.caffemodel", "flow_seed6.caffemodel",
                    "--ensemble", "UCF101_split3", "rgb_seed4.caffemodel", "flow_seed7.caffemodel"]
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                rc = calcSig_wOF.main(argv, net_factory=factory)
            ens_t.append(time.perf_counter() - t0)
            assert rc == 0
        same = open(csv, "rb").read() == open(os.path.join(root, "ens1", "video", "UCF101_split1", "rgb_global_pool_features.csv"), "rb").read()
        # ONE run as a fresh process -- `python calcSig_wOF.py ...` as a user types it (a child, never a re-exec): interpreter start, torch
        # import, HIP context, first allocations, extractors built from the weight cache, default device-pool cap.  Same bytes.
        fresh = None
        try:
            import subprocess
            for name, ch in (("rgb", 3), ("flow", 10)):
                with open(os.path.join(root, name + ".prototxt"), "w") as f:
                    f.write(bn_inception.to_prototxt(bn_inception.bn_inception(ch)))
            cli = os.path.join(ROOT, "video-query-algorithms_amd", "calcSig_wOF.py")
            env = {k: v for k, v in os.environ.items() if k not in ("VQ_DEVICE_POOL_GB", "RANK", "WORLD_SIZE", "LOCAL_RANK")}
            argv = [sys.executable, cli, os.path.join(root, "frames"), os.path.join(root, "rgb.prototxt"), "synthetic:2", os.path.join(root, "flow.prototxt"),
                    "synthetic:5", "--outFeatures_dir", os.path.join(root, "fresh"), "--modelname", "UCF101_split1", "--num_worker", "16",
                    "--gpus", str(device_index), "--device_jpeg"]
            # The child stands for ONE invocation -- but it starts beside THIS process, which holds its block pool and has just released the
            # 41 GB database of configs[3]: its hipMalloc of 2 x 16 GB waits for the driver by box (1.0-2.4 s here against 0.95-1.05 s for
            # the same command with a quiet GPU, tools/fresh_runs.py; a few idle seconds before it did not help, draining this process's
            # pool first -- vq_device_pool_trim -- made it 3.6 s).  The figure is an upper bound of what a user's invocation costs.
            t0 = time.perf_counter()
            r = subprocess.run(argv, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=600)
            fresh_s = time.perf_counter() - t0
            if os.environ.get("VQ_CLI_TRACE") == "1":                       # tools/e2e_trace.py: the child's own stamps
                sys.stderr.write("fresh process (%.3f s):\n%s" % (fresh_s, r.stderr.decode(errors="replace")[-3000:]))
            if r.returncode == 0:
                fcsv = os.path.join(root, "fresh", "video", "UCF101_split1", "rgb_global_pool_features.csv")
                body = lambda path: open(path, "rb").read().split(b"\n", 1)[1]          # the header names the weights file: rows only
                fresh = {"seconds": fresh_s, "value": n_clips / fresh_s, "unit": "clips/s", "rows_equal_in_process_run": body(fcsv) == body(csv)}
            else:
                fresh = {"error": r.stderr.decode(errors="replace")[-300:]}
        except Exception as e:          # noqa: BLE001 -- the in-process figures stand on their own
            fresh = {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    runs = sorted(times[1:4])
    times = [times[0], runs[1], times[4]]
    steady = (n_clips - 32) / max(times[1] - times[2], 1e-9)
    return {"metric": "clips/sec end to end through the drop-in command line (JPEG frame tree -> CSV tree), two-stream, T=25; warm: median of "
                      "in-process repeats of main() (device block pool, decoder pool, weight and page caches warm)", "value": n_clips / times[1],
            "unit": "clips/s", "clips": n_clips, "seconds": times[1], "seconds_of_the_three_runs": runs, "first_run_seconds": times[0],
            "fresh_process": fresh,
            "seconds_32_clips": times[2], "csv_rows": rows,
            "ensemble3": {"value": 3 * n_clips / ens_t[1], "unit": "(clip, member)/s", "seconds": ens_t[1], "first_run_seconds": ens_t[0],
                          "vs_three_runs": (3 * n_clips / ens_t[1]) / (n_clips / times[1]), "member_1_bytes_equal_single_run": bool(same)},
            "steady_state": {"value": steady, "unit": "clips/s",
                             "note": "(256 - 32 clips) / (time of the 256-clip run - time of a 32-clip run): what a long job sees once the two "
                                     "network handles exist (building them -- packed weights read from the cache, upload, buffers -- is %.2f s of every "
                                     "run; the first run on a machine also reads, folds and transforms the weights)" % max(times[2] - 32 / max(steady, 1e-9), 0.0)},
            "files_read_per_clip": 25 + 250, "mean_file_kb": {"rgb": sum(map(len, rgb_blobs)) / 8 / 1024, "flow": sum(map(len, grey_blobs)) / 8 / 1024},
            "config": {"workload": "calcSig_wOF.py --device_jpeg --num_worker 16, 256 clips x 30 frames of 340x256, T=25 (the reference's default), "
                                   "both streams, one GPU, network handles rebuilt per run (packed weights from the cache next to the library, as on every "
                                   "run after a machine's first)"},
            "note": "whole process time of main(): directory parsing, reading 70 400 files (8 batches of 32 clips per stream), JPEG decoding, resize + crop, 800 + 800 crops "
                    "through the two networks (which also loads / folds / uploads the weights), CSV writing"}


def bench_e2e_wof(device_index, max_pairs=64):
    """The drop-in build_wof_clips.py command line end to end on ONE GPU: a "video" (a directory of its 257 frames of 340 x 256, the form
    the command line takes where cv2.VideoCapture is missing) -> grey conversion -> warped TV-L1 flow in windows of 64 pairs -> 768
    img_ / flow_x_ / flow_y_ JPEG files (encoded by --num_worker host threads) -> clip directories.  Needs Pillow; absent -> skipped."""
    try:
        import io
        from PIL import Image
    except ImportError:
        return None
    import contextlib
    import shutil
    import tempfile
    from video_query_algorithms_amd import build_wof_clips
    n_frames, h, w = 257, 256, 340
    rng = np.random.default_rng(33)
    tex = rng.random((h + 128, w + 128))
    for _ in range(6):
        tex = (tex + np.roll(tex, 1, 0) + np.roll(tex, -1, 0) + np.roll(tex, 1, 1) + np.roll(tex, -1, 1)) / 5.0
    tex = (tex - tex.min()) / (tex.max() - tex.min()) * 255.0
    root = tempfile.mkdtemp(prefix="vq_wof_")
    try:
        src = os.path.join(root, "src", "pan")
        os.makedirs(src)
        for t in range(n_frames):                            # a slow pan over a smooth texture
            x0, y0 = 32 + (t * 3) // 8, 32 + t // 8
            g = np.clip(tex[y0:y0 + h, x0:x0 + w] + rng.normal(0, 1.0, (h, w)), 0, 255).astype(np.uint8)
            Image.fromarray(np.repeat(g[:, :, None], 3, 2)).save(os.path.join(src, "frame_%05d.jpg" % t), "JPEG", quality=95, subsampling=2)
        times = []
        for rep in range(2):
            out = os.path.join(root, "out%d" % rep)
            t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                rc = build_wof_clips.main([os.path.join(root, "src"), out, "--fps", "15", "--clip_time", "10", "--num_worker", "16",
                                           "--starting_gpu", str(device_index)] + (["--max_pairs", str(max_pairs)] if max_pairs else []))
            times.append(time.perf_counter() - t0)
            assert rc == 0
        clips = sorted(os.listdir(os.path.join(root, "out1", "pan")))
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return {"metric": "frames/sec end to end through the drop-in build_wof_clips.py (frames -> warped TV-L1 flow -> img / flow_x / flow_y JPEG files -> clips)",
            "value": (n_frames - 1) / times[1], "unit": "frames/s", "frames": n_frames - 1, "seconds": times[1], "first_run_seconds": times[0],
            "clip_directories": clips,
            "config": {"workload": "one 257-frame video of 340x256 given as a directory of frames, --num_worker 16 (--max_pairs: %s), one GPU"
                                   % (max_pairs or "the command line's default")},
            "note": "whole process time of main(): reading and decoding the frames (host image library), grey conversion, 256 warped flows on the "
                    "GPU, 768 JPEG encodings on 16 host threads, the clip regrouping"}


def bench_rounds(device_index, with_cpu):
    """BASELINE configs[0] and configs[4] on the B seam, through the drop-in Ticket / Hyperparameter / TargetClip objects over a resident
    FeatureDB (SURVEY.md 8(d) cfg 1 data: 10 000 clips x 2 streams x 3 splits x 1024 fp32, |N(0,1)| x 3.8 / x 1.2, reference clip = row 7,
    broker defaults):
      * one query = compute_similarities + compute_scores + select_clips_to_review (ticket.py:120-180,311-356) -- the three calls the
        unmodified reference was timed on (oracle/time_reference_cfg1.py);
      * 100 weight-update rounds = {compute_similarities, compute_scores, 20 labels off the top of the ranking, optimize_weights
        (hyperparameter.py:29-76), compute_scores, select_clips_to_review} on the same resident database.
    cpu_baseline: the oracle's FAITHFUL restatement of the same reference lines (dicts of Python lists, np.dot per (clip, stream, split),
    one thread -- as the reference is) on a bounded sample of the clips, scaled; beside it the unmodified reference's own rate as timed in
    the build container this round (it cannot travel to the GPU box)."""
    import random
    os.environ.setdefault("COMPUTE_EPS", "0.000003")
    streams, splits, weights0 = ("rgb", "warped_optical_flow"), (1, 2, 3), {"rgb": 1.0, "warped_optical_flow": 1.5}
    n, d = 10000, 1024
    rng = np.random.default_rng(0)                       # SURVEY.md 8(d) cfg 1 (= oracle/sim_oracle.cfg1_features)
    x = np.abs(rng.standard_normal((3, 2, n, d), dtype=np.float32))
    x[:, 0] *= np.float32(3.8)
    x[:, 1] *= np.float32(1.2)
    x = np.ascontiguousarray(x.transpose(2, 1, 0, 3))    # [N, S, E, D]
    clip_ids = np.arange(1, n + 1, dtype=np.int64)
    db = vqa.FeatureDB.from_arrays(x, clip_ids=clip_ids, device=device_index)
    db.stream_names, db.slot_splits = list(streams), [list(splits)] * 2
    ref_row = 7
    ref_records = [{"dnn_stream_id": st, "dnn_stream_split": sp, "name": "global_pool", "video_clip_id": int(clip_ids[ref_row]),
                    "feature_vector": x[ref_row, si, ei].astype(np.float64).tolist()} for si, st in enumerate(streams) for ei, sp in enumerate(splits)]

    def ticket():
        hp = vqa.Hyperparameter(weights0, 0.8, 0.0, 0.35, 0.0, streams, "global_pool", 1, 0.7, "bagging", 3)      # broker.py:36-59
        tk = vqa.Ticket({"query_id": 1, "video_id": 1, "ref_clip": 0, "ref_clip_id": int(clip_ids[ref_row]), "search_set": 1,
                         "number_of_matches_to_review": 20, "dynamic_target_adjustment": False, "user_matches": {}},
                        records=ref_records, feature_db=db, device=device_index)
        tk.target = vqa.TargetClip(tk, hp)
        tk.target.get_target_features()
        return tk, hp
    # ---- configs[0]: one query
    tk, hp = ticket()
    lat, parts = [], []
    for rep in range(60):
        random.seed(a="73459912436")
        t0 = time.perf_counter()
        tk.compute_similarities(hp)
        t1 = time.perf_counter()
        tk.compute_scores(weights0)
        t2 = time.perf_counter()
        tk.select_clips_to_review(0.8, 20, 0.35)
        t3 = time.perf_counter()
        if rep >= 10:
            lat.append(t3 - t0)
            parts.append((t1 - t0, t2 - t1, t3 - t2))
    med = float(np.median(lat))
    first_matches = dict(tk.matches)
    avg0 = tk._avg.copy()
    query = {"metric": "queries/sec weighted-cosine query on a resident 10k x 2 x 3 x 1024 database (similarities + scores + review set)",
             "value": 1.0 / med, "unit": "queries/s", "ms_per_query": med * 1e3, "queries_timed": len(lat),
             "ms": dict(zip(("compute_similarities", "compute_scores", "select_clips_to_review"), (np.median(np.array(parts), axis=0) * 1e3).tolist())),
             "config": {"workload": "configs[0]: compute_matches.py weighted cosine on 10k x 1024 fp32 features (2 streams x 3 splits), broker defaults"}}
    # ---- configs[4]: 100 weight-update rounds on the resident database
    tk, hp = ticket()
    lab_rng = np.random.default_rng(5)
    random.seed(a="73459912436")
    R, L = 100, 20
    stage = {"similarities": 0.0, "optimize_weights": 0.0, "scores": 0.0, "select": 0.0}
    t_all = time.perf_counter()
    for r in range(R):
        t0 = time.perf_counter()
        tk.compute_similarities(hp)
        stage["similarities"] += time.perf_counter() - t0
        if r == 0:
            hp.weights, hp.threshold = dict(hp.default_weights), hp.default_threshold
        t0 = time.perf_counter()
        tk.compute_scores(hp.weights)                    # under the previous round's weights: what the user reviewed
        stage["scores"] += time.perf_counter() - t0
        rows, vals = db.topk(L)
        tk.matches = [{"video_clip": int(clip_ids[row]), "user_match": bool(lab_rng.random() < 0.5 + 0.4 * (i < L // 2)), "is_match": bool(v >= hp.threshold)}
                      for i, (row, v) in enumerate(zip(rows, vals))]
        t0 = time.perf_counter()
        hp.optimize_weights(tk)
        stage["optimize_weights"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        tk.compute_scores(hp.weights)
        stage["scores"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        tk.select_clips_to_review(hp.threshold, 20, hp.near_miss_default)
        stage["select"] += time.perf_counter() - t0
    total = time.perf_counter() - t_all
    rounds = {"metric": "weight-update rounds/sec on a resident 10k-clip database (similarities, optimize_weights on 20 labels, scores, review set)",
              "value": R / total, "unit": "rounds/s", "rounds": R, "ms_per_round": total / R * 1e3, "ms": {k: v / R * 1e3 for k, v in stage.items()},
              "final_weight": float(hp.weights[streams[1]]), "final_threshold": float(hp.threshold),
              "config": {"workload": "configs[4], the query half: 100 iterative compute_matches weight updates on 10k resident clips "
                                     "(the extraction half: tools/e2e_cfg5.py, profiles/r0N_e2e_cfg5.json)"}}
    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import sim_oracle as so
        ns = 2000                                         # bounded sample: the first 2 000 clips (the faithful walk is linear in the clips)
        recs = [{"dnn_stream_id": st, "dnn_stream_split": sp, "name": "global_pool", "video_clip_id": int(c), "feature_vector": x[c - 1, si, ei].astype(np.float64).tolist()}
                for ei, sp in enumerate(splits) for si, st in enumerate(streams) for c in clip_ids[:ns]]
        target = so.faithful_scaled_ref_clip_features(so.faithful_clip_features(ref_records, streams, "global_pool")[0])
        t0 = time.perf_counter()
        cand = so.faithful_candidate_features(recs, list(splits), streams, "global_pool")
        sims = so.faithful_similarities(target, cand)
        t_sim = time.perf_counter() - t0
        t0 = time.perf_counter()
        sc = so.faithful_scores(sims, weights0)
        random.seed(a="73459912436")
        so.faithful_select(sc, int(clip_ids[ref_row]), {}, 0.8, 20, 0.35)
        t_sc = time.perf_counter() - t0
        ranked = sorted(sc.items(), key=lambda kv: kv[1], reverse=True)[:L]
        labelled = [{"video_clip": c, "user_match": bool(i % 3 != 2), "is_match": bool(v >= 0.8)} for i, (c, v) in enumerate(ranked)]
        t0 = time.perf_counter()
        so.faithful_optimize_weights(sims, labelled, streams, 0.0, float(os.environ["COMPUTE_EPS"]))
        t_opt = time.perf_counter() - t0
        scale = n / ns
        # live parity of the timed path: the product's averaged similarities of the sample against the faithful walk
        err = max(abs(avg0[c - 1, si] - sims[int(c)][st][0]) for c in clip_ids[:ns] for si, st in enumerate(streams))
        ref_box = {}
        rpath = os.path.join(ROOT, "profiles", "r05_reference_cfg1_container.json")
        if os.path.exists(rpath):
            with open(rpath) as f:
                ref_box = json.load(f)
        note = "faithful restatement (oracle/sim_oracle.py: dicts of lists, np.dot per (clip, stream, split)) of %d of the 10 000 clips, one thread, scaled x%d" % (ns, scale)
        query["cpu_baseline"] = {"value": 1.0 / ((t_sim + t_sc) * scale), "unit": "queries/s", "cores": 1, "kind": "port", "sample": note,
                                 "reference_unmodified_in_build_container": _pick(ref_box, ["queries_per_s", "date", "cpu_count"])}
        rounds["cpu_baseline"] = {"value": 1.0 / ((t_sim + 2 * t_sc + t_opt) * scale), "unit": "rounds/s", "cores": 1, "kind": "port", "sample": note,
                                  "reference_unmodified_in_build_container": _pick(ref_box, ["rounds_per_s", "date", "cpu_count"])}
        query["parity_max_abs_err_vs_oracle"] = float(err)
        query["review_set_size"] = len(first_matches)
    db.close()
    return {"query": query, "weight_updates": rounds}


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _r(x, nd=4):
    """Round floats (recursively) so that the line stays short; 4 significant decimals are what the numbers are good for."""
    if isinstance(x, float):
        return float("%.*g" % (nd + 2, x))
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


LINE_BYTES = 4000       # the driver's record keeps the tail of stdout: the line stays under 4 KB


def compact(out):
    """The ONE line the driver records: every number of DESIGN.md section 5, kernel names and config.workload -- no prose (the notes
    live in DESIGN.md and behind --verbose).  Stays under 4 KB so that the driver's record keeps all of it."""
    line = _pick(out, ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"])
    line["vs_baseline"] = out.get("vs_baseline")
    cfg = out["config"]
    line["config"] = {"workload": "configs[1]: TSN BN-Inception RGB, 224x224x3, T=3, B=32/GPU (96 crops/step), crops in HBM",
                      "global_batch": cfg["global_batch"], "parallelism": cfg["parallelism"]}
    if cfg.get("distributed"):
        line["config"]["distributed"] = _pick(cfg["distributed"], ["backend", "world_size", "rccl_version", "rehearsal_on_one_gpu"])
    roof = out["roofline"]
    line["roofline"] = _pick(roof, ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_frac", "matrix_pipe_frac", "avg_launch_ms",
                                    "launches_per_step", "conv_ms_per_step", "other_kernels_ms_per_step", "all_gather_ms_per_step", "profiled_steps",
                                    "traffic_per_step"])
    line["roofline"].setdefault("traffic", None)
    line["roofline"]["kernel"] = "%d conv launches/step: conv_igemm(_pipe) / pool_gemm + wino_f2x2_3x3 kernels, v_mfma_f32_32x32x2" % roof["launches_per_step"]
    line["roofline"]["families"] = {k: _pick(v, ["matrix_pipe_frac", "ms_per_step", "launches"]) for k, v in roof.get("families", {}).items()}
    if isinstance(roof.get("rank_ms_per_step"), dict):
        line["roofline"]["rank_ms_per_step"] = _pick(roof["rank_ms_per_step"], ["min", "max", "all"])
    if "pmc_matrix_pipe_utilisation" in roof:
        line["roofline"]["pmc_matrix_pipe"] = roof["pmc_matrix_pipe_utilisation"]
    line["config"]["timed_mode"] = "vq_tsn_forward default: 2 sub-batch streams, no events; per-kernel fields: " + str(roof.get("kernel_fields_from"))
    if "single_stream" in out:
        line["single_stream"] = _pick(out["single_stream"], ["value", "ms_per_step", "frac", "profiled_steps"])
        line["single_stream"].update(_pick(out["single_stream"].get("kernels") or {}, ["conv_ms_per_step", "kernel_frac", "matrix_pipe_frac"]))
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = _pick(cb, ["value", "unit", "cores", "kind"])
        line["cpu_baseline"]["sample"] = cb["sample"].split(" of the")[0]
        if "ten_crop_variant" in cb:
            line["cpu_baseline"]["ten_crop_value"] = cb["ten_crop_variant"]["value"]
    if "parity_vs_oracle_rel_err" in out:
        line["parity_vs_oracle_rel_err"] = out["parity_vs_oracle_rel_err"]
    ts = out.get("two_stream")
    if ts:
        line["two_stream"] = _pick(ts, ["value", "unit", "ms_per_step", "parity_vs_oracle_rel_err"])
        line["two_stream"]["config"] = {"workload": "configs[2]: RGB + 10-ch flow stack, T=7, B=64 (448+448 crops)"}
        line["two_stream"]["roofline"] = _pick(ts["roofline"], ["bound", "frac", "kernel_frac", "matrix_pipe_frac", "conv_ms_per_step"])
        if "single_stream" in ts:
            line["two_stream"]["single_stream"] = ts["single_stream"]["value"]
        if "cpu_baseline" in ts:
            line["two_stream"]["cpu_baseline"] = _pick(ts["cpu_baseline"], ["value", "cores", "kind"])
    sim = out.get("similarity")
    if sim:
        c = _pick(sim, ["value", "unit", "ms_per_query", "steps", "scaling"])
        c["config"] = {"workload": "configs[3]: 1 query x 1M x (2 x 5) x 1024 fp32 = 40.96 GB, tiled in place",
                       "rows_per_gpu": sim["config"]["rows_per_gpu"]}
        c["roofline"] = _pick(sim["roofline"], ["bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms",
                                                "score_all_gather_ms_per_query", "tile_in_place_ms"])
        c["roofline"]["row_major_frac"] = sim["roofline"]["row_major"]["frac"]
        if "batched" in sim:
            c["batched"] = _pick(sim["batched"], ["value", "ms_per_pass", "hbm_frac", "traffic"])
        if "cpu_baseline" in sim:
            c["cpu_baseline"] = _pick(sim["cpu_baseline"], ["value", "unit", "cores", "kind"])
            c["cpu_baseline"]["single_thread"] = sim["cpu_baseline"]["single_thread"]["value"]
        line["similarity"] = c
    rd = out.get("rounds")
    if rd:
        c = {}
        for key, cfgname in (("query", "configs[0]: 1 query, resident 10k x 2 x 3 x 1024"), ("weight_updates", "configs[4]: 100 weight updates, 10k clips")):
            q = rd[key]
            c[key] = _pick(q, ["value", "unit", "ms_per_query", "ms_per_round", "parity_max_abs_err_vs_oracle"])
            c[key]["config"] = {"workload": cfgname}
            if "cpu_baseline" in q:
                cb = q["cpu_baseline"]
                c[key]["cpu_baseline"] = _pick(cb, ["value", "cores", "kind"])
                ref = cb.get("reference_unmodified_in_build_container") or {}
                c[key]["cpu_baseline"]["reference_unmodified"] = [ref.get("queries_per_s", ref.get("rounds_per_s")), ref.get("date")]
        line["rounds"] = c
    fl = out.get("flow")
    if fl:
        c = _pick(fl, ["value", "unit", "ms_per_batch", "iteration_launches_per_batch"])
        c["warped"] = fl["warped"]["value"]
        c["roofline"] = _pick(fl["roofline"], ["bound", "achieved", "peak", "unit", "frac", "traffic"])
        if "cpu_baseline" in fl:
            c["cpu_baseline"] = _pick(fl["cpu_baseline"], ["value", "cores", "kind"])
        line["flow"] = c
    jp = out.get("jpeg")
    if jp:
        c = _pick(jp, ["value", "unit", "ms_per_batch", "bit_identical_to_libjpeg_turbo"])
        c["small_batch"] = jp["small_batch"]["value"]
        c["cpu_baseline"] = jp["cpu_baseline"]["value"]
        line["jpeg"] = c
    e2e = out.get("e2e_cli")
    if e2e:
        line["e2e_cli"] = _pick(e2e, ["value", "unit", "seconds", "first_run_seconds"])
        line["e2e_cli"]["mode"] = "warm in-process"
        line["e2e_cli"]["fresh_process"] = _pick(e2e.get("fresh_process") or {}, ["seconds", "value", "error"])
        if "error" in line["e2e_cli"]["fresh_process"]:                     # the line must stay under 4 KB whatever went wrong
            line["e2e_cli"]["fresh_process"]["error"] = str(line["e2e_cli"]["fresh_process"]["error"])[-80:]
        line["e2e_cli"]["ensemble3"] = _pick(e2e["ensemble3"], ["value", "vs_three_runs", "member_1_bytes_equal_single_run"])
        line["e2e_cli"]["steady"] = e2e["steady_state"]["value"]
    wof = out.get("e2e_wof_cli")
    if wof:
        line["e2e_wof_cli"] = _pick(wof, ["value", "unit", "seconds"])
    line = _r(line)
    # the bound is enforced, not hoped for (an 8-rank line carries per-rank times and the distributed block): the least important numbers go first
    for path in (("jpeg", "cpu_baseline"), ("jpeg", "small_batch"), ("flow", "cpu_baseline"), ("e2e_wof_cli", "seconds"), ("two_stream", "cpu_baseline"),
                 ("similarity", "cpu_baseline"), ("roofline", "families"), ("single_stream", "matrix_pipe_frac"), ("jpeg",), ("e2e_wof_cli",),
                 ("e2e_cli", "ensemble3"), ("flow", "roofline"), ("similarity", "batched")):
        if len(json.dumps(line, separators=(",", ":"))) <= LINE_BYTES:
            break
        node = line
        for key in path[:-1]:
            node = node.get(key, {}) if isinstance(node, dict) else {}
        if isinstance(node, dict):
            node.pop(path[-1], None)
    return line


def self_launch(args):
    """``python bench.py --gpus N`` with no launcher: start N fresh children -- one rank per GPU over 127.0.0.1 -- BEFORE this process makes
    any GPU call, let rank 0 print the line, return the first non-zero exit code.  (Under torchrun / as a child: do the work.)"""
    from video_query_algorithms_amd import fanout
    if args.gpus <= 1 or "RANK" in os.environ or fanout.is_child():
        return None
    envs = fanout.rank_envs(args.gpus, max(1, host_cores() // args.gpus))
    return fanout.run_children(os.path.abspath(__file__), sys.argv[1:], envs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--skip-sim", action="store_true")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-flow", action="store_true")
    ap.add_argument("--skip-two-stream", action="store_true")
    ap.add_argument("--verbose", action="store_true", help="print the full object (every note, per-stream rooflines, sub-benchmarks' samples) instead of the "
                                                          "compact line the driver records")
    ap.add_argument("--details", default=None, help="also write the full object to this JSON file")
    ap.add_argument("--profile-two-stream", action="store_true",
                    help="with --profile-only: keep the configs[2] section in the process (for a rocprofv3 summary of the 448-crop launches); "
                         "by default --profile-only runs configs[1] (and the scan) only, so that the profiler's per-kernel averages are those "
                         "of the timed region")
    ap.add_argument("--profile-only", action="store_true",
                    help="for rocprofv3 comparisons: every forward of the process runs like the timed region (one stream, "
                         "VQ_TSN_SPLIT=1) and the un-profiled production-mode pass is skipped")
    ap.add_argument("--tune-cache", default="0", help="directory of conv tiling tables to read / extend (default 0: every run times the tilings "
                                                        "itself; tools/profile_round.sh shares one directory between its runs so that the "
                                                        "run under rocprofv3 traces no autotune launch)")
    ap.add_argument("--tiles", default=None, help="JSON file: load the conv tiling table if it exists, else write it")
    args = ap.parse_args()
    rc = self_launch(args)                                   # before ANY GPU call of this process
    if rc is not None:
        sys.exit(rc)
    # This process builds and closes a dozen extractors back to back (two streams, three ensemble members, several command-line runs):
    # their device blocks go round through the library's pool instead of the driver (a hipMalloc of tens of GB right behind a hipFree
    # of as much stalls ~1 s on these hosts, and stalls the queues with it).  A command line of its own never needs this.
    os.environ.setdefault("VQ_DEVICE_POOL_GB", "120")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # VQ_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend -- rehearses the N > 1 control flow (sharding,
    # barriers, gathers, max-over-ranks timing) on a one-GPU box; the numbers of such a run mean nothing.
    rehearse = os.environ.get("VQ_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("VQ_BENCH_FAIL_INIT_RANK") == str(rank):      # tests: a rank that dies where RCCL would be initialised
            raise SystemExit("rank %d told to fail before the process group exists" % rank)
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    stream = torch.cuda.Stream(device=device)
    dist_info = None
    if world > 1:
        # the run must be what --gpus says it is: N ranks, one per GPU, over RCCL (backend "nccl" IS RCCL on ROCm)
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        assert rehearse or dist.get_backend() == "nccl", dist.get_backend()
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:      # noqa: BLE001
            ver = "unavailable (%s)" % type(e).__name__
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_version": ver,
                     "devices_visible": torch.cuda.device_count(), "rank0_device": torch.cuda.get_device_name(device),
                     "rehearsal_on_one_gpu": rehearse}

    dt, roof, model, crops, feats = bench_tsn(args, rank, world, device, stream)
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        per_rank = torch.zeros(world, dtype=torch.float64, device=device)
        per_rank[rank] = roof["rank_ms_per_step"]
        dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
        roof["rank_ms_per_step"] = {"min": float(per_rank.min().item()), "max": float(per_rank.max().item()),
                                    "all": [float(v) for v in per_rank.cpu().tolist()]}
    dt = float(tmax.item())
    value = world * B_CLIPS * args.steps / dt
    out = {"metric": "clips/sec TSN feature-extract", "value": value, "unit": "clips/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[1]: TSN BN-Inception RGB stream, 224x224x3 uint8 crops resident in HBM, "
                                  "T=3 segments, B=32 clips per GPU (96 crops per step), random-init weights",
                      "global_batch": world * B_CLIPS, "crops_per_step_per_gpu": B_CLIPS * T_SEG,
                      "parallelism": "dp%d" % world, "collective": "all_gather feature blocks (RCCL)" if world > 1 else None,
                      "distributed": dist_info},
           "roofline": roof}
    # algorithmic FLOPs of a step (per GPU) over the whole timed step
    roof["achieved"] = roof["flops_per_step"] / out["ms_per_step"] / 1e9
    roof["frac"] = roof["achieved"] / PEAK_FP32_MFMA_TFLOPS
    out["config"]["timed_mode"] = ("all steps on one stream (--profile-only)" if args.profile_only else
                                   "vq_tsn_forward's default: 2 sub-batches on 2 HIP streams, no launch carries an event; per-kernel fields from the " + roof["kernel_fields_from"])
    up = torch.tensor([roof.pop("single_stream_ms_per_step")], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(up, op=dist.ReduceOp.MAX)
    if not args.profile_only:
        out["single_stream"] = {"value": world * B_CLIPS / float(up.item()) * 1e3, "unit": "clips/s", "ms_per_step": float(up.item()),
                                "frac": roof["flops_per_step"] / float(up.item()) / 1e9 / PEAK_FP32_MFMA_TFLOPS,
                                "kernels": {k: roof[k] for k in ("conv_ms_per_step", "other_kernels_ms_per_step", "kernel_frac", "matrix_pipe_frac")},
                                "profiled_steps": roof["profiled_steps"],
                                "note": "same K steps, same bracketing, ALL on one stream (events on every %dth): the timed mode of rounds "
                                        "1-4, like for like with BENCH_r01..r04's value" % sample_every(args.steps)}
    if rank == 0 and world == 1 and not args.skip_cpu:
        base, ps_cpu = cpu_baseline_tsn(crops.cpu().numpy(), (model.graph, tsn_net.synthetic_weights(model.graph, seed=2)))
        out["cpu_baseline"] = base
        # the baseline run doubles as a live parity check of the timed path (first clips of the batch)
        ncl = ps_cpu.shape[0] // T_SEG
        ref = ps_cpu.astype(np.float64).reshape(ncl, T_SEG, -1).mean(axis=1)
        got = feats[:ncl].cpu().numpy()
        out["parity_vs_oracle_rel_err"] = float(np.abs(got - ref).max() / np.abs(ref).max())
    model.close()
    del crops
    if rank == 0 and world == 1 and not args.skip_two_stream and (not args.profile_only or args.profile_two_stream):
        out["two_stream"] = bench_two_stream(args, device, stream, not args.skip_cpu)

    if not args.skip_sim:
        sdt, ssteps, sroof, db, row0, rows = bench_sim(args, rank, world, device, stream)
        tmax = torch.tensor([sdt], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        sdt = float(tmax.item())
        sim = {"metric": "queries/sec similarity (1M x 1024)", "value": ssteps / sdt, "unit": "queries/s",
               "ms_per_query": sdt / ssteps * 1e3, "steps": ssteps, "scaling": "strong", "dtype": "f64 accumulate over f32 features",
               "config": {"workload": "configs[3]: 1 query x 1M clips x (2 streams x 5 splits) x 1024 fp32 = 40.96 GB, "
                                      "row-sharded over %d GPU(s), weighted score fused, score slices all-gathered" % world,
                          "rows_per_gpu": rows},
               "roofline": sroof}
        batched = sroof.pop("batched", None)
        if batched:
            if world > 1:                                     # per-rank passes run side by side: the job's rate is that of the slowest
                bt = torch.tensor([batched["ms_per_pass"]], dtype=torch.float64, device=device)
                dist.all_reduce(bt, op=dist.ReduceOp.MAX)
                batched["ms_per_pass"] = float(bt.item())
                batched["value"] = batched["queries_per_pass"] / batched["ms_per_pass"] * 1e3
            sim["batched"] = batched
        if rank == 0 and world == 1 and not args.skip_cpu:
            sim["cpu_baseline"] = cpu_baseline_sim(db, row0)
        out["similarity"] = sim
        db.close()
    if rank == 0 and world == 1 and not args.skip_sim and not args.profile_only:
        out["rounds"] = bench_rounds(local_rank, not args.skip_cpu)
    if rank == 0 and world == 1 and not args.skip_flow and not args.profile_only:
        out["flow"] = bench_flow(local_rank, not args.skip_cpu)
        jp = bench_jpeg(local_rank)
        if jp:
            out["jpeg"] = jp
        e2e = bench_e2e_cli(local_rank)
        if e2e:
            out["e2e_cli"] = e2e
        wof = bench_e2e_wof(local_rank)
        if wof:
            out["e2e_wof_cli"] = wof
    if rank == 0:
        if args.details:
            os.makedirs(os.path.dirname(os.path.abspath(args.details)) or ".", exist_ok=True)
            with open(args.details, "w") as f:
                json.dump(out, f)
        print(json.dumps(out if args.verbose else compact(out), separators=(",", ":")), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
