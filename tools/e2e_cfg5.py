"""BASELINE configs[4] end to end on the hot path (SURVEY.md 8(d) cfg 5):

  N synthetic clips (T = 7 snippets, RGB + 10-channel flow stacks, crops resident in HBM)
    -> TSN features on the GPU for E = 3 weight seeds ("splits"), written straight into the resident feature DB
       (A -> B hand-off without a host round trip) and, for the first --csv-clips clips, into the reference's
       data/features CSV tree (then read back with the load_db parser rules and compared)
    -> R rounds of {compute_similarities, optimize_weights on 20 seeded labels, compute_scores, select_clips_to_review}
       through the drop-in Ticket / Hyperparameter API.

    python tools/e2e_cfg5.py [--clips 10000] [--rounds 100] [--csv-clips 200] [--out DIR]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/e2e_cfg5.py ...
        N > 1: the clips are sharded for the extraction and the feature blocks STAY where they were produced -- every rank adopts its own
        block as its rows of a ShardedFeatureDB ("produced in place by A"), no feature ever crosses xGMI; the rounds run on
        all ranks (query broadcast, local scans, score slices gathered: sharded_db.py), rank 0 prints.  --gather-features keeps the
        round-3 form (blocks all-gathered over RCCL, rounds on rank 0's GPU).  VQ_DIST_BACKEND=gloo + VQ_CFG5_ONE_CARD=1: every rank on
        cuda:0 (a rehearsal of the N > 1 control flow on a one-GPU box)

Prints one JSON object with the wall time of every stage.
"""
import argparse
import json
import os
import random
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

os.environ.setdefault("COMPUTE_EPS", "0.000003")
import video_query_algorithms_amd as vqa
from video_query_algorithms_amd.shard import all_gather_rows, shard_range
from video_query_algorithms_amd.tsn import bn_inception, feature_csv, net as tsn_net

STREAMS = ("rgb", "warped_optical_flow")
SPLITS = (1, 2, 3)
SEED = "73459912436"


class _DevArray:
    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def extract(args, device, rank, world, timings):
    """[n_mine, S, E, D] fp32 on the device: this rank's clips through 2 streams x 3 weight seeds."""
    first, count = shard_range(args.clips, world, rank)
    T, B = args.segments, args.batch_clips
    out = torch.zeros((count, len(STREAMS), len(SPLITS), 1024), dtype=torch.float32, device=device)
    for si, (mode, ch, mean) in enumerate((("rgb", 3, tsn_net.RGB_MEAN), ("flow", 10, tsn_net.FLOW_MEAN))):
        g = bn_inception.bn_inception(ch)
        tuned = {}                                       # tiling tables of the first weight seed, reused by the others
        for ei, split in enumerate(SPLITS):
            t0 = time.perf_counter()
            model = tsn_net.TsnNet(g, tsn_net.synthetic_weights(g, seed=100 + split), max_crops=B * T, device=device.index)
            for n_crops, tiles in tuned.items():
                model.set_layer_tiles(n_crops, tiles)
            fptr, _ = model.feat_devptr()
            for b0 in range(0, count, B):
                nb = min(B, count - b0)
                gen = torch.Generator(device=device).manual_seed(1000003 * (si + 1) + first + b0)   # per clip block, any sharding
                crops = torch.randint(0, 256, (nb * T, 224, 224, ch), dtype=torch.uint8, device=device, generator=gen)
                model.forward_device(crops.data_ptr(), nb * T, T, mean)
                feat = torch.as_tensor(_DevArray(fptr, (nb, 1024), "<f8"), device=device)
                out[b0:b0 + nb, si, ei] = feat.to(torch.float32)
            torch.cuda.synchronize(device)
            if not tuned:
                for n_crops in {B * T, B * T // 2, (count % B) * T, (count % B) * T // 2} - {0}:
                    tuned[n_crops] = model.layer_tiles(n_crops)
            model.close()
            timings["extract_%s_split%d_s" % (mode, split)] = time.perf_counter() - t0
    return first, count, out


def main(argv=None, observer=None):
    """``observer`` (tests): an object whose ``start(feats, clip_ids, ticket, hp)`` is called once before the rounds and
    ``round(r, ticket, hp, prev_weights, prev_threshold, labels, rng_state_before_select)`` after each one."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=10000)
    ap.add_argument("--segments", type=int, default=7)
    ap.add_argument("--batch-clips", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=100)
    ap.add_argument("--labels", type=int, default=20)
    ap.add_argument("--csv-clips", type=int, default=200)
    ap.add_argument("--out", default=None)
    ap.add_argument("--gather-features", action="store_true", help="N > 1: all-gather the feature blocks and run the rounds on rank 0 only")
    args = ap.parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("VQ_CFG5_ONE_CARD") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VQ_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    sharded = world > 1 and not args.gather_features
    timings = {"clips": args.clips, "segments": args.segments, "world": world}
    t_all = time.perf_counter()
    first, count, mine = extract(args, device, rank, world, timings)
    t0 = time.perf_counter()
    if sharded:
        feats = mine                                      # this rank's rows; nothing is gathered
    else:
        feats = all_gather_rows(mine.reshape(count, -1), args.clips).reshape(args.clips, len(STREAMS), len(SPLITS), 1024) if world > 1 else mine
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    timings["all_gather_s"] = time.perf_counter() - t0
    timings["extract_total_s"] = time.perf_counter() - t_all
    timings["extract_clips_per_s"] = args.clips * len(SPLITS) / timings["extract_total_s"]   # (clip, split) pairs, both streams
    timings["features_gathered"] = bool(world > 1 and not sharded)
    if rank != 0 and not sharded:
        dist.destroy_process_group()
        return 0
    clip_ids = np.arange(1, args.clips + 1, dtype=np.int64)
    if sharded:
        args.csv_clips = min(args.csv_clips, count if rank == 0 else 0)      # the CSV sample comes from rank 0's own rows

    # ---- data/features layout for the first clips, and back through the load_db parser rules -----------------
    n_csv = min(args.csv_clips, args.clips)
    if n_csv:
        t0 = time.perf_counter()
        out_dir = args.out or tempfile.mkdtemp(prefix="vq_features_")
        host = feats[:n_csv].to(torch.float64).cpu().numpy()
        names = ["clip_%04d" % c for c in clip_ids[:n_csv]]
        for ei, split in enumerate(SPLITS):
            feature_csv.write_features(out_dir, "synthetic_video", "/synthetic/", "UCF101_split%d" % split, "global_pool", names,
                                       {"rgb": host[:, 0, ei], "warped_optical_flow": host[:, 1, ei]},
                                       {"rgb": "synthetic:%d" % (100 + split), "warped_optical_flow": "synthetic:%d" % (100 + split)})
        timings["csv_write_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        back = np.zeros_like(host)
        for ei, split in enumerate(SPLITS):
            nsplit, per_stream = feature_csv.read_split_dir(os.path.join(out_dir, "synthetic_video", "UCF101_split%d" % split))
            assert nsplit == split
            for si, st in enumerate(STREAMS):
                clips, vals, _meta = per_stream[st]
                assert (clips == clip_ids[:n_csv]).all()
                back[:, si, ei] = vals
        timings["csv_read_s"] = time.perf_counter() - t0
        assert (back == host).all(), "CSV round trip changed a value"       # repr(float64) is shortest round-trip
        timings["csv_clips"] = n_csv

    # ---- resident DB (the device block A produced, adopted without a copy) + query rounds ---------------------
    t0 = time.perf_counter()
    feats = feats.contiguous()
    if sharded:
        from video_query_algorithms_amd.sharded_db import ShardedFeatureDB
        mine_db = vqa.FeatureDB(count, len(STREAMS), len(SPLITS), 1024, np.float32, device.index, clip_ids[first:first + count])
        mine_db.adopt_device(feats.data_ptr(), keepalive=feats)
        mine_db.stream_names = list(STREAMS)
        mine_db.slot_splits = [list(SPLITS)] * len(STREAMS)
        db = ShardedFeatureDB(mine_db, args.clips, first, clip_ids)           # SPMD: every rank runs the rounds below
    else:
        db = vqa.FeatureDB(args.clips, len(STREAMS), len(SPLITS), 1024, np.float32, device.index, clip_ids)
        db.adopt_device(feats.data_ptr(), keepalive=feats)
        db.stream_names = list(STREAMS)
        db.slot_splits = [list(SPLITS)] * len(STREAMS)
    timings["db_adopt_s"] = time.perf_counter() - t0
    ref_row = 7
    ref = db.read_rows([ref_row])[0].astype(np.float64)
    ref_records = [{"dnn_stream_id": st, "dnn_stream_split": sp, "name": "global_pool", "video_clip_id": int(clip_ids[ref_row]),
                    "feature_vector": ref[si, ei].tolist()} for si, st in enumerate(STREAMS) for ei, sp in enumerate(SPLITS)]
    hp = vqa.Hyperparameter({"rgb": 1.0, "warped_optical_flow": 1.5}, 0.8, 0.0, 0.35, 0.0, STREAMS, "global_pool", 1, 0.7, "bagging", 3)
    tk = vqa.Ticket({"query_id": 1, "video_id": 1, "ref_clip": 0, "ref_clip_id": int(clip_ids[ref_row]), "search_set": 1,
                     "number_of_matches_to_review": 20, "dynamic_target_adjustment": False, "user_matches": {}},
                    records=ref_records, feature_db=db, device=device.index)
    tk.target = vqa.TargetClip(tk, hp)
    tk.target.get_target_features()
    stage = {"similarities": 0.0, "optimize_weights": 0.0, "scores": 0.0, "select": 0.0}
    rng = np.random.default_rng(5)
    random.seed(a=SEED)
    if observer is not None:
        observer.start(feats, clip_ids, tk, hp)
    t_rounds = time.perf_counter()
    for r in range(args.rounds):
        t0 = time.perf_counter()
        tk.compute_similarities(hp)
        stage["similarities"] += time.perf_counter() - t0
        if r == 0:
            hp.weights, hp.threshold = dict(hp.default_weights), hp.default_threshold
        prev_weights, prev_threshold = dict(hp.weights), hp.threshold
        t0 = time.perf_counter()
        tk.compute_scores(hp.weights)                 # scores under the previous round's weights: what the user reviewed
        stage["scores"] += time.perf_counter() - t0
        # 20 seeded labels on the current top scores (the user's review of the previous round)
        top_rows, top_vals = db.topk(args.labels)
        tk_labels = tk.matches = [{"video_clip": int(clip_ids[row]), "user_match": bool(rng.random() < 0.5 + 0.4 * (i < args.labels // 2)),
                       "is_match": bool(v >= hp.threshold)} for i, (row, v) in enumerate(zip(top_rows, top_vals))]
        t0 = time.perf_counter()
        hp.optimize_weights(tk)
        stage["optimize_weights"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        tk.compute_scores(hp.weights)
        stage["scores"] += time.perf_counter() - t0
        rng_state = random.getstate() if observer is not None else None
        t0 = time.perf_counter()
        tk.select_clips_to_review(hp.threshold, 20, hp.near_miss_default)
        stage["select"] += time.perf_counter() - t0
        if observer is not None:
            observer.round(r, tk, hp, prev_weights, prev_threshold, tk_labels, rng_state)
    timings["rounds"] = args.rounds
    timings["rounds_total_s"] = time.perf_counter() - t_rounds
    for k, v in stage.items():
        timings["round_%s_ms" % k] = v / max(args.rounds, 1) * 1e3
    timings["final_weights"] = {k: float(v) for k, v in hp.weights.items()}
    timings["final_threshold"] = float(hp.threshold)
    timings["final_matches"] = len(tk.matches)
    timings["total_s"] = time.perf_counter() - t_all
    if rank == 0:
        print(json.dumps(timings))
    db.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
