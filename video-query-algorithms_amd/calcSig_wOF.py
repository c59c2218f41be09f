"""Drop-in for the reference's ``src/features_GPU_compute/calcSig_wOF.py`` command line.

Same positional arguments, flags and defaults (calcSig_wOF.py:156-176), same clip order (:200), same snippet
sampling (:67-72), same consensus (:82), same output tree and CSV bytes (:116-134).  Differences, all internal:
one network per stream is built once (not per video, :205-210), all B*T crops of a batch of clips go through the
network in one pass instead of one 10-crop forward per snippet (only crop 0 was ever kept, :95), and with
``torchrun`` the clips of a video are sharded over the ranks (one process per GPU) and the per-GPU feature blocks
are all-gathered over RCCL (from device memory); rank 0 writes the files.  ``--num_worker`` sizes the pool of decoder
threads (the reference's worker processes, calcSig_wOF.py:204-210).

Started by plain ``python`` the way calcSig_wOF_ensemble.sh:13-19 spells it (``--num_worker 24 --gpus 0 1 2 3 4 5 6 7``, no
launcher), the command line maps workers to GPUs as the reference does (worker i -> ``gpu_list[(i - 1) % len]``, :47-55) and,
when that names more than one GPU, starts itself once per GPU as fresh child processes BEFORE making any GPU call
(``fanout.py``): the children are the ranks of the sharded path above, the parent only waits and returns the first
non-zero exit code.  The CSV tree is byte-identical to the one-GPU run.

    python calcSig_wOF.py frames/ rgb.prototxt rgb_weights.npz flow.prototxt flow_weights.npz \
        --outFeatures_dir features/ --modelname UCF101_split1 [--num_frame_per_video 25] [--gpus 0]
"""
from __future__ import annotations

import argparse
import glob
import os
import sys
import time

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.tsn import frames
    from video_query_algorithms_amd.tsn.caffe_net import CaffeNet
    from video_query_algorithms_amd.tsn.ingest import FrameIngest
    from video_query_algorithms_amd.tsn.feature_csv import format_rows, write_feature_file, write_features
    from video_query_algorithms_amd.shard import all_gather_rows, shard_range
    from video_query_algorithms_amd import fanout
else:
    from . import fanout
    from .tsn import frames
    from .tsn.caffe_net import CaffeNet
    from .tsn.ingest import FrameIngest
    from .tsn.feature_csv import format_rows, write_feature_file, write_features
    from .shard import all_gather_rows, shard_range


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('frame_path', type=str, help="root directory holding the frames")
    parser.add_argument('net_proto_rgb', type=str)
    parser.add_argument('net_weights_rgb', type=str)
    parser.add_argument('net_proto_flow', type=str)
    parser.add_argument('net_weights_flow', type=str)
    parser.add_argument('--rgb_prefix', type=str, help="prefix of RGB frames", default='img_')
    parser.add_argument('--flow_x_prefix', type=str, help="prefix of x direction flow images", default='flow_x_')
    parser.add_argument('--flow_y_prefix', type=str, help="prefix of y direction flow images", default='flow_y_')
    parser.add_argument('--num_frame_per_video', type=int, default=25, help="number of frames to evaluate in each video")
    parser.add_argument('--num_worker', type=int, default=1, help="decoder threads (the reference's worker processes; the GPU side is batched instead)")
    parser.add_argument("--gpus", type=int, nargs='+', default=None, help='GPU ids; one process per GPU under torchrun')
    parser.add_argument('--outFeatures_dir', type=str, default=None, help='Specify directory to write out feature files')
    parser.add_argument('--delimiter', type=str, default=',', help='delimiter used in feature files')
    parser.add_argument('--modelname', type=str, default=None)
    parser.add_argument('--featureBlob', type=str, default='global_pool', help='name of blob to extract as a feature')
    parser.add_argument('--featureBlob_size', type=int, default=1024, help='expected size of the feature blob')
    # additions (not in the reference)
    parser.add_argument('--frame_ext', type=str, default='.jpg', help='frame file extension (.jpg needs cv2 or PIL)')
    parser.add_argument('--batch_clips', type=int, default=32,
                        help='clips per forward pass (32 x T = 800 crops per stream at the default T; 16 runs the networks as efficiently with '
                             'half the activation memory and a shorter start, but no faster end to end: profiles/README.md)')
    parser.add_argument('--host_resize', action='store_true',
                        help='resize + crop the frames on the host (numpy) instead of on the GPU; same bytes either way')
    parser.add_argument('--exact_resize', action='store_true',
                        help="resize with exact fp64 bilinear weights instead of cv2.resize's 11-bit fixed-point rule (the default, "
                             "what the reference's dependency computes); differs by one grey level on about one pixel in eight")
    parser.add_argument('--device_jpeg', action='store_true',
                        help='decode the .jpg frames with the library (Huffman on host threads, IDCT / upsampling / colour on the '
                             'GPU; the pixels libjpeg gives cv2.imread) and resize them where they land -- no host image library')
    parser.add_argument('--ensemble', nargs=3, action='append', metavar=('MODELNAME', 'WEIGHTS_RGB', 'WEIGHTS_FLOW'),
                        help="a further ensemble member (repeatable): the SAME pass over the frame tree -- every frame read, decoded and resized "
                             "once -- also runs these weights and writes their <modelname> directories; the three runs of "
                             "calcSig_wOF_ensemble.sh:13-37 become one command with two --ensemble options, byte-identical files")
    parser.add_argument('--number_format', choices=('repr', 'g12'), default='repr',
                        help="how str(numpy.float64) printed under the numpy the reference ran with: shortest round-trip "
                             "(numpy >= 1.14, lossless) or 12 significant digits (numpy < 1.14); the reference ships files of both kinds")
    return parser


def _join_world(device: int):
    """(rank, world) of this process.  Under torchrun the rank binds its GPU BEFORE the communicator exists
    (``set_device`` + ``device_id``: RCCL then builds the communicator on that device instead of guessing at the first
    collective); backend "nccl" = RCCL, or gloo on a box without GPUs / with VQ_DIST_BACKEND=gloo (a rehearsal of the
    N > 1 path on one card)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" not in os.environ or world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        backend = os.environ.get("VQ_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(device)
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    return dist.get_rank(), dist.get_world_size()


def _stack_rows(blocks, width):
    """Per-batch feature blocks (numpy, or torch tensors still on the GPU) -> one [rows, width] block of the same kind."""
    if not blocks:
        return np.zeros((0, width))
    if isinstance(blocks[0], np.ndarray):
        return np.concatenate(blocks, axis=0)
    import torch
    return torch.cat(blocks, dim=0)


def _fan_out(args, argv, program):
    """The reference's worker -> GPU map (calcSig_wOF.py:47-55) as child processes, one per distinct GPU.  Returns the exit
    code of the fan-out, or None when this process is to do the work itself (one GPU, a rank of a launcher, a child)."""
    if "RANK" in os.environ or fanout.is_child() or program is None:
        return None
    devices = fanout.worker_devices(args.gpus, args.num_worker, None if args.gpus else fanout.visible_gpus())
    if len(devices) < 2:
        return None
    per_rank = -(-max(1, args.num_worker) // len(devices))            # the decoder threads are shared out over the ranks
    envs = fanout.rank_envs(len(devices), per_rank)
    for e, g in zip(envs, devices):
        e["VQ_FANOUT_WORKERS"] = str(per_rank)
        e["VQ_FANOUT_DEVICE"] = str(g)
    print('{} workers on GPUs {}: one process per GPU'.format(args.num_worker, devices), flush=True)
    return fanout.run_children(program, argv, envs)


class _CropPipeline:
    """--device_jpeg: the device crops of the batches AHEAD of the network, across the streams.  The (stream, batch) items form one
    work list (``order``); up to ``ahead`` of them beyond the one the network asks for are in preparation on
    two threads -- lists of undecoded files in, device crops out (tsn/ingest.py; the library's calls release the interpreter; the two
    lanes own a decoder and a HIP stream each, so one batch's host half -- reading the files, stripping the byte stuffing --
    overlaps the other's device half).  Nothing here needs an extractor: the first batches are prepared while the extractors are
    still being built, and the flow stream's first batches while the RGB stream's last ones are in the network."""

    def __init__(self, ingests, batches, order, load_clip, file_pool, ahead=3):
        from concurrent.futures import ThreadPoolExecutor
        self.ingests, self.batches, self.order, self.load_clip, self.file_pool, self.ahead = ingests, batches, order, load_clip, file_pool, ahead
        self.pool = ThreadPoolExecutor(max_workers=2)
        self.jobs, self.submitted = {}, 0

    def _prepare(self, k):
        si, bi = self.order[k]
        lists = [f.result() for f in [self.file_pool.submit(self.load_clip, si, u) for u in self.batches[bi]]]
        return self.ingests[si].crops_from_jpegs([f for c in lists for f in c], lane=k % 2)

    def start(self):
        self._fill(0)

    def _fill(self, k):
        while self.submitted < min(k + self.ahead + 1, len(self.order)):
            self.jobs[self.submitted] = self.pool.submit(self._prepare, self.submitted)
            self.submitted += 1

    def get(self, k):
        self._fill(k)
        return self.jobs.pop(k).result()

    def close(self):
        self.pool.shutdown()
        for ing in self.ingests:
            ing.close()


def _main(argv=None, net_factory=None, program=None):
    """``net_factory(net_proto, net_weights, device, max_crops=, feature_blob=, resize_rule=)`` builds the per-stream extractor
    (default: the HIP ``CaffeNet``; the CPU tests of the sharding logic pass a stand-in).  ``program``: the script the
    per-GPU children are started from (default: this file; a caller that passes its own ``net_factory`` names its own
    script here, or gets no fan-out)."""
    t_main = time.perf_counter()
    trace = os.environ.get("VQ_CLI_TRACE") == "1"

    def stamp(what):
        if trace:
            print("trace: %.3f s  %s" % (time.perf_counter() - t_main, what), file=sys.stderr, flush=True)
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if program is None and net_factory is None:
        program = os.path.abspath(__file__)
    rc = _fan_out(args, argv, program)                                  # before ANY GPU call of this process
    if rc is not None:
        return rc
    if fanout.is_child() and "VQ_FANOUT_WORKERS" in os.environ:
        args.num_worker = int(os.environ["VQ_FANOUT_WORKERS"])
    net_factory = net_factory or CaffeNet
    if args.modelname is None:                                                         # calcSig_wOF.py:179-180
        args.modelname = args.net_weights_rgb.split('/')[-1][:-11] + '_' + args.net_weights_flow.split('/')[-1][:-11]
    T = args.num_frame_per_video
    gpu_list = args.gpus
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = gpu_list[local % len(gpu_list)] if gpu_list else local                    # calcSig_wOF.py:50-55
    if fanout.is_child() and "VQ_FANOUT_DEVICE" in os.environ:                         # the GPU the parent mapped this rank to
        device = int(os.environ["VQ_FANOUT_DEVICE"])
    rank, world = _join_world(device)
    if world == 1 and "torch" not in sys.modules:
        # one rank: nothing here needs torch (device buffers and streams come from the library, tsn/devmem.py) -- 0.8 s of import for a
        # process that takes 2.3 s for 256 clips.  Decided before the library is loaded: it must share ONE HIP runtime with torch if torch
        # is in the process (VQ_NO_TORCH=0 keeps torch).
        os.environ.setdefault("VQ_NO_TORCH", "1")
    backend_device = None                                # where a rank's blocks must live for the collective (RCCL: its GPU)
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            import torch
            backend_device = torch.device("cuda", device)
    frame_path = args.frame_path if args.frame_path[-1] == '/' else args.frame_path + '/'
    streamCNN = [{'modality': 'rgb', 'mode': 'rgb', 'net_proto': args.net_proto_rgb, 'cnt_indexer': 1, 'stack_depth': 1},
                 {'modality': 'flow', 'mode': 'warped_optical_flow', 'net_proto': args.net_proto_flow, 'cnt_indexer': 2,
                  'stack_depth': 5}]                                                   # calcSig_wOF.py:185-189
    # the ensemble: the command line's own weights first, then every --ensemble member; ONE pass over the frame tree feeds them all
    members = [{'modelname': args.modelname, 'rgb': args.net_weights_rgb, 'flow': args.net_weights_flow}]
    for name, w_rgb, w_flow in args.ensemble or []:
        members.append({'modelname': name, 'rgb': w_rgb, 'flow': w_flow})
    if len({m['modelname'] for m in members}) != len(members):
        raise SystemExit("--ensemble: every member needs its own --modelname (they name the output directories)")
    rule = "exact" if args.exact_resize else "cv2"
    device_jpeg = args.device_jpeg and args.frame_ext.lower() in ('.jpg', '.jpeg') and not args.host_resize
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, args.num_worker))
    io_pool = ThreadPoolExecutor(max_workers=2)          # feature files are formatted and written here
    csv_jobs = []
    build_pool = ThreadPoolExecutor(max_workers=2)

    # The clips of the videos form one list (video order, then clip number: calcSig_wOF.py:193-200) and THAT is what the ranks share
    # out: contiguous ranges of it, whole batches of --batch_clips whatever the videos' sizes (the reference's fixtures are 87- and
    # 114-clip videos: per-video sharding left 11-14 clips per rank on 8 GPUs and one collective per video and stream).  The list is
    # cut into GROUPS of whole videos of at least 16 batches per rank: a group is shared out, extracted, all-gathered (one collective
    # per stream and ensemble member) and its files are queued before the next group starts -- the reference writes a video's files
    # inside its per-video loop (calcSig_wOF.py:195-222), so a bad frame or a kill near the end of a long tree leaves every finished
    # video behind, and what a rank keeps in memory is bounded by a group, not by the tree.  The files are one per video, stream and
    # member, with the bytes of the one-GPU run.
    videos = []
    for video_path in sorted(glob.glob(frame_path + '*/')):                            # calcSig_wOF.py:193-195
        f_info = frames.parse_directory(video_path, args.rgb_prefix, args.flow_x_prefix, args.flow_y_prefix)
        clip_list = sorted(list(f_info[0]), key=lambda clip: int(clip[-4:]))           # calcSig_wOF.py:199-200
        videos.append((video_path, f_info, clip_list))
    group_clips = int(os.environ.get("VQ_CLI_GROUP_CLIPS", "0")) or 16 * args.batch_clips * world
    groups, acc, n_acc = [], [], 0                       # groups of video indices
    for vi, (_p, _f, clip_list) in enumerate(videos):
        acc.append(vi)
        n_acc += len(clip_list)
        if n_acc >= group_clips:
            groups.append(acc)
            acc, n_acc = [], 0
    if acc:
        groups.append(acc)
    group_units = [[(vi, vid) for vi in g for vid in videos[vi][2]] for g in groups]
    shards = [shard_range(len(u), world, rank) for u in group_units]
    count = sum(c for _f, c in shards)                   # clips of this rank in the whole tree
    stamp("frame tree parsed: %d videos, %d clips, %d group(s)" % (len(videos), sum(len(u) for u in group_units), len(groups)))
    on_gpu = world > 1                                   # blocks that will be all-gathered never visit the host
    host_rule = {'rule': rule} if args.host_resize else {}             # the host loaders resize; the others hand frames to the GPU

    def build(s, m):
        return net_factory(s['net_proto'], m[s['modality']], device, max_crops=args.batch_clips * T, feature_blob=args.featureBlob, resize_rule=rule)
    # both streams' extractors are built side by side: the flow nets' weights are folded and uploaded while the RGB stream is running
    net_jobs = {(s['modality'], mi): build_pool.submit(build, s, m) for s in streamCNN for mi, m in enumerate(members)} if count else {}

    def load_clip(si, unit):
        s = streamCNN[si]
        f_info, vid = videos[unit[0]][1], unit[1]
        frame_cnt = f_info[s['cnt_indexer']][vid]
        ticks = frames.frame_ticks(frame_cnt, T, s['stack_depth'])
        if s['modality'] == 'rgb':
            load = frames.load_rgb_jpegs if device_jpeg else frames.load_rgb_snippets if args.host_resize else frames.load_rgb_frames
            return load(f_info[0][vid], ticks, args.rgb_prefix, args.frame_ext, **host_rule)
        load = frames.load_flow_jpegs if device_jpeg else frames.load_flow_snippets if args.host_resize else frames.load_flow_frames
        return load(f_info[0][vid], ticks, frame_cnt, s['stack_depth'], args.flow_x_prefix, args.flow_y_prefix, args.frame_ext, **host_rule)

    batches, group_batches = [], []                      # this rank's batches of all groups; per group: their indices
    for units, (first, cnt) in zip(group_units, shards):
        group_batches.append(list(range(len(batches), len(batches) + -(-cnt // args.batch_clips))))
        batches += [units[b0:min(b0 + args.batch_clips, first + cnt)] for b0 in range(first, first + cnt, args.batch_clips)]
    n_streams = len(streamCNN)
    # The order of the work.  Host decoding: stream by stream, as the reference does (all RGB batches, then all flow batches).
    # --device_jpeg: batch by batch, the RGB and the flow half of a batch behind each other -- the preparation of a flow batch (8 000
    # files at the reference's defaults) then has a whole batch of BOTH networks to hide behind instead of one flow forward, and a long
    # job is bound by the networks from its first batch to its last.  Features do not depend on the order.
    crop_pipe = None
    if device_jpeg and count:
        # (letting the flow stream trail the RGB stream by a batch or two, so that its extractor and first files are surely ready at its
        # first turn, was measured: 0.70-0.77 s against 0.63-0.65 for 256 clips -- the uneven ends cost more than the stall)
        group_order = [[(si, bi) for bi in gb for si in range(n_streams)] for gb in group_batches]
        order = [x for go in group_order for x in go]
        crop_pipe = _CropPipeline([FrameIngest(3 if s['modality'] == 'rgb' else 2 * s['stack_depth'], device, rule) for s in streamCNN],
                                  batches, order, load_clip, pool)
        # (building the extractors BEFORE the readers start -- 12 ms alone against 45+ ms beside sixteen reader threads -- was measured in
        # round 5: no difference end to end, 537-545 against 538-561 clips/s on one box: the job is bound by the networks)
        crop_pipe.start()                            # the first batches are read and decoded while the extractors are being built
    else:
        group_order = [[(si, bi) for si in range(n_streams) for bi in gb] for gb in group_batches]
        order = [x for go in group_order for x in go]
    nets = [None] * n_streams
    mine = [[[] for _ in members] for _ in streamCNN]    # the feature blocks of the group in work
    waited = [{'files': 0.0, 'crops': 0.0, 'nets': 0.0} for _ in streamCNN]     # VQ_CLI_TRACE=1: where the loop waited, per stream
    pending = {}                                     # host decoding: (si, bi) -> futures of the batch's clips, one batch ahead

    def submit_files(k):
        if k < len(order) and not crop_pipe:
            si, bi = order[k]
            pending[(si, bi)] = [pool.submit(load_clip, si, u) for u in batches[bi]]

    # One rank: a batch's rows are on the host as soon as its forward returns, so they are FORMATTED (half a million float reprs per 256
    # clips and stream) batch by batch on the writer threads while the next batch is on the GPU; what is left for the end of a group
    # is a header and a write per file.  (Several ranks: the rows of a video come together in the all-gather; formatted then.)
    early_rows = world == 1
    row_jobs = [[{} for _ in members] for _ in streamCNN]        # [stream][member]{video index: [futures of formatted row blocks, in clip order]}
    cursor = {}

    def through_the_nets(si, make, batch_key):
        """One batch (or a part of it: clips of another frame size, a chunk of the ensemble path) through every member's network of
        stream si: ``make(net)`` hands a net the crops (decoded, resized and cropped once, whatever the number of members)."""
        t0 = time.perf_counter()
        for mi, net in enumerate(nets[si]):
            block = make(net)
            mine[si][mi].append(block)
            if early_rows and isinstance(block, np.ndarray):
                units_here = batches[batch_key[1]][cursor.get(batch_key, 0):cursor.get(batch_key, 0) + block.shape[0]]
                lo = 0
                while lo < len(units_here):                        # a batch may straddle videos: one row block per (video, clip range)
                    hi = lo
                    while hi < len(units_here) and units_here[hi][0] == units_here[lo][0]:
                        hi += 1
                    nos = np.asarray([int(vid[-4:]) for _vi, vid in units_here[lo:hi]], dtype=np.int64)     # calcSig_wOF.py:131
                    row_jobs[si][mi].setdefault(units_here[lo][0], []).append(io_pool.submit(format_rows, block[lo:hi], nos, args.number_format))
                    lo = hi
                if mi == len(nets[si]) - 1:
                    cursor[batch_key] = cursor.get(batch_key, 0) + block.shape[0]
        waited[si]['nets'] += time.perf_counter() - t0

    def flush(gi):
        """Group gi is through the networks: gather its blocks (one collective per stream and member), queue its videos' files."""
        n_units = len(group_units[gi])
        for si, s in enumerate(streamCNN):
            for mi, m in enumerate(members):
                width = nets[si][mi].feature_dim if nets[si] else args.featureBlob_size
                local_feat = _stack_rows(mine[si][mi], width)
                mine[si][mi] = []
                if world > 1:
                    import torch
                    if isinstance(local_feat, np.ndarray):       # host blocks, or a rank that owns no clip
                        local_feat = torch.from_numpy(np.ascontiguousarray(local_feat, dtype=np.float64))
                    if backend_device is not None and local_feat.device != backend_device:
                        local_feat = local_feat.to(backend_device)
                    local_feat = all_gather_rows(local_feat, n_units).cpu().numpy()
                numFeatures = local_feat.shape[1] if n_units else args.featureBlob_size
                assert numFeatures == args.featureBlob_size                                  # calcSig_wOF.py:219-220
                if rank != 0:
                    continue
                if early_rows and (row_jobs[si][mi] or not n_units):
                    for vi in groups[gi]:
                        video_path, _f, clip_list = videos[vi]
                        blocks = row_jobs[si][mi].pop(vi, [])
                        if clip_list:
                            csv_jobs.append(io_pool.submit(lambda a_: write_feature_file(*a_[:-1], [f.result() for f in a_[-1]]),
                                                           (args.outFeatures_dir, video_path.split('/')[-2], video_path, m['modelname'], args.featureBlob,
                                                            s['mode'], {'rgb': m['rgb'], 'warped_optical_flow': m['flow']}[s['mode']], blocks)))
                    continue
                row = 0
                for video_path, _f, clip_list in (videos[vi] for vi in groups[gi]):
                    block, row = local_feat[row:row + len(clip_list)], row + len(clip_list)
                    if not clip_list:
                        continue
                    # a stream's file is formatted and written (half a million float reprs per 256 clips) by a thread of its own while the
                    # next stream is on the GPU; the same files as writing both at the end (:116-134)
                    csv_jobs.append(io_pool.submit(write_features, args.outFeatures_dir, video_path.split('/')[-2], video_path, m['modelname'],
                                                   args.featureBlob, clip_list, {s['mode']: block},
                                                   {'rgb': m['rgb'], 'warped_optical_flow': m['flow']}, args.number_format))

    t_loop = time.perf_counter()
    submit_files(0)
    k = -1
    for gi, go in enumerate(group_order):
        for si, bi in go:
            k += 1
            s, batch = streamCNN[si], batches[bi]
            if nets[si] is None:
                nets[si] = [net_jobs[(s['modality'], mi)].result() for mi in range(len(members))]
                stamp("%s extractor(s) ready" % s['modality'])
            crops = []
            if not crop_pipe:
                # --num_worker decoder threads (the reference runs that many worker PROCESSES, each with its own net,
                # calcSig_wOF.py:204-210); the batch after the one on the GPU is decoded meanwhile
                t0 = time.perf_counter()
                crops = [f.result() for f in pending.pop((si, bi))]
                waited[si]['files'] += time.perf_counter() - t0
                submit_files(k + 1)
            for _vi, vid in batch:
                print('video {} for {} modality done'.format(vid, s['modality']))
            if crop_pipe:
                t0 = time.perf_counter()
                dev_crops = crop_pipe.get(k)
                waited[si]['crops'] += time.perf_counter() - t0
                through_the_nets(si, lambda net: net.extract_clips_from_crops(dev_crops, T, on_device=on_gpu), (si, bi))
            elif not crops:
                continue
            elif args.host_resize:
                block = np.concatenate(crops, axis=0)
                through_the_nets(si, lambda net: net.extract_clips(block, T, on_device=on_gpu), (si, bi))
            else:
                # resize + crop on the GPU, once per batch; clips of different frame sizes in one batch go one by one
                same_size = [np.concatenate(crops, axis=0)] if len({c.shape[1:] for c in crops}) == 1 else crops
                for g in same_size:
                    if len(nets[si]) == 1:
                        through_the_nets(si, lambda net: net.extract_clips_from_frames(g, T, on_device=on_gpu), (si, bi))
                    else:
                        per = args.batch_clips * T                      # = max_crops of the extractors
                        for i in range(0, g.shape[0], per):
                            dev_crops = nets[si][0].crops_from_frames(g[i:i + per])
                            nets[si][0].sync_ingest()
                            through_the_nets(si, lambda net: net.extract_clips_from_crops(dev_crops, T, on_device=on_gpu), (si, bi))
        flush(gi)
    stamp("last batch through the networks")
    if trace:
        for si, s in enumerate(streamCNN):
            print("trace: %s stream, %d batches (loop of both streams: %.3f s): waited %.3f s for file lists, %.3f s for device crops, %.3f s in the networks"
                  % (s['modality'], len(batches), time.perf_counter() - t_loop, waited[si]['files'], waited[si]['crops'], waited[si]['nets']),
                  file=sys.stderr, flush=True)

    stamp("features gathered, feature files queued")
    for per_stream in nets:
        for n in per_stream or []:
            n.close()
    stamp("networks closed")
    if crop_pipe:
        crop_pipe.close()
    stamp("ingest closed")
    pool.shutdown()
    build_pool.shutdown()
    stamp("extractors closed")
    for job in csv_jobs:
        job.result()                                                     # a writer's exception is the command's
    io_pool.shutdown()
    stamp("feature files written (torch imported: %s)" % ("torch" in sys.modules))
    return 0



def main(argv=None, net_factory=None, program=None):
    """The command line's entry point.  A one-rank run asks for the torch-less mode by setting VQ_NO_TORCH in the environment BEFORE the
    library is loaded (it is latched there: _lib.TORCHLESS); the variable is put back when the call returns, so that a host that calls
    main() in-process does not leak it to its later children."""
    before = os.environ.get("VQ_NO_TORCH")
    try:
        return _main(argv=argv, net_factory=net_factory, program=program)
    finally:
        if before is None:
            os.environ.pop("VQ_NO_TORCH", None)
        else:
            os.environ["VQ_NO_TORCH"] = before


if __name__ == "__main__":
    sys.exit(main())
