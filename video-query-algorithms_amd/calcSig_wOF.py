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

import numpy as np

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import video_query_algorithms_amd  # noqa: F401
    from video_query_algorithms_amd.tsn import frames
    from video_query_algorithms_amd.tsn.caffe_net import CaffeNet
    from video_query_algorithms_amd.tsn.feature_csv import write_features
    from video_query_algorithms_amd.shard import all_gather_rows, shard_range
    from video_query_algorithms_amd import fanout
else:
    from . import fanout
    from .tsn import frames
    from .tsn.caffe_net import CaffeNet
    from .tsn.feature_csv import write_features
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
    parser.add_argument('--batch_clips', type=int, default=32, help='clips per forward pass')
    parser.add_argument('--host_resize', action='store_true',
                        help='resize + crop the frames on the host (numpy) instead of on the GPU; same bytes either way')
    parser.add_argument('--exact_resize', action='store_true',
                        help="resize with exact fp64 bilinear weights instead of cv2.resize's 11-bit fixed-point rule (the default, "
                             "what the reference's dependency computes); differs by one grey level on about one pixel in eight")
    parser.add_argument('--device_jpeg', action='store_true',
                        help='decode the .jpg frames with the library (Huffman on host threads, IDCT / upsampling / colour on the '
                             'GPU; the pixels libjpeg gives cv2.imread) and resize them where they land -- no host image library')
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


def main(argv=None, net_factory=None, program=None):
    """``net_factory(net_proto, net_weights, device, max_crops=, feature_blob=, resize_rule=)`` builds the per-stream extractor
    (default: the HIP ``CaffeNet``; the CPU tests of the sharding logic pass a stand-in).  ``program``: the script the
    per-GPU children are started from (default: this file; a caller that passes its own ``net_factory`` names its own
    script here, or gets no fan-out)."""
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
    backend_device = None                                # where a rank's blocks must live for the collective (RCCL: its GPU)
    if world > 1:
        import torch.distributed as dist
        if dist.get_backend() == "nccl":
            import torch
            backend_device = torch.device("cuda", device)
    frame_path = args.frame_path if args.frame_path[-1] == '/' else args.frame_path + '/'
    streamCNN = [{'modality': 'rgb', 'mode': 'rgb', 'net_proto': args.net_proto_rgb, 'net_weights': args.net_weights_rgb,
                  'cnt_indexer': 1, 'stack_depth': 1},
                 {'modality': 'flow', 'mode': 'warped_optical_flow', 'net_proto': args.net_proto_flow,
                  'net_weights': args.net_weights_flow, 'cnt_indexer': 2, 'stack_depth': 5}]   # calcSig_wOF.py:185-189
    nets = {}
    rule = "exact" if args.exact_resize else "cv2"
    device_jpeg = args.device_jpeg and args.frame_ext.lower() in ('.jpg', '.jpeg') and not args.host_resize
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=max(1, args.num_worker))
    prep_pool = ThreadPoolExecutor(max_workers=2)        # --device_jpeg: the stages in front of the network (see the batch loop)
    io_pool = ThreadPoolExecutor(max_workers=len(streamCNN))             # feature files are formatted and written here
    csv_jobs = []
    build_pool = ThreadPoolExecutor(max_workers=len(streamCNN))
    net_jobs = {}                                        # both extractors are built side by side when the first video turns up: the flow
                                                         # net's weights are folded and uploaded while the RGB stream is already running
    for video_path in sorted(glob.glob(frame_path + '*/')):                            # calcSig_wOF.py:193-195
        f_info = frames.parse_directory(video_path, args.rgb_prefix, args.flow_x_prefix, args.flow_y_prefix)
        clip_list = sorted(list(f_info[0]), key=lambda clip: int(clip[-4:]))           # calcSig_wOF.py:199-200
        first, count = shard_range(len(clip_list), world, rank)
        features = {}
        if not net_jobs:
            for s in streamCNN:
                net_jobs[s['modality']] = build_pool.submit(net_factory, s['net_proto'], s['net_weights'], device, max_crops=args.batch_clips * T,
                                                            feature_blob=args.featureBlob, resize_rule=rule)
        for s in streamCNN:
            if s['modality'] not in nets:
                nets[s['modality']] = net_jobs[s['modality']].result()
            net = nets[s['modality']]
            mine = []

            host_rule = {'rule': rule} if args.host_resize else {}         # the host loaders resize; the others hand frames to the GPU

            def load_clip(vid, s=s):
                frame_cnt = f_info[s['cnt_indexer']][vid]
                ticks = frames.frame_ticks(frame_cnt, T, s['stack_depth'])
                if s['modality'] == 'rgb':
                    load = frames.load_rgb_jpegs if device_jpeg else frames.load_rgb_snippets if args.host_resize else frames.load_rgb_frames
                    return load(f_info[0][vid], ticks, args.rgb_prefix, args.frame_ext, **host_rule)
                load = frames.load_flow_jpegs if device_jpeg else frames.load_flow_snippets if args.host_resize else frames.load_flow_frames
                return load(f_info[0][vid], ticks, frame_cnt, s['stack_depth'], args.flow_x_prefix, args.flow_y_prefix, args.frame_ext, **host_rule)

            # --num_worker decoder threads (the reference runs that many worker PROCESSES, each with its own net,
            # calcSig_wOF.py:204-210); the batch after the one on the GPU is decoded meanwhile
            batches = [clip_list[b0:min(b0 + args.batch_clips, first + count)] for b0 in range(first, first + count, args.batch_clips)]
            pending = [pool.submit(load_clip, vid) for vid in batches[0]] if batches else []
            on_gpu = world > 1                           # blocks that will be all-gathered never visit the host
            staged = []                                  # --device_jpeg: the crops of the batches in front of the network, being made by prep_pool
            for bi, vids in enumerate(batches):
                crops = [f.result() for f in pending]
                pending = [pool.submit(load_clip, vid) for vid in batches[bi + 1]] if bi + 1 < len(batches) else []
                for vid in vids:
                    print('video {} for {} modality done'.format(vid, s['modality']))
                if not crops:
                    continue
                if device_jpeg:
                    # Three stages: two threads read, decode, resize and crop batches b + 1 and b + 2 on the GPU (lists of undecoded files in,
                    # device crops out; the library's calls release the interpreter; each thread has a decoder and a stream of its own, so
                    # one batch's host half -- reading the files, stripping the byte stuffing -- overlaps the other's device half) while
                    # this one runs batch b through the network
                    staged.append(prep_pool.submit(net.crops_from_jpegs, [f for c in crops for f in c], lane=bi % 2))
                    if len(staged) > 2:
                        mine.append(net.extract_clips_from_crops(staged.pop(0).result(), T, on_device=on_gpu))
                elif args.host_resize:
                    mine.append(net.extract_clips(np.concatenate(crops, axis=0), T, on_device=on_gpu))
                elif len({c.shape[1:] for c in crops}) == 1:
                    mine.append(net.extract_clips_from_frames(np.concatenate(crops, axis=0), T, on_device=on_gpu))   # resize + crop on the GPU
                else:                                        # clips of different frame sizes in one batch
                    mine += [net.extract_clips_from_frames(c, T, on_device=on_gpu) for c in crops]
            for job in staged:
                mine.append(net.extract_clips_from_crops(job.result(), T, on_device=on_gpu))
            local_feat = _stack_rows(mine, net.feature_dim)
            if world > 1:
                import torch
                if isinstance(local_feat, np.ndarray):       # host blocks, or a rank that owns no clip of this video
                    local_feat = torch.from_numpy(np.ascontiguousarray(local_feat, dtype=np.float64))
                if backend_device is not None and local_feat.device != backend_device:
                    local_feat = local_feat.to(backend_device)
                local_feat = all_gather_rows(local_feat, len(clip_list)).cpu().numpy()
            features[s['mode']] = local_feat
            numFeatures = local_feat.shape[1] if len(clip_list) else args.featureBlob_size
            assert numFeatures == args.featureBlob_size                                  # calcSig_wOF.py:219-220
            if rank == 0 and clip_list:
                # a stream's file is formatted and written (half a million float reprs per 256 clips) by a thread of its own while the
                # next stream -- or the next video -- is on the GPU; the same two files as writing both at the end (:116-134)
                video = video_path.split('/')[-2]
                csv_jobs.append(io_pool.submit(write_features, args.outFeatures_dir, video, video_path, args.modelname, args.featureBlob, clip_list,
                                               {s['mode']: local_feat}, {'rgb': args.net_weights_rgb, 'warped_optical_flow': args.net_weights_flow},
                                               args.number_format))
    pool.shutdown()
    prep_pool.shutdown()
    build_pool.shutdown()
    for n in nets.values():                                              # releasing the device buffers overlaps the last file's formatting
        n.close()
    for job in csv_jobs:
        job.result()                                                     # a writer's exception is the command's
    io_pool.shutdown()
    return 0


if __name__ == "__main__":
    sys.exit(main())
