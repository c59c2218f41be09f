"""Frame / warped-optical-flow preparation: the drop-in for the reference's ``build_wof_clips.py`` command line.

Reference: src/features_GPU_compute/build_wof_clips.py -- per video it (1) shells out to the third-party
``extract_warp_gpu -b 20 -t 1 -s 1`` for ``flow_x_NNNNN.jpg`` / ``flow_y_NNNNN.jpg`` (:55-76), (2) dumps the RGB frames with
cv2, skipping the first one, as ``img_NNNNN.jpg`` (:24-53), (3) regroups everything into ``clip_NNNN`` directories of
``clip_time * fps`` frames, keeping a shorter last clip if it lasts at least 2 s (:78-128).  Same positional arguments and
flags (:132-152); ``--df_path`` and ``--out_format`` are accepted and ignored (there is no external binary; output is
always directories).

What runs where: the flow is ``Tvl1Flow.warped_consecutive`` (csrc/vq_flow.hip: TV-L1, camera motion from corners moved by
the first-pass flow, second pass; the SURF matches the binary also uses are not built -- parity unpinned either way, the
binary is not in the reference tree).  Grey conversion follows cv2's 8-bit BGR2GRAY fixed-point rule.  Files are written
with cv2 or Pillow at cv2.imwrite's default JPEG quality (95), or as .ppm / .pgm when neither is importable
(``calcSig_wOF.py --frame_ext .ppm`` reads those).

Video decoding needs cv2 (``VideoCapture``); where it is missing (this image), a "video" may be a directory
``<src_dir>/<name>/`` of ALL its frames in order (``frame_00000.jpg`` ..., any extension ``tsn/frames.imread`` reads): frame
0 plays the part of the initial frame the reference skips, so ``img_00001`` is the second frame and ``flow_x_00001`` the flow
from the first to the second, exactly the numbering the reference's two passes produce.
"""
from __future__ import annotations

import argparse
import glob
import os
import sys

if __package__ in (None, ""):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import video_query_algorithms_amd  # noqa: F401  (registers the package under its importable name)
    from video_query_algorithms_amd.tsn import frames as frames_mod
    from video_query_algorithms_amd.tsn.flow import Tvl1Flow
    from video_query_algorithms_amd import fanout
else:
    from . import fanout
    from .tsn import frames as frames_mod
    from .tsn.flow import Tvl1Flow

import numpy as np


def bgr_to_grey(frame: np.ndarray) -> np.ndarray:
    """cv2.cvtColor(frame, COLOR_BGR2GRAY) for uint8: (B * 1868 + G * 9617 + R * 4899 + 8192) >> 14."""
    f = frame.astype(np.int32)
    return ((f[..., 0] * 1868 + f[..., 1] * 9617 + f[..., 2] * 4899 + 8192) >> 14).astype(np.uint8)


def _writer():
    """(extension, write(path_without_ext, image)) for the best image writer available."""
    try:
        import cv2
        return ".jpg", lambda p, img: cv2.imwrite(p + ".jpg", img)
    except ImportError:
        pass
    try:
        from PIL import Image

        def save(p, img):
            im = Image.fromarray(img if img.ndim == 2 else np.ascontiguousarray(img[:, :, ::-1]))
            if img.ndim == 3:
                im.save(p + ".jpg", "JPEG", quality=95, subsampling=2)         # cv2.imwrite's defaults: quality 95, 4:2:0
            else:
                im.save(p + ".jpg", "JPEG", quality=95)
        return ".jpg", save
    except ImportError:
        return ".ppm", lambda p, img: frames_mod.write_pnm(p + ".ppm", img)


def iter_video(path: str):
    """The frames of a video one by one as BGR uint8 [h, w, 3]: a video file through cv2.VideoCapture, or a directory of
    frames.  Nothing is kept: a 10-minute video never sits in host memory as a whole."""
    if os.path.isdir(path):
        names = sorted(n for n in os.listdir(path) if os.path.splitext(n)[1].lower() in (".jpg", ".jpeg", ".ppm", ".pnm", ".npy", ".png"))
        if not names:
            raise IOError("no frames in " + path)
        for n in names:
            yield frames_mod.imread(os.path.join(path, n), True)
        return
    try:
        import cv2
    except ImportError:
        raise ImportError("decoding %s needs cv2.VideoCapture; without cv2 give a directory of the video's frames instead" % path)
    video = cv2.VideoCapture(path)
    while True:
        ret, frame = video.read()
        if not ret:
            break
        yield frame


def iter_video_ahead(path: str, readers: int = 4, ahead: int = 32):
    """``(frame, grey)`` of a video in order, decoded (and converted to grey) up to ``ahead`` frames in front of the consumer: a
    directory of frames by ``readers`` threads, a video file by one (cv2.VideoCapture is sequential).  The consumer's GPU calls and
    the decoding of the next window's frames then overlap instead of taking turns; memory stays bounded by ``ahead`` frames."""
    import collections
    from concurrent.futures import ThreadPoolExecutor

    def prepared(frame):
        return frame, bgr_to_grey(frame)
    if os.path.isdir(path):
        names = sorted(n for n in os.listdir(path) if os.path.splitext(n)[1].lower() in (".jpg", ".jpeg", ".ppm", ".pnm", ".npy", ".png"))
        if not names:
            raise IOError("no frames in " + path)
        with ThreadPoolExecutor(max_workers=max(1, readers)) as pool:
            pending = collections.deque()
            todo = iter(names)
            for n in todo:
                pending.append(pool.submit(lambda n=n: prepared(frames_mod.imread(os.path.join(path, n), True))))
                if len(pending) >= ahead:
                    break
            while pending:
                item = pending.popleft().result()
                n = next(todo, None)
                if n is not None:
                    pending.append(pool.submit(lambda n=n: prepared(frames_mod.imread(os.path.join(path, n), True))))
                yield item
        return
    import queue
    import threading
    q = queue.Queue(maxsize=max(2, ahead))
    stop = threading.Event()

    def pump():
        try:
            for frame in iter_video(path):
                if stop.is_set():
                    return
                q.put(prepared(frame))
            q.put(None)
        except BaseException as e:      # noqa: BLE001 -- handed to the consumer
            q.put(e)
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    try:
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, BaseException):
                raise item
            yield item
    finally:
        stop.set()
        while th.is_alive():            # unblock a producer waiting on a full queue
            try:
                q.get_nowait()
            except queue.Empty:
                th.join(0.01)


def read_video(path: str) -> np.ndarray:
    """All frames of a (short) video as BGR uint8 [n, h, w, 3]."""
    return np.stack(list(iter_video(path)))


def process_video(vid_path: str, out_path: str, flow_for, new_size=(0, 0), seed: int = 0, max_pairs: int = 64, writers=None) -> int:
    """Steps (1) and (2) for one video: ``img_NNNNN`` for frames 1.. (resized to ``new_size`` when given, as the reference's
    dump_frames does, build_wof_clips.py:40-42), ``flow_x/y_NNNNN`` for the warped flow (k-1 -> k) of the frames AS DECODED
    (the reference hands the video file itself to extract_warp_gpu, :70-73: the flow never sees --new_width / --new_height).
    The video streams through in windows of ``max_pairs`` frame pairs.  ``flow_for(h, w)`` gives the flow workspace for a
    frame size.  ``writers``: a thread pool that encodes and writes the image files (three JPEG encodings per frame are more host
    time than the flow is GPU time; --num_worker threads, the reference's "CPU workers", share them) -- all files of the video are
    on disk when the call returns.  Returns the number of frames written."""
    vid_name = os.path.basename(os.path.normpath(vid_path)).split('.')[0]
    out_full_path = os.path.join(out_path, vid_name)
    os.makedirs(out_full_path, exist_ok=True)
    ext, write = _writer()
    flow = None
    window, greys = [], []                  # frames k0 .. of the current window; greys[0] is the last frame of the one before
    done = 0                                # frame pairs written so far

    jobs = []

    def put(path, img):
        if writers is None:
            write(path, img)
        else:
            jobs.append(writers.submit(write, path, img))

    def write_rgb(path, frame):
        write(path, frame if new_size == (0, 0) else frames_mod.resize_bilinear(frame, new_size))

    def flush():
        nonlocal done
        if len(greys) < 2:
            return
        fx, fy = flow.warped_consecutive(np.stack(greys), seed=seed + done)
        for j, frame in enumerate(window):
            k = done + j + 1
            if writers is None:
                write_rgb('{}/img_{:05d}'.format(out_full_path, k), frame)
            else:
                jobs.append(writers.submit(write_rgb, '{}/img_{:05d}'.format(out_full_path, k), frame))
            put('{}/flow_x_{:05d}'.format(out_full_path, k), fx[j])
            put('{}/flow_y_{:05d}'.format(out_full_path, k), fy[j])
        done += len(window)
        # the encoders are slower than the flow: at most two windows of images wait for them (each pending job holds a frame or a
        # flow plane), so a long video never sits in host memory as a whole
        while len(jobs) > 2 * 3 * max_pairs:
            jobs.pop(0).result()

    for i, (frame, g) in enumerate(iter_video_ahead(vid_path)):
        if flow is None:
            flow = flow_for(g.shape[0], g.shape[1])
        elif g.shape != (flow.h, flow.w):
            raise ValueError("%s: frame %d is %dx%d, the first one %dx%d" % (vid_name, i, g.shape[1], g.shape[0], flow.w, flow.h))
        greys.append(g)
        if i > 0:                                                              # the reference skips the initial frame (:34)
            window.append(frame)
        if len(window) == max_pairs:
            flush()
            window, greys = [], [g]
    flush()
    for j in jobs:
        j.result()                                                              # an encoder's exception is the command's
    return done


def create_clip(vid_path: str, out_path: str, frames_per_clip: int = 150, frames_per_second: int = 15) -> int:
    """Step (3), build_wof_clips.py:78-128: move the frames of one video into clip_NNNN directories.  Returns the clips made."""
    vid_name = os.path.basename(os.path.normpath(vid_path)).split('.')[0]
    out_video_path = os.path.join(out_path, vid_name)
    rgb = sorted(glob.glob(os.path.join(out_video_path, 'img_*')))
    if not rgb:
        return 0
    ext = os.path.splitext(rgb[0])[1]
    plan, dropped = frames_mod.clip_plan(len(rgb), frames_per_clip, frames_per_second)
    kinds = ('img', 'flow_x', 'flow_y')
    for clip, first, last in plan:
        clip_dir = os.path.join(out_video_path, 'clip_{:04d}'.format(clip))
        os.mkdir(clip_dir)
        for iframe in range(first, last + 1):
            for kind in kinds:
                os.replace(os.path.join(out_video_path, '{}_{:05d}{}'.format(kind, iframe, ext)),
                           os.path.join(clip_dir, '{}_{:05d}{}'.format(kind, iframe - first + 1, ext)))
    for iframe in range(len(rgb) - dropped + 1, len(rgb) + 1):                 # a tail shorter than 2 s is deleted (:123-126)
        for kind in kinds:
            os.remove(os.path.join(out_video_path, '{}_{:05d}{}'.format(kind, iframe, ext)))
    return len(plan)


def _main(argv=None, program=None) -> int:
    if "torch" not in sys.modules:
        # nothing here holds a tensor: the library runs on the system's HIP runtime and the 0.8 s import is saved (VQ_NO_TORCH=0 keeps torch;
        # decided before the library is loaded, _lib._preload_torch_hip)
        os.environ.setdefault("VQ_NO_TORCH", "1")
    parser = argparse.ArgumentParser(description="Extract rgb and warped optical flow frames")
    parser.add_argument("src_dir", help="directory with video files")
    parser.add_argument("out_dir")
    parser.add_argument("--fps", type=int, default=15, help="frames per second, default = 15")
    parser.add_argument("--clip_time", type=int, default=10, help="clip time in seconds, default = 10")
    parser.add_argument("--num_worker", type=int, default=16, help="CPU workers, default=16 (here: the threads that encode and write the image files)")
    parser.add_argument("--df_path", type=str, default='./lib/dense_flow/', help='accepted for compatibility; no external toolbox is used')
    parser.add_argument("--out_format", type=str, default='dir', choices=['dir', 'zip'], help='format of output, default=dir')
    parser.add_argument("--ext", type=str, default='mp4', choices=['avi', 'mp4'], help='video file extensions, default = mp4')
    parser.add_argument("--new_width", type=int, default=0, help='resize image width')
    parser.add_argument("--new_height", type=int, default=0, help='resize image height')
    parser.add_argument("--num_gpu", type=int, default=1, help='number of GPU, default = 1 (one process per GPU, videos dealt round-robin)')
    parser.add_argument("--starting_gpu", type=int, default=0, help='ID of first GPU to use, default = 0')
    parser.add_argument("--max_pairs", type=int, default=64, help='frame pairs per flow batch on the GPU (not in the reference)')
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parser.parse_args(argv)
    new_size = (args.new_width, args.new_height)
    assert new_size == (0, 0) or (new_size[0] != 0 and new_size[1] != 0)
    if args.out_format != 'dir':
        raise SystemExit("only --out_format dir is produced")
    os.makedirs(args.out_dir, exist_ok=True)
    vid_list = sorted(glob.glob(args.src_dir + '/*.' + args.ext))
    if not vid_list:                                                            # frame directories stand in for video files
        vid_list = sorted(d for d in glob.glob(args.src_dir + '/*') if os.path.isdir(d))
    if not fanout.is_child():
        print("number of videos found = {}".format(len(vid_list)))
    if args.num_gpu > 1 and not fanout.is_child():
        # the reference spreads its workers over the GPUs (worker i -> (i - 1) % NUM_GPU + START_GPU, build_wof_clips.py:66);
        # here: one fresh process per GPU, started before this one makes any GPU call, videos dealt round-robin
        envs = [{"VQ_FANOUT_RANK": str(i), "VQ_FANOUT_WORLD": str(args.num_gpu)} for i in range(args.num_gpu)]
        return fanout.run_children(program or os.path.abspath(__file__), argv, envs)
    rank, world = int(os.environ.get("VQ_FANOUT_RANK", "0")), int(os.environ.get("VQ_FANOUT_WORLD", "1"))
    device = args.starting_gpu + rank
    flows = {}                                                                  # (h, w) -> flow workspace

    def flow_for(h, w):
        if (h, w) not in flows:
            flows[(h, w)] = Tvl1Flow(args.max_pairs, h, w, device=device)
        return flows[(h, w)]

    from concurrent.futures import ThreadPoolExecutor
    writers = ThreadPoolExecutor(max_workers=max(1, args.num_worker // world)) if args.num_worker > 1 else None
    mine = [(vid_id, vid_path) for vid_id, vid_path in enumerate(vid_list) if vid_id % world == rank]
    for vid_id, vid_path in mine:                                               # the seed is the video's GLOBAL index: the
        n = process_video(vid_path, args.out_dir, flow_for, new_size, seed=(vid_id * 2654435761) & 0xFFFFFFFF, max_pairs=args.max_pairs, writers=writers)   # files do not depend on --num_gpu
        print('warp + rgb for {} {} done ({} frames)'.format(vid_id, os.path.basename(os.path.normpath(vid_path)), n))
        sys.stdout.flush()
    for _, vid_path in mine:                                                    # build_wof_clips.py:188-191
        create_clip(vid_path, args.out_dir, frames_per_clip=args.clip_time * args.fps, frames_per_second=args.fps)
    if writers is not None:
        writers.shutdown()
    for f in flows.values():
        f.close()
    if os.environ.get("VQ_CLI_TRACE") == "1":
        print("trace: done (torch imported: %s)" % ("torch" in sys.modules), file=sys.stderr, flush=True)
    return 0



def main(argv=None, program=None) -> int:
    """The command line's entry point: VQ_NO_TORCH (set for a one-rank run before the library is loaded, where it is latched:
    _lib.TORCHLESS) is put back when the call returns."""
    before = os.environ.get("VQ_NO_TORCH")
    try:
        return _main(argv=argv, program=program)
    finally:
        if before is None:
            os.environ.pop("VQ_NO_TORCH", None)
        else:
            os.environ["VQ_NO_TORCH"] = before


if __name__ == '__main__':
    sys.exit(main())
