"""Frame ingest in front of the TSN extractor: decoded frames or undecoded JPEG files -> uint8 device crops
``[n, 224, 224, C]``, the form the network's first kernel reads.  What the reference does per snippet on the host --
``cv2.imread`` (calcSig_wOF.py:92,105-106) and, inside ``predict_single_frame`` / ``predict_single_flow_stack(...,
frame_size=(340, 256))`` (:94,111), ``cv2.resize`` + crop 0 of the 10-crop over-sample -- runs on the GPU for whole batches.

A :class:`FrameIngest` depends on the stream's channel count, the device and the resize rule only -- not on any weights -- so the
command line creates it before (and prepares batches while) the extractors are being built, keeps preparing the next stream's
first batches while the current stream's last ones are in the network, and feeds several ensemble members from one decode.
"""
from __future__ import annotations

import atexit
import os
import threading

import numpy as np

from . import frames


# Decoders of closed FrameIngest objects, per device: a decoder's buffers (pinned host memory for the unstuffed streams / coefficients,
# device planes and pixels: 1-2 GB at the command line's batch sizes) cost 40 ms to free and as much to allocate again, so a process that
# runs the command line's main() again (the bench, a broker's worker) takes its decoders from here.  At most _POOL_KEEP idle ones per device.
_POOL_KEEP = 4
_idle_decoders = {}
_idle_lock = threading.Lock()


def _take_decoder(device, h, w):
    with _idle_lock:
        idle = _idle_decoders.get(device, [])
        for k, dec in enumerate(idle):
            if dec.max_h >= h and dec.max_w >= w:
                return idle.pop(k)
    return None


def _give_decoder(device, dec):
    with _idle_lock:
        idle = _idle_decoders.setdefault(device, [])
        if len(idle) < _POOL_KEEP:
            idle.append(dec)
            return
    dec.close()


def drain_decoder_pool():
    """Close the idle decoders (also at interpreter exit)."""
    with _idle_lock:
        idle = [d for ds in _idle_decoders.values() for d in ds]
        _idle_decoders.clear()
    for dec in idle:
        dec.close()


atexit.register(drain_decoder_pool)


class FrameIngest:
    def __init__(self, channels: int, device: int = 0, resize_rule: str = "cv2"):
        if resize_rule not in frames.RESIZE_RULES:
            raise ValueError("resize_rule must be 'cv2' or 'exact'")
        self._channels, self.device, self._resize_rule = int(channels), int(device), resize_rule
        self._lanes = {}

    def crops_from_frames(self, frames_: np.ndarray, frame_size=(340, 256), crop=224):
        """Decoded frames -> device crops (torch uint8 [n, crop, crop, C]) through vq_resize_crop: RGB frames
        [n, H, W, 3], or flow planes [n, C, H, W] (grey x/y frames in stack order).  Same bytes as frames.crop0."""
        import ctypes as C
        from .._lib import call
        from . import devmem
        f = np.ascontiguousarray(frames_, dtype=np.uint8)
        if f.ndim != 4:
            raise ValueError("frames must be [n,H,W,3] (RGB) or [n,C,H,W] (flow planes)")
        n = f.shape[0]
        out = devmem.empty_u8((n, crop, crop, self._channels), self.device)
        stream = devmem.current_stream_handle(self.device)
        if self._channels == 3:
            if f.shape[3] != 3:
                raise ValueError("RGB frames must be [n,H,W,3]")
            call("vq_resize_crop", f.ctypes.data_as(C.c_void_p), 0, n, f.shape[1], f.shape[2], 3, frame_size[0], frame_size[1], crop, frames.RESIZE_RULES[self._resize_rule],
                 C.c_void_p(out.data_ptr()), 3, 0, self.device, C.c_void_p(stream))
        else:
            if f.shape[1] != self._channels:
                raise ValueError("flow planes must be [n,%d,H,W]" % self._channels)
            for k in range(self._channels):
                plane = np.ascontiguousarray(f[:, k])
                call("vq_resize_crop", plane.ctypes.data_as(C.c_void_p), 0, n, f.shape[2], f.shape[3], 1, frame_size[0], frame_size[1],
                     crop, frames.RESIZE_RULES[self._resize_rule], C.c_void_p(out.data_ptr()), self._channels, k, self.device, C.c_void_p(stream))
        return out

    def sync(self):
        """Wait for the resize / crop work of ``crops_from_frames`` (queued on torch's current stream) before an extractor's own
        stream reads the crops."""
        from . import devmem
        devmem.synchronize_current(self.device)

    def crops_from_jpegs(self, files, frame_size=(340, 256), crop=224, lane=0):
        """JPEG file contents -> device crops (torch uint8 [n, crop, crop, C]) without the frames ever visiting the host:
        entropy decoding on the library's host threads, IDCT / upsampling / colour on the device (tsn/jpeg.py), resize +
        crop 0 straight from the decoder's device buffer.  RGB net: n files; flow net: n * C files in stack order per
        snippet (x0, y0, x1, y1, ...).  The pixels are libjpeg's (what cv2.imread returns), bit for bit.  ``lane``: calls of
        different lanes own different decoders and streams and may run at the same time in different threads (the command line
        keeps two batches in preparation: one's host half -- reading, unstuffing -- overlaps the other's device half)."""
        from . import devmem
        ch = self._channels
        per_snip = 1 if ch == 3 else ch
        if len(files) % per_snip:
            raise ValueError("flow net: %d files is not a multiple of the %d planes of a snippet" % (len(files), ch))
        n = len(files) // per_snip
        out = devmem.empty_u8((n, crop, crop, ch), self.device)
        st = self._lanes.setdefault(lane, {"stream": None, "jpeg": None, "lock": threading.Lock()})
        with st["lock"]:                                 # a lane's decoder buffer and stream serve one call at a time
            return self._crops_from_jpegs_on(st, files, n, ch, frame_size, crop, out)

    def _crops_from_jpegs_on(self, st, files, n, ch, frame_size, crop, out):
        import ctypes as C
        from .._lib import call
        from . import devmem, jpeg
        # a stream of its own (non-blocking): the call may run in a thread of its own for a LATER batch while the network works on the
        # current one on the default stream -- decoding a batch of flow files keeps a few CUs busy for tens of milliseconds
        if st["stream"] is None:
            # (a high-priority stream was tried for the decode kernels in round 4: no gain, 385-391 against 394-412 clips/s end to end.
            # Round 6, VQ_INGEST_PRIORITY=low|normal|high on one box: 576 / 576 / 577 and 575 / 531 / 571 clips/s -- the queue priority of the
            # ingest kernels does not change what the networks get; the default stays normal.)
            prio = {"low": -1, "normal": 0, "high": 1}.get(os.environ.get("VQ_INGEST_PRIORITY", "normal"), 0)
            st["stream"] = devmem.new_stream(self.device, prio)
        ingest = st["stream"]
        stream = ingest.cuda_stream
        h, w, _ = jpeg.info(files[0])
        # files per decoder call (its buffers grow to what a call needs; a call addresses its component planes with 32 bits)
        cap = max(1, min(8192, int(3.0e9 // (2 * (h + 16) * (w + 16)))))
        dec = st["jpeg"]
        if dec is None or dec.max_h < h or dec.max_w < w:
            if dec is not None:
                dec.close()
            dec = st["jpeg"] = _take_decoder(self.device, h, w) or jpeg.JpegDecoder(cap, h, w, self.device)
        cap = min(cap, dec.max_frames)
        same_size = (w, h) == tuple(frame_size) and crop % 2 == 0 and crop <= min(h, w)
        if same_size and ch in (3, 10):
            # frames as build_wof_clips.py writes them: resize is the identity, crop 0 the top-left pixels -- decoded and cropped without the
            # whole-frame pixel pass in between (8 000 grey frames per flow batch: 0.7 GB written and read again for the 57 % that survive)
            per = cap if ch == 3 else max(1, cap // ch)
            for i in range(0, n, per):
                m = min(per, n - i)
                group = files[i:i + m] if ch == 3 else [f for k in range(ch) for f in files[i * ch + k:(i + m) * ch:ch]]
                dec.decode_to_crops(group, ch == 3, crop, out[i:i + m].data_ptr(), planes=ch, stream=stream)
            return out
        if ch == 3:
            for i in range(0, n, cap):
                ptr, (m, _, _) = dec.decode_to_device(files[i:i + cap], color=True, stream=stream)
                call("vq_resize_crop", C.c_void_p(ptr), 1, m, h, w, 3, frame_size[0], frame_size[1], crop, frames.RESIZE_RULES[self._resize_rule],
                     C.c_void_p(out[i:i + m].data_ptr()), 3, 0, self.device, C.c_void_p(stream))
                ingest.synchronize()                                  # the decoder's buffer is reused by its next call
        else:
            # the grey frames of `per` snippets in ONE decoder call, plane-major (all x0 frames, then all y0 frames, ...): a batch of
            # 32 clips x 25 snippets is 8 000 small files -- the size at which the entropy decoding runs on the device -- and every
            # plane's frames are contiguous for the resize that interleaves them into the 10-channel crops
            per = max(1, cap // ch)
            for i in range(0, n, per):
                m = min(per, n - i)
                group = [f for k in range(ch) for f in files[i * ch + k:(i + m) * ch:ch]]
                ptr, _ = dec.decode_to_device(group, color=False, stream=stream)
                if ch == 10 and crop % 2 == 0:
                    # the ten planes of the stacks in ONE launch (taps once per pixel, whole-word stores; frames that have the size already
                    # are copied): 10 launches of 0.53 ms per 800 crops took 5.3 ms of the GPU per batch beside the networks
                    call("vq_resize_crop_planes", C.c_void_p(ptr), m, h, w, ch, m * h * w, frame_size[0], frame_size[1], crop,
                         frames.RESIZE_RULES[self._resize_rule], C.c_void_p(out[i:i + m].data_ptr()), self.device, C.c_void_p(stream))
                else:
                    for k in range(ch):
                        call("vq_resize_crop", C.c_void_p(ptr + k * m * h * w), 1, m, h, w, 1, frame_size[0], frame_size[1], crop,
                             frames.RESIZE_RULES[self._resize_rule], C.c_void_p(out[i:i + m].data_ptr()), ch, k, self.device, C.c_void_p(stream))
                ingest.synchronize()
        return out

    def close(self):
        for st in self._lanes.values():
            if st["jpeg"] is not None:
                _give_decoder(self.device, st["jpeg"])             # idle, not freed: see _idle_decoders
                st["jpeg"] = None
