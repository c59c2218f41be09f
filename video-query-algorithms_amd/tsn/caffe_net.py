"""``CaffeNet``-shaped front door for the MI355X TSN extractor.

The reference touches exactly four things of ``pyActionRecog.action_caffe.CaffeNet``:
``CaffeNet(net_proto, net_weights, device_id)`` (calcSig_wOF.py:52,55), ``predict_single_frame([frame], score_name,
frame_size=(340,256))`` (:94), ``predict_single_flow_stack(flow_stack, score_name, frame_size=(340,256))`` (:111) and
``net._net.blobs[blob].data[0]`` (:95,112).  This class offers the same four, so the reference's per-snippet loop runs
unchanged on top of it, plus ``extract_clips`` -- the batched path (all B*T crops in one forward) the drop-in
command line uses.

Weights: a ``.caffemodel`` (decoded by ``tsn/caffemodel.py`` without Caffe or a protobuf schema; the BN blob order is
"parity unpinned", see there), an ``.npz`` with ``<layer>/W``, ``<layer>/b`` for convolutions and
``<layer>/scale|shift|mean|var`` for the frozen BN layers, or ``synthetic:<seed>`` (random init of the right
architecture).
"""
from __future__ import annotations

import numpy as np

from . import bn_inception, frames
from .ingest import FrameIngest
from .net import FLOW_MEAN, RGB_MEAN, TsnNet, synthetic_weights


def load_weights(graph, spec: str):
    if spec.startswith("synthetic:"):
        return synthetic_weights(graph, seed=int(spec.split(":", 1)[1]))
    if spec.endswith(".npz"):
        z = np.load(spec, allow_pickle=False)
        out = {}
        for key in z.files:
            layer, field = key.rsplit("/", 1)
            out.setdefault(layer, {})[field] = z[key]
        return out
    if spec.endswith(".caffemodel"):
        from .caffemodel import weights_from_caffemodel
        return weights_from_caffemodel(spec, graph)
    raise ValueError("weights file %r: expected .caffemodel, .npz or synthetic:<seed>" % spec)


def weights_digest(spec: str) -> str:
    """Names the CONTENT of a weights specification: ``synthetic:<seed>`` as it is, a file by a digest of its bytes."""
    if spec.startswith("synthetic:"):
        return spec
    try:                                   # 8 ms for a 41 MB .caffemodel; SHA-1 (40 ms) where the module is missing
        import xxhash
        h, tag = xxhash.xxh3_128(), "file:xxh3:"
    except ImportError:
        import hashlib
        h, tag = hashlib.sha1(), "file:"
    with open(spec, "rb") as f:
        for piece in iter(lambda: f.read(1 << 22), b""):
            h.update(piece)
    return tag + h.hexdigest()


def save_weights(path: str, weights):
    np.savez(path, **{"%s/%s" % (layer, f): a for layer, d in weights.items() for f, a in d.items()})


class _Blob:
    def __init__(self, data):
        self.data = data


class _NetView:
    def __init__(self):
        self.blobs = {}


class CaffeNet:
    def __init__(self, net_proto, net_weights, device_id=0, max_crops=96, feature_blob="global_pool", resize_rule="cv2"):
        """resize_rule: "cv2" = OpenCV's fixed-point INTER_LINEAR, what the reference's ``cv2.resize(frame, (340, 256))``
        computes (default); "exact" = exact fp64 bilinear weights (tsn/frames.py)."""
        if resize_rule not in frames.RESIZE_RULES:
            raise ValueError("resize_rule must be 'cv2' or 'exact'")
        self._resize_rule = resize_rule
        self._graph = bn_inception.load_prototxt(net_proto) if isinstance(net_proto, str) else net_proto
        self._blob = feature_blob
        self._channels = self._graph.input_shape[0]
        self._mean = RGB_MEAN if self._channels == 3 else tuple([128.0] * self._channels)
        if isinstance(net_weights, str):
            # read (and fold, and lay out) only if the packed form of exactly these weights is not in the cache next to the library
            graph = self._graph
            self._model = TsnNet(graph, lambda: load_weights(graph, net_weights), max_crops=max_crops, device=device_id, feature_blob=feature_blob,
                                 cache_key=weights_digest(net_weights))
        else:
            self._model = TsnNet(self._graph, net_weights, max_crops=max_crops, device=device_id, feature_blob=feature_blob)
        self._net = _NetView()
        self._ingest = FrameIngest(self._channels, device_id, resize_rule)

    # -- the reference's per-snippet interface ------------------------------------------------------------
    def predict_single_frame(self, frame, score_name=None, over_sample=True, frame_size=(340, 256)):
        crop = frames.crop0(frame[0], frame_size, rule=self._resize_rule)[None]                     # crop 0 of the 10-crop over-sample
        _, ps = self._model.forward(crop, 1, self._mean)
        self._net.blobs[self._blob] = _Blob(ps.reshape(1, -1, 1, 1))        # .data[0] is what calcSig reads
        return None

    def predict_single_flow_stack(self, frame, score_name=None, over_sample=True, frame_size=(340, 256)):
        crop = np.stack([frames.crop0(f, frame_size, rule=self._resize_rule) for f in frame], axis=-1)[None]
        _, ps = self._model.forward(crop, 1, self._mean)
        self._net.blobs[self._blob] = _Blob(ps.reshape(1, -1, 1, 1))
        return None

    # -- the batched path ------------------------------------------------------------------------------------
    def extract_clips(self, crops: np.ndarray, T: int, on_device: bool = False):
        """crops uint8 [B*T, 224, 224, C] -> consensus features [B, D] float64 (calcSig_wOF.py:82); with
        ``on_device`` a torch tensor that never left the GPU (the block a rank contributes to the all-gather)."""
        out = []
        per = (self._model.max_crops // T) * T
        if per == 0:
            raise ValueError("max_crops (%d) is smaller than T (%d)" % (self._model.max_crops, T))
        for i in range(0, crops.shape[0], per):
            if on_device:
                import torch
                chunk = torch.from_numpy(np.ascontiguousarray(crops[i:i + per], dtype=np.uint8)).to(torch.device("cuda", self._model.device))
                self._model.forward_device(chunk.data_ptr(), chunk.shape[0], T, self._mean)
                out.append(self._model.features_tensor(chunk.shape[0] // T).clone())
            else:
                feat, _ = self._model.forward(crops[i:i + per], T, self._mean, want_per_snippet=False)
                out.append(feat)
        if on_device:
            import torch
            return torch.cat(out, dim=0)
        return np.concatenate(out, axis=0)

    def crops_from_frames(self, frames_: np.ndarray, frame_size=(340, 256), crop=224):
        """Decoded frames -> device crops; see :class:`tsn.ingest.FrameIngest` (this extractor's own instance)."""
        return self._ingest.crops_from_frames(frames_, frame_size, crop)

    def sync_ingest(self):
        self._ingest.sync()

    def crops_from_jpegs(self, files, frame_size=(340, 256), crop=224, lane=0):
        """JPEG file contents -> device crops; see :class:`tsn.ingest.FrameIngest`."""
        return self._ingest.crops_from_jpegs(files, frame_size, crop, lane)

    def extract_clips_from_jpegs(self, files, T: int, frame_size=(340, 256), on_device: bool = False):
        """JPEG file contents of B*T snippets (flow: * C planes) -> consensus features [B, D]; see crops_from_jpegs."""
        from . import devmem
        per_snip = 1 if self._channels == 3 else self._channels
        per = (self._model.max_crops // T) * T
        if per == 0:
            raise ValueError("max_crops (%d) is smaller than T (%d)" % (self._model.max_crops, T))
        out = []
        for i in range(0, len(files) // per_snip, per):
            crops = self.crops_from_jpegs(files[i * per_snip:(i + per) * per_snip], frame_size)
            devmem.synchronize_current(self._model.device)
            nb = crops.shape[0]
            if on_device:
                self._model.forward_device(crops.data_ptr(), nb, T, self._mean)
                out.append(self._model.features_tensor(nb // T).clone())
            else:
                out.append(self._model.forward_device(crops.data_ptr(), nb, T, self._mean, np.empty((nb // T, self._model.feature_dim), dtype=np.float64)))
        if on_device:
            import torch
            return torch.cat(out, dim=0)
        return np.concatenate(out, axis=0)

    def extract_clips_from_crops(self, crops, T: int, on_device: bool = False):
        """Device crops (torch uint8 [B*T, crop, crop, C], e.g. from ``crops_from_jpegs`` run by another thread for the NEXT batch while
        this one is in the network) -> consensus features [B, D]."""
        nb = crops.shape[0]
        if nb > self._model.max_crops or nb % T:
            raise ValueError("%d crops: at most max_crops (%d), a multiple of T (%d)" % (nb, self._model.max_crops, T))
        if on_device:
            self._model.forward_device(crops.data_ptr(), nb, T, self._mean)
            return self._model.features_tensor(nb // T).clone()
        # the features come back behind the forward on the extractor's own stream: the call does not wait for the decode kernels of the
        # batches ahead, which other threads have in flight on the ingest streams
        return self._model.forward_device(crops.data_ptr(), nb, T, self._mean, np.empty((nb // T, self._model.feature_dim), dtype=np.float64))

    def extract_clips_from_frames(self, frames_: np.ndarray, T: int, frame_size=(340, 256), on_device: bool = False):
        """Decoded frames of B*T snippets -> consensus features [B, D]: resize + crop 0 on the device, then the
        batched forward on the resident crops (no host-side image processing at all).  ``on_device``: see extract_clips."""
        from . import devmem
        per = (self._model.max_crops // T) * T
        if per == 0:
            raise ValueError("max_crops (%d) is smaller than T (%d)" % (self._model.max_crops, T))
        out = []
        for i in range(0, frames_.shape[0], per):
            crops = self.crops_from_frames(frames_[i:i + per], frame_size)
            devmem.synchronize_current(self._model.device)
            nb = crops.shape[0]
            if on_device:
                self._model.forward_device(crops.data_ptr(), nb, T, self._mean)
                out.append(self._model.features_tensor(nb // T).clone())
            else:
                out.append(self._model.forward_device(crops.data_ptr(), nb, T, self._mean, np.empty((nb // T, self._model.feature_dim), dtype=np.float64)))
        if on_device:
            import torch
            return torch.cat(out, dim=0)
        return np.concatenate(out, axis=0)

    @property
    def feature_dim(self):
        return self._model.feature_dim

    def close(self):
        self._ingest.close()
        self._model.close()
