"""The on-disk feature layout both sides of the reference agree on.

Writer: what ``writeFeatures`` produces (src/features_GPU_compute/calcSig_wOF.py:116-134) --
``<out>/<video>/<modelname>/<stream>_<blob>_features.csv``, one header line
``video =<v>, video url =<path>, CNN stream =<mode>, feature blob =<blob>, caffe model =<weights>`` and one row
per clip: ``int(clip_dir[-4:])`` then the 1024 values as ``str(numpy.float64)``, LF line ends, trailing newline.

What ``str(numpy.float64)`` prints depends on the numpy the script ran under, and the reference ships BOTH kinds:
``data/features/.../DowntownBrooklynDrive_480p`` holds shortest-round-trip reprs (numpy >= 1.14, up to 17 digits),
``SHRP2_Forward_clips_features/S06NDS_Sample_120406_1451_00186_Forward`` holds 12 significant digits (numpy < 1.14: ``'%.12g'`` plus ``'.0'``
for integral values).  ``number_format="repr"`` (default: lossless, so ``load_db`` gets the exact features) and
``"g12"`` reproduce one each byte for byte (tests/test_feature_files.py).
Reader: the parsing rules of ``load_db`` (src/api/api_load_records.py:41-58).
"""
from __future__ import annotations

import csv
import os
from typing import Dict, List, Sequence, Tuple

import numpy as np

STREAM_MODES = ("rgb", "warped_optical_flow")          # calcSig_wOF.py:120


def _g12(v: float) -> str:
    """``str(numpy.float64)`` of numpy < 1.14: 12 significant digits, integral values keep a ``.0``."""
    s = "%.12g" % v
    return s if ("." in s or "e" in s or "n" in s) else s + ".0"


NUMBER_FORMATS = {"repr": repr, "g12": _g12}


def format_rows(feat: np.ndarray, clip_numbers: np.ndarray, number_format: str = "repr") -> bytes:
    """``"<clip>,<v0>,...\\n"`` per row, every value printed like ``str(numpy.float64)`` -- by the library (a C loop that holds
    no interpreter lock: the writer thread of the command line runs beside the threads that feed the GPU).  ``NUMBER_FORMATS``
    above are the same two rules in Python; the tests hold one against the other."""
    import ctypes as C
    from .._lib import call
    if number_format not in NUMBER_FORMATS:
        raise KeyError(number_format)
    feat = np.ascontiguousarray(feat, dtype=np.float64)
    nos = np.ascontiguousarray(clip_numbers, dtype=np.int64)
    if feat.ndim != 2 or nos.shape != (feat.shape[0],):
        raise ValueError("feat must be [n, D] with one clip number per row")
    if feat.shape[0] == 0:
        return b""
    cap = feat.shape[0] * (feat.shape[1] * 26 + 22)
    buf = np.empty(cap, dtype=np.uint8)              # not cleared (a ctypes buffer is: 7 MB per 256 x 1024 file), copied once below
    n = C.c_int64()
    call("vq_format_feature_rows", feat.ctypes.data_as(C.c_void_p), feat.shape[0], feat.shape[1], nos.ctypes.data_as(C.c_void_p),
         0 if number_format == "repr" else 1, buf.ctypes.data_as(C.c_void_p), cap, C.byref(n))
    return buf[:n.value].tobytes()


def write_features(out_dir: str, video: str, video_path: str, modelname: str, blob: str, clip_names: Sequence[str],
                   features: Dict[str, np.ndarray], weights_files: Dict[str, str], number_format: str = "repr") -> List[str]:
    """features[mode] is [n_clips, D] float64 in clip order; returns the files written."""
    f_output_dir = os.path.join(out_dir, video, modelname)
    os.makedirs(f_output_dir, exist_ok=True)
    written = []
    for mode in STREAM_MODES:
        if mode not in features:
            continue
        header_txt = 'video =' + video + ', video url =' + video_path + ', CNN stream =' + mode \
                     + ', feature blob =' + blob + ', caffe model =' + weights_files[mode]
        outfile = os.path.join(f_output_dir, mode + "_" + blob + "_features.csv")
        feat = np.ascontiguousarray(features[mode], dtype=np.float64)
        clip_nos = np.asarray([int(vid[-4:]) for vid in clip_names], dtype=np.int64)          # calcSig_wOF.py:131
        with open(outfile, mode='wb') as fout:
            fout.write((header_txt + "\n").encode())
            fout.write(format_rows(feat, clip_nos, number_format))
        written.append(outfile)
    return written


def write_feature_file(out_dir: str, video: str, video_path: str, modelname: str, blob: str, mode: str, weights_file: str,
                       row_blocks: Sequence[bytes]) -> str:
    """One stream's file of one video from rows that are ALREADY formatted (``format_rows`` of consecutive clip ranges, in clip
    order): the same bytes as ``write_features`` on the whole block.  The command line formats a batch's rows while the next batch
    is on the GPU, so that a job's last feature files cost a header and a write."""
    f_output_dir = os.path.join(out_dir, video, modelname)
    os.makedirs(f_output_dir, exist_ok=True)
    header_txt = 'video =' + video + ', video url =' + video_path + ', CNN stream =' + mode + ', feature blob =' + blob + ', caffe model =' + weights_file
    outfile = os.path.join(f_output_dir, mode + "_" + blob + "_features.csv")
    with open(outfile, mode='wb') as fout:
        fout.write((header_txt + "\n").encode())
        for rows in row_blocks:
            fout.write(rows)
    return outfile


def read_features(csv_path: str) -> Tuple[dict, np.ndarray, np.ndarray]:
    """api_load_records.py:45-58: header fields via ``split('=')[-1]``; rows ``clip, f0, f1, ...``.
    Returns (header dict, clip numbers [n], features [n, D] float64)."""
    with open(csv_path, 'r') as f:
        reader = csv.reader(f)
        header = next(reader)
        meta = {"video": header[0].split('=')[-1], "dnn_stream": header[2].split('=')[-1],
                "feature_name": header[3].split('=')[-1], "dnn_weights_file_uri": header[4].split('=')[-1]}
        clips, rows = [], []
        for row in reader:
            clips.append(int(row[0]))
            rows.append([float(x) for x in row[1:]])
    return meta, np.asarray(clips, dtype=np.int64), np.asarray(rows, dtype=np.float64)


def read_split_dir(split_path: str):
    """One ``<video>/<split>`` directory -> (split number, {stream: (clips, features)}); the split number is the
    LAST CHARACTER of the directory name (api_load_records.py:43)."""
    nsplit = int(split_path.rstrip("/")[-1])
    out = {}
    for entry in sorted(os.scandir(split_path), key=lambda e: e.name):
        if entry.is_file() and entry.name.endswith('.csv') and not entry.name.startswith('.'):
            meta, clips, feats = read_features(entry.path)
            out[meta["dnn_stream"]] = (clips, feats, meta)
    return nsplit, out
