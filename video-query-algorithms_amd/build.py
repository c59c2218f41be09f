"""Build libvqamd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python video-query-algorithms_amd/build.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SOURCES = ["csrc/vq_sim.hip", "csrc/vq_tsn.hip", "csrc/vq_wino.hip", "csrc/vq_boot.hip", "csrc/vq_frames.hip", "csrc/vq_comm.hip", "csrc/vq_flow.hip",
           "csrc/vq_jpeg.hip"]
# host-only translation units (no HIP include): plain C++, also built with sanitizers by tests/sanitize/Makefile
HOST_SOURCES = ["csrc/host/vq_csv.cc", "csrc/host/vq_jpeg_host.cc", "csrc/host/vq_corners.cc", "csrc/host/vq_block_pool.cc"]
HEADERS = ["csrc/vq_common.h", "csrc/vq_tsn_kernels.h", "csrc/host/vq_host.h", "csrc/host/vq_jpeg_host.h", "csrc/host/vq_corners.h", "csrc/host/vq_block_pool.h", "../include/vq_amd.h"]
OUT = os.path.join(HERE, "libvqamd.so")
# -ffp-contract=off: score arithmetic must round like the reference's numpy scalars; FMAs are explicit
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in architectural VGPRs.  Left to itself the register allocator parks part of
# a large tile's accumulators in AGPRs and copies them in and out on EVERY K step (128 v_accvgpr_read/write per 64 MFMAs in the
# 128x64 tile's loop) -- VALU work that comes straight out of the fp32 matrix pipe's time.
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-Wall", "-Wno-unused-result",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]
FLAGS += os.environ.get("VQ_EXTRA_HIPCC_FLAGS", "").split()     # experiments (e.g. -DVQ_WINO_SKEW=1); empty for the product


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, p)) > t for p in SOURCES + HOST_SOURCES + HEADERS + ["build.py"])


def build(force=False, verbose=True):
    if not force and not _stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [hipcc, "-c"] + FLAGS + [os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    host_flags = ["-O3", "-std=c++17", "-fPIC", "-Wall", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
                  "-I" + os.path.join(HERE, "csrc", "host")]
    for src in HOST_SOURCES:
        obj = os.path.join(HERE, "build", os.path.basename(src) + ".o")
        objs.append(obj)
        cmd = [os.environ.get("CXX", "g++"), "-c"] + host_flags + [os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("compilation failed")
    cmd = [hipcc, "-shared", "--offload-arch=gfx950", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd + ["-ldl", "-lpthread"])
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
