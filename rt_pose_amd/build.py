"""Build librtp_hip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

    python -m rt_pose_amd.build            # incremental
    python -m rt_pose_amd.build --force

hipcc cross-compiles without a GPU, so this also runs in the CPU-only authoring container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(HERE, "lib", "librtp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-Wno-int-to-pointer-cast",
         # no SLP vectoriser: it packs the epilogues' fp32 adds / multiplies into v_pk_add_f32 / v_pk_fma_f32, which cost more issue
         # time beside the matrix pipe than the scalar pairs they replace (fused data gradient of conv_tiled: 77.6 -> 72.8 us)
         "-fno-slp-vectorize"] + os.environ.get("RTP_HIPCC_EXTRA", "").split()
LIB = os.environ.get("RTP_BUILD_LIB", LIB)   # A/B builds: another output library (load it with RTP_LIB)
OBJ = os.environ.get("RTP_BUILD_OBJ", OBJ)


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "rtp.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src, force, hdr_m):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    srcp = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(srcp), hdr_m):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    hdr_m = _deps_mtime()
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: _compile(s, force, hdr_m), sources()))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
