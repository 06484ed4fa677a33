#!/bin/bash
# tools/tiled_variant.sh <suffix> <extra hipcc flags...>: rt_pose_amd/lib/librtp_hip_<suffix>.so = the product library with
# conv_tiled.hip recompiled with the given -D switches (RTP_TILED_PROF, RTP_TILED_TY=2, RTP_EXP_*).  Use with RTP_LIB=...
set -e
cd "$(dirname "$0")/.."
suf=$1; shift
python -m rt_pose_amd.build > /dev/null
O=rt_pose_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -Wno-int-to-pointer-cast -fno-slp-vectorize "$@" -Iinclude -c rt_pose_amd/csrc/conv_tiled.hip -o $O/conv_tiled_$suf.o_ 2>&1 | grep -E "error" || true
objs=$(ls $O/*.o | grep -v conv_tiled.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rt_pose_amd/lib/librtp_hip_$suf.so $objs $O/conv_tiled_$suf.o_
echo built rt_pose_amd/lib/librtp_hip_$suf.so
