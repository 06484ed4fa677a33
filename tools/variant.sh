#!/bin/bash
# tools/variant.sh <source stem> <suffix> <extra hipcc flags...>: rt_pose_amd/lib/librtp_hip_<suffix>.so = the product library with
# csrc/<stem>.hip recompiled with the given switches (-DRTP_WGT_PROF, ...).  Use with RTP_LIB=...
set -e
cd "$(dirname "$0")/.."
stem=$1; suf=$2; shift 2
python -m rt_pose_amd.build > /dev/null
O=rt_pose_amd/lib/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-inline-asm -Wno-int-to-pointer-cast -fno-slp-vectorize "$@" -Iinclude -c rt_pose_amd/csrc/$stem.hip -o $O/${stem}_$suf.o_ 2>&1 | grep -E "error" || true
objs=$(ls $O/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rt_pose_amd/lib/librtp_hip_$suf.so $objs $O/${stem}_$suf.o_
echo built rt_pose_amd/lib/librtp_hip_$suf.so
