#!/bin/bash
# An A/B build of the encode-path kernels (toolame_hip.hip): tools/variant_hip.sh NAME "-DTL_P2_NS=0 ..."  -> build/lib_NAME.so
# (that translation unit recompiled with the flags, every other object the product's: build/obj)
set -eu
R=${GRAFT_REPO_ROOT:-$PWD}; N=$1; F=${2:-}; O=$R/build/obj_h_$N
make -s -C $R/odr-audioenc_amd/csrc > /dev/null 2>&1
mkdir -p $O; cp $R/build/obj/*.o $O/; rm -f $O/toolame_hip.o
make -s -C $R/odr-audioenc_amd/csrc OBJ=$O EXTRA="$F -Wno-pass-failed" $O/toolame_hip.o > /dev/null 2>&1
cd $R/odr-audioenc_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -pthread -Wl,--version-script=exports.map -o $R/build/lib_$N.so $O/*.o
python3 $R/tools/check_isa.py $R/build/lib_$N.so --no-fail | grep frame_kernelILi1ELb0ELi2
