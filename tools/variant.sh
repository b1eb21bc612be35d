#!/bin/bash
# Variant build for timing experiments: tools/variant.sh <name> <source.hip> <extra flags...>  ->  var/libwam_<name>.so
# (the other objects come from the regular build; run with WAM_LIB_PATH=var/libwam_<name>.so)
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/web-audio-mastering_amd/csrc
mkdir -p $R/var/obj
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable "$@" -c $C/$SRC -o $R/var/obj/$NAME.o
OBJS=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/var/libwam_$NAME.so $OBJS $R/var/obj/$NAME.o
echo built var/libwam_$NAME.so
