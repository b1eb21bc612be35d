#!/bin/bash
# Builds timing variants of libwam_hip.so: tools/variants.sh <name> <file.hip> <flags...>  ->  var/<name>.so
# (one translation unit recompiled with extra -D flags, the rest taken from csrc/build)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; file=$2; shift 2
mkdir -p $R/var /tmp/var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $R/web-audio-mastering_amd/csrc/$file.hip -o /tmp/var_$name/$file.o
objs=""
for o in $R/web-audio-mastering_amd/csrc/build/*.o; do b=$(basename $o); if [ "$b" = "$file.o" ]; then objs="$objs /tmp/var_$name/$file.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/var/$name.so $objs
echo built $name
