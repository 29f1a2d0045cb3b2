#!/bin/bash
# usage: build_variant.sh NAME "EXTRA FLAGS" file.hip [file.hip ...]  ->  build/variants/libidocp_hip_NAME.so  (the other objects come from build/obj)
set -e
cd /root/repo
name=$1; flags=$2; shift 2
mkdir -p build/variants/$name
objs=""
for o in build/obj/*.o; do
  base=$(basename $o .o)
  skip=0; for f in "$@"; do [ "$(basename $f)" = "$base" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $o"
done
for f in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Iinclude -Iidocp_amd/csrc $flags -x hip -c $f -o build/variants/$name/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libidocp_hip_$name.so $objs build/variants/$name/*.o
ls -la build/variants/libidocp_hip_$name.so
