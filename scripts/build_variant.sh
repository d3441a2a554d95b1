#!/bin/bash
# dev: build a variant library build/var/lib_<name>.so where ONE source is recompiled with extra flags and the other
# objects are taken from the product build (usage: scripts/build_variant.sh <name> <file.hip> "<extra flags>")
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; extra=$3
make -C mythos_amd/csrc -j8 >/dev/null
mkdir -p build/var/obj_$name
flags=$(make -s -C mythos_amd/csrc print-flags)
(cd mythos_amd/csrc && /opt/rocm/bin/hipcc $flags $extra -c $src -o ../../build/var/obj_$name/${src%.hip}.o)
objs=""
for o in build/csrc/*.o; do
  b=$(basename $o)
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs build/var/obj_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o build/var/lib_$name.so
echo "built build/var/lib_$name.so"
