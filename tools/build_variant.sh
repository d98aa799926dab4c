#!/bin/bash
# A/B builds of one kernel file: tools/build_variant.sh <name> <file.hip | git-rev:file.hip> [extra hipcc flags...]
#   -> tools/bin/lib_<name>.so  (the other objects are the in-tree build's; run with STRQ_LIB=tools/bin/lib_<name>.so)
set -eu
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/tools/bin/obj_$name
base=$(basename ${src#*:})
if [[ "$src" == *:* ]]; then
  rev=${src%%:*}; mkdir -p $root/tools/bin/src_$name
  cp $root/strique_amd/csrc/*.h $root/tools/bin/src_$name/
  git -C $root show $rev:strique_amd/csrc/$base > $root/tools/bin/src_$name/$base
  file=$root/tools/bin/src_$name/$base; inc="-I$root/include"
else
  file=$root/strique_amd/csrc/$base; inc=""
fi
extra=""
[[ "$base" == viterbi_kernels.hip && -z "${NOILP:-}" ]] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"      # NOILP=1: the default scheduler
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fno-fast-math -Wall -Wno-unused-function $extra $inc "$@" -c $file -o $root/tools/bin/obj_$name/${base%.hip}.o
objs=""
for o in $root/strique_amd/lib/obj/*.o; do
  if [[ "$(basename $o)" == "${base%.hip}.o" ]]; then objs="$objs $root/tools/bin/obj_$name/${base%.hip}.o"; else objs="$objs $o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $root/tools/bin/lib_$name.so $objs -lz
echo $root/tools/bin/lib_$name.so
