#!/bin/bash
# Builds an experiment variant of the library: tools/build_variant.sh <name> <file.hip> <extra compiler flags...>
# recompiles one source with the flags and links it with the default build's other objects into flac_amd/lib/exp/<name>.so
set -e
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../flac_amd/csrc"
mkdir -p ../lib/exp
base=$(basename "$src" .hip)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I../../include -I. -Wall -Wno-unused-function "$@" -c "$src" -o ../lib/exp/$name-$base.o
objs=$(ls ../lib/obj/*.o | grep -v "/$base.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/exp/$name.so $objs ../lib/exp/$name-$base.o
rm -f ../lib/exp/$name-$base.o
echo built flac_amd/lib/exp/$name.so
