#!/bin/bash
# Builds the host-side file parsers with AddressSanitizer + UBSan and runs tools/fuzz_host.cpp on them (CPU only).
#   tools/fuzz_host.sh <iterations> <seed> <base .flac files...>
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
out=${TMPDIR:-/tmp}/flac_amd_fuzz
mkdir -p "$out"
g++ -std=c++17 -g -O1 -fsanitize=address,undefined -fno-sanitize=enum -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I"$here/include" -I"$here/flac_amd/csrc" "$here/tools/fuzz_host.cpp" \
    "$here"/flac_amd/csrc/{metadata_chain,metadata_object,stream_common,stream_decoder,pcm_io,md5}.cpp -o "$out/fuzz_host"
iters=$1; seed=$2; shift 2
"$out/fuzz_host" "$iters" "$seed" "$out" "$@"
