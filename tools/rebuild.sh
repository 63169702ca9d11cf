#!/bin/bash
# developer: recompile the named translation units of gsdf_amd/csrc (default: all that changed is up to you) and relink libgsdfhip.so
#   bash tools/rebuild.sh abi_mesh.hip [abi_eval.hip ...] [-DFLAG ...]
cd "$(dirname "$0")/../gsdf_amd/csrc" || exit 1
python gen_embedded.py embedded_src.inc
FLAGS="-O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I../../include"
UNITS=(); DEFS=()
for a in "$@"; do case "$a" in -D*) DEFS+=("$a");; *) UNITS+=("$a");; esac; done
pids=()
for u in "${UNITS[@]}"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "${DEFS[@]}" -c "$u" -o "${u%.*}.o" & pids+=($!); done
for p in "${pids[@]}"; do wait $p || exit 1; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC abi_eval.o abi_mesh.o abi_comm.o abi_host.o compile.o specialize.o -lhiprtc -ldl -o ${OUT:-libgsdfhip.so} && echo "linked ${OUT:-libgsdfhip.so}"
