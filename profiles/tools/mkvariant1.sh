#!/bin/bash
# build an A/B variant of libmds.so from PART 1 (the step / rollout kernels): mkvariant1.sh <name> [-Dflags ...]  ->  abl/libmds_<name>.so  (part 2 reused from build/)
name=$1; shift
mkdir -p abl
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-pass-failed -fPIC -DMDS_PART=1 "$@" -c -o build/mds_part1_$name.o multidronesim_amd/csrc/mds_api.hip && \
hipcc --offload-arch=gfx950 -fPIC -shared -o abl/libmds_$name.so build/mds_part1_$name.o build/mds_part2.o -Wl,-rpath,/opt/rocm/lib && echo "built abl/libmds_$name.so"
