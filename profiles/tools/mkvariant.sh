#!/bin/bash
# build an A/B variant of libmds.so: mkvariant.sh <name> [-Dflags for part 2 ...]  ->  abl/libmds_<name>.so  (part 1 is reused from build/)
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-pass-failed -fPIC -DMDS_PART=2 "$@" -c -o build/mds_part2_$name.o multidronesim_amd/csrc/mds_api.hip && \
hipcc --offload-arch=gfx950 -fPIC -shared -o abl/libmds_$name.so build/mds_part1.o build/mds_part2_$name.o -Wl,-rpath,/opt/rocm/lib && echo "built abl/libmds_$name.so"
