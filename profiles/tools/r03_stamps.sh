#!/bin/bash
# per-stage shader-clock stamps of k_cbf_rollout on the three C4 scenes.  The stamps are compiled out of the product: this needs
#   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-pass-failed -fPIC -shared -DMDS_ROLL_STAMPS=1 \
#         -o multidronesim_amd/libmds_stamps.so multidronesim_amd/csrc/mds_api.hip -Wl,-rpath,/opt/rocm/lib
# (round 4: bash profiles/tools/mkvariant.sh stamps -DMDS_ROLL_STAMPS=1  ->  abl/libmds_stamps.so)
# (MDS_TUNE_ROLL_STAMPS=1 then prints them and synchronises the stream).  bash profiles/tools/r03_stamps.sh [steps per launch] [scenes]
mkdir -p gpurun_out/r3c
for sc in ${2:-far under level}; do
  echo "== scene $sc"
  MDS_LIB_PATH=$PWD/${MDS_STAMPS_LIB:-abl/libmds_stamps.so} MDS_TUNE_ROLL_STAMPS=1 timeout -k 10 300 python bench.py --workload c4 --c4-scene $sc --no-cpu-baseline --no-extras --fused-rollout ${1:-50} --steps 200 --warmup 50 2>&1 >/dev/null | grep "roll stamps" | tail -11
done
