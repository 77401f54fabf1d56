#!/bin/bash
# Build variant libraries of libpvr_hip.so that differ in chain_wave128.hip's compile-time knobs (timing knock-outs give WRONG results: timing only).
# usage: scripts/r06_cw8_variants.sh name "-DCW8_KNOCK=8" [name2 "flags2" ...]   ->  pvr_habitat_amd/lib/libpvr_hip_<name>.so
#        SRC=stem scripts/r06_cw8_variants.sh stamp "-DSTEM_STAMP"                  (another source file of csrc/ instead of chain_wave128.hip)
set -e
cd "$(dirname "$0")/../pvr_habitat_amd/csrc"
make -j8 >/dev/null
SRC=${SRC:-chain_wave128}
OTHERS=$(ls build/*.o | grep -v $SRC.o)
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result $flags -c $SRC.hip -o build/cw8_$name.obj -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "ScratchSize" | sort | uniq -c | head -3
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpvr_hip_$name.so $OTHERS build/cw8_$name.obj
  echo built $name
done
