#!/bin/bash
# Timing experiments on conv_w4: builds libpvr_hip_knock<k>.so variants (conv_w4.hip with -DW4_KNOCK=k: pieces of the step removed,
# results wrong, timing only) next to the real library.  Run scripts/w4_stamps.py with PVR_LIB pointing at each on the GPU box.
set -e
cd "$(dirname "$0")/../pvr_habitat_amd/csrc"
make -s
for k in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DW4_KNOCK=$k -c conv_w4.hip -o build/conv_w4_knock$k.oo &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libpvr_hip_knock$k.so $(ls build/*.o | grep -v conv_w4.o) build/conv_w4_knock$k.oo ) &
done
wait
ls -la ../lib/
