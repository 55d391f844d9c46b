#!/bin/bash
# Timing-only builds of the stash-writing pipelined forward (wrong stash): which part of the stash costs what.
#   tools/ablate_stash.sh   (here: builds latent-space-normalizing-flow_amd/_ablate/s_<parts>.so; parts: 1 masks, 2 sigma tiles, 4 rows)
#   on the GPU box: for p in 0 1 2 4 7; do LSNF_LIB_PATH=latent-space-normalizing-flow_amd/_ablate/s_$p.so python tools/ablate_stash.py; done
set -e
cd "$(dirname "$0")/../latent-space-normalizing-flow_amd/csrc"
make -j8 >/dev/null
mkdir -p ../_ablate _build_abl
OBJS=$(ls _build/*.o | grep -v lsnf_fwd3p.o)
for v in ${PARTS:-0 1 2 4 7}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -DLSNF_STASH_PARTS=$v $EXTRA -c lsnf_fwd3p.hip -o _build_abl/fwd3p_s$v.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS _build_abl/fwd3p_s$v.o -o ../_ablate/s_$v.so
  echo built s_$v.so
done
