#!/bin/bash
# Timing-only builds of the bf16x3 batch contraction (lsnf_params3.hip; wrong gradients): X3_ABL bits 1 no atomics, 2 no MFMA,
# 4 no global loads after the first two stages, 8 no split.  GPU box: for v in 0 1 2 4 8 ...; do LSNF_LIB_PATH=.../_ablate/x3_$v.so python tools/ablate_x3.py; done
set -e
cd "$(dirname "$0")/../latent-space-normalizing-flow_amd/csrc"
make -j8 >/dev/null
mkdir -p ../_ablate _build_abl
OBJS=$(ls _build/*.o | grep -v lsnf_params3.o)
for v in ${ABLS:-0 1 2 4 8 6 14 15}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function -DX3_ABL=$v $EXTRA -c lsnf_params3.hip -o _build_abl/params3_$v.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS _build_abl/params3_$v.o -o ../_ablate/x3_$v.so
  echo built x3_$v.so
done
