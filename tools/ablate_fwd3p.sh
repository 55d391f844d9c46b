#!/bin/bash
# Timing-only ablation builds of lsnf_fwd3p.hip (wrong numbers, some racy): which part of the kernel costs what.
#   tools/ablate_fwd3p.sh            (here: builds latent-space-normalizing-flow_amd/_ablate/p_*.so)
#   python tools/ablate_fwd3p.py     (GPU box: times each build at B = 65 536)
set -e
cd "$(dirname "$0")/../latent-space-normalizing-flow_amd/csrc"
make -j8 >/dev/null
mkdir -p ../_ablate _build_abl
OBJS=$(ls _build/*.o | grep -v lsnf_fwd3p.o)
for v in BASE NOFILL NOMFMA NOSYNC NODMA NOSTAGGER "NOFILL -DLSNF_ABL_NOSYNC" "NOSTAGGER -DLSNF_ABL_NOFILL"; do
  name=$(echo $v | sed 's/ -DLSNF_ABL_/_/g')
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DLSNF_ABL_$v -c lsnf_fwd3p.hip -o _build_abl/fwd3p_$name.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS _build_abl/fwd3p_$name.o -o ../_ablate/p_$name.so
  echo built p_$name.so
done
