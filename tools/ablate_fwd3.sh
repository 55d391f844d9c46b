#!/bin/bash
# Builds timing-only variants of the library (each removes one cost of the split-bf16 forward) into
# latent-space-normalizing-flow_a../_ablate/ (git-ignored, shipped to the GPU box); run tools/ablate_fwd3.py there.
set -e
cd "$(dirname "$0")/../latent-space-normalizing-flow_amd/csrc"
mkdir -p ../_ablate
if [ $# -eq 0 ]; then
  set -- "base:" "nosplit:-DLSNF_ABLATE_SPLIT" "noio:-DLSNF_ABLATE_IO" "noepi:-DLSNF_ABLATE_EPILOGUE" \
         "mfmaonly:-DLSNF_ABLATE_SPLIT -DLSNF_ABLATE_EPILOGUE -DLSNF_ABLATE_IO -DLSNF_ABLATE_DMA -DLSNF_ABLATE_BARRIER"
fi
for v in "$@"; do
  name=${v%%:*}; flags=${v#*:}
  make -j8 BUILD=_build_f3_$name OUT=../_ablate/f3_$name.so EXTRA="$flags" > /dev/null
  echo built $name
done
