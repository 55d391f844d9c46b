#!/bin/bash
# Builds timing-only variants of the library (each removes one cost of the split-bf16 forward) into
# latent-space-normalizing-flow_a../_ablate/ (git-ignored, shipped to the GPU box); run tools/ablate_fwd3.py there.
set -e
cd "$(dirname "$0")/../latent-space-normalizing-flow_amd/csrc"
mkdir -p ../_ablate
for v in "base:" "nosplit:-DLSNF_ABLATE_SPLIT" "nomfma:-DLSNF_ABLATE_MFMA" "noio:-DLSNF_ABLATE_IO" "nodma:-DLSNF_ABLATE_DMA" \
         "nobar:-DLSNF_ABLATE_BARRIER" "noepi:-DLSNF_ABLATE_EPILOGUE" "nosplit_noepi:-DLSNF_ABLATE_SPLIT -DLSNF_ABLATE_EPILOGUE" \
         "mfmaonly:-DLSNF_ABLATE_SPLIT -DLSNF_ABLATE_EPILOGUE -DLSNF_ABLATE_IO -DLSNF_ABLATE_DMA -DLSNF_ABLATE_BARRIER"; do
  name=${v%%:*}; flags=${v#*:}
  make -j8 BUILD=_build_f3_$name OUT=../_ablate/f3_$name.so EXTRA="$flags" > /dev/null
  echo built $name
done
