#!/bin/bash
for rep in 1 2; do
for cfg in 0,0 1,2 1,4 1,8 1,16 1,32 2,4 2,16 3,4 3,16; do
  LSNF_STAGGER=$cfg python tools/run_fwd.py 60 | sed "s/^/stagger=$cfg /"
done; done
