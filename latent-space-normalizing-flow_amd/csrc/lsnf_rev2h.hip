// lsnf_rev2h.hip -- the throughput reverse (sampling) pass (lsnf_rev3.hip, reference model.py:484-498 / :424-456) on the
// two-term fp16 split of lsnf_fwd2h.hip: three fp16 MFMAs per product instead of six bf16 ones, fp16's range guarded by
// the NaN probe after every GEMM stage and the bf16x3 kernel queued behind as the fix-up pass (same protocol, same
// guard words).  The kernel source is lsnf_rev3.hip, compiled here with the other split (lsnf_l16.h, LSNF_L16_PARTS).
#define LSNF_L16_PARTS 2
#define LSNF_REV3_ENTRY lsnf_launch_reverse2h
#define lsnf_rev3_kernel lsnf_rev2h_kernel
#include "lsnf_rev3.hip"
