// lsnf_fwd2h.hip -- the throughput forward (lsnf_fwd3.hip's 16x16x32 "L16" kernel, reference model.py:474-483 + train.py:317-319)
// with the operands split into TWO fp16 terms instead of three bf16 terms (LSNF_MATH_FP16X2):
//   w = w1 + w2, x = x1 + x2 (each the round-to-nearest fp16 of what the previous one left: 11 + 11 significand bits),
//   w*x ~= w1*x1 + w1*x2 + w2*x1  on v_mfma_f32_16x16x32_f16 with fp32 accumulation;
// the dropped w2*x2 and the split's own remainder are <= 2^-22 |w||x| each -- below the accumulated fp32 rounding of a
// K >= 32 dot product (tools/study/split_bf16_accuracy.py: log-prob error vs float64 equal to the fp32 kernel's).
// Half the MFMAs and 2 instead of 4.5 VALU per split element of the bf16x3 kernel; the price is fp16's RANGE: an
// activation with |x| >= 65520 becomes inf (bf16 has fp32's exponent) -- which is why bf16x3 stays the default.
// The kernel source is lsnf_fwd3.hip, compiled here with the other split (lsnf_l16.h, LSNF_L16_PARTS).
#define LSNF_L16_PARTS 2
#define LSNF_FWD3_ENTRY lsnf_launch_forward2h
#define lsnf_fwd3b_kernel lsnf_fwd2h_kernel
#include "lsnf_fwd3.hip"
