// lsnf_bwd3w.hip -- the f_width-128 instantiation of the bf16x3 throughput backward (lsnf_bwd3.hip), as its own translation
// unit so that it can be compiled without packed fp32 math (Makefile: -fno-slp-vectorize; the measurement is quoted in lsnf_bwd3.hip).
#define LSNF_BWD3_WIDE_TU
#include "lsnf_bwd3.hip"
