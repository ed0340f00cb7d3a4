/* ORACLE (test infrastructure): the double-precision instantiation of dcn_ref.c (dcn_forward_ref_f64,
 * dcn_backward_ref_f64), checker of the operator's fp64 entry points (cdfo_dcn_forward_dt / _backward_dt). */
#define REAL double
#define NAME(x) x##_f64
#define FLOOR_ floor
#include "dcn_ref.c"
