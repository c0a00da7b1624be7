/* kmdiff_hip_test.h -- test hooks of libkmdiff_hip.so.  NOT part of the interface a host binds (kmdiff_hip.h):
 * they expose, on the host, the correctly rounded log / exp (kmdiff_amd/csrc/kmd_ddmath.h) and Cephes igamc(1/2, x)
 * over them, which decide the rows whose p-value lies within 1e-8 of the threshold (KMD_CNT_NEAR_THRESHOLD) -- so that
 * tests/test_rounded_math.py and tests/test_gpu_threshold.py can hold them to mpmath and to the reference's alglib. */
#ifndef KMDIFF_HIP_TEST_H
#define KMDIFF_HIP_TEST_H
#include "kmdiff_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
double kmd_test_log_rounded(double x);
double kmd_test_exp_rounded(double x);
double kmd_test_igamc_half_rounded(double x);
double kmd_test_row_pvalue_rounded(const kmd_model* m, uint64_t sum_control, uint64_t sum_case);
/* the reference's log-factorial running sum (log_factorial_table.cpp:13-22) of each d_k[i] < 2^20 as the device repeats
 * it: term by term (d_plain, may be NULL) and through the table of logarithms + the binade-wise exact integer sums of
 * kmd_pvalues_refine (d_fast).  Device pointers; asynchronous on `stream`. */
int kmd_test_running_sums(const uint64_t* d_k, size_t n, double* d_plain, double* d_fast, void* stream);
#ifdef __cplusplus
}
#endif
#endif
