/*
 * ss_oracle_batch.c -- batch drivers over the oracle, used only by bench.py's
 * cpu_baseline leg and by tests.  TEST INFRASTRUCTURE ONLY (see ss_oracle.h).
 * OpenMP over independent proofs; `threads` <= 0 means "all cores".
 */
#include "ss_oracle.h"
#include <omp.h>

int so_num_procs(void) { return omp_get_num_procs(); }

void so_s101_verify_batch(const so_s101_proof *proofs, size_t n, uint32_t *status, int threads)
{
    if (threads <= 0) threads = omp_get_num_procs();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < (long)n; i++) status[i] = so_s101_verify(&proofs[i], 0);
}

void so_stwo_verify_batch(const so_stwo_cfg *cfg, const so_stwo_proof *proofs, size_t n, int mode,
                          uint32_t *status, int threads)
{
    if (threads <= 0) threads = omp_get_num_procs();
#pragma omp parallel for num_threads(threads) schedule(static)
    for (long i = 0; i < (long)n; i++) status[i] = so_stwo_verify(cfg, &proofs[i], mode, 0);
}
