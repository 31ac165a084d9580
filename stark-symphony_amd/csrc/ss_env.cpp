// ss_process_defaults (include/ss_verify.h, ABI 2.4): the process-wide HIP runtime setting this library's multi-stream
// submissions want, as an EXPLICIT call of the host program -- rounds 4-5 set it from a constructor when the library was
// loaded, i.e. a shared library changed the runtime of its whole host process (torch, RCCL, the caller's own streams)
// behind its back; nothing happens at load time any more.
//
// GPU_MAX_HW_QUEUES: the runtime multiplexes a process's streams onto this many hardware queues (default 4) and streams that
// share a queue serialise.  Measured at 4 / 8 / 12 / 16 / 24 / 32 (profiles/r06_hw_queues_sweep.txt): the metric batch
// (65 536 proofs per pass) does not care; one GPU's 8 192-proof share with the accept reduce on RCCL 3.36 / 3.45 / 3.44 /
// 3.16 / 3.44 / 3.44 M proofs/s; stark101 x 4 096 on 16 independent streams 44.7 / 45.7 / 50.6 / 51.0 / 61.1 / 61.6 M; what a
// foreign stream of the process waits while the 16 are busy is the same from 12 up (median 0.1-0.2 ms).  24 is the smallest
// value that holds every rate, so that is the default asked for; a value already in the environment wins.
#include <stdlib.h>

#include "../../include/ss_verify.h"

extern "C" int ss_process_defaults(void)
{
    if (!getenv("SS_KEEP_ENV")) setenv("GPU_MAX_HW_QUEUES", "24", /*overwrite*/ 0);
    const char *v = getenv("GPU_MAX_HW_QUEUES");
    return v ? atoi(v) : 0;
}
