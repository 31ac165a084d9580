// Process defaults the library sets when it is LOADED -- before the HIP runtime initialises, which it does lazily at the
// first HIP call of the process, not when libamdhip64.so is mapped.
//
// GPU_MAX_HW_QUEUES: the runtime multiplexes a process's streams onto this many hardware queues (default 4), and streams
// that share a queue serialise.  The pipelines of this library (HEAD kernels beside TAIL kernels, 16 passes of small
// stark101 batches in flight: stark-symphony_amd/verifier.py Pipeline / IndependentStreams, csrc/ss_ingest_dev.hip) exist
// for the overlap that loses (stark101 x 4 096: 44.7 M -> 60.6 M proofs/s; an 8 192-proof stwo share: 2.55 -> 2.38 ms,
// profiles/r04_accept_reduce.txt).  A C / Rust caller that links the library gets the setting without knowing about it; a
// value the caller has put into the environment wins, SS_KEEP_ENV=1 switches this off, and a process that has already
// initialised HIP before loading the library (Python with a warm torch) is not affected either way.
#include <stdlib.h>

__attribute__((constructor)) static void ss_env_defaults()
{
    if (getenv("SS_KEEP_ENV")) return;
    setenv("GPU_MAX_HW_QUEUES", "24", /*overwrite*/ 0);
}
