// Host worker threads of the library: one process-wide pool, started on first use, as many threads as the
// cores the process may really use (scheduler affinity capped by the cgroup quota).  The text entry points
// stage and (for non-canonical texts) parse on it chunk after chunk, so threads are not re-created per chunk.
#pragma once
#include <stddef.h>

#include <functional>

namespace ss {

unsigned effective_cpus();  // scheduler affinity capped by the cgroup CPU quota (ss_ingest.cpp)

// f(i) for every i in [0, n), items handed out one at a time to at most `max_threads` threads (the
// caller is one of them).  Calls from several threads are served one after the other.
void parallel_for(size_t n, const std::function<void(size_t)> &f, size_t max_threads = 16);

}  // namespace ss
