// Does ss_process_defaults() (csrc/ss_env.cpp; an explicit call since ABI 2.4, a constructor before) reach the HIP runtime of a
// program that LINKS the library (the C / Rust caller of INTEGRATION.md 3)?  16 stark101 passes of 4 096 proofs in flight, each
// on its own stream, through the C ABI; run by tools/evidence.sh: the program calls ss_process_defaults() first (24), the same
// with SS_KEEP_ENV=1 or with `nodefaults` as second argument (the runtime's default, 4), GPU_MAX_HW_QUEUES set by the caller.
//   hipcc -O2 -Iinclude tools/probes/hw_queues_probe.hip -o build/hw_queues_probe -Lstark-symphony_amd -lss_verify -Wl,-rpath,$PWD/stark-symphony_amd
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "ss_verify.h"

#define CHECK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define OK(x) do { int rc_ = (x); if (rc_ != SS_OK) { fprintf(stderr, "libss_verify: %s (%d) at line %d\n", ss_last_error(), rc_, __LINE__); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: hw_queues_probe tests/golden/stark101_proof.json [nodefaults]\n"); return 2; }
    if (argc < 3) ss_process_defaults();  // before the process's first HIP call
    std::ifstream f(argv[1]);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    ss_s101_shape sh = {10, 13};
    std::vector<uint32_t> rec(ss_s101_record_words(&sh));
    if (ss_s101_parse(text.data(), text.size(), SS_TEXT_JSON, &sh, rec.data()) != 0) { fprintf(stderr, "not a stark101 proof\n"); return 2; }
    sh.max_layers = 10; sh.max_path = 13;
    const size_t n = 4096, slots = 16;
    std::vector<const uint32_t *> ptrs(n, rec.data());
    std::vector<uint32_t> batch(ss_s101_batch_words(&sh, n));
    OK(ss_s101_pack(&sh, n, ptrs.data(), batch.data()));
    ss_ctx *ctx = nullptr;
    OK(ss_ctx_create(0, &ctx));
    const size_t wsb = ss_s101_workspace_bytes(&sh, n);
    uint32_t *batch_dev;
    CHECK(hipMalloc(&batch_dev, batch.size() * 4));
    CHECK(hipMemcpy(batch_dev, batch.data(), batch.size() * 4, hipMemcpyHostToDevice));
    std::vector<hipStream_t> st(slots);
    std::vector<void *> ws(slots);
    std::vector<uint32_t *> status(slots);
    for (size_t k = 0; k < slots; k++) {
        CHECK(hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking));
        CHECK(hipMalloc(&ws[k], wsb));
        CHECK(hipMalloc(&status[k], n * 4));
    }
    auto pass = [&](size_t steps) -> int {
        for (size_t i = 0; i < steps; i++) {
            const size_t k = i % slots;
            OK(ss_s101_verify_batch_dev(ctx, &sh, n, batch_dev, ws[k], wsb, status[k], nullptr, st[k]));
        }
        for (size_t k = 0; k < slots; k++) CHECK(hipStreamSynchronize(st[k]));
        return 0;
    };
    if (pass(64)) return 2;
    std::vector<uint32_t> host(n);
    CHECK(hipMemcpy(host.data(), status[0], n * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        if (host[i]) { fprintf(stderr, "proof %zu rejected\n", i); return 1; }
    const size_t steps = 1920;
    const auto t0 = std::chrono::steady_clock::now();
    if (pass(steps)) return 2;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const char *e = getenv("GPU_MAX_HW_QUEUES");
    printf("%.1f M proofs/s   (GPU_MAX_HW_QUEUES in the environment now: %s)\n", steps * n / dt / 1e6, e ? e : "unset");
    return 0;
}
