// A one-lane kernel that samples the shader clock while other kernels run (tools only):
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/clock_monitor.hip -o build/libclock_monitor.so
// s_memtime counts shader cycles, s_memrealtime a constant 100 MHz (tools/clock_probe.hip); sample i = (realtime, cycles).
// The loop is bounded by n samples of `period` ticks each: the wave always ends by itself.
#include <hip/hip_runtime.h>

__global__ void __launch_bounds__(64) clock_monitor_kernel(unsigned long long *samples, unsigned n, unsigned period)
{
    if (threadIdx.x || blockIdx.x) return;
    for (unsigned i = 0; i < n; i++) {
        const unsigned long long r = __builtin_amdgcn_s_memrealtime(), c = __builtin_amdgcn_s_memtime();
        samples[2 * i] = r;
        samples[2 * i + 1] = c;
        while (__builtin_amdgcn_s_memrealtime() - r < period) __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" int cm_launch(unsigned long long *samples_dev, unsigned n, unsigned period_ticks, void *stream)
{
    hipLaunchKernelGGL(clock_monitor_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, samples_dev, n, period_ticks);
    return (int)hipGetLastError();
}
