// xcd_flag_probe.hip -- does a spin loop on an agent-scope atomic LOAD see a flag that a workgroup on ANOTHER XCD publishes later?
// (the hand-over of the ELBO scan's last round relies on it: elbo_qf_kernel.hip).  Workgroup b runs on XCD b % 8.  Workgroup 0 waits
// `delay` sleeps, writes 64 payload doubles and releases the flag; workgroups 1 .. n-1 poll (bounded) and report the number of polls
// until they saw it, whether they timed out, and whether the payload they then read was the published one.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_flag_probe xcd_flag_probe.hip && ./xcd_flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Out { unsigned polls, timeout, bad; };
template <int MODE>   // 0: acquire load in the loop; 1: relaxed load in the loop + one acquire fence after; 2: fetch_add(0) in the loop
__global__ void probe(unsigned *flag, double *payload, Out *out, unsigned epoch, int delay, unsigned maxpoll) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0) {
        if (tid == 0) for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(64);
        __syncthreads();
        if (tid < 64) __hip_atomic_store(payload + tid, (double)epoch + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    __shared__ unsigned s_polls, s_to;
    if (tid == 0) {
        unsigned n = 0, seen = 0;
        while (n < maxpoll) {
            unsigned f;
            if (MODE == 0) f = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            else if (MODE == 1) f = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else f = __hip_atomic_fetch_add(flag, 0u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            ++n;
            if (f == epoch) { seen = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        if (MODE == 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);
        s_polls = n; s_to = seen ? 0u : 1u;
    }
    __syncthreads();
    unsigned bad = 0;
    if (tid < 64) {
        const double v = __hip_atomic_load(payload + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad = (v != (double)epoch + tid) ? 1u : 0u;
    }
    bad = __syncthreads_or((int)bad);
    if (tid == 0) { out[b].polls = s_polls; out[b].timeout = s_to; out[b].bad = s_to ? 0u : bad; }
}
int main() {
    const int n = 33;
    unsigned *flag; double *payload; Out *out;
    hipMalloc(&flag, 256); hipMalloc(&payload, 64 * 8); hipMalloc(&out, n * sizeof(Out));
    hipMemset(flag, 0, 256); hipMemset(payload, 0, 64 * 8);
    std::vector<Out> h(n);
    unsigned epoch = 0;
    for (int mode = 0; mode < 3; ++mode)
        for (int delay : {0, 50, 2000}) {
            unsigned to = 0, bad = 0; unsigned long long polls = 0; unsigned maxp = 0;
            for (int rep = 0; rep < 20; ++rep) {
                ++epoch;
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(n), dim3(256), 0, 0, flag, payload, out, epoch, delay, 2000000u);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(n), dim3(256), 0, 0, flag, payload, out, epoch, delay, 2000000u);
                else hipLaunchKernelGGL(probe<2>, dim3(n), dim3(256), 0, 0, flag, payload, out, epoch, delay, 2000000u);
                if (hipDeviceSynchronize() != hipSuccess) { printf("sync failed\n"); return 1; }
                hipMemcpy(h.data(), out, n * sizeof(Out), hipMemcpyDeviceToHost);
                for (int b = 1; b < n; ++b) { to += h[b].timeout; bad += h[b].bad; polls += h[b].polls; if (h[b].polls > maxp) maxp = h[b].polls; }
            }
            printf("mode %d (%s) delay %4d sleeps: timeouts %u, stale payloads %u of %d, polls mean %.1f max %u\n", mode,
                   mode == 0 ? "acquire load" : mode == 1 ? "relaxed load + fence" : "fetch_add 0", delay, to, bad, 20 * (n - 1),
                   (double)polls / (20 * (n - 1)), maxp);
        }
    return 0;
}
