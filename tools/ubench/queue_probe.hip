// Which streams of a process get in each other's way?  Two probes per ordered pair (X, Y) of {null stream, three high-priority
// side streams}, optionally with unused streams created in between (they take hardware queue ids):
//   run:  a 300 us kernel on X, a 1 us kernel on Y right behind it -- how long until Y's kernel is done?  (shared queue: 300 us)
//   wait: Y waits (hipStreamWaitEvent) for an event that a helper stream records behind a 300 us kernel; a 1 us kernel on X --
//         how long until X's kernel is done?  (a queue whose packet processor sits on Y's barrier: 300 us)
// Build: hipcc --offload-arch=gfx950 -O2 -o queue_probe queue_probe.hip      Run: ./queue_probe [unused streams before aux] [.. aux2] [.. aux3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
__global__ void spin(long long ticks, int *out) {      // s_memrealtime: 100 MHz
    const long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    if (out) *out = 1;
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double probe_run(hipStream_t x, hipStream_t y, int *buf) {
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, x, 30000LL, buf);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, y, 100LL, buf + 1);
    const double t0 = now_us();
    (void)hipStreamSynchronize(y);
    return now_us() - t0;
}
static double probe_wait(hipStream_t x, hipStream_t y, hipStream_t helper, hipEvent_t late, int *buf) {
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, helper, 30000LL, buf);
    (void)hipEventRecord(late, helper);
    (void)hipStreamWaitEvent(y, late, 0);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, y, 100LL, buf + 2);
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, x, 100LL, buf + 1);
    const double t0 = now_us();
    (void)hipStreamSynchronize(x);
    return now_us() - t0;
}
int main(int argc, char **argv) {
    const int da = argc > 1 ? atoi(argv[1]) : 0, db = argc > 2 ? atoi(argv[2]) : 0, dc = argc > 3 ? atoi(argv[3]) : 0;
    int *buf; (void)hipMalloc(&buf, 64);
    int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t d, helper, s[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < da; i++) (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking);
    (void)hipStreamCreateWithPriority(&s[1], hipStreamNonBlocking, hi);
    for (int i = 0; i < db; i++) (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking);
    (void)hipStreamCreateWithPriority(&s[2], hipStreamNonBlocking, hi);
    for (int i = 0; i < dc; i++) (void)hipStreamCreateWithFlags(&d, hipStreamNonBlocking);
    (void)hipStreamCreateWithPriority(&s[3], hipStreamNonBlocking, hi);
    (void)hipStreamCreateWithPriority(&helper, hipStreamNonBlocking, lo);
    hipEvent_t late; (void)hipEventCreateWithFlags(&late, hipEventDisableTiming);
    const char *nm[4] = {"null", "aux", "aux2", "aux3"};
    probe_run(s[0], s[1], buf); probe_wait(s[0], s[1], helper, late, buf);
    printf("unused streams %d %d %d: [us]\n", da, db, dc);
    for (int x = 0; x < 4; x++) { printf("  run  X=%-5s", nm[x]); for (int y = 0; y < 4; y++) if (x != y) printf("  Y=%-5s %6.0f", nm[y], probe_run(s[x], s[y], buf)); printf("\n"); }
    for (int x = 0; x < 4; x++) { printf("  wait X=%-5s", nm[x]); for (int y = 0; y < 4; y++) if (x != y) printf("  Y=%-5s %6.0f", nm[y], probe_wait(s[x], s[y], helper, late, buf)); printf("\n"); }
    return 0;
}
