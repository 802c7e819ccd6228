// Reduction of the hipStreamEndCapture crash recorded in profiles/round3/graph_probe.txt (stages 4 / 5 of
// benchmarks/graph_probe.py: forward + loss + backward with the companion weight-gradient streams).  What that capture does
// and the passing ones (ILVLM_WGRAD_STREAMS=0; the two-stream forward) do not: a second stream W is forked from the origin
// stream S MANY times -- every weight-gradient GEMM is ordered behind its operands by `record(e, S); wait(W, e)`
// (csrc/block.hip order_after, 4 x 24 times per step, through a ring of 64 events that is re-recorded inside the capture) --
// and joined back ONCE, at the end of the tower.  This program replays exactly that pattern with trivial kernels:
//   graph_fork_probe <forks> <ring> <nested> <thread>
//     forks   number of record(S) / wait(W) pairs before the one join
//     ring    number of distinct events cycled through (0 = a fresh event per fork)
//     nested  1 = S itself is a fork of the capture's origin stream O (the text tower's stream), joined back at the end
//     thread  1 = the forks are issued from a second host thread (autograd's worker) than the one that began the capture
//     two     1 = (with nested) the origin O forks a companion W2 of its own as well, many times, joined once: both towers
// Run it against the HIP runtime torch bundles too (LD_LIBRARY_PATH=<site-packages>/torch/lib): that is the library the crashing
// capture runs on, and it is not the one hipcc links by default (/opt/rocm/lib).
// RESULT (profiles/round4/graph_fork_probe_both_runtimes.txt): on /opt/rocm's 7.2 runtime every variant passes; on the runtime the
// torch wheel bundles (roc-7.0.2, LD_PRELOAD=<site-packages>/torch/lib/libamdhip64.so) every nested = 1 variant -- ONE fork of W
// from a forked S is enough -- segfaults in hipStreamEndCapture (hip::Stream::EndCapture() recursing without end), nested = 0 passes.
// Prints "ok <result>" when capture, instantiate and launch succeed.  Build: hipcc -O2 -o graph_fork_probe graph_fork_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("error %d (%s) at line %d: %s\n", (int)e_, hipGetErrorString(e_), __LINE__, #x); \
            fflush(stdout);                                                                     \
            exit(2);                                                                            \
        }                                                                                       \
    } while (0)

__global__ void bump(float* p, float v) { atomicAdd(p, v); }

int main(int argc, char** argv) {
    const int forks = argc > 1 ? atoi(argv[1]) : 4, ring = argc > 2 ? atoi(argv[2]) : 0, nested = argc > 3 ? atoi(argv[3]) : 0,
              thr = argc > 4 ? atoi(argv[4]) : 0, two = argc > 5 ? atoi(argv[5]) : 0;
    float* d;
    CK(hipMalloc(&d, 4));
    CK(hipMemset(d, 0, 4));
    hipStream_t O, S, W, W2;
    CK(hipStreamCreateWithFlags(&W2, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev2(forks);
    for (auto& e : ev2) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t join_w2;
    CK(hipEventCreateWithFlags(&join_w2, hipEventDisableTiming));
    CK(hipStreamCreateWithFlags(&O, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&W, hipStreamNonBlocking));
    std::vector<hipEvent_t> ev(ring > 0 ? ring : forks);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    hipEvent_t fork_s, join_w, join_s;
    CK(hipEventCreateWithFlags(&fork_s, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_w, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&join_s, hipEventDisableTiming));
    // eager warm-up of the same pattern (the probe warms up before it captures; the ring events carry eager records)
    auto body = [&](hipStream_t s) {
        for (int i = 0; i < forks; ++i) {
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, s, d, 1.0f);
            hipEvent_t e = ev[ring > 0 ? i % ring : i];
            CK(hipEventRecord(e, s));
            CK(hipStreamWaitEvent(W, e, 0));
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, W, d, 100.0f);
        }
        CK(hipEventRecord(join_w, W));
        CK(hipStreamWaitEvent(s, join_w, 0));
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, s, d, 10000.0f);
    };
    body(S);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d, 0, 4));
    hipStream_t origin = nested ? O : S;
    CK(hipStreamBeginCapture(origin, hipStreamCaptureModeGlobal));
    if (nested) {
        hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, O, d, 0.5f);
        CK(hipEventRecord(fork_s, O));
        CK(hipStreamWaitEvent(S, fork_s, 0));
    }
    if (thr) {
        std::thread t([&] { body(S); });
        t.join();
    } else {
        body(S);
    }
    if (nested && two) {           // the vision tower on the origin stream, with its own companion
        for (int i = 0; i < forks; ++i) {
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, O, d, 0.0f);
            CK(hipEventRecord(ev2[i], O));
            CK(hipStreamWaitEvent(W2, ev2[i], 0));
            hipLaunchKernelGGL(bump, dim3(1), dim3(1), 0, W2, d, 0.0f);
        }
        CK(hipEventRecord(join_w2, W2));
        CK(hipStreamWaitEvent(O, join_w2, 0));
    }
    if (nested) {
        CK(hipEventRecord(join_s, S));
        CK(hipStreamWaitEvent(O, join_s, 0));
    }
    printf("capture body issued (forks %d ring %d nested %d thread %d two %d)\n", forks, ring, nested, thr, two);
    fflush(stdout);
    hipGraph_t g;
    CK(hipStreamEndCapture(origin, &g));
    printf("end capture ok\n");
    fflush(stdout);
    hipGraphExec_t ge;
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, origin));
    CK(hipStreamSynchronize(origin));
    float h = 0;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("ok %.1f (expected %.1f)\n", h, forks * 101.0f + 10000.0f + (nested ? 0.5f : 0.0f));
    return 0;
}
