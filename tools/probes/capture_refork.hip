// Stand-alone repro for the hipStreamEndCapture crash on ROCm 7.2 (gfx950) that keeps REFIL_HIPGRAPH off the default
// four-stream schedule: a stream that was forked into a capture, JOINED back, and is then forked into the same capture
// a second time. Build: hipcc --offload-arch=gfx950 -O2 tools/probes/capture_refork.hip -o capture_refork ; run: ./capture_refork MODE
//   MODE 0  origin -> fork A -> join A -> end                                   (control)
//   MODE 1  origin -> fork A -> join A into origin -> fork A again from origin -> join -> end
//   MODE 2  origin -> fork A, fork B -> join A into B -> fork A again from B -> join A, B into origin -> end
//           (the learner's pattern: A = weight-gradient stream carrying the forward's mask-word kernel, B = hypernet chain)
//   MODE 3  origin -> fork A -> join A into origin TWICE, no node in between (a duplicate edge) -> end
//   MODE 4  origin -> fork A -> join A -> origin kernel -> join A again (A idle in between: a redundant edge) -> fork A again -> join -> end
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void bump(int* p) { atomicAdd(p, 1); }
static int after(hipStream_t from, hipStream_t to, hipEvent_t e) { CK(hipEventRecord(e, from)); CK(hipStreamWaitEvent(to, e, 0)); return 0; }
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1;
    hipStream_t o, a, b; hipEvent_t ev[8]; int* p;
    CK(hipStreamCreateWithFlags(&o, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    CK(hipMalloc(&p, 4)); CK(hipMemset(p, 0, 4));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(o, hipStreamCaptureModeThreadLocal));
    bump<<<1, 1, 0, o>>>(p);
    if (after(o, a, ev[0])) return 1;
    bump<<<1, 1, 0, a>>>(p);
    if (mode == 2) {
        if (after(o, b, ev[1])) return 1;
        bump<<<1, 1, 0, b>>>(p);
        if (after(a, b, ev[2])) return 1;              // join A into B
        bump<<<1, 1, 0, b>>>(p);
        if (after(b, a, ev[3])) return 1;              // A forked a second time, from B
        bump<<<1, 1, 0, a>>>(p);
        if (after(b, o, ev[4])) return 1;
    } else {
        if (after(a, o, ev[2])) return 1;              // join A into the origin
        if (mode == 3) { if (after(a, o, ev[7])) return 1; }
        bump<<<1, 1, 0, o>>>(p);
        if (mode == 4) {
            if (after(a, o, ev[6])) return 1;
            bump<<<1, 1, 0, o>>>(p);
        }
        if (mode == 1 || mode == 4) {
            if (after(o, a, ev[3])) return 1;          // A forked a second time
            bump<<<1, 1, 0, a>>>(p);
        }
    }
    if (mode) { if (after(a, o, ev[5])) return 1; }
    printf("mode %d: ending capture\n", mode); fflush(stdout);
    CK(hipStreamEndCapture(o, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, o)); CK(hipStreamSynchronize(o));
    int h = 0; CK(hipMemcpy(&h, p, 4, hipMemcpyDeviceToHost));
    printf("mode %d: graph replayed, %d kernel nodes ran\n", mode, h);
    return 0;
}
