// How long does a chain of N dependent, (nearly) empty kernels take on this box -- launched into a stream, replayed as a
// hipGraph captured from the stream, and replayed as a hipGraph whose nodes were added explicitly?  (what a
// kernel boundary costs inside the step's graph.)   hipcc -O2 --offload-arch=gfx950 launch_chain_probe.hip -o launch_chain_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(float* p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main() {
    const int N = 40, REP = 50;
    float* buf; CK(hipMalloc(&buf, 1 << 20)); CK(hipMemset(buf, 0, 1 << 20));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int blocks : {1, 256, 2048}) {
        // (a) stream launches
        for (int w = 0; w < 3; ++w) { for (int k = 0; k < N; ++k) tiny<<<blocks, 256, 0, st>>>(buf, blocks * 256); }
        CK(hipStreamSynchronize(st));
        float ms_stream = 0;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) for (int k = 0; k < N; ++k) tiny<<<blocks, 256, 0, st>>>(buf, blocks * 256);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_stream, e0, e1));
        // (b) captured graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < N; ++k) tiny<<<blocks, 256, 0, st>>>(buf, blocks * 256);
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        float ms_graph = 0;
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < REP; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms_graph, e0, e1));
        printf("blocks %4d: %d dependent kernels: stream %.2f us per kernel, captured graph %.2f us per kernel\n", blocks, N,
               1e3 * ms_stream / (REP * N), 1e3 * ms_graph / (REP * N));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
