// anyorder_probe.cpp -- what is the floor of ONE small launch per step, and is it the launch or the memory system?
// A stand-in of the headline launch (50 crops x 128 rows: 3200 workgroups of 2 waves, one wave per output row, four 3-byte
// taps per output pixel, 4.9 MB of planar fp32 non-temporal stores) is replayed from a 256-launch HIP graph in scenarios
// that add the headline's difficulties one at a time:
//   empty        the kernel returns at once                               (launch + drain floor of the grid)
//   dense/hot    fixed 4x-shrunk crops of ONE frame                       (taps hit the L2 after the first replay)
//   dense/cold   the same from a rotation of 24 frames (600 MB)           (taps come from HBM)
//   var/hot      variable crops (origin + step from a kernel-argument table, w~U[32,512], h~U[64,1024]) of one frame
//   var/cold     variable crops of the 24-frame rotation, another crop table per launch  (= the headline's access pattern)
// and, for the submission model, eager launches with and without hipExtAnyOrderLaunch (AQL barrier bit) and hand-built
// graphs whose kernel nodes have no dependency edges.
//   hipcc -O2 --offload-arch=gfx950 tools/anyorder_probe.cpp -o tools/bin/anyorder_probe && tools/bin/anyorder_probe
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <random>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

namespace {
constexpr int CROPS = 50, DW = 64, DH = 128, FW = 3840, FH = 2160, ROT = 8, FRAMES = 24, TABS = 16;

struct CropTab { // per-crop origin and step, in the kernel arguments like K1's inline plane table
    int x0[CROPS], y0[CROPS];
    float fx[CROPS], fy[CROPS];
};
enum { DENSE = 0, EMPTY = 1, VAR = 2 };

__global__ __launch_bounds__(128) void standin(const unsigned char* __restrict__ frame, float* __restrict__ out, int mode, CropTab t) {
    if (mode == EMPTY) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 2 + wave, crop = blockIdx.y;
    int sx = crop * 7 + lane * 4, sy = crop * 3 + row * 4; // 4x shrink
    if (mode == VAR) {
        sx = t.x0[crop] + (int)(lane * t.fx[crop]);
        sy = t.y0[crop] + (int)(row * t.fy[crop]);
    }
    const unsigned char* p = frame + ((size_t)sy * FW + sx) * 3;
    const unsigned char* q = p + (size_t)FW * 3;
    float v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = 0.25f * ((float)p[k] + (float)p[k + 3] + (float)q[k] + (float)q[k + 3]);
    float* o = out + ((size_t)crop * 3 * DH + row) * DW + lane;
#pragma unroll
    for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(v[k] * 0.3f - 1.f, o + (size_t)k * DH * DW);
}

double ms_between(hipEvent_t a, hipEvent_t b) {
    float t = 0.f;
    (void)hipEventElapsedTime(&t, a, b);
    return t;
}
} // namespace

int main() {
    std::vector<unsigned char*> frames(FRAMES);
    float* out[ROT];
    for (auto& f : frames) {
        CK(hipMalloc(&f, (size_t)FW * FH * 3));
        CK(hipMemset(f, 7, (size_t)FW * FH * 3));
    }
    for (auto& o : out) CK(hipMalloc(&o, (size_t)CROPS * 3 * DH * DW * 4));
    std::vector<CropTab> tabs(TABS);
    std::mt19937 rng(20240229);
    for (auto& t : tabs)
        for (int i = 0; i < CROPS; ++i) {
            const int w = 32 + (int)(rng() % 481), h = 64 + (int)(rng() % 961);
            t.x0[i] = (int)(rng() % (unsigned)(FW - w - 1));
            t.y0[i] = (int)(rng() % (unsigned)(FH - h - 1));
            t.fx[i] = (float)(w - 1) / DW;
            t.fy[i] = (float)(h - 1) / DH;
        }
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(DH / 2, CROPS), block(128);

    auto launch = [&](int i, int mode, bool cold, unsigned flags) {
        const unsigned char* fr = frames[cold ? (size_t)(i % FRAMES) : 0];
        if (flags) hipExtLaunchKernelGGL(standin, grid, block, 0, s, nullptr, nullptr, flags, fr, out[i % ROT], mode, tabs[(size_t)(i % TABS)]);
        else hipLaunchKernelGGL(standin, grid, block, 0, s, fr, out[i % ROT], mode, tabs[(size_t)(i % TABS)]);
    };
    for (int i = 0; i < 20000; ++i) launch(i, DENSE, false, 0); // clock ramp
    CK(hipStreamSynchronize(s));

    struct Scn { const char* name; int mode; bool cold; };
    const Scn scn[] = {{"empty     ", EMPTY, false}, {"dense/hot ", DENSE, false}, {"dense/cold", DENSE, true}, {"var/hot   ", VAR, false}, {"var/cold  ", VAR, true}};
    for (const Scn& sc : scn) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < 264; ++i) launch(i, sc.mode, sc.cold, 0); // 264 = 11 x 24 frames
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 4; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            std::printf("%s in-order graph(264): %.3f us/launch\n", sc.name, ms_between(e0, e1) * 1e3 / (8 * 264));
        }
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    // submission model: eager in-order / any-order (host-bound on this runtime), graphs without dependency edges
    for (unsigned flags : {0u, (unsigned)hipExtAnyOrderLaunch}) {
        for (int rep = 0; rep < 2; ++rep) {
            const int N = 2048;
            CK(hipEventRecord(e0, s));
            const auto h0 = std::chrono::steady_clock::now();
            for (int i = 0; i < N; ++i) launch(i, VAR, true, flags);
            const auto h1 = std::chrono::steady_clock::now();
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            std::printf("var/cold   eager %s: %.3f us/launch (host enqueue %.3f us)\n", flags ? "any-order" : "in-order ", ms_between(e0, e1) * 1e3 / N,
                        std::chrono::duration<double, std::micro>(h1 - h0).count() / N);
        }
    }
    for (int chains : {264, 4, 2}) {
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipGraphCreate(&g, 0));
        std::vector<hipGraphNode_t> last((size_t)chains, nullptr);
        int mode = VAR;
        for (int i = 0; i < 264; ++i) {
            const unsigned char* fr = frames[(size_t)(i % FRAMES)];
            float* o = out[i % ROT];
            void* args[4] = {(void*)&fr, (void*)&o, (void*)&mode, (void*)&tabs[(size_t)(i % TABS)]};
            hipKernelNodeParams kp{};
            kp.func = (void*)standin;
            kp.gridDim = grid;
            kp.blockDim = block;
            kp.kernelParams = args;
            hipGraphNode_t n;
            hipGraphNode_t& prev = last[(size_t)(i % chains)];
            CK(hipGraphAddKernelNode(&n, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = n;
        }
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int w = 0; w < 4; ++w) CK(hipGraphLaunch(ge, s));
        CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < 8; ++r) CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            std::printf("var/cold   graph of %3d independent chains: %.3f us/launch\n", chains, ms_between(e0, e1) * 1e3 / (8 * 264));
        }
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
