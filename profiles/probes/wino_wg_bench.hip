// prototype harness: Winograd wgrad — correctness (small) and timing (ResNet-50 shapes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../myconvnet_amd/csrc/wino_kernels.h"
#include "wino_w4_wgrad_kernel.h"
void mcn_set_error(const char*, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
static int g_w8 = 0;
static double run(int N, int H, int W, int Cin, int Cout, int reps, bool check) {
    size_t xn = (size_t)N * H * W * Cin, wn = (size_t)9 * Cin * Cout, yn = (size_t)N * H * W * Cout;
    std::vector<float> hx(xn), hdy(yn), hdw(wn);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hdy) v = (rand() % 2001 - 1000) / 1000.f;
    WinoWgradParams p; memset(&p, 0, sizeof(p));
    p.H = H; p.W = W; p.Cs = Cin; p.Cin = Cin; p.ldy = Cout; p.Nn = Cout; p.TH = (H + 1) / 2; p.TW = (W + 1) / 2; p.ntiles = N * p.TH * p.TW;
    p.nbc = (Cin + 63) / 64; p.nbn = (Cout + 63) / 64;
    int nblk = p.nbc * p.nbn;
    int splits = 256 / nblk; if (splits < 1) splits = 1;
    int groups = (p.ntiles + 31) / 32;
    if (splits > groups) splits = groups;
    int gps = (groups + splits - 1) / splits;
    splits = (groups + gps - 1) / gps;
    p.tiles_per_split = gps * 32;
    float *x, *dy, *slab, *dw;
    CK(hipMalloc(&x, xn * 4)); CK(hipMalloc(&dy, yn * 4)); CK(hipMalloc(&slab, (size_t)splits * wn * 4)); CK(hipMalloc(&dw, wn * 4));
    CK(hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dy, hdy.data(), yn * 4, hipMemcpyHostToDevice));
    CK(hipMemset(slab, 0xff, (size_t)splits * wn * 4));
    p.x = x; p.dy = dy; p.slab = slab; p.x_bytes = (unsigned)(xn * 4); p.dy_bytes = (unsigned)(yn * 4);
    const int lds = g_w8 ? WINO_WG_LDS_W8 : 2 * WINO_WG_STAGE;
    CK(hipFuncSetAttribute((const void*)conv_wino_wgrad_f3k2, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)conv_wino_wgrad_f3k2_w8, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int grid = nblk * splits;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto go = [&]() {
        if (g_w8) hipLaunchKernelGGL(conv_wino_wgrad_f3k2_w8, dim3(grid), dim3(512), lds, 0, p);
        else hipLaunchKernelGGL(conv_wino_wgrad_f3k2, dim3(grid), dim3(256), lds, 0, p);
        hipLaunchKernelGGL((wgrad_reduce_linear_kernel<1>), dim3(2048), dim3(256), 0, 0, (const float*)slab, dw, splits, (long)wn / 4, 1.0f);
    };
    go(); go();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) go();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (check) {
        CK(hipMemcpy(hdw.data(), dw, wn * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0;
        for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) for (int c = 0; c < Cin; ++c) for (int co = 0; co < Cout; ++co) {
            double s = 0;
            for (int n = 0; n < N; ++n) for (int oy = 0; oy < H; ++oy) for (int ox = 0; ox < W; ++ox) {
                int iy = oy + r - 1, ix = ox + q - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                s += (double)hx[(((size_t)n * H + iy) * W + ix) * Cin + c] * hdy[(((size_t)n * H + oy) * W + ox) * Cout + co];
            }
            double d = hdw[((size_t)(r * 3 + q) * Cin + c) * Cout + co] - s;
            num += d * d; den += s * s;
        }
        printf("check N%d H%d W%d %d->%d splits %d: rel l2 %.3e\n", N, H, W, Cin, Cout, splits, sqrt(num / den));
    }
    CK(hipFree(x)); CK(hipFree(dy)); CK(hipFree(slab)); CK(hipFree(dw));
    return ms / reps;
}
int main(int argc, char** argv) {
    g_w8 = argc > 1 ? atoi(argv[1]) : 0;
    run(2, 8, 8, 64, 64, 1, true);
    run(3, 7, 7, 32, 96, 1, true);
    run(5, 14, 10, 72, 40, 1, true);
    struct { int H, Cin, Cout; } L[] = {{56, 64, 64}, {28, 128, 128}, {14, 256, 256}, {7, 512, 512}};
    for (auto& l : L) {
        double fl = 2.0 * 256 * l.H * l.H * 9 * l.Cin * l.Cout;
        double ms = run(256, l.H, l.H, l.Cin, l.Cout, 10, false);
        printf("wino wgrad H%3d %4d->%4d : %7.1f us  %6.1f TF (direct-equivalent)\n", l.H, l.Cin, l.Cout, ms * 1e3, fl / ms / 1e9);
    }
    return 0;
}
