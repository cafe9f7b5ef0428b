// prototype harness: Winograd F(2x2,3x3) fp32 forward — correctness against a CPU loop (small) and timing (ResNet-50 shapes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../myconvnet_amd/csrc/wino_kernels.h"
#include "wino_w4_kernel.h"
void mcn_set_error(const char*, ...) {}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static int g_w8 = 0;
template <int DBG>
static double run(int N, int H, int W, int Cin, int Cout, int reps, bool check) {
    size_t xn = (size_t)N * H * W * Cin, wn = (size_t)9 * Cin * Cout, yn = (size_t)N * H * W * Cout;
    std::vector<float> hx(xn), hw(wn), hy(yn);
    for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
    for (auto& v : hw) v = (rand() % 2001 - 1000) / 3000.f;
    float *x, *w, *u, *y;
    const int nb = (Cout + 63) / 64, ns = Cin / 32;
    size_t un = (size_t)nb * ns * 16 * 2048;
    CK(hipMalloc(&x, xn * 4)); CK(hipMalloc(&w, wn * 4)); CK(hipMalloc(&u, un * 4)); CK(hipMalloc(&y, yn * 4));
    CK(hipMemcpy(x, hx.data(), xn * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), wn * 4, hipMemcpyHostToDevice));
    CK(hipMemset(y, 0xff, yn * 4));
    WinoParams p; memset(&p, 0, sizeof(p));
    p.in = x; p.u = u; p.out = y; p.H = H; p.W = W; p.Cs = Cin; p.Cin = Cin; p.TH = (H + 1) / 2; p.TW = (W + 1) / 2; p.ntiles = N * p.TH * p.TW;
    p.Nn = Cout; p.ldo = Cout; p.in_bytes = (unsigned)(xn * 4); p.u_bytes = (unsigned)(un * 4); p.out_bytes = (unsigned)(yn * 4);
    const long tot = (long)nb * ns * 2048;
    const int lds = 2 * WINO_STAGE;
    CK(hipFuncSetAttribute((const void*)&conv_wino_f2k3<DBG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void*)&conv_wino_f2k3_w8<DBG, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    static const int g_pers = getenv("WINO_PERS") ? atoi(getenv("WINO_PERS")) : 0;
    const int total = ((p.ntiles + 63) / 64) * nb;
    p.sk_slices = 1; p.sk_body = total; (void)g_pers;
    const int grid = total;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) {
        hipLaunchKernelGGL(wino_filter_transform_kernel, dim3((tot + 255) / 256), dim3(256), 0, 0, w, u, Cin, Cout, 0);
        if (g_w8) hipLaunchKernelGGL((conv_wino_f2k3_w8<DBG, 0>), dim3(grid), dim3(512), lds, 0, p);
        else hipLaunchKernelGGL((conv_wino_f2k3<DBG, 0>), dim3(grid), dim3(256), lds, 0, p);
    }
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) {
        if (g_w8) hipLaunchKernelGGL((conv_wino_f2k3_w8<DBG, 0>), dim3(grid), dim3(512), lds, 0, p);
        else hipLaunchKernelGGL((conv_wino_f2k3<DBG, 0>), dim3(grid), dim3(256), lds, 0, p);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (check) {
        CK(hipMemcpy(hy.data(), y, yn * 4, hipMemcpyDeviceToHost));
        double num = 0, den = 0, mx = 0;
        for (int n = 0; n < N; ++n) for (int oy = 0; oy < H; ++oy) for (int ox = 0; ox < W; ++ox) for (int co = 0; co < Cout; ++co) {
            double s = 0;
            for (int r = 0; r < 3; ++r) for (int q = 0; q < 3; ++q) {
                int iy = oy + r - 1, ix = ox + q - 1;
                if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                const float* xp = &hx[(((size_t)n * H + iy) * W + ix) * Cin];
                const float* wp = &hw[((size_t)(r * 3 + q) * Cin) * Cout + co];
                for (int c = 0; c < Cin; ++c) s += (double)xp[c] * wp[(size_t)c * Cout];
            }
            double d = hy[(((size_t)n * H + oy) * W + ox) * Cout + co] - s;
            num += d * d; den += s * s; if (fabs(d) > mx) mx = fabs(d);
        }
        printf("check N%d H%d W%d %d->%d : rel l2 %.3e  max abs %.3e\n", N, H, W, Cin, Cout, sqrt(num / den), mx);
    }
    CK(hipFree(x)); CK(hipFree(w)); CK(hipFree(u)); CK(hipFree(y));
    return ms / reps;
}

int main(int argc, char** argv) {
    int g_dbg = argc > 1 ? atoi(argv[1]) : 0;
    g_w8 = argc > 2 ? atoi(argv[2]) : 0;
    run<0>(2, 8, 8, 32, 64, 1, true);
    run<0>(3, 7, 7, 64, 128, 1, true);
    run<0>(2, 14, 10, 64, 96, 1, true);
    struct { int H, Cin, Cout; } L[] = {{56, 64, 64}, {28, 128, 128}, {14, 256, 256}, {7, 512, 512}};
    for (auto& l : L) {
        double fl = 2.0 * 256 * l.H * l.H * 9 * l.Cin * l.Cout;
        double ms = 0;
        switch (g_dbg) {
            case 0: ms = run<0>(256, l.H, l.H, l.Cin, l.Cout, 10, false); break;
            case 1: ms = run<1>(256, l.H, l.H, l.Cin, l.Cout, 10, false); break;
            case 2: ms = run<2>(256, l.H, l.H, l.Cin, l.Cout, 10, false); break;
        }
        printf("wino H%3d %4d->%4d : %7.1f us  %6.1f TF (direct-equivalent)  %6.1f TF executed\n", l.H, l.Cin, l.Cout, ms * 1e3, fl / ms / 1e9, fl / ms / 1e9 * 16 / 36 * ((l.H + 1) / 2 * 2.0 / l.H) * ((l.H + 1) / 2 * 2.0 / l.H));
    }
    return 0;
}
