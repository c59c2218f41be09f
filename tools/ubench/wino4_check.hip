// The F(4x4,3x3) experiment (tools/ubench/wino4_kernel.hip) on its own: correctness against a direct fp64 convolution on small
// maps (heights that are not multiples of 4, a ragged last tile block), then the launch time at the layers' shapes.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -Iinclude \
//         -Ivideo-query-algorithms_amd/csrc tools/ubench/wino4_check.hip -o tools/ubench/wino4_check
//   tools/ubench/wino4_check [H Cin Cout crops]
// -DVQ_WINO_PHASES adds per-workgroup phase stamps; -DVQ_EXP_NOLOAD / _NOU / _NOREAD / _NOMFMA drop one ingredient of the K
// loop (timing only: WINO4_TIME_ANYWAY=1 skips the failing checks); WINO4_IDENT=1 WINO4_DUMP=1 runs an identity filter and
// prints outputs (which input value landed where).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "wino4_kernel.hip"

namespace vq {
std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
}  // namespace vq

static const double Gm[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};

// w[o][c][3][3] -> [Cin/8][36][Cout][8]
static std::vector<float> transform_filters(const std::vector<double>& w, int Cout, int Cin) {
    std::vector<float> u((size_t)36 * Cout * Cin);
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c) {
            const double* g = &w[((size_t)o * Cin + c) * 9];
            double t[6][3];
            for (int i = 0; i < 6; ++i)
                for (int b = 0; b < 3; ++b) t[i][b] = Gm[i][0] * g[0 * 3 + b] + Gm[i][1] * g[1 * 3 + b] + Gm[i][2] * g[2 * 3 + b];
            for (int i = 0; i < 6; ++i)
                for (int j = 0; j < 6; ++j) {
                    const double v = t[i][0] * Gm[j][0] + t[i][1] * Gm[j][1] + t[i][2] * Gm[j][2];
                    u[(((size_t)(c / 8) * 36 + i * 6 + j) * Cout + o) * 8 + c % 8] = (float)v;
                }
        }
    return u;
}

static int run_case(int H, int W, int Cin, int Cout, int crops, bool check, int reps) {
    const size_t n_in = (size_t)crops * H * W * Cin, n_out = (size_t)crops * H * W * Cout;
    std::vector<float> hin(n_in), hb(Cout);
    std::vector<double> w((size_t)Cout * Cin * 9);
    unsigned s = 12345u + H * 7 + Cin;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (double)((s >> 8) & 0xFFFF) / 65536.0 - 0.5; };
    for (auto& v : hin) v = (float)(rnd() * 2.0);
    for (auto& v : w) v = (double)(float)(rnd() * 0.2);
    for (auto& v : hb) v = (float)(rnd() * 0.5);
    if (getenv("WINO4_IDENT")) {   // centre tap = identity on (o % Cin == c): the output shows which input value landed where
        for (auto& v : w) v = 0;
        for (int o = 0; o < Cout; ++o) w[((size_t)o * Cin + o % Cin) * 9 + 4] = 1.0;
        for (auto& v : hb) v = 0;
        for (int n = 0; n < crops; ++n)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    for (int c = 0; c < Cin; ++c) hin[(((size_t)n * H + y) * W + x) * Cin + c] = (float)(n * 10000 + y * 100 + x * 10 + c + 1);
    }
    const std::vector<float> hu = transform_filters(w, Cout, Cin);
    float *din, *dout, *du, *db;
    hipMalloc(&din, n_in * 4);
    hipMalloc(&dout, n_out * 4);
    hipMalloc(&du, hu.size() * 4);
    hipMalloc(&db, Cout * 4);
    hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice);
    hipMemcpy(du, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice);
    hipMemset(dout, 0xFF, n_out * 4);   // NaN: an unwritten output shows
    vq::WinoGroup g{};
    g.n_jobs = 1;
    vq::WinoJob& a = g.job[0];
    a.in = din; a.u = du; a.bias = db; a.out = dout;
    a.H = H; a.W = W; a.Cs_in = Cin; a.coff_in = 0; a.Cin = Cin;
    a.Cs_out = Cout; a.coff_out = 0; a.Cout = Cout;
    a.th = (H + 3) / 4; a.tw = (W + 3) / 4; a.P = crops * a.th * a.tw; a.relu = 1;
    a.in_bytes = (unsigned)(n_in * 4); a.u_bytes = (unsigned)(hu.size() * 4); a.out_bytes = (unsigned)(n_out * 4);
#ifdef VQ_WINO_PHASES
    const int nwg_max = (((a.P + 15) / 16) * (Cout / 32) + 7) & ~7;
    hipMalloc(&a.phases, (size_t)nwg_max * 6 * sizeof(long long));
    hipMemset(a.phases, 0, (size_t)nwg_max * 6 * sizeof(long long));
#endif
    if (vq::launch_wino4_group(g, nullptr, nullptr, nullptr) != 0) {
        printf("launch failed: %s\n", vq::last_error_ref().c_str());
        return 1;
    }
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    int rc = 0;
    if (check) {
        std::vector<float> hout(n_out);
        hipMemcpy(hout.data(), dout, n_out * 4, hipMemcpyDeviceToHost);
        double worst = 0, ymax = 0;
        size_t nan = 0;
        int ndump = 0;
        for (int n = 0; n < crops; ++n)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    for (int o = 0; o < Cout; ++o) {
                        double acc = hb[o];
                        for (int dy = 0; dy < 3; ++dy)
                            for (int dx = 0; dx < 3; ++dx) {
                                const int yy = y + dy - 1, xx = x + dx - 1;
                                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                                const float* px = &hin[(((size_t)n * H + yy) * W + xx) * Cin];
                                for (int c = 0; c < Cin; ++c) acc += (double)px[c] * w[((size_t)o * Cin + c) * 9 + dy * 3 + dx];
                            }
                        acc = std::max(acc, 0.0);
                        const float got = hout[(((size_t)n * H + y) * W + x) * Cout + o];
                        if (!(got == got)) ++nan;
                        if (getenv("WINO4_DUMP") && (getenv("WINO4_IDENT") || std::fabs((double)got - acc) > 1e-3) && ndump < 60 && n == 0 && (getenv("WINO4_IDENT") ? (o == 0 || o == 9) && (x == 0 || x == 5) : (o % 4) == 0)) {
                            printf("  y=%d x=%d o=%d got %.5f want %.5f\n", y, x, o, got, acc);
                            ++ndump;
                        }
                        worst = std::max(worst, std::fabs((double)got - acc));
                        ymax = std::max(ymax, std::fabs(acc));
                    }
        printf("check %dx%d Cin=%d Cout=%d crops=%d: max |err| = %.3e = %.3e max|y|, %zu NaN%s\n", H, W, Cin, Cout, crops, worst, worst / ymax, nan,
               (nan == 0 && worst <= 2e-5 * ymax) ? "" : "   <-- FAIL");
        rc = (nan == 0 && worst <= 2e-5 * ymax) ? 0 : 1;
    }
    if (reps > 0) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int r = 0; r < 3; ++r) vq::launch_wino4_group(g, nullptr, nullptr, nullptr);
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) vq::launch_wino4_group(g, nullptr, nullptr, nullptr);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= reps;
        const double flops = 2.0 * crops * H * W * Cout * (double)Cin * 9;
#ifdef VQ_WINO_PHASES
        {
            std::vector<long long> ph((size_t)nwg_max * 6);
            hipMemcpy(ph.data(), a.phases, ph.size() * sizeof(long long), hipMemcpyDeviceToHost);
            double pro = 0, loop = 0, epi = 0;
            int live = 0;
            for (int i = 0; i < nwg_max; ++i)
                if (ph[6 * i]) {
                    ++live;
                    pro += ph[6 * i + 1] - ph[6 * i];
                    loop += ph[6 * i + 2] - ph[6 * i + 1];
                    epi += ph[6 * i + 3] - ph[6 * i + 2];
                }
            printf("  mean per workgroup [ticks]: prologue %.0f, K loop %.0f (%d steps, %.1f/step), epilogue %.0f\n", pro / live, loop / live, Cin / 8,
                   loop / live / (Cin / 8), epi / live);
        }
#endif
        printf("time %dx%d Cin=%d Cout=%d crops=%d: %d workgroups, %.1f us per launch, algorithmic %.1f TFLOP/s, executed (x36/144) %.1f TFLOP/s\n", H, W,
               Cin, Cout, crops, g.total_units, ms * 1e3, flops / ms / 1e9, flops / ms / 1e9 / 4);
    }
    hipFree(din); hipFree(dout); hipFree(du); hipFree(db);
    return rc;
}

int main(int argc, char** argv) {
    int rc = 0;
    rc |= run_case(8, 8, 8, 32, 2, true, 0);
    rc |= run_case(12, 20, 16, 64, 3, true, 0);
    rc |= run_case(14, 16, 24, 32, 5, true, 0);     // height not a multiple of 4: clipped tiles
    rc |= run_case(28, 28, 64, 96, 2, true, 0);
    rc |= run_case(7, 4, 16, 32, 1, true, 0);
    if (rc && !getenv("WINO4_TIME_ANYWAY")) return rc;
    if (argc > 4) {
        run_case(atoi(argv[1]), atoi(argv[1]), atoi(argv[2]), atoi(argv[3]), atoi(argv[4]), false, 20);
    } else {
        run_case(56, 56, 64, 192, 96, false, 20);
        run_case(28, 28, 64, 64, 96, false, 20);
        run_case(28, 28, 64, 96, 96, false, 20);
        run_case(28, 28, 96, 96, 96, false, 20);
        run_case(16, 16, 128, 160, 96, false, 20);
        run_case(56, 56, 64, 192, 448, false, 10);
        run_case(28, 28, 96, 96, 448, false, 10);
    }
    return 0;
}
