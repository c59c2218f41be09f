// Where a workgroup of the Winograd kernel spends its time: s_memtime stamps at kernel entry, K-loop entry,
// K-loop exit and kernel exit (compiled from the product source with VQ_WINO_PHASES).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DVQ_WINO_PHASES -Iinclude \
//         -Ivideo-query-algorithms_amd/csrc tools/ubench/wino_phases.hip -o tools/ubench/wino_phases
//   tools/ubench/wino_phases H Cin Cout crops variant
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../video-query-algorithms_amd/csrc/vq_wino.hip"

namespace vq {
std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
}  // namespace vq

int main(int argc, char** argv) {
    const int H = argc > 1 ? atoi(argv[1]) : 28, Cin = argc > 2 ? atoi(argv[2]) : 96, Cout = argc > 3 ? atoi(argv[3]) : 96;
    const int crops = argc > 4 ? atoi(argv[4]) : 96, variant = argc > 5 ? atoi(argv[5]) : 0;
    const size_t n_in = (size_t)crops * H * H * Cin, n_out = (size_t)crops * H * H * Cout, n_u = (size_t)16 * Cout * Cin;
    std::vector<float> hin(n_in), hu(n_u), hb(Cout, 0.1f);
    for (size_t i = 0; i < n_in; ++i) hin[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    for (size_t i = 0; i < n_u; ++i) hu[i] = (float)((i * 40503u) % 2000) * 1e-4f - 0.1f;
    float *din, *dout, *du, *db;
    hipMalloc(&din, n_in * 4);
    hipMalloc(&dout, n_out * 4);
    hipMalloc(&du, n_u * 4);
    hipMalloc(&db, Cout * 4);
    hipMemcpy(din, hin.data(), n_in * 4, hipMemcpyHostToDevice);
    hipMemcpy(du, hu.data(), n_u * 4, hipMemcpyHostToDevice);
    hipMemcpy(db, hb.data(), Cout * 4, hipMemcpyHostToDevice);
    vq::WinoGroup g{};
    g.n_jobs = 1;
    vq::WinoJob& a = g.job[0];
    a.in = din; a.u = du; a.bias = db; a.out = dout;
    a.H = a.W = H; a.Cs_in = Cin; a.coff_in = 0; a.Cin = Cin;
    a.Cs_out = Cout; a.coff_out = 0; a.Cout = Cout;
    a.th = a.tw = (H + 1) / 2; a.P = crops * a.th * a.tw; a.relu = 1;
    a.in_bytes = (unsigned)(n_in * 4); a.u_bytes = (unsigned)(n_u * 4); a.out_bytes = (unsigned)(n_out * 4);
    const int nb = variant == 1 ? 2 : 1;                     // variant 1 owns 64 output channels per workgroup
    int nwg = ((((a.P + 31) / 32) * ((Cout + 32 * nb - 1) / (32 * nb))) + 7) & ~7;
    hipMalloc(&a.phases, (size_t)nwg * 6 * sizeof(long long));
    hipMemset(a.phases, 0, (size_t)nwg * 6 * sizeof(long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) vq::launch_wino_group(g, variant, nullptr, nullptr, nullptr);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    if (vq::launch_wino_group(g, variant, nullptr, nullptr, nullptr) != 0) {
        printf("launch failed: %s\n", vq::last_error_ref().c_str());
        return 1;
    }
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> ph((size_t)nwg * 6);
    hipMemcpy(ph.data(), a.phases, ph.size() * sizeof(long long), hipMemcpyDeviceToHost);
    // padding workgroups (the grid is rounded up to a multiple of 8) leave their stamps at zero: drop them
    {
        std::vector<long long> live;
        for (int i = 0; i < nwg; ++i)
            if (ph[6 * i] != 0) live.insert(live.end(), ph.begin() + 6 * i, ph.begin() + 6 * i + 6);
        ph.swap(live);
    }
    const int nwg_all = nwg;
    (void)nwg_all;
    long long t0 = ph[0], t1 = ph[3];
    const int nwg_live = (int)(ph.size() / 6);
    for (int i = 0; i < nwg_live; ++i) {
        t0 = std::min(t0, ph[6 * i]);
        t1 = std::max(t1, ph[6 * i + 3]);
    }
    nwg = nwg_live;
    const double span = (double)(t1 - t0), tick_us = ms * 1e3 / span;   // event time includes launch overhead: upper bound
    double pro = 0, loop = 0, epi = 0;
    for (int i = 0; i < nwg; ++i) {
        pro += ph[6 * i + 1] - ph[6 * i];
        loop += ph[6 * i + 2] - ph[6 * i + 1];
        epi += ph[6 * i + 3] - ph[6 * i + 2];
    }
    const int steps = Cin / 8;
    printf("H=%d Cin=%d Cout=%d crops=%d BN=%d: %d workgroups, kernel %.1f us (events), stamp span %.0f ticks (<= %.4f us/tick)\n", H, Cin,
           Cout, crops, 32 * nb, nwg, ms * 1e3, span, tick_us);
    printf("  mean per workgroup [ticks]: prologue %.0f, K loop %.0f (%d steps, %.1f/step), epilogue %.0f\n", pro / nwg, loop / nwg, steps,
           loop / nwg / (steps - 1), epi / nwg);
    // start-time histogram: how many workgroups start in each tenth of the kernel
    int hist[10] = {0};
    for (int i = 0; i < nwg; ++i) hist[std::min(9, (int)((ph[6 * i] - t0) * 10 / span))]++;
    printf("  starts per tenth of the span:");
    for (int b = 0; b < 10; ++b) printf(" %d", hist[b]);
    printf("\n");
    // timeline of XCD 0 (stamps of different XCDs are not comparable): workgroups resident in each 1/20 of its span
    {
        long long x0 = -1, x1 = 0;
        int nx = 0;
        for (int i = 0; i < nwg; ++i)
            if ((ph[6 * i + 5] & 15) == 0) {
                x0 = x0 < 0 ? ph[6 * i] : std::min(x0, ph[6 * i]);
                x1 = std::max(x1, ph[6 * i + 3]);
                ++nx;
            }
        printf("  XCD 0: %d workgroups, span %lld ticks; resident workgroups per 1/20 span:", nx, x1 - x0);
        for (int b = 0; b < 20; ++b) {
            const long long t = x0 + (x1 - x0) * (2 * b + 1) / 40;
            int r = 0;
            for (int i = 0; i < nwg; ++i)
                if ((ph[6 * i + 5] & 15) == 0 && ph[6 * i] <= t && t < ph[6 * i + 3]) ++r;
            printf(" %d", r);
        }
        printf("\n");
        // one CU of XCD 0: its workgroups in start order
        long long hw0 = -1;
        for (int i = 0; i < nwg && hw0 < 0; ++i)
            if ((ph[6 * i + 5] & 15) == 0) hw0 = ph[6 * i + 4] & 0xFF00;   // cu_id | sh_id | se_id
        printf("  one CU (hw_id & 0xff00 = %llx): [start, K loop, end] relative to XCD start:", hw0);
        for (int i = 0; i < nwg; ++i)
            if ((ph[6 * i + 5] & 15) == 0 && (ph[6 * i + 4] & 0xFF00) == hw0)
                printf(" [%lld %lld %lld]", ph[6 * i] - x0, ph[6 * i + 1] - x0, ph[6 * i + 3] - x0);
        printf("\n");
    }
    const double flops = 2.0 * crops * H * H * Cout * (double)Cin * 9;
    printf("  algorithmic %.1f TFLOP/s, executed (x16/36) %.1f TFLOP/s\n", flops / ms / 1e9, flops / ms / 1e9 * 16 / 36);
    return 0;
}
