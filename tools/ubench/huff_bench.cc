// Host entropy decoding alone (csrc/host/vq_jpeg_host.cc:decode_scan), one thread: MB/s of scan data and ns per coefficient block.
//   g++ -O3 -std=c++17 -pthread -I include -I video-query-algorithms_amd/csrc/host tools/ubench/huff_bench.cc \
//       video-query-algorithms_amd/csrc/host/vq_jpeg_host.cc -o tools/ubench/build/huff_bench && tools/ubench/build/huff_bench <dir of .jpg> [reps]
#include <dirent.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vq_jpeg_host.h"

namespace vq {
std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
}  // namespace vq

using namespace vq::jpeg;

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    std::vector<std::vector<uint8_t>> files;
    if (DIR* d = opendir(dir.c_str())) {
        while (dirent* e = readdir(d)) {
            const std::string nm = e->d_name;
            if (nm.size() < 4 || nm.substr(nm.size() - 4) != ".jpg") continue;
            FILE* f = fopen((dir + "/" + nm).c_str(), "rb");
            if (!f) continue;
            fseek(f, 0, SEEK_END);
            const long n = ftell(f);
            fseek(f, 0, SEEK_SET);
            std::vector<uint8_t> v((size_t)n);
            if (fread(v.data(), 1, (size_t)n, f) == (size_t)n) files.push_back(std::move(v));
            fclose(f);
        }
        closedir(d);
    }
    if (files.empty()) return 1;
    std::vector<Frame> fr(files.size());
    std::vector<std::vector<int16_t>> coef(files.size());
    std::vector<std::vector<size_t>> off(files.size(), std::vector<size_t>(3, 0));
    size_t bytes = 0, blocks = 0;
    for (size_t i = 0; i < files.size(); ++i) {
        if (parse_headers(files[i].data(), files[i].size(), fr[i]) != VQ_OK) return 2;
        const size_t nb = place_blocks(fr[i], fr[i].H, fr[i].W);
        size_t at = 0;
        for (int c = 0; c < fr[i].nc; ++c) {
            off[i][c] = at;
            at += (size_t)fr[i].comp[c].bw * fr[i].comp[c].bh;
        }
        coef[i].assign(nb * 64, 0);
        bytes += files[i].size() - fr[i].scan;
        blocks += nb;
    }
    unsigned long long sum = 0;
    for (int pass = 0; pass < 3; ++pass) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (size_t i = 0; i < files.size(); ++i) {
                memset(coef[i].data(), 0, coef[i].size() * 2);
                if (decode_scan(files[i].data(), files[i].size(), fr[i], coef[i].data(), off[i].data()) != VQ_OK) return 3;
            }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (size_t i = 0; i < files.size(); ++i)
            for (int16_t v : coef[i]) sum = sum * 1315423911ull + (unsigned short)v;
        printf("%zu files, %.1f KB of scan each: %.1f MB/s, %.1f ns per block, %.3f ms per file (checksum %016llx)\n", files.size(),
               bytes / 1024.0 / files.size(), bytes * reps / s / 1e6, s * 1e9 / (blocks * (double)reps), s * 1e3 / (files.size() * (double)reps), sum);
        sum = 0;
    }
    return 0;
}
