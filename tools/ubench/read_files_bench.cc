// How fast the library's worker threads read a batch of small files (csrc/host/vq_jpeg_host.cc:read_files): host only.
//   g++ -O2 -std=c++17 -pthread -I include -I video-query-algorithms_amd/csrc/host tools/ubench/read_files_bench.cc \
//       video-query-algorithms_amd/csrc/host/vq_jpeg_host.cc -o gpurun_out/read_files_bench && gpurun_out/read_files_bench <dir> [files] [bytes]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "vq_jpeg_host.h"

namespace vq {
std::string& last_error_ref() {
    static thread_local std::string s;
    return s;
}
}  // namespace vq

int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const int n = argc > 2 ? atoi(argv[2]) : 8000;
    const size_t bytes = argc > 3 ? (size_t)atol(argv[3]) : 13000;
    std::vector<std::string> paths;
    std::vector<uint8_t> junk(bytes, 0x5a);
    for (int i = 0; i < n; ++i) {
        paths.push_back(dir + "/clip_" + std::to_string(i / 250) + "_flow_" + std::to_string(i) + ".jpg");
        FILE* f = fopen(paths.back().c_str(), "wb");
        if (!f) return 1;
        fwrite(junk.data(), 1, junk.size(), f);
        fclose(f);
    }
    std::vector<const char*> cp;
    for (auto& p : paths) cp.push_back(p.c_str());
    for (int workers : {1, 2, 4, 8, 16, vq::jpeg::batch_workers(n)}) {
        std::vector<std::vector<uint8_t>> data;          // reused by the repetitions, as the decoder handle reuses its own
        for (int rep = 0; rep < 3; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            const int rc = vq::jpeg::read_files(cp.data(), n, data, workers);
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            printf("rc %d: %d files of %zu bytes on %d threads in %.2f ms%s\n", rc, n, bytes, workers, ms, rep ? "" : "  (fresh memory)");
        }
    }
    for (auto& p : paths) remove(p.c_str());
    return 0;
}
