// Feature rows as text, exactly as the reference's writer prints them (src/features_GPU_compute/calcSig_wOF.py:128-133:
// str(int(clip[-4:])) + "," + ",".join(map(str, feature)), one line per clip).  str(numpy.float64) is Python's float repr --
// the SHORTEST decimal string that reads back to the same double, fixed notation for 1e-4 <= |x| < 1e16, else d.ddde+XX -- under
// numpy >= 1.14 and '%.12g' (plus ".0" for integral values) before; the reference ships files of both kinds
// (tsn/feature_csv.py).  Formatting half a million doubles per 256-clip file in the interpreter took 0.1 s per file and held
// the interpreter lock against the threads that feed the GPU; here it is a C loop that holds nothing.
//
// Host-only translation unit (no HIP): also built by tests/sanitize/Makefile with -fsanitize=address,undefined.
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

#include "vq_amd.h"
#include "vq_host.h"

namespace {

// Python's repr(float): digits and exponent of the shortest round-trip form, then float_repr_style 'r' layout.
char* put_repr(char* p, double v) {
    if (std::isnan(v)) {
        memcpy(p, "nan", 3);
        return p + 3;
    }
    if (std::isinf(v)) {
        if (v < 0) *p++ = '-';
        memcpy(p, "inf", 3);
        return p + 3;
    }
    if (std::signbit(v)) {
        *p++ = '-';
        v = -v;
    }
    if (v == 0.0) {
        memcpy(p, "0.0", 3);
        return p + 3;
    }
    char sci[40];                                        // d[.ddd]e[+-]XX: shortest digits that round-trip
    const auto r = std::to_chars(sci, sci + sizeof sci, v, std::chars_format::scientific);
    char digits[24];
    int nd = 0;
    const char* q = sci;
    for (; q < r.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int exp10 = 0;
    if (q < r.ptr) {
        ++q;
        const bool neg = *q == '-';
        if (*q == '+' || *q == '-') ++q;
        for (; q < r.ptr; ++q) exp10 = exp10 * 10 + (*q - '0');
        if (neg) exp10 = -exp10;
    }
    const int decpt = exp10 + 1;                         // position of the decimal point relative to the digit string
    if (decpt > -4 && decpt <= 16) {                     // repr: fixed notation for 1e-4 <= v < 1e16
        if (decpt <= 0) {
            *p++ = '0';
            *p++ = '.';
            for (int i = 0; i < -decpt; ++i) *p++ = '0';
            memcpy(p, digits, nd);
            return p + nd;
        }
        if (decpt >= nd) {
            memcpy(p, digits, nd);
            p += nd;
            for (int i = nd; i < decpt; ++i) *p++ = '0';
            *p++ = '.';
            *p++ = '0';
            return p;
        }
        memcpy(p, digits, decpt);
        p += decpt;
        *p++ = '.';
        memcpy(p, digits + decpt, nd - decpt);
        return p + (nd - decpt);
    }
    *p++ = digits[0];
    if (nd > 1) {
        *p++ = '.';
        memcpy(p, digits + 1, nd - 1);
        p += nd - 1;
    }
    *p++ = 'e';
    int e = decpt - 1;
    *p++ = e < 0 ? '-' : '+';
    if (e < 0) e = -e;
    char eb[8];
    int ne = 0;
    do {
        eb[ne++] = (char)('0' + e % 10);
        e /= 10;
    } while (e);
    if (ne < 2) eb[ne++] = '0';                          // at least two exponent digits: 1e-05
    while (ne) *p++ = eb[--ne];
    return p;
}

// str(numpy.float64) of numpy < 1.14: '%.12g', integral values keep a ".0"
char* put_g12(char* p, double v) {
    const int n = snprintf(p, 32, "%.12g", v);
    bool plain = true;
    for (int i = 0; i < n; ++i)
        if (p[i] == '.' || p[i] == 'e' || p[i] == 'n') plain = false;       // 'n': nan, inf
    p += n;
    if (plain) {
        *p++ = '.';
        *p++ = '0';
    }
    return p;
}

}  // namespace

extern "C" int vq_format_feature_rows(const double* feats, int64_t n_rows, int32_t dim, const int64_t* clip_numbers, int32_t number_format,
                                      char* out, int64_t cap, int64_t* written) {
    if (!feats || !clip_numbers || !out || !written) return vq::host_fail(VQ_E_INVALID, "NULL argument");
    if (n_rows < 0 || dim <= 0) return vq::host_fail(VQ_E_INVALID, "n_rows >= 0 and dim > 0 required");
    if (number_format != 0 && number_format != 1) return vq::host_fail(VQ_E_INVALID, "number_format: 0 = repr, 1 = 12 significant digits");
    // worst case per value: sign + 17 digits + point + e-308 = 25 bytes, + the comma; per row 21 bytes of clip number + newline
    const int64_t need = n_rows * ((int64_t)dim * 26 + 22);
    if (cap < need) {
        *written = need;
        return vq::host_fail(VQ_E_INVALID, "output buffer of %lld bytes is smaller than the %lld the rows may need", (long long)cap, (long long)need);
    }
    auto rows = [&](int64_t r0, int64_t r1, char* p) {
        for (int64_t r = r0; r < r1; ++r) {
            p += snprintf(p, 22, "%lld", (long long)clip_numbers[r]);
            const double* row = feats + r * dim;
            for (int32_t k = 0; k < dim; ++k) {
                *p++ = ',';
                p = number_format == 0 ? put_repr(p, row[k]) : put_g12(p, row[k]);
            }
            *p++ = '\n';
        }
        return p;
    };
    // Up to four threads, each on a run of rows, each writing where its first row would start if every row before it had the worst-case
    // length; the runs are then moved together in order (256 x 1024 values: 13 ms on one thread, the last thing a command-line run waits for).
    const int64_t row_cap = (int64_t)dim * 26 + 22;
    const int workers = (int)std::min<int64_t>(4, n_rows * dim / 32768);
    char* p = out;
    if (workers <= 1) {
        p = rows(0, n_rows, out);
    } else {
        std::vector<char*> end((size_t)workers);
        std::vector<std::thread> pool;
        auto first_row = [&](int k) { return n_rows * k / workers; };
        for (int k = 1; k < workers; ++k)
            pool.emplace_back([&, k] { end[(size_t)k] = rows(first_row(k), first_row(k + 1), out + first_row(k) * row_cap); });
        end[0] = rows(0, first_row(1), out);
        for (std::thread& th : pool) th.join();
        p = end[0];
        for (int k = 1; k < workers; ++k) {
            char* from = out + first_row(k) * row_cap;
            const size_t len = (size_t)(end[(size_t)k] - from);
            memmove(p, from, len);
            p += len;
        }
    }
    *written = p - out;
    return VQ_OK;
}
