// Host-only AddressSanitizer / UBSan build of the COCO RLE codec (ampis_amd/csrc/rle_host.hip is plain C++: no device code), driven
// with valid round trips and with hostile input (the reference has no sanitizer run; GPU sanitizers are not available on
// the pool, so the host codec is what can be covered).  Built and run by tests/test_sanitize.py:
//   g++ -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -I<rocm>/include rle_sanitize_main.cpp rle_host.hip
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/ampis_hip.h"

namespace amp {   // the library defines these in context.hip (which needs the HIP runtime); the codec only reports through them
static char g_err[1024];
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace amp

static unsigned long long rng_state = 88172645463325252ull;
static unsigned int rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return (unsigned int)(rng_state >> 11); }

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "CHECK failed line %d: %s (%s)\n", __LINE__, #cond, amp::g_err); return 1; } } while (0)

int main() {
    // 1. round trips on random masks of awkward sizes: encode -> string -> from_string -> decode, area, iou, merge
    for (int it = 0; it < 200; ++it) {
        const int h = 1 + rnd() % 70, w = 1 + rnd() % 90;
        std::vector<uint8_t> m((size_t)h * w), m2((size_t)h * w), back((size_t)h * w);
        const unsigned int dens = rnd() % 100;
        for (auto& v : m) v = (rnd() % 100) < dens;
        for (auto& v : m2) v = (rnd() % 100) < 50;
        std::vector<uint32_t> c((size_t)h * w + 2), c2((size_t)h * w + 2), c3((size_t)h * w + 2), cm(2 * ((size_t)h * w + 2));
        int n = 0, n2 = 0, n3 = 0, nm = 0;
        CHECK(amp_rle_encode(m.data(), h, w, c.data(), (int)c.size(), &n) == AMP_OK);
        CHECK(amp_rle_encode(m2.data(), h, w, c2.data(), (int)c2.size(), &n2) == AMP_OK);
        std::vector<char> s((size_t)n * 7 + 8);
        size_t len = 0;
        CHECK(amp_rle_to_string(c.data(), n, s.data(), s.size(), &len) == AMP_OK);
        CHECK(amp_rle_from_string(s.data(), len, c3.data(), (int)c3.size(), &n3) == AMP_OK);
        CHECK(n3 == n && memcmp(c.data(), c3.data(), (size_t)n * 4) == 0);
        CHECK(amp_rle_decode(c3.data(), n3, h, w, back.data()) == AMP_OK);
        CHECK(memcmp(back.data(), m.data(), m.size()) == 0);
        unsigned long long area = 0, want = 0;
        for (auto v : m) want += v;
        CHECK(amp_rle_area(c.data(), n, &area) == AMP_OK && area == want);
        double iou = -1;
        CHECK(amp_rle_iou(c.data(), n, c2.data(), n2, 0, &iou) == AMP_OK && iou >= 0.0 && iou <= 1.0);
        CHECK(amp_rle_merge2(c.data(), n, c2.data(), n2, it & 1, cm.data(), (int)cm.size(), &nm) == AMP_OK);
        unsigned long long am = 0, wm = 0;
        for (size_t i = 0; i < m.size(); ++i) wm += (it & 1) ? (m[i] && m2[i]) : (m[i] || m2[i]);
        CHECK(amp_rle_area(cm.data(), nm, &am) == AMP_OK && am == wm);
        // too-small buffers are reported, not overrun
        if (n > 1) CHECK(amp_rle_encode(m.data(), h, w, c3.data(), n - 1, &n3) != AMP_OK);
        if (len > 1) CHECK(amp_rle_to_string(c.data(), n, s.data(), len - 1, &len) != AMP_OK);
    }
    // 2. polygons, including degenerate ones
    for (int it = 0; it < 100; ++it) {
        const int h = 5 + rnd() % 60, w = 5 + rnd() % 60, k = 3 + rnd() % 12;
        std::vector<double> xy(2 * k);
        for (auto& v : xy) v = (double)(rnd() % 2000) / 20.0 - 20.0;      // partly outside the canvas
        std::vector<uint32_t> c((size_t)h * w + 2);
        int n = 0;
        CHECK(amp_rle_from_polygon(xy.data(), k, h, w, c.data(), (int)c.size(), &n) == AMP_OK);
        unsigned long long tot = 0;
        for (int i = 0; i < n; ++i) tot += c[i];
        CHECK(tot == (unsigned long long)h * w);
    }
    // 3. hostile counts strings: every byte value, endless continuation groups, truncation, huge deltas -- an error code, never UB
    std::vector<uint32_t> c(64);
    int n = 0;
    for (int b = 0; b < 256; ++b) {
        char s1[1] = {(char)b};
        (void)amp_rle_from_string(s1, 1, c.data(), 64, &n);
    }
    std::string endless(40, (char)(48 + 0x3f));                           // continuation bit set on all 40 groups
    CHECK(amp_rle_from_string(endless.data(), endless.size(), c.data(), 64, &n) != AMP_OK);
    std::string trunc = std::string(1, (char)(48 + 0x21));                // "more" but nothing follows
    CHECK(amp_rle_from_string(trunc.data(), trunc.size(), c.data(), 64, &n) != AMP_OK);
    std::string neg = std::string(1, (char)(48 + 0x10));                  // a negative first run
    CHECK(amp_rle_from_string(neg.data(), neg.size(), c.data(), 64, &n) != AMP_OK);
    for (int it = 0; it < 2000; ++it) {                                   // random byte soup
        char buf[24];
        const int l = rnd() % 24;
        for (int i = 0; i < l; ++i) buf[i] = (char)(rnd() % 256);
        (void)amp_rle_from_string(buf, (size_t)l, c.data(), 64, &n);
    }
    CHECK(amp_rle_from_string("0", 1, c.data(), 0, &n) != AMP_OK);       // cap = 0
    // decode with runs that overshoot h*w must be refused
    uint32_t big[2] = {10, 1000};
    std::vector<uint8_t> small(12);
    CHECK(amp_rle_decode(big, 2, 3, 4, small.data()) != AMP_OK);
    printf("RLE SANITIZE OK\n");
    return 0;
}
