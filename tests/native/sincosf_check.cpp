// tests/test_libm_sincosf.py builds and runs this on the CPU: csrc/libm_sincosf.hpp against the system's sinf / cosf, bit for
// bit.  The library's Line end points and Gaussian rotations rest on that identity.
#include "libm_sincosf.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static float from_bits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

int main() {
    long checked = 0, bad = 0;
    auto check = [&](float y) {
        float s, c;
        pcrhip::libm::sincosf(y, s, c);
        const float rs = sinf(y), rc = cosf(y);
        const bool same = (bits(s) == bits(rs) || (s != s && rs != rs)) && (bits(c) == bits(rc) || (c != c && rc != rc));
        ++checked;
        if (!same && bad++ < 10) std::printf("y = %.9g (0x%08x): sin %.9g vs %.9g, cos %.9g vs %.9g\n", y, bits(y), s, rs, c, rc);
    };
    std::mt19937_64 rng(5);
    // every float of [pi/4, 2 pi) in steps of 5 (4.6e6 values), of [2^-13, pi/4) in steps of 97, and of [100, 140) (the seam at 120)
    for (uint32_t u = bits(0.78539816f); u < bits(6.2831855f); u += 5) { check(from_bits(u)); check(-from_bits(u)); }
    for (uint32_t u = bits(1.2e-4f); u < bits(0.78539816f); u += 97) { check(from_bits(u)); check(-from_bits(u)); }
    for (uint32_t u = bits(100.0f); u < bits(140.0f); u += 3) check(from_bits(u));
    // the neighbourhoods of the multiples of pi/2 (where the reduction cancels)
    for (int k = 1; k < 80; ++k) {
        const uint32_t mid = bits((float)(k * 1.5707963267948966));
        for (uint32_t u = mid - 20000; u < mid + 20000; ++u) check(from_bits(u));
    }
    // random arguments of every magnitude, both signs; specials
    std::uniform_int_distribution<uint32_t> any(0u, 0xffffffffu);
    for (int i = 0; i < 12000000; ++i) check(from_bits(any(rng)));
    std::uniform_real_distribution<float> mid(-1000.f, 1000.f);
    for (int i = 0; i < 4000000; ++i) check(mid(rng));
    for (float y : {0.0f, -0.0f, INFINITY, -INFINITY, NAN, 120.0f, 3.4e38f, 1e-45f, 0.78539819f, 0.78539813f}) check(y);
    std::printf("checked %ld arguments, %ld differ\n", checked, bad);
    return bad ? 1 : 0;
}
