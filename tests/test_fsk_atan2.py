"""fsk_atan2f (radiosonde_auto_rx_amd/csrc/sonde_fsk_dev.h): the fine-timing angle of the modem (fsk.c:705, `atan2f` of the timing sum) as ONE dependent chain — few registers on the
device — must return what `(float)atan2((double)y, (double)x)` returns, bit for bit, signed zeros and axes included: the modem's soft decisions and frame lengths hang on it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#define SONDE_FSK_EMU
#include "%s/radiosonde_auto_rx_amd/csrc/sonde_fsk_dev.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
static int same(float a, float b) { unsigned ua, ub; memcpy(&ua, &a, 4); memcpy(&ub, &b, 4); return ua == ub; }
int main() {
    long bad = 0, n = 0;
    const float sp[] = { 0.f, -0.f, 1.f, -1.f, 1e-30f, -1e-30f, 3e38f, -3e38f, 0.41421357f, 2.4142137f, 1e-45f };
    for (float y : sp) for (float x : sp) { n++; if (!same(fsk_atan2f(y, x), (float)std::atan2((double)y, (double)x))) { bad++; printf("special %%g %%g\n", y, x); } }
    std::mt19937_64 g(7); std::uniform_real_distribution<double> u(-1, 1);
    for (long i = 0; i < 3000000; i++) {
        const float y = (float)(u(g) * std::pow(10.0, u(g) * 12)), x = (float)(u(g) * std::pow(10.0, u(g) * 12));
        n++; if (!same(fsk_atan2f(y, x), (float)std::atan2((double)y, (double)x))) { bad++; if (bad < 5) printf("%%.9g %%.9g\n", y, x); }
    }
    for (long i = 0; i < 1000000; i++) {        // near the range-reduction boundaries |y| / |x| = tan(pi/8), 1, 1 / tan(pi/8)
        const float x = (float)(u(g) * 1000.0), r = (float)(((i %% 3) == 0 ? 0.41421356 : (i %% 3) == 1 ? 1.0 : 2.41421356) * (1.0 + 1e-6 * u(g))), y = x * r * ((i & 8) ? -1.f : 1.f);
        n++; if (!same(fsk_atan2f(y, x), (float)std::atan2((double)y, (double)x))) { bad++; if (bad < 5) printf("b %%.9g %%.9g\n", y, x); }
    }
    printf("%%ld %%ld\n", bad, n);
    return bad != 0;
}
'''


def test_fsk_atan2f_equals_libm_double_rounded_to_float(tmp_path):
    src = tmp_path / "at.cpp"
    src.write_text(SRC % ROOT)
    exe = tmp_path / "at"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-o", str(exe), str(src)])
    r = subprocess.run([str(exe)], capture_output=True, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-500:]
    bad, n = r.stdout.decode().split()[-2:]
    assert int(bad) == 0 and int(n) > 4_000_000
