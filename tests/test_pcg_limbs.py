"""pcg64_dev.h pcg_muladd (the generator's 128-bit step written limb by limb for the device compiler) == the unsigned __int128 expression, on the host."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r"""
#include <cstdio>
#include <cstdint>
#include <random>
#include "pcg64_dev.h"
int main() {
    std::mt19937_64 g(1);
    long bad = 0;
    for (long n = 0; n < 4000000; n++) {
        uint64_t w[6];
        for (auto &x : w) {
            x = g();
            const int k = (int)(g() % 8);
            if (k == 0) x = ~0ull; else if (k == 1) x = 0; else if (k == 2) x |= 0xffffffff00000000ull; else if (k == 3) x |= 0xffffffffull;
        }
        const mi::u128 m = mi::make_u128(w[0], w[1]), p = mi::make_u128(w[2], w[3]), s = mi::make_u128(w[4], w[5]);
        if (mi::pcg_muladd(s, m, p) != (mi::u128)(m * s + p)) bad++;
        mi::Pcg64 a, b;  // the generator's own step, both ways
        a.state = b.state = s, a.inc = b.inc = p | 1u;
        a.step();
        b.state = b.state * mi::pcg_mult() + b.inc;
        if (a.state != b.state) bad++;
    }
    std::printf("bad %ld\n", bad);
    return bad != 0;
}
"""


def test_limbwise_step_equals_int128(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "gymnasium_amd", "csrc"), "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "bad 0", out.stdout + out.stderr
