"""CPU suite, part 4: the FP64 log/exp/pow/acos used by the HIP day step (splash_math.hpp is
plain C++), compiled with g++ and checked against libm: < 1 ulp (log, exp, acos), special values
(pow(0,y)=0, pow(x<0,y)=NaN, exp(-inf)=0 ...) that the reference's NaN-as-control-flow needs."""
import os
import subprocess
import tempfile

from conftest import ROOT


def test_device_math_against_libm():
    src = os.path.join(ROOT, "tests", "csrc", "test_math.cpp")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "test_math")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        r = subprocess.run([exe], capture_output=True, text=True)
        print(r.stdout)
        assert r.returncode == 0, r.stdout + r.stderr
