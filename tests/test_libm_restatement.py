"""csrc/libm_flt32.h restates the host C library's single-precision powf / logf (glibc >= 2.28) so that the device
reproduces the reference's std::pow(float,float) / std::log(float) results bit for bit.  The header is compiled for the
host here and compared with the host libm on 10^7 arguments (oracle/libm_check.cpp); the GPU tests then check the device
evaluation of the same header through a9 (dependent error probabilities bit-identical to the oracle's libm calls)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_restatement_matches_host_libm(tmp_path):
    exe = str(tmp_path / "libm_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "oracle", "libm_check.cpp"), "-lm"],
                   check=True)
    out = subprocess.run([exe, "10000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "powf mismatches 0 logf mismatches 0 expf mismatches 0 log1pf mismatches 0 fallbacks 0" in out.stdout
