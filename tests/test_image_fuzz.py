"""Corrupt-blob robustness of the REDA reader, under AddressSanitizer + UBSan (CPU build; the
GPU kernels chase the image's indices unchecked, so the reader is where corruption must stop)."""
import os
import subprocess

from golden_util import GOLD

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reader_survives_corrupt_blobs(tmp_path):
    exe = str(tmp_path / "image_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cpp", "image_fuzz.cpp"),
                    os.path.join(ROOT, "one_amd", "csrc", "dfa_image.cpp"), "-o", exe], check=True)
    blobs = [os.path.join(GOLD, "dfas", n + ".reda")
             for n in ("err", "uri", "num3", "newyork", "aab", "log100", "set5")]
    out = subprocess.run([exe, "1500"] + blobs, capture_output=True, text=True,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "image fuzz ok" in out.stdout
