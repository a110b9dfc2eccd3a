"""CPU: the oracle's C sources under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 lists sanitizer
runs among the reference's auxiliary checks; the GPU pool has no sanitizer support, so the CPU restatement - the code every
GPU result is compared with - is what gets them).  A child python preloads libasan, loads oracle/liboracle_asan.so through
MXX_ORACLE_LIB and runs the oracle's own test files; any report aborts the child (-fno-sanitize-recover)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_tests_pass_under_asan_and_ubsan():
    from oracle import oracle as O

    lib = O.build_sanitized()
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    assert os.path.isabs(libasan), "gcc has no libasan"
    env = dict(os.environ, MXX_ORACLE_LIB=lib, LD_PRELOAD=libasan, OMP_NUM_THREADS="2",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle.py"), os.path.join(ROOT, "tests", "test_oracle_sampling.py"),
                          "-k", "not sanitized"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert " passed" in out.stdout
