"""Sanitizer run of the CPU restatement ("ASan on the CPU restatement"; GPU ASan is not available on the
pool): oracle/liboracle_asan.so is the same dn_oracle.c built with -fsanitize=address,undefined, and the golden-fixture
replays of tests/test_oracle_golden.py run under it in a child process with libasan preloaded.  Any out-of-bounds access,
use of uninitialised stack, signed overflow or misaligned access in the restatement aborts that child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_golden_replays_are_clean_under_asan_and_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("gcc has no libasan.so here")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_asan.so"])
    env = dict(os.environ, DN_ORACLE_LIB="liboracle_asan.so", LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    probe = subprocess.run([sys.executable, "-c", "from oracle import oracle as O; O.lib(); m = open('/proc/self/maps').read(); "
                            "print('liboracle_asan.so' in m and 'libasan' in m and 'liboracle.so' not in m)"],
                           env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert probe.stdout.strip() == "True", probe.stdout + probe.stderr      # the child really runs the sanitizer build
    # the fixtures that drive the most code: a scripted sequence (reward / gates / resets), a closed-loop trajectory with the
    # normaliser, the action chain, the PID family and random spawn
    sel = "test_scripted_teacher_forced or test_closed_loop_vec_env or test_action_chain_bit_exact or " \
          "test_pid_family_action_types_match_reference or test_random_spawn_geometry_matches_reference or test_normalize_observation"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_golden.py"), "-x", "-q", "-k", sel,
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
