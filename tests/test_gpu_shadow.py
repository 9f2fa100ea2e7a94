"""CPU: the bodies of the `-m gpu` tests with the CPU twin standing in for the HIP library (conftest.py, CFX_SHADOW_GPU).
Says nothing about the kernels; says that no GPU-only test has gone stale against the host it calls through — the one way
a round's GPU suite turned red without a kernel being wrong (round 3)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_test_bodies_hold_on_the_twin():
    env = dict(os.environ, CFX_SHADOW_GPU="1")
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests"), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider"]
    try:
        import pytest_timeout  # noqa: F401  (optional plugin: without it the option would be a usage error)
        cmd += ["--timeout", "300"]
    except ImportError:
        pass
    try:
        import xdist  # noqa: F401
        cmd += ["-n", "4"]
    except ImportError:
        pass
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    tail = "\n".join((r.stdout + r.stderr).strip().splitlines()[-25:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail
