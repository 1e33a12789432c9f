"""The open-launch protocol (host appends jobs to a kernel that is running; the kernel closes itself when it runs dry;
a Dekker-style handshake decides every append one way or the other) raced on the CPU: tests/protocol_model restates the
two sides' steps with the orders and memory orders of csrc/rl_api.hip (session_append) and csrc/rl_kernels.hip.h (the
OPEN variant's refill) and lets real threads run them against each other -- tens of thousands of appends that arrive
while a launch is closing.  The GPU legs of the same protocol are tests/test_gpu_multi.py and tools/open_launch_stress.py."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "protocol_model", "open_protocol_model.cpp")
EXE = os.path.join(HERE, "protocol_model", "_build", "open_protocol_model")


@pytest.fixture(scope="module")
def model():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < os.path.getmtime(SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-o", EXE, SRC], check=True)
    return EXE


@pytest.mark.parametrize("callers,per_caller", [(2, 10000), (8, 5000), (16, 2500)])
def test_every_job_is_handed_out_exactly_once_by_the_launch_that_accepted_it(model, callers, per_caller):
    run = subprocess.run([model, str(callers), str(per_caller)], capture_output=True, timeout=300)
    out = run.stdout.decode()
    assert run.returncode == 0 and out.startswith("ok: %d jobs" % (callers * per_caller)), out + run.stderr.decode()
    launches = int(out.split(" over ")[1].split()[0])
    turned_down = int(out.split("launches, ")[1].split()[0])
    assert 1 <= launches <= callers * per_caller and turned_down <= launches - 1   # a launch begins with a refusal (or a full job table)
