"""Child process of a `-m gpu` test session (started by conftest.pytest_configure before the session touches the GPU): runs the
N > 1 forms of bench.py one after the other as FRESH processes -- this script itself never initialises HIP -- with both ranks on
the box's one GPU over gloo (RCCL refuses two ranks on one device), and leaves per-case stdout / stderr / return code in the
directory given as argv[1].  tests/test_parallel.py waits for `done.json` and checks the lines.

    plain      python bench.py --gpus 2 ...            (no launcher, no WORLD_SIZE: bench.py starts its own ranks)
    mixed      python bench.py --gpus 2 --mixed ...    (configs[4] shape: recorded steps per layout under the gradient reducer)
    launcher   python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...    (the form the driver uses)
"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def main(out_dir):
    env = dict(os.environ, DGDM_BENCH_ONE_DEVICE="1", DGDM_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    bench = os.path.join(ROOT, "bench.py")
    small = ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-gather"]
    cases = {
        "plain": [sys.executable, bench, "--gpus", "2", "--nodes", "500", "--edges", "2000", *small],
        "mixed": [sys.executable, bench, "--gpus", "2", "--mixed", "--batch", "2", *small],
        "launcher": [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(free_port()), bench, "--gpus", "2", "--nodes", "500", "--edges", "2000", *small],
    }
    status = {}
    for name, cmd in cases.items():
        with open(os.path.join(out_dir, f"{name}.out"), "w") as o, open(os.path.join(out_dir, f"{name}.err"), "w") as e:
            try:
                rc = subprocess.run(cmd, stdout=o, stderr=e, env=env, cwd=ROOT, timeout=600).returncode
            except subprocess.TimeoutExpired:
                rc = -9
        status[name] = {"rc": rc, "cmd": " ".join(cmd)}
        with open(os.path.join(out_dir, "progress.json"), "w") as f:
            json.dump(status, f)
    os.replace(os.path.join(out_dir, "progress.json"), os.path.join(out_dir, "done.json"))


if __name__ == "__main__":
    main(sys.argv[1])
