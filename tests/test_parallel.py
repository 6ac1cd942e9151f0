"""N>1 path on CPU: world_size-2 gloo processes run the flat-buffer gradient all-reduce and the
slide sharding helpers (the GPU path uses the same code with backend nccl == RCCL)."""
import os
import socket

import json

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dgdm_histopath_lab_amd.parallel import BalancedSlideLoader, FlatGradAllReducer, balance_slides, shard_slides, slide_cost


def free_port() -> int:
    """A port the OS hands out (bind to 0) -- no arithmetic on the pid that two concurrent runs could share."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(target, world, *extra):
    """Spawn `world` ranks, collect ONE plain-python result per rank (no tensors on the queue: a tensor travels as a shared-memory
    handle that needs its producer alive), then join."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q, *extra)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return sorted(res, key=lambda t: t[0])


class Tiny(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 4)
        self.dead = torch.nn.Linear(3, 3)      # never used: must not enter the buffer (D9)
        self.b = torch.nn.Linear(4, 1, bias=False)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = Tiny()
    red = FlatGradAllReducer(m, world)
    data = torch.arange(24, dtype=torch.float32).view(4, 6) / 10.0
    mine = data[list(shard_slides(4, rank, world))]
    for _ in range(3):  # later steps reuse the buffer and start bucket 0 under the backward
        m.zero_grad(set_to_none=True)
        m(mine).sum().backward()
        red.all_reduce()
    views = all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(red.live, red.views))     # .grad IS the buffer: nothing copied back
    order = [k for p in red.live for k, q_ in m.named_parameters() if q_ is p]
    q.put((rank, {k: p.grad.tolist() for k, p in m.named_parameters() if p.grad is not None}, red.nbytes, red.bucket_nbytes,
           dict(red.stats), views, order))
    dist.destroy_process_group()


def test_flat_grad_all_reduce_two_ranks_gloo():
    res = run_ranks(_worker, 2)
    torch.manual_seed(0)
    m = Tiny()
    data = torch.arange(24, dtype=torch.float32).view(4, 6) / 10.0
    (m(data[:2]).sum() / 2 + m(data[2:]).sum() / 2).backward()  # mean over ranks of per-rank sums
    for rank, grads, nbytes, buckets, stats, views, order in res:
        assert set(grads) == {"a.weight", "a.bias", "b.weight"}          # dead params stay out
        assert nbytes == 4 * (24 + 4 + 4) and sum(buckets) == nbytes and len(buckets) == 2 and min(buckets) > 0
        assert order[0] == "b.weight"                                     # laid out in completion order of the backward: last layer first
        assert stats == {"early_launches": 2, "steps": 3} and views      # steps 2 and 3 started bucket 0 from the gradient hook
        for k, g in grads.items():
            torch.testing.assert_close(torch.tensor(g), dict(m.named_parameters())[k].grad, rtol=1e-6, atol=1e-7)


def _phase_worker(rank, world, port, q):
    """A phase switch changes which parameters carry gradients: the reducer raises unless it is reset on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = Tiny()
    red = FlatGradAllReducer(m, world)
    x = torch.full((2, 6), float(rank + 1))
    m(x).sum().backward()
    red.all_reduce()
    first = red.nbytes
    m.zero_grad(set_to_none=True)
    (m(x).sum() + m.dead(torch.ones(1, 3)).sum()).backward()      # `dead` is live now
    try:
        red.all_reduce()
        raised = False
    except RuntimeError:
        raised = True
    red.reset()
    red.all_reduce()
    q.put((rank, raised, first, red.nbytes, m.dead.bias.grad.tolist()))
    dist.destroy_process_group()


def test_reducer_reset_on_phase_switch_two_ranks_gloo():
    res = run_ranks(_phase_worker, 2)
    for rank, raised, first, second, dead_bias_grad in res:
        assert raised and first == 4 * (24 + 4 + 4) and second == first + 4 * (9 + 3)
        assert dead_bias_grad == [1.0, 1.0, 1.0]       # identical on both ranks: mean == value


def _abandon_worker(rank, world, port, q):
    """A step abandoned between backward and all_reduce() (exception, skipped non-finite step), then a normal step, then a step
    with gradient accumulation (two backwards before one all_reduce()): ADVICE r2 (low) -- the early-launch state is reset per step
    and accumulation runs without overlap."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    m = Tiny()
    red = FlatGradAllReducer(m, world)
    data = torch.arange(48, dtype=torch.float32).view(8, 6) / 10.0
    mine = data[list(shard_slides(8, rank, world))]            # 4 rows per rank
    m.zero_grad(set_to_none=True); m(mine).sum().backward(); red.all_reduce()           # step 1: learns the layout
    m.zero_grad(set_to_none=True); m(mine * 3.0).sum().backward()                      # step 2: bucket 0 leaves early ... and the step is dropped
    early_after_abandoned = red.stats["early_launches"]
    m.zero_grad(set_to_none=True); m(mine).sum().backward(); red.all_reduce()           # step 3: a normal step again (launches early)
    g3 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    early_after_normal = red.stats["early_launches"]
    m.zero_grad(set_to_none=True)                                                       # step 4: two micro-batches, one exchange
    m(mine[:2]).sum().backward()
    m(mine[2:]).sum().backward()
    red.all_reduce()
    g4 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m(mine[:2]).sum().backward()                                                        # step 5: accumulate INTO the flat buffer's views
    m(mine[2:]).sum().backward()
    red.all_reduce()
    g5 = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    q.put((rank, early_after_abandoned, early_after_normal, red.stats["early_launches"], {k: v.tolist() for k, v in g3.items()},
           {k: v.tolist() for k, v in g4.items()}, {k: v.tolist() for k, v in g5.items()}))
    dist.destroy_process_group()


def test_reducer_abandoned_step_and_gradient_accumulation_two_ranks_gloo():
    res = run_ranks(_abandon_worker, 2)
    torch.manual_seed(0)
    m = Tiny()
    data = torch.arange(48, dtype=torch.float32).view(8, 6) / 10.0
    (m(data).sum() / 2).backward()          # mean over the two ranks of the per-rank sums
    want = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    for rank, e_ab, e_norm, e_end, g3, g4, g5 in res:
        # steps 2 and 3 launched early; in the accumulating steps 4 and 5 the first micro-batch's launch is waited for and discarded at
        # the second forward (overlap off for the rest of the step): correct values below are what matters
        assert e_ab == 1 and e_norm == 2 and e_end == 4
        for k in want:
            torch.testing.assert_close(torch.tensor(g3[k]), want[k], rtol=1e-6, atol=1e-6)
            torch.testing.assert_close(torch.tensor(g4[k]), want[k], rtol=1e-6, atol=1e-6)
            # step 5 added the same local gradients onto step 4's averaged ones (in the buffer) and averaged again
            torch.testing.assert_close(torch.tensor(g5[k]), 2 * want[k], rtol=1e-6, atol=1e-6)


def test_single_rank_always_flag_runs_the_collective():
    """`always=True` (bench rehearsal knob) must not return early with one rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        m = Tiny()
        m(torch.ones(2, 6)).sum().backward()
        g = m.a.weight.grad.clone()
        red = FlatGradAllReducer(m, 1, always=True)
        red.all_reduce()
        assert red.nbytes == 4 * (24 + 4 + 4) and torch.equal(m.a.weight.grad, g)
        lazy = FlatGradAllReducer(m, 1)
        lazy.all_reduce()
        assert lazy.nbytes == 0
    finally:
        dist.destroy_process_group()


def test_sharding_helpers():
    assert [list(shard_slides(32, r, 8)) for r in range(8)][3] == [12, 13, 14, 15]
    assert sum(len(shard_slides(10, r, 4)) for r in range(4)) == 10 and list(shard_slides(10, 3, 4)) == [8, 9]
    costs = [slide_cost(n, 5 * n) for n in (10000, 1000, 9000, 2000, 8000, 3000, 7000, 4000)]
    bins = balance_slides(costs, 4)
    loads = [sum(costs[i] for i in b) for b in bins]
    assert sorted(i for b in bins for i in b) == list(range(8))
    assert max(loads) / (sum(loads) / 4) < 1.25   # attention-dominated N^2 costs still balance


def test_balanced_slide_loader_partitions_every_step_and_balances_the_cost():
    """configs[4] shape: slides of 1k..10k nodes (E = 5 N) in global batches of 32 over 8 ranks: every slide of a step goes to
    exactly one rank, all ranks derive the same plan, the worst rank load stays within 25 % of the mean (attention cost ~ N^2),
    and a naive equal-count split of the same steps does not."""
    from dgdm_histopath_lab_amd import GraphData
    g = torch.Generator().manual_seed(77)
    ns = torch.randint(1000, 10001, (8 * 32,), generator=g).tolist()
    slides = [GraphData(x=torch.empty(n, 1), edge_index=torch.empty(2, 5 * n, dtype=torch.long)) for n in ns]
    loaders = [BalancedSlideLoader(slides, 32, 8, r) for r in range(8)]
    plans = [ld.plan() for ld in loaders]
    assert all(p == plans[0] for p in plans) and len(plans[0]) == len(loaders[0]) == 8
    for s, bins in enumerate(plans[0]):
        assert sorted(i for b in bins for i in b) == list(range(32 * s, 32 * (s + 1))) and all(len(b) >= 1 for b in bins)
    assert loaders[0].max_over_mean_load() < 1.25
    naive = max(max(sum(slide_cost(ns[i], 5 * ns[i]) for i in range(32 * s + 4 * r, 32 * s + 4 * r + 4)) for r in range(8)) /
                (sum(slide_cost(ns[i], 5 * ns[i]) for i in range(32 * s, 32 * s + 32)) / 8) for s in range(8))
    assert naive > 1.25
    b0 = next(iter(BalancedSlideLoader([GraphData(x=torch.randn(n, 4), edge_index=torch.randint(0, n, (2, 3 * n))) for n in (5, 9, 7, 3)], 4, 2, 1)))
    assert b0.num_graphs == 2 and b0.x.size(0) in (12, 14, 16, 10, 8)


def _rehearsal_case(name):
    from conftest import BENCH_REHEARSAL
    if not BENCH_REHEARSAL:
        pytest.skip("rehearsal not started (run as `pytest -m gpu` on a GPU box)")
    proc, d = BENCH_REHEARSAL["proc"], BENCH_REHEARSAL["dir"]
    try:
        proc.wait(1500)
    except Exception:
        proc.kill()
        raise
    done = os.path.join(d, "done.json")
    assert os.path.exists(done), f"rehearsal driver died:\n{open(os.path.join(d, 'driver.log')).read()[-3000:]}"
    st = json.load(open(done))[name]
    out, err = open(os.path.join(d, f"{name}.out")).read(), open(os.path.join(d, f"{name}.err")).read()
    assert st["rc"] == 0, f"{st['cmd']} exited with {st['rc']}:\n{err[-3000:]}"
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must hold exactly one line, got {len(lines)}:\n{out[:2000]}"
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("form", ["plain", "launcher"])
def test_bench_two_rank_rehearsal_prints_one_json_line(form):
    """The N > 1 branch of bench.py (init_process_group, split recording around the collective, per-rank timing, max over ranks,
    ONE JSON line from rank 0) run unattended as fresh child processes started before this session touched the GPU
    (DGDM_BENCH_ONE_DEVICE=1: both ranks on the box's one GPU; gloo, since RCCL refuses two ranks on one device).
    `plain`: `python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE -- the script starts its own
    ranks (the reference: pl.Trainer(devices=gpus), cli/train.py:346-359), relays the line and the return code; `launcher`:
    `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2 ...`, the form the driver uses."""
    r = _rehearsal_case(form)
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["unit"] == "slides/s"
    assert r["config"]["global_batch"] == 8 and r["config"]["parallelism"] == "dp2"
    assert r["value"] > 0 and abs(r["value"] - 8 * 1e3 / r["ms_per_step"]) < 1e-2 * r["value"]
    pr, ge = r["per_rank"], r["gradient_exchange"]
    assert len(pr["ms_per_step_by_rank"]) == 2 and pr["ms_per_step_max"] <= r["ms_per_step"] * 1.001
    assert ge["bytes"] > 1e6 and ge["messages_per_step"] in (1, 2) and sum(ge["bucket_bytes"]) == ge["bytes"]
    assert "HIP graph replay" in r["config"]["launch"], r["config"]["launch"]      # the split recording was taken, not the eager fallback


@pytest.mark.gpu
def test_bench_two_rank_rehearsal_of_the_mixed_stream_replays_graphs():
    """`python bench.py --gpus 2 --mixed` (BASELINE configs[4] shape) -- every layout of the stream recorded
    under the gradient reducer (training.GraphedStepCache on one flat buffer), one message per step."""
    r = _rehearsal_case("mixed")
    assert r["n_gpus"] == 2 and r["config"]["global_batch"] == 4 and "MIXED-SIZE STREAM" in r["config"]["workload"]
    assert "HIP graph replay, one recording per layout (8 layouts" in r["config"]["launch"] and "all-reduce" in r["config"]["launch"]
    assert r["gradient_exchange"]["messages_per_step"] == 1 and r["gradient_exchange"]["early_launches"] == 0
    assert r["value"] > 0 and len(r["per_rank"]["ms_per_step_by_rank"]) == 2 and r["config"]["max_over_mean_rank_load"] < 1.3


def test_bench_starts_its_own_ranks_and_relays_their_failure_without_a_gpu():
    """Host logic of `python bench.py --gpus N` without a launcher (CPU container: no GPU, so the ranks can only fail): the parent
    refuses a node with fewer GPUs than ranks (rc 2, nothing on stdout), and with the one-device rehearsal knob it starts
    `torch.distributed.run`, whose ranks exit with bench.py's "needs a GPU" -- return code and empty stdout are relayed."""
    import subprocess
    import sys
    if torch.cuda.device_count() > 0:
        pytest.skip("CPU-side check (the GPU run has the rehearsal tests)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DGDM_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 2 and r.stdout == "" and "shows 0 GPU" in r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       env=dict(env, DGDM_BENCH_ONE_DEVICE="1"), timeout=300)
    assert r.returncode != 0 and r.stdout == "" and "needs a GPU" in r.stderr
