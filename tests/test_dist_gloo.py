"""N>1 plumbing on CPU: two gloo ranks run the same timing harness bench.py uses (barrier, K timed
steps, MAX over ranks) with a stand-in step, and the proof sharding covers every proof exactly once.
The GPU step itself is covered by the -m gpu tests; no collective exists on the data path."""
import os
import socket
import time

import torch.multiprocessing as mp

import plonky2_lib_amd.dist as gdist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    g = gdist.init_from_env(use_cuda=False)
    done = []

    def step():
        time.sleep(0.02 * (rank + 1))          # rank 1 is the slow one
        done.append(1)
    dt = gdist.timed_steps(g, step, steps=3, warmup=1)
    mine = list(gdist.proofs_for_rank(7, rank, world))
    out.put((rank, dt, len(done), mine))
    g.close()


def test_two_rank_timing_and_sharding():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, dt0, n0, m0), (r1, dt1, n1, m1) = res
    assert n0 == n1 == 4                        # 1 warmup + 3 timed on every rank
    assert abs(dt0 - dt1) < 1e-9                # both ranks report the MAX
    assert dt0 >= 3 * 0.04 * 0.9                # ... which is the slow rank's time
    assert sorted(m0 + m1) == list(range(7)) and abs(len(m0) - len(m1)) <= 1


def _share_worker(rank, world, port, path, out):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      OMP_NUM_THREADS="1")
    import numpy as np
    import plonky2_lib_amd.synth as synth
    from oracle import oracle
    g = gdist.init_from_env(use_cuda=False)
    built = []

    def build():
        built.append(rank)
        return synth.arith_circuit(6, synth.Config.standard_recursion_config(), seed=3)
    desc, cf, t_build = gdist.shared_circuit(g, build, path, seed=77)
    oc = oracle.OracleCircuit(desc)                               # CPU checker: the rank's own witness must be a valid witness
    rc, proof = oc.prove(wires=np.ascontiguousarray(desc.wires))
    ok = rc == 0 and oc.verify(proof) == 0
    out.put((rank, built, bool(ok), np.asarray(desc.wires).tobytes(), np.asarray(desc.sigmas).tobytes(), proof.tobytes(), os.path.exists(path)))
    if cf is not None:
        desc.wires = None
        cf.close()
    g.close()


def test_eight_ranks_share_one_circuit_file(tmp_path):
    """bench.py on a multi-GPU node: rank 0 builds the circuit and writes it once as a hand-off file, ranks 1..7 map it (no second
    build) and put rank-specific values into the unconstrained padding rows: eight different, valid witnesses of one circuit; the
    file's name is gone once everyone has mapped it."""
    world, port = 8, _free_port()
    path = str(tmp_path / "shared.glpc")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_share_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [[0]] + [[]] * 7                 # only rank 0 ran build()
    assert all(r[2] for r in res)                                  # every rank's witness proves and verifies (oracle)
    assert len({r[3] for r in res}) == 8 and len({r[5] for r in res}) == 8      # eight different witnesses, eight different proofs
    assert len({r[4] for r in res}) == 1                           # one circuit
    assert not any(r[6] for r in res) and not os.path.exists(path)


def test_single_rank_needs_no_process_group():
    g = gdist.Group()
    assert gdist.timed_steps(g, lambda: None, 2, 1) >= 0.0
    assert list(gdist.proofs_for_rank(5, 0, 1)) == [0, 1, 2, 3, 4]


def test_bench_gpus_flag_is_honoured_without_a_launcher():
    """`python bench.py --gpus N` (no torch.distributed.run around it) must start N ranks itself or fail loudly -- never run one
    rank and report n_gpus = 1.  Without N visible GPUs that is an error naming the flag."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    # the cheapest workload there is (2^12-row stand-in, one step, no CPU baseline, no variants): on a host that does have three
    # GPUs this unmarked test must not run the headline bench on them for minutes
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--no-cpu-baseline", "--no-variants", "--circuit", "stand-in",
                        "--log-n", "12", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    import torch
    if torch.cuda.device_count() < 3:
        assert r.returncode != 0 and "--gpus 3" in r.stderr and "visible" in r.stderr
    assert '"n_gpus": 1' not in r.stdout
