"""bench.py contract checks on a GPU box: `--gpus N` without a launcher starts N ranks (here two gloo ranks sharing the one
GPU of the test box, small trace), every rank proves a different witness and verifies its own proof, rank 0 prints ONE JSON
line with n_gpus = N."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_two_ranks_started_by_bench_itself():
    d = _run(["--gpus", "2", "--force-device", "0", "--dist-backend", "gloo", "--log-n", "12", "--steps", "2", "--warmup", "1",
              "--no-variants", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["verified"] is True and d["steps"] == 2
    assert d["value"] > 0 and abs(d["value"] - 2 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]      # whole-job throughput
    assert d["scaling"] == "weak" and d["vs_baseline"] is None


def test_four_ranks_share_the_circuit_file():
    """The real-circuit bench on several ranks (here four gloo ranks on the one GPU of the box; the box allows six GPU processes, the
    test runner is one): rank 0 builds the 2^17-row circuit and writes it as a hand-off file, ranks 1..3 map it -- one Python build
    instead of four -- and every rank verifies its own, different proof."""
    d = _run(["--gpus", "4", "--force-device", "0", "--dist-backend", "gloo", "--log-n", "17", "--steps", "1", "--warmup", "1",
              "--no-variants", "--no-cpu-baseline"])
    assert d["n_gpus"] == 4 and d["verified"] is True and d["signatures_per_proof"] == 1
    assert "mapped it" in d["config"]["circuit_hand_off"]


def test_single_rank_line_has_roofline_verified_and_variants():
    d = _run(["--log-n", "13", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"])
    assert d["n_gpus"] == 1 and d["verified"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    v = d["variants"]
    assert v["witness_from_host_memory"]["same_proof_as_resident"] is True
    assert v["gpu_witness_fill_advice_columns"]["same_proof_after_fill"] is True


def test_headline_line_on_the_real_circuit():
    """--log-n 17: the smallest instance of the real secp256k1 circuit (one signature per proof) through the default code path"""
    d = _run(["--log-n", "17", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"])
    assert d["verified"] is True and d["config"]["signatures_per_proof"] == 1 and "98 687 rows per signature" in d["config"]["workload"]
    assert d["variants"]["gate_mix_stand_in"]["verified"] is True
    assert d["variants"]["gpu_witness_fill_advice_columns"]["same_proof_after_fill"] is True


def test_zkdsa_batch_workload_line():
    d = _run(["--workload", "zkdsa-batch", "--batch", "64", "--sub-batch", "32", "--threads", "2", "--steps", "1", "--warmup", "1"])
    assert d["verified"] is True and d["value"] > 0 and "glp_prove_batch" in d["config"]["workload"]


def test_keccak256_workload_line():
    """BASELINE config 2 through bench.py: the real Keccak-256 circuit, every proof verified and its public inputs = the digest."""
    d = _run(["--workload", "keccak256", "--batch", "4", "--steps", "1", "--warmup", "1", "--hasher", "keccak"])
    assert d["verified"] is True and d["value"] > 0 and "KeccakGoldilocksConfig" in d["config"]["workload"]
    assert "2^13 rows" in d["config"]["workload"]


def test_smt_workload_line():
    """BASELINE config 4 through bench.py: the real 16-level inclusion circuit, every proof verified and bound to the tree's root."""
    d = _run(["--workload", "smt", "--batch", "32", "--sub-batch", "16", "--steps", "1", "--warmup", "1"])
    assert d["verified"] is True and d["value"] > 0 and "PoseidonGate x22" in d["config"]["workload"]      # 20 of the gadget + 2 for the 12 public inputs' hash


def test_c99_consumer_proves_the_sample_circuit_file(tmp_path):
    """A plain C program against include/glp.h (csrc/examples/abi_smoke.c): reads tests/golden/zkdsa_2_3.glpc, creates the circuit,
    proves the file's witness and verifies the proof -- the path a Rust / Go host takes, without Python in between."""
    pkg = os.path.join(ROOT, "plonky2-lib_amd")
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(pkg, "csrc", "examples", "abi_smoke.c"), "-L", pkg, "-lglprover", "-Wl,-rpath," + pkg, "-o", exe])
    r = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "zkdsa_2_3.glpc")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "circuit file ok: 2^3 rows, 135 wires" in r.stdout and "proved and verified" in r.stdout and "abi_smoke ok" in r.stdout
    assert "staged proof equal; batch verifier: ok, ok, rejected (" in r.stdout
