"""Data-parallel step on the GPU with MORE THAN ONE PROCESS (BASELINE config 4's logic on the hardware that exists: one
MI355X).  Two fresh child processes share cuda:0 over gloo and run the real CSTS `SegmentedTrainStep` -- HIP-graph chain,
flat gradient buckets, eager collectives between the graphs, EgoNCE all-gather with gradient -- one clip each; the
gradients the optimizer reads must equal ONE process on the concatenated B=2 batch, which the reference fixture
tests/golden/model_T8_B2.npz pins (tools/train_avgaze_net.py:70-99 under DDP, slowfast/utils/distributed.py:15-49,
slowfast/models/build.py:44-46; the reference's rank-0 slicing bug in AllGather_multi.backward is not reproduced)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

if not torch.cuda.is_available():
    pytest.skip("needs a GPU", allow_module_level=True)

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run_ranks(tmp_path, trunk_cut, compute, world=2, timeout=900, bucket_dtype="fp32", factors="factors"):
    port = _free_port()
    outs = [str(tmp_path / f"rank{r}.npz") for r in range(world)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), str(world), port, str(trunk_cut), compute, outs[r], bucket_dtype, factors],
                              env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            logs.append(out)
    finally:
        for p in procs:                      # exactly the children started above
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed (code {p.returncode}):\n{logs[r][-4000:]}"
    return [np.load(o, allow_pickle=False) for o in outs]


@pytest.mark.parametrize("trunk_cut", [3, 0])
def test_two_process_graph_chain_equals_single_process_reference_fixture(tmp_path, trunk_cut):
    g = np.load(os.path.join(GOLDEN, "model_T8_B2.npz"), allow_pickle=False)
    r0, r1 = _run_ranks(tmp_path, trunk_cut, "fp32")
    assert int(r0["n_buckets"]) == (3 if trunk_cut else 2)
    assert int(r0["n_factor_params"]) == 3        # the fusion-conv weight gradients were exchanged as factors (all-gather), not all-reduced
    # EgoNCE runs over the GATHERED embeddings: the same global value on both ranks == the reference's B=2 value;
    # KLDiv is per rank, its mean over ranks is the reference's batch mean
    assert abs(float(r0["nce"]) - float(r1["nce"])) < 1e-6
    assert abs(float(r0["nce"]) - float(g["nce"])) < 1e-3
    assert abs(0.5 * (float(r0["kld"]) + float(r1["kld"])) - float(g["kld"])) < 1e-4
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - float(g["loss"])) < 1e-4
    # averaged buckets: identical on both ranks, equal to the single-process gradients of the reference
    names = [str(n) for n in r0["grad_names"]]
    assert np.array_equal(r0["grad_norms"], r1["grad_norms"])
    norm_of = dict(zip(names, r0["grad_norms"]))
    worst = 0.0
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if n == "classifier.bias":           # softmax is shift-invariant: rounding noise only
            continue
        err = abs(norm_of[n] - ref_norm) / ref_norm
        worst = max(worst, err)
        assert err <= 2e-3, (n, norm_of[n], ref_norm)
        ref_slice = g[n.replace(".", "_") + "_g"]
        sl = r0["g__" + n][:ref_slice.size]
        assert np.array_equal(sl, r1["g__" + n][:ref_slice.size]), n
        e = np.linalg.norm(sl.astype(np.float64) - ref_slice) / np.linalg.norm(ref_slice.astype(np.float64))
        assert e < 5e-3, (n, e)
    tot, ref_tot = float(r0["grad_total_norm"]), float(g["grad_total_norm"])
    assert abs(tot - ref_tot) < 1e-3 * ref_tot
    assert abs(float(r0["clip_norm_seen"]) - ref_tot) < 1e-3 * ref_tot      # what the captured clip kernel measured
    assert float(r0["weights_moved_by_lr0"]) == 0.0
    print(f"\n[2 processes, gloo, one GPU, graph chain, trunk_cut={trunk_cut}] worst gradient-norm error vs the reference's "
          f"single-process B=2 fixture {worst:.2e}; total norm {tot:.6f} vs {ref_tot:.6f}; EgoNCE {float(r0['nce']):.5f} vs {float(g['nce']):.5f}")
    # after one real update the replicas are bit-identical
    for k in ("param_sum", "param_abs_sum", "param_heads"):
        assert np.array_equal(r0[k], r1[k]), k
    assert np.isfinite(float(r0["loss_b"])) and abs(float(r0["loss_b"]) - float(r0["loss"])) < 1e-5     # step B sees step A's weights


def test_two_process_graph_chain_bf16_mode(tmp_path):
    """The benchmarked compute mode through the same two-process chain: stated bf16 bars against the reference fixture."""
    g = np.load(os.path.join(GOLDEN, "model_T8_B2.npz"), allow_pickle=False)
    r0, r1 = _run_ranks(tmp_path, 3, "bf16")
    assert abs(0.5 * (float(r0["loss"]) + float(r1["loss"])) - float(g["loss"])) < 1e-2 * float(g["loss"])
    assert np.array_equal(r0["grad_norms"], r1["grad_norms"])
    norm_of = dict(zip([str(n) for n in r0["grad_names"]], r0["grad_norms"]))
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if n != "classifier.bias":
            assert abs(norm_of[n] - ref_norm) <= 4e-2 * ref_norm, (n, norm_of[n], ref_norm)
    tot, ref_tot = float(r0["grad_total_norm"]), float(g["grad_total_norm"])
    assert abs(tot - ref_tot) < 1e-2 * ref_tot
    for k in ("param_sum", "param_abs_sum", "param_heads"):
        assert np.array_equal(r0[k], r1[k]), k


def test_two_process_graph_chain_dense_fusion_gradients(tmp_path):
    """CSTS_AMD.FUSION_GRAD_FACTORS False: the three fusion-conv weight gradients travel inside the all-reduced head bucket (151 MB
    each) instead of as all-gathered rank-(B T') factors -- the default the other tests of this file run.  Same bars."""
    g = np.load(os.path.join(GOLDEN, "model_T8_B2.npz"), allow_pickle=False)
    r0, r1 = _run_ranks(tmp_path, 3, "fp32", factors="dense")
    assert int(r0["n_factor_params"]) == 0
    norm_of = dict(zip([str(n) for n in r0["grad_names"]], r0["grad_norms"]))
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if n != "classifier.bias":
            assert abs(norm_of[n] - ref_norm) <= 2e-3 * ref_norm, (n, norm_of[n], ref_norm)
    for k in ("param_sum", "param_abs_sum", "param_heads"):
        assert np.array_equal(r0[k], r1[k]), k


def test_two_process_graph_chain_16bit_buckets(tmp_path):
    """CSTS_AMD.GRAD_BUCKET_DTYPE bf16 through the real chain: the buckets leave each backward graph in 16 bits (half the xGMI
    bytes), are averaged in 16 bits and read by the optimizer kernels directly (fp32 accumulation, csts_opt_args.grad_dt).  The
    averaged gradients equal the reference's B = 2 gradients to the bf16-mode bars, the clip kernel sees the same norm, the
    replicas stay bit-identical."""
    g = np.load(os.path.join(GOLDEN, "model_T8_B2.npz"), allow_pickle=False)
    r0, r1 = _run_ranks(tmp_path, 3, "bf16", bucket_dtype="bf16")
    assert str(r0["bucket_dtype"]) == "torch.bfloat16"
    assert np.array_equal(r0["grad_norms"], r1["grad_norms"])
    norm_of = dict(zip([str(n) for n in r0["grad_names"]], r0["grad_norms"]))
    for n, ref_norm in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if n != "classifier.bias":
            assert abs(norm_of[n] - ref_norm) <= 4e-2 * ref_norm, (n, norm_of[n], ref_norm)
    tot, ref_tot = float(r0["grad_total_norm"]), float(g["grad_total_norm"])
    assert abs(tot - ref_tot) < 1e-2 * ref_tot
    assert abs(float(r0["clip_norm_seen"]) - ref_tot) < 1e-2 * ref_tot
    assert float(r0["weights_moved_by_lr0"]) == 0.0
    for k in ("param_sum", "param_abs_sum", "param_heads"):
        assert np.array_equal(r0[k], r1[k]), k


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` with no launcher starts its own N ranks; with fewer than N devices it must exit non-zero
    instead of printing a one-GPU number under an N-GPU label."""
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "refusing" in p.stderr


def test_bench_self_launch_two_ranks_gloo_on_one_gpu():
    """The self-launch path end to end: the parent never touches the GPU, two children come up as ranks 0 / 1 (gloo, one
    device), the graph chain runs, rank 0 prints ONE line with n_gpus == rccl_ranks == 2."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "2",
                        "--frames", "8", "--batch-per-gpu", "1", "--median-steps", "3", "--no-roofline", "--no-loss-check"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["rccl_ranks"] == 2 and out["config"]["global_batch"] == 2
    assert out["config"]["step"].startswith("hip_graph_chain") and out["value"] > 0
