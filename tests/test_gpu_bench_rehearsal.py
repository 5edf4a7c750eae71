"""GPU: bench.py --gpus 2 end to end, both ranks on the one GPU of the test box (BENCH_REHEARSE_ON_ONE_GPU=1: gloo control plane,
every rank on device 0) -- never a scaling result, but everything an N > 1 run does besides owning its own GPU runs here: the
ranks are spawned before anything touches a GPU, the job is sharded (--strong) or replicated (weak), every rank checks its own
pictures against the oracle and the verdicts are reduced, clocks are reduced with MAX, rank 0 prints ONE line, the end-to-end leg
splits the host's cores between the ranks and says so, the single-process engine runs over two contexts.  What the driver's
SCALE run will print is this line with n_gpus = 1, 2, 4, 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra):
    env = dict(os.environ, BENCH_REHEARSE_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--placement-trials", "0", "--ordinary-buffers", "--cli-pictures", "0", "--e2e-repeats", "1"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_two_ranks_one_line(mode):
    extra = ["--frames", "48", "--e2e-pictures", "96"] if mode == "weak" else ["--strong", "96"]
    d = _run(extra)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "end_to_end", "engine_multi_context"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "macroblocks/s"
    assert d["scaling"] == mode and d["vs_baseline"] is None and d["higher_is_better"] is True
    assert d["rehearsal_all_ranks_on_one_gpu"] is True
    assert d["value"] > 0 and d["ms_per_step"] > 0
    cfg = d["config"]
    assert cfg["bit_exact_vs_oracle"] is True                       # every rank's verdict, reduced
    assert cfg["frames_per_gpu_per_step"] == 48
    assert cfg["macroblocks_per_step"] == 96 * 8160                 # whole job: both ranks
    assert abs(d["value"] - cfg["macroblocks_per_step"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"])) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and 0 < rf["frac"] < 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    e2e = d["end_to_end"]
    assert e2e["n_gpus"] == 2 and e2e["pictures"] == 96 and e2e["bit_exact_vs_oracle"] is True and e2e["value"] > 0
    host = e2e["host"]
    assert host["ranks_on_this_host"] == 2 and host["entropy_threads_per_rank"] >= 1
    assert abs(host["host_cores_per_rank"] * 2 - host["host_cores_visible"]) < 1e-9
    assert 0.0 <= e2e["wall_not_entropy"]["share_of_wall"] <= 1.0
    multi = d["engine_multi_context"]
    assert "error" not in multi, multi
    assert multi["contexts"] == 2 and multi["bit_exact_vs_oracle"] is True and multi["rehearsal_contexts_share_devices"] is True


def test_every_rccl_call_of_an_n_gpu_run_with_a_world_of_one():
    """The N > 1 run initialises RCCL on the rank's device (`init_process_group("nccl", device_id=...)`), opens a gloo side group,
    and uses barriers and MAX / MIN reductions of device tensors.  No second GPU exists on the test box, so this is the most of
    that path it can run: the same calls, in the same order, with world_size = 1 (BENCH_RCCL_ONE_RANK=1) -- RCCL loads, the
    communicator comes up on the device, the collectives return, the line is printed and the group is destroyed."""
    env = dict(os.environ, BENCH_RCCL_ONE_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533",
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
           "--placement-trials", "0", "--cli-pictures", "0", "--e2e-repeats", "1", "--frames", "48", "--e2e-pictures", "48"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_calls_with_one_rank"] is True and d["rehearsal_all_ranks_on_one_gpu"] is None
    assert d["config"]["bit_exact_vs_oracle"] is True and d["end_to_end"]["bit_exact_vs_oracle"] is True and d["value"] > 0
