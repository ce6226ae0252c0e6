"""bench.py itself: the JSON contract at N = 1 and the N > 1 code path (two ranks sharing the box's one GPU over gloo:
AOS2_BENCH_BACKEND / AOS2_BENCH_SHARE_GPU are test hooks; the driver's multi-GPU runs use RCCL)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env_extra=None, timeout=600):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_contract_single_gpu(gpu):
    d = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "32", "--cpu-frames", "8", "--no-extra"])
    assert d["metric"] == "frames/sec (extract+match+localBA) TUM 640x480" and d["unit"] == "frames/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["value"] > 0 and abs(d["value"] - 32 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert set(d["cpu_baseline"]["ms_per_frame"]) >= {"extract", "search_by_projection_last", "pose_optimization", "search_local_points", "local_ba"}
    m = d["config"]["matches_per_frame_mean"]
    assert m["search_by_projection_last"] > 100 and m["inliers_2"] > 100 and d["config"]["local_ba_iterations"][0] > 0
    # the last timed step was checked against the oracle: every batch position and every LocalBA window of it
    pc = d["parity_checked"]
    assert pc["ok"] is True and pc["n_mismatches"] == 0 and pc["frames"] == 32 and pc["distinct_frame_pairs"] == 32   # no tiling
    assert pc["local_ba_windows"] == 4 and pc["distinct_local_ba_problems"] == 4   # every window a different problem
    assert d["config"]["local_ba_windows_per_step"] == 4 and d["config"]["local_ba_steps_per_call"] == 1 and d["config"]["local_ba_windows_per_call"] == 4
    assert max(pc["worst_abs_diff"][k] for k in ("mTcw", "lba_pose", "lba_point")) <= 1e-5   # the worst difference is reported
    ls = d["extra"]["local_ba_lock_step"]
    assert ls["trial_slots_enqueued"] >= 15 and 1.0 <= ls["slots_over_trials"] <= 1.03 and d["config"]["local_ba_mix"] == "heterogeneous"
    assert d["config"]["step_runner"].startswith("native") and d["config"]["local_ba_handles_in_flight"] == 3 and d["config"]["images"].startswith("resident")
    assert d["exchange"]["bytes_to_rank0_per_step_at_8_ranks"] == 7 * 32 * d["exchange"]["slot_bytes"] and d["exchange"]["pack_kernel_us"] > 0
    assert len(d["config"]["frames_per_s_per_rank"]) == 1 and d["config"]["host_threads_per_rank"]["local_ba_workers_per_handle"] >= 1
    ts = d["extra"]["timed_steps"]   # where the enqueueing thread waited during the timed steps, how long the LocalBA calls took
    assert set(ts["host_thread_waits_ms_per_step"]) == {"local_ba", "keyframe_legs", "tracking"} and ts["local_ba_call_wall_ms_min_median_max"][1] > 0
    # host placement is left to the scheduler by default (AOS2_BENCH_NUMA=1 binds to the GPU's NUMA node), one LocalBA program per handle
    assert d["config"]["host_cpus_bound_to_the_gpus_numa_node"] == 0 and d["config"]["local_ba_window_groups_per_handle"] == 1
    # ... and the keyframe legs: the BoW searches and LocalMapping's (keyframe, neighbour) searches
    assert pc["reference_keyframe_bow_frames"] >= 1 and pc["keyframe_neighbour_pairs"] == 4 * 30   # 10 first-order + 20 second-order Fuse targets per keyframe
    kl = d["config"]["keyframe_legs_per_step"]
    assert kl["triangulation_pairs"] == 4 * 10 and kl["fuse_pairs"] == 4 * 30
    assert {"search_for_triangulation", "fuse", "search_by_bow"} <= set(d["cpu_baseline"]["ms_per_frame"])


def test_bench_kitti_slice(gpu):
    """BASELINE configs[2] + [3] as a bench line (--workload kitti): a 2-step slice -- both eyes' extraction on two handles,
    ComputeStereoMatches and the stereo Frame members on the device, the tracking chain, the keyframe's BoW search and the 20-keyframe
    LocalBA windows; the last timed step equals the oracle (mvuRight / mvDepth from the oracle's ComputeStereoMatches, bit for bit)"""
    d = _run([sys.executable, "bench.py", "--workload", "kitti", "--steps", "2", "--warmup", "1", "--batch", "16", "--cpu-frames", "8"])
    assert d["metric"].startswith("stereo frames/sec") and "KITTI" in d["config"]["workload"] and d["n_gpus"] == 1
    assert abs(d["value"] - 16 * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]
    pc = d["parity_checked"]
    assert pc["ok"] is True and pc["frames"] == 16 and pc["distinct_frame_pairs"] == 16 and pc["local_ba_windows"] == 2
    assert pc["reference_keyframe_bow_frames"] == 2 and pc["keyframe_neighbour_pairs"] == 0
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 1444097 * 16 and 0 < d["roofline"]["frac"] < 1
    assert d["cpu_baseline"]["value"] > 0 and "extract_both_eyes_and_stereo_matches" in d["stage_ms"]
    assert d["config"]["keypoints_per_frame_mean"] > 1500 and d["config"]["matches_per_frame_mean"]["inliers_2"] > 100


def test_bench_with_the_large_batch_pose_optimization_form(gpu):
    """batches of more than 256 frames run PoseOptimization with 128 threads per frame (all frames resident at once); forced here on a
    small batch: the last timed step still equals the oracle (inlier counts, outlier flags bit-identical, poses 1e-5)"""
    d = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-extra"],
             {"AOS2_PO_THREADS": "128"})
    assert d["parity_checked"]["ok"] is True and d["parity_checked"]["frames"] >= 8


def test_bench_local_ba_windows_of_two_steps_per_call(gpu):
    """AOS2_BENCH_LBA_STEPS_PER_CALL=2: a LocalBA call every second step with both steps' windows (8 different problems here), an odd step
    count gets a last call for the rest; every window of the last call equals the oracle"""
    d = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "32", "--no-cpu-baseline", "--no-extra"],
             {"AOS2_BENCH_LBA_STEPS_PER_CALL": "2"})
    pc = d["parity_checked"]
    assert pc["ok"] is True and pc["local_ba_windows"] == 8 and pc["distinct_local_ba_problems"] == 8
    assert d["config"]["local_ba_windows_per_step"] == 4 and d["config"]["local_ba_steps_per_call"] == 2 and d["config"]["local_ba_windows_per_call"] == 8


def test_bench_detects_a_wrong_result(gpu):
    """the --verify leg is a real check: with one map point of the table moved after the oracle inputs were fixed
    (AOS2_BENCH_FAULT, a test hook) the line reports the mismatch and the process fails"""
    env = dict(os.environ)
    env["AOS2_BENCH_FAULT"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "8", "--no-cpu-baseline", "--no-extra"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 3 and len(lines) == 1, r.stdout[-2000:] + r.stderr[-2000:]
    pc = json.loads(lines[0])["parity_checked"]
    assert pc["ok"] is False and pc["n_mismatches"] > 0


@pytest.mark.parametrize("workload,launcher", [("tum", "torchrun"), ("euroc8", "torchrun"), ("tum", "self"), ("euroc8", "self")])
def test_bench_two_ranks_share_the_gpu(gpu, workload, launcher):
    """N = 2 both ways: under torch.distributed.run (the driver's form) and as plain `python bench.py --gpus 2`, which starts
    its own ranks"""
    tail = ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "16", "--no-extra", "--no-cpu-baseline",
            "--workload", workload, "--verify"]
    if launcher == "torchrun":
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port())] + tail
    else:
        cmd = [sys.executable] + tail
    d = _run(cmd, {"AOS2_BENCH_BACKEND": "gloo", "AOS2_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["parity_checked"]["ok"] is True
    if workload == "tum":
        assert len(d["config"]["frames_per_s_per_rank"]) == 2
    if workload == "euroc8":
        assert d["parity_checked"]["frames"] == 8 and d["parity_checked"]["gathered_slots"] == 8
    if workload == "tum":
        assert d["scaling"] == "weak" and d["exchange"]["headers_ok"] is True and d["exchange"]["bytes_to_rank0_per_step"] > 0
        assert abs(d["value"] - 2 * 16 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    else:
        assert d["scaling"] == "strong" and d["config"]["headers_ok"] is True and d["config"]["frames_per_rank"] == 4


def test_bench_refuses_a_rank_count_that_is_not_the_gpu_count(gpu):
    """--gpus N is what the line reports: a launcher that started another number of ranks is an error, not a mislabelled line"""
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


@pytest.mark.parametrize("workload", ["tum", "euroc8"])
def test_bench_rccl_path_on_one_gpu(gpu, workload):
    """The exchange step over RCCL itself (torch.distributed backend "nccl" = RCCL): a one-rank process group on the box's GPU runs
    the code the multi-GPU ranks run -- process group bound to the device, the step's slots packed by the device kernel and
    gathered as DEVICE tensors on the step's own stream, the all-reduce of the timing, the barriers -- and rank 0 checks the
    gathered slots (AOS2_BENCH_FORCE_DIST is a test hook; with N GPUs the driver's torchrun launch takes the same path)."""
    d = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "16", "--no-extra", "--no-cpu-baseline",
              "--workload", workload, "--verify"], {"AOS2_BENCH_FORCE_DIST": "1", "MASTER_PORT": str(_free_port())})
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["parity_checked"]["ok"] is True
    if workload == "tum":
        assert d["exchange"]["backend"] == "nccl" and d["exchange"]["headers_ok"] is True
    else:
        assert d["config"]["backend"] == "nccl" and d["config"]["headers_ok"] is True and d["parity_checked"]["gathered_slots"] == 8


def test_bench_on_a_recorded_sequence(gpu, tmp_path):
    """$TUM_FR1_DESK set: bench.py reads its frame pairs from the dataset directory, says so in `data`, and the last timed step
    still equals the oracle"""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import __graft_entry__ as g
    from test_datasets_cpu import _mini_tum
    _mini_tum(g.load_package(), tmp_path, n=12)
    d = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--batch", "16", "--cpu-frames", "8", "--no-extra"],
             {"TUM_FR1_DESK": str(tmp_path)})
    assert d["data"].startswith("real") and d["parity_checked"]["ok"] is True and d["parity_checked"]["frames"] >= 8
    assert d["config"]["matches_per_frame_mean"]["search_by_projection_last"] > 50
