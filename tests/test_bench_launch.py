"""`python bench.py --gpus N` with no launcher around it starts its N ranks itself (child processes through
torch.distributed.run, rendezvous on 127.0.0.1), forwards rank 0's JSON line and exits with the ranks' status.

Without a GPU the ranks cannot integrate anything (the engine has no CPU path), so the CPU test checks the
launch itself: N ranks really start with RANK / WORLD_SIZE set, each refuses for the right reason, and the
failure reaches the caller's exit status.  On the GPU box the same entry runs the N = 2 control flow end to end
(both ranks on the one device, gloo instead of RCCL: RATSDF_BENCH_DEVICE / RATSDF_BENCH_BACKEND)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _run(args, extra_env, timeout):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(extra_env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=str(ROOT))


def test_gpus_n_starts_n_ranks_and_propagates_their_status():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by test_two_ranks_through_the_plain_entry_point")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {}, 300)
    out = r.stdout + r.stderr
    assert r.returncode != 0, out[-2000:]
    # both ranks started and stopped at the engine's "no GPU" refusal (not at an argument or rendezvous error)
    assert out.count("needs an MI355X") >= 2, out[-2000:]
    assert "launch N > 1 with" not in out


@pytest.mark.gpu
def test_two_ranks_through_the_plain_entry_point():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reps", "1", "--frames-per-step", "12",
              "--cpu-frames", "0", "--host-frames", "0", "--no-secondary", "--streams", "0"],
             {"RATSDF_BENCH_DEVICE": "0", "RATSDF_BENCH_BACKEND": "gloo"}, 600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, rank 0's
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["streams"] == 2
    assert d["directory_blocks_all_ranks"] > 0          # the directory exchange ran


@pytest.mark.gpu
def test_one_stream_over_two_subvolume_ranks_through_the_plain_entry_point():
    """BASELINE configs[3] end to end through `python bench.py --gpus 2 --shard`: rank 0 owns the stream, the frames
    reach rank 1 by broadcast (inside the timed region), each rank integrates its subvolume (checked against the
    sharded CPU oracle on the bytes it received), the directory deltas are all-gathered, and every block of the
    union sits on its owner and nowhere else.  Both ranks on the one device, gloo instead of RCCL."""
    r = _run(["--gpus", "2", "--shard", "--steps", "2", "--warmup", "1", "--reps", "1", "--frames-per-step", "12",
              "--bcast-chunk", "4", "--cpu-frames", "4"],
             {"RATSDF_BENCH_DEVICE": "0", "RATSDF_BENCH_BACKEND": "gloo"}, 600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2 and d["config"]["streams"] == 1 and d["value"] > 0
    fb = d["frame_broadcast"]
    assert fb["in_timed_region"] and fb["chunk_frames"] == 4 and fb["frames_ahead"] == 8 and fb["broadcast_gbps"] > 0
    assert fb["bytes_per_frame"] == 640 * 480 * 15 + 64
    # the directory exchange ran, and the union of the ranks' block sets is what check_sharded_directories expects:
    # every block on its owner, on no other rank
    assert d["directory_blocks_all_ranks"] == d["directory_union_blocks"] > 0
    assert len(d["shards"]) == 2 and all(s["active_blocks"] > 0 for s in d["shards"])
    assert sum(s["active_blocks"] for s in d["shards"]) == d["directory_union_blocks"]
    assert d["parity"]["frames"] == 4 and d["parity"]["max_abs_tsdf"] == 0.0 and d["parity"]["max_abs_prob"] < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["streams", "shard"])
def test_the_multi_rank_code_path_over_rccl_with_a_group_of_one(mode):
    """The calls an 8-GPU run makes, over RCCL itself: RCCL refuses two ranks on one device, so the one-GPU box takes
    bench.py's N > 1 path with a process group of ONE rank (RATSDF_BENCH_ONE_RANK_GROUP=1): init_process_group("nccl",
    device_id=...), the directory exchange on device tensors (all_gather_into_tensor), barrier, all_reduce(MAX) of the
    step time -- and, with --shard, the frame broadcast on its side stream and all_gather_object of the shard table."""
    args = ["--gpus", "1", "--steps", "2", "--warmup", "1", "--reps", "1", "--frames-per-step", "12", "--cpu-frames", "4"]
    args += ["--shard", "--bcast-chunk", "4"] if mode == "shard" else ["--host-frames", "0", "--no-secondary", "--streams", "0"]
    r = _run(args, {"RATSDF_BENCH_ONE_RANK_GROUP": "1"}, 600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    assert d["directory_blocks_all_ranks"] > 0          # the RCCL all-gather of the directory ran
    if mode == "shard":
        assert d["scaling"] == "strong" and d["frame_broadcast"]["backend"] == "nccl"
        assert d["frame_broadcast"]["in_timed_region"] and d["frame_broadcast"]["broadcast_gbps"] > 0
        assert d["directory_union_blocks"] == d["directory_blocks_all_ranks"] and len(d["shards"]) == 1
        assert d["parity"]["frames"] == 4 and d["parity"]["max_abs_tsdf"] == 0.0
    else:
        assert d["config"]["streams"] == 1 and d["config"]["directory_allgather_every_frames"] == 12
