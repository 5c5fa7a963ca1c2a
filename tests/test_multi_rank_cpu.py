"""N > 1 path on CPU: two ranks over gloo shard the clip set, 'transcribe' their shard (here with the
CPU oracle on the nano model — the checker standing in for the GPU call), gather the fixed-stride
records and must reproduce the single-process result in clip order."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from whisper_rust_ort_amd import modelspec as ms  # noqa: E402
from whisper_rust_ort_amd import sharding  # noqa: E402

CLIPS_PER_RANK, MAX_NEW, PROMPT, EOT = 2, 6, [3, 5, 7, 9], 2


def _transcribe(clip_id):
    from oracle import oracle as orc
    dims = ms.PRESETS["nano"]
    w = ms.flatten_state_dict(dims, ms.synth_state_dict(dims, 7))
    pcm = ms.synth_clip(clip_id)[:48000]                      # 3 s, zero-padded window like :899-905
    mel = orc.window_mel(orc.log_mel(pcm, 80), 0, 3000)
    toks, _ = orc.decode_greedy(dims, w, orc.encoder(dims, w, mel), PROMPT, MAX_NEW, EOT)
    return toks


def _worker(rank, world, port, q):
    os.environ["OMP_NUM_THREADS"] = "2"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ids = sharding.shard_clip_ids(rank, world, CLIPS_PER_RANK)
    toks = [_transcribe(i) for i in ids]
    dist.barrier()
    elapsed = sharding.max_over_ranks(dist, 1.0 + rank)        # MAX over ranks
    rec = sharding.pack_records(ids, toks, len(PROMPT) + MAX_NEW)
    allrec = sharding.gather_records(dist, rec)
    q.put((rank, elapsed, allrec))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo_gather_matches_single_process():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = [_transcribe(i) for i in range(world * CLIPS_PER_RANK)]
    for rank, elapsed, allrec in got:
        assert elapsed == 2.0                                   # max(1.0, 2.0)
        rows = sharding.unpack_records(allrec)
        assert [cid for cid, _ in rows] == list(range(world * CLIPS_PER_RANK))
        for (cid, tk), r in zip(rows, ref):
            assert tk.tolist() == r.tolist()


def test_record_packing_roundtrip_and_bounds():
    toks = [np.array([1, 2, 3]), np.array([], np.int64), np.arange(10)]
    rec = sharding.pack_records([7, 3, 5], toks, 10)
    assert rec.shape == (3, 12) and rec.dtype == np.int32
    rows = sharding.unpack_records(rec)
    assert [c for c, _ in rows] == [3, 5, 7]
    assert rows[2][1].tolist() == [1, 2, 3] and rows[0][1].tolist() == []
    with pytest.raises(ValueError):
        sharding.pack_records([0], [np.arange(11)], 10)
    assert sharding.shard_clip_ids(3, 8, 64) == list(range(192, 256))


def test_bench_self_launches_one_rank_per_gpu():
    """`python bench.py --gpus N` (the driver's plain command) must start the N ranks itself: the parent never
    touches HIP, the children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*, rank 0's line is relayed."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["launch_check"] and line["world"] == 2 and line["max_rank"] == 1.0
    assert [c for c, _ in line["records"]] == [0, 1, 2, 3]
    assert [t for _, t in line["records"]] == [[0, 0], [0, 0], [1, 1], [1, 1]]     # [rank, local_rank] per clip
    # a launch whose world size disagrees with --gpus is refused
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check"],
                         env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "does not match --gpus" in bad.stderr


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}


def test_bench_self_launch_at_eight_ranks_and_strong_scaling_split():
    """The driver's N = 8 command shape on CPU (gloo rendezvous, no GPU work): eight children, every rank's records
    gathered, and `--total-clips 512` (BASELINE configs[2]/[4]: 512 clips sharded over the GPUs) dealt as 64 per rank
    with "scaling": "strong"; a clip count the ranks cannot share evenly is refused."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check", "--total-clips", "512"],
                         env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["world"] == 8 and line["max_rank"] == 7.0
    assert line["scaling"] == "strong" and line["clips_per_gpu"] == 64
    assert [c for c, _ in line["records"]] == list(range(16))
    assert [t[0] for _, t in line["records"]] == [r for r in range(8) for _ in range(2)]
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--launch-check", "--total-clips", "100"],
                         env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert bad.returncode != 0 and "not a multiple" in bad.stderr


def test_bench_launcher_fails_when_one_rank_dies():
    """One child exits with status 3 before the rendezvous: the launcher must return non-zero promptly, name the rank, and
    end the other ranks (they would wait in the rendezvous for ever) instead of hanging."""
    import subprocess
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--launch-check", "--fail-rank", "2"],
                         env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "rank 2 exited with 3" in out.stderr
    assert not any(l.startswith("{") for l in out.stdout.splitlines())      # no result line from a failed job
    assert time.time() - t0 < 120


def test_nccl_init_failure_falls_back_to_gloo_cleanly():
    """bench.py's process-group initialisation (init_group): asked for RCCL on a box without a GPU, the "nccl" backend fails; the
    fallback must leave no half-initialised group behind and gather over gloo — both ranks finish, rank 0 reports the backend and
    the error it fell back from.  Also: the per-rank host-thread cap divides the cgroup's cores by the world size."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check", "--prefer-nccl"], env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["world"] == 2 and line["backend"] == "gloo" and line["nccl_error"]
    assert "falling back to gloo" in out.stderr
    assert [c for c, _ in line["records"]] == [0, 1, 2, 3]
    assert 1 <= line["host_threads_per_rank"] <= max(1, (os.cpu_count() or 1) // 2)


def test_cli_devices_flag_maps_to_one_model_per_device_and_contexts_per_stream():
    """whisper_bench --devices 0-7 --streams-per-gpu 2 --print-plan: the plan the CLI derives from the flags (one model per device, one
    context and host thread per (device, stream)) without touching a device; list and range forms, a bad range is an error."""
    import json
    import subprocess
    cli = os.path.join(ROOT, "whisper-rust-ort_amd", "whisper_bench")
    r = subprocess.run([cli, "--onnx-dir", "synthetic:base:1", "--devices", "0-7", "--streams-per-gpu", "2", "--max-batch", "64", "--print-plan"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout)
    assert plan["devices"] == list(range(8)) and plan["models"] == 8 and plan["host_threads"] == 16 and plan["max_batch"] == 64
    assert [(c["context"], c["device"], c["stream"]) for c in plan["contexts"]] == [(2 * d + s, d, s) for d in range(8) for s in range(2)]
    r = subprocess.run([cli, "--onnx-dir", "synthetic:base:1", "--devices", "1,3-4", "--print-plan"], capture_output=True, text=True, timeout=60)
    assert json.loads(r.stdout)["devices"] == [1, 3, 4]
    r = subprocess.run([cli, "--onnx-dir", "synthetic:base:1", "--devices", "5-2", "--print-plan"], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "bad --devices range" in r.stderr
