#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native Whisper hot path.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: plain `python bench.py --gpus N` starts one rank per GPU itself; under
     `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` it joins the ranks it is given)

Workload (BASELINE.json configs[2] scaled to the GPUs present; at N=1 it is configs[1]'s model
and dtype on one GPU's share of the clip set): whisper-base dims, bf16 MFMA, hash-seeded synthetic
weights, synthetic 30 s clips (modelspec.synth_clip), greedy decode of exactly 128 new tokens per
clip (EOT placed in suppress_tokens — the reference's own mechanism, src/main.rs:765,817), clips
sharded across ranks with no data-path collective; one STEP = one pass of
log-mel → encoder → cross-KV → 128-token greedy decode over one batch of `--clips` clips per GPU,
PCM already resident in HBM.

Prints ONE JSON line (rank 0).  `value` = audio seconds transcribed per wall second by the whole
job (the "× real time" figure of BASELINE.json, rtfx = 1/rtf with rtf = src/main.rs:1191).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from whisper_rust_ort_amd import binding as wb  # noqa: E402
from whisper_rust_ort_amd import modelspec as ms  # noqa: E402

# chip partition defaults (profiles/r03_cu_partition_sweep.txt); --pipeline / --enc-cus override
DEFAULT_PIPELINE = 0
DEFAULT_ENC_CUS = 64

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA
MFMA_F32_PEAK_TF = 157.3


Hip = wb.HipRuntime   # device-resident PCM (hipMalloc / hipMemcpy of the runtime the library already loaded)


def algorithmic_work(dims, n_clips: int, n_prompt: int, max_new: int, kv_esz: int, cross_es: bool = False) -> dict:
    """SURVEY.md §8d formulas, per launch / per batch."""
    d, F, T, Ld, Le, V, M = dims.d_model, dims.ffn, dims.n_audio_ctx, dims.dec_layers, dims.enc_layers, dims.vocab, dims.n_mels
    enc_flop = 2 * d * M * 3 * 3000 + 2 * d * d * 3 * T + Le * (2 * 4 * T * d * d + 2 * 2 * T * T * d + 2 * 2 * T * d * F)
    cross_kv_flop = 2 * Ld * 2 * T * d * d
    positions = n_prompt + max_new - 1
    per_pos = Ld * 2 * (6 * d * d + 2 * d * F + 2 * T * d)
    dec_flop = positions * per_pos + max_new * 2 * d * V
    return {
        "enc_flop_per_clip": enc_flop,
        "cross_kv_flop_per_clip": cross_kv_flop,
        "dec_flop_per_clip": dec_flop,
        # one cross-attention launch reads K and V of one layer for every clip of the batch, once — or, on the encoder
        # states (wh_cross_es.hip), the T x d states themselves: the kernel's own algorithmic bytes, half of the K + V form
        "cross_attn_bytes_per_launch": (1 if cross_es else 2) * T * d * kv_esz * n_clips,   # kv_esz: 4 f32, 2 bf16, 1 e4m3
        "cross_attn_launches": positions * Ld,
        "mel_bytes_per_clip": 480000 * 4 + M * 3000 * 4,
    }


def cpu_baseline(dims, seed: int, prompt, eot, max_new: int, levels=None) -> dict:
    """The oracle (a port of the reference's algorithm — the Rust/ORT binary cannot be built here) timed on this box's
    host cores in the reference's benchmarked shape: clip-parallel, ONE thread per clip
    (`--chunk-parallelism N --intra-op 1`, run_benchmark_without_hf_pipeline_rust.sh:7-9; src/main.rs:884-919), at
    N = 4, 8 and all usable cores; each level transcribes N clips of the same workload concurrently, once."""
    import threading
    from oracle import oracle as orc
    w = ms.flatten_state_dict(dims, ms.synth_state_dict(dims, seed))
    cores = orc.cpu_budget()
    L = orc.lib()
    if levels is None:
        levels = sorted({min(4, cores), min(8, cores), cores})

    def one(i, out):
        L.orc_set_threads(1)   # OpenMP's team size is per calling thread: every clip runs on exactly one core
        pcm = ms.synth_clip(i)
        t0 = time.perf_counter()
        mel = orc.log_mel(pcm, dims.n_mels)
        t1 = time.perf_counter()
        enc = orc.encoder(dims, w, mel)
        t2 = time.perf_counter()
        toks, _ = orc.decode_greedy(dims, w, enc, prompt, max_new, eot, suppress=[eot])
        out[i] = (t1 - t0, t2 - t1, time.perf_counter() - t2, [int(t) for t in toks[:8]])

    sweep = []
    for n in levels:
        out = {}
        th = [threading.Thread(target=one, args=(i, out)) for i in range(n)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        el = time.perf_counter() - t0
        per = np.asarray([v[0] + v[1] + v[2] for v in out.values()])
        sweep.append({"clips_in_parallel": n, "threads": n, "seconds": el, "rtfx": 30.0 * n / el,
                      "p95_s_per_clip": float(np.percentile(per, 95)),
                      "mel_enc_dec_s_clip0": [round(x, 3) for x in out[0][:3]]})
    best = max(sweep, key=lambda r: r["rtfx"])
    return {"value": best["rtfx"], "unit": "x real time (audio s / wall s)", "cores": best["threads"], "kind": "port",
            "sample": f"{best['clips_in_parallel']} clips (30 s each) transcribed concurrently, one thread per clip, whisper-base fp32, "
                      f"{max_new} new tokens, C oracle; {best['seconds']:.1f} s wall",
            "seconds": best["seconds"], "clip_parallel_sweep": sweep, "tokens": out[0][3]}


def cpu_budget(cap: int = 64) -> int:
    """Host cores this process may use: min(affinity mask, cgroup CPU quota, cap) — the GPU boxes expose every hardware thread of the
    host but grant a share of them."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt[0] != "max":
            n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, min(n, cap))


def init_group(rank: int, world: int, prefer_nccl: bool, local_rank: int):
    """The process group of the result gather: RCCL (backend "nccl") when asked for and available, else gloo — the gather is a few KB.
    A failed RCCL initialisation must leave nothing half-built behind before gloo is tried.  Returns (dist, backend, error text or None)."""
    import torch.distributed as dist
    err = None
    if prefer_nccl:
        try:
            import torch
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world)  # RCCL on ROCm
            dist.barrier()
            return dist, "nccl", None
        except Exception as e:  # RCCL unavailable
            err = f"{type(e).__name__}: {e}"
            print(f"[bench] nccl init failed ({err}); falling back to gloo", file=sys.stderr)
            if dist.is_initialized():
                dist.destroy_process_group()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist, "gloo", err


def self_launch(n: int) -> int:
    """One process per GPU: start n copies of this script as children with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, relay rank 0's stdout (the JSON line), return non-zero if any rank fails.
    Children are separate processes (subprocess, never exec) and are ended by their exact PIDs on failure."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    rc = 0
    out0 = b""
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                try:
                    if r == 0:
                        o, _ = procs[0].communicate(timeout=0.5)
                        out0 += o or b""
                    else:
                        procs[r].wait(timeout=0.5)
                except subprocess.TimeoutExpired:
                    continue
                pending.discard(r)
                if procs[r].returncode != 0:
                    rc = rc or procs[r].returncode or 1
                    print(f"[bench] rank {r} exited with {procs[r].returncode}", file=sys.stderr)
                    for q in pending:      # a dead rank would leave the others waiting in a barrier
                        procs[q].terminate()
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    # rank 0's stdout may carry library chatter (gloo prints its connection banner there): relay the JSON line(s) only
    for line in out0.decode(errors="replace").splitlines():
        if line.startswith("{"):
            sys.stdout.write(line + "\n")
        elif line.strip():
            print(line, file=sys.stderr)
    sys.stdout.flush()
    return rc


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--clips", type=int, default=0,
                    help="clips per GPU per step (one device batch resident in HBM); default: the library's largest batch for the preset — "
                         "2048 for whisper-base (≈60 GB of workspace + caches in bf16, ≈125 GB in the f16x3 mode; a context of 1024 clips and more may hold a second or third workspace for a moment while it is created: the placement step, DESIGN.md 5f), 256 for whisper-large-v3 (≈99 GB); "
                         "a quarter of that in the exact-f32 mode")
    ap.add_argument("--total-clips", type=int, default=0,
                    help="strong scaling (BASELINE configs[2]/[4]: '512 clips sharded over N GPUs'): the whole job's clips per step, "
                         "dealt evenly to the ranks; overrides --clips and reports \"scaling\": \"strong\"")
    ap.add_argument("--streams", type=int, default=1, help="independent HIP streams (contexts) per GPU; clips are split evenly")
    ap.add_argument("--pipeline", type=int, default=None, choices=[0, 1],
                    help="1: chip partition — step i runs log-mel + encoder of step i+1's batch on the encoder stream (its own compute "
                         "units) beside step i's token loop (wh_transcribe_batch_device_next); 0: one stream, one phase at a time")
    ap.add_argument("--enc-cus", type=int, default=-1,
                    help="compute units of the encoder stream in pipeline mode (multiple of 8: the same count from every XCD); "
                         "0 = no CU masks, two plain streams; default: the measured best (profiles/r03_cu_partition_sweep.txt)")
    ap.add_argument("--dec-cus", type=int, default=-1, help="compute units of the token-loop stream (default: the other 256 - enc-cus)")
    ap.add_argument("--cross-es", default="auto", choices=["auto", "on", "off"],
                    help="token-loop cross-attention on the encoder states (wh_cross_es.hip) instead of the projected K / V cache; "
                         "auto = the library's rule (bf16 whisper-base geometry, contexts of >= 256 clips)")
    ap.add_argument("--preset", default="base")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "f32", "fp8", "f16x3"])
    ap.add_argument("--max-new-tokens", type=int, default=128)
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-batch1", action="store_true", help="skip the untimed side measurements: batch-1 latency, 64- / 256- / 1024-clip batches, host-resident clips, "
                                                              "in-tolerance precisions, strong scaling (profiling runs)")
    ap.add_argument("--no-row-check", action="store_true",
                    help="skip the untimed one-clip / pipelined-vs-plain identity checks (profiling runs: their launches would dilute per-kernel averages)")
    ap.add_argument("--test-single-device", action="store_true",
                    help="rehearsal on a one-GPU box: every rank uses device 0 and the gather goes over gloo")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group even for world size 1")
    ap.add_argument("--graph-timed", action="store_true",
                    help="experiment: no HIP events in the timed region (decode replays its hipGraph); roofline then comes from the profiled pass")
    ap.add_argument("--profile-all", action="store_true", help="event-time every kernel group in the timed region")
    ap.add_argument("--launch-check", action="store_true",
                    help="no GPU work: every rank joins a gloo group, gathers one dummy record and rank 0 prints what it saw "
                         "(CPU test of the launcher and the rendezvous)")
    ap.add_argument("--fail-rank", type=int, default=-1, help="launcher test: with --launch-check, this rank exits with status 3 before the rendezvous")
    ap.add_argument("--prefer-nccl", action="store_true", help="launcher test: with --launch-check, try the RCCL backend first (on a CPU box: exercises the fallback to gloo)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher.  It has not touched HIP (the
        # binding loads libwhisper_hip.so lazily) and never will: it starts one child per GPU and relays rank 0.
        raise SystemExit(self_launch(a.gpus))
    if world != a.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {a.gpus}: launch one rank per GPU "
                         f"(python bench.py --gpus N starts them itself)")
    scaling = "weak"
    if a.total_clips:   # strong scaling: the job's clip set is fixed and dealt evenly to the ranks (src/main.rs:884-919 deals a fixed window set to its workers)
        if a.total_clips % world:
            raise SystemExit(f"--total-clips {a.total_clips} is not a multiple of the {world} ranks")
        a.clips = a.total_clips // world
        scaling = "strong"
    if a.launch_check:
        if a.fail_rank == rank:
            print(f"[bench] rank {rank}: --fail-rank asked this rank to die before the rendezvous", file=sys.stderr)
            raise SystemExit(3)
        from whisper_rust_ort_amd import sharding
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist, lc_backend, lc_err = init_group(rank, world, a.prefer_nccl, local_rank)
        ids = sharding.shard_clip_ids(rank, world, 2)
        rec = sharding.gather_records(dist, sharding.pack_records(ids, [np.array([rank, local_rank])] * 2, 4))
        tmax = sharding.max_over_ranks(dist, float(rank))
        if rank == 0:
            print(json.dumps({"launch_check": True, "world": world, "max_rank": tmax, "scaling": scaling, "clips_per_gpu": a.clips,
                              "backend": lc_backend, "nccl_error": lc_err, "host_threads_per_rank": max(1, cpu_budget() // world),
                              "records": [[c, t.tolist()] for c, t in sharding.unpack_records(rec)]}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    dist = None
    backend = "none"
    if world > 1 or a.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if a.test_single_device:
            local_rank = 0
        dist, backend, _ = init_group(rank, world, not a.test_single_device, local_rank)

    dims = ms.PRESETS[a.preset]
    prec = wb.PRECISIONS[a.precision]
    esz = 4 if prec in (wb.WH_PREC_F32, wb.WH_PREC_F16X3) else 2
    if wb.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: libwhisper_hip has no CPU fallback")
    if a.clips <= 0:   # the largest device batch whose workspace + caches fit comfortably
        # whisper-base: 2048 = the library's largest context (WH_MAX_BATCH); with the cross-attention on the encoder states a clip
        # holds 1.5 MB instead of 18.4 MB of decode-side state, and the latency-bound decode GEMMs amortise further
        a.clips = {"base": 2048, "large-v3": 256}.get(a.preset, 256)
        if prec == wb.WH_PREC_F32 and a.preset in ("base", "large-v3"):
            a.clips //= 4
        if prec == wb.WH_PREC_F16X3 and a.preset == "large-v3":
            a.clips //= 2   # f32-sized activations and K / V cache: ≈ 0.8 GB per clip
    if a.pipeline is None:
        a.pipeline = DEFAULT_PIPELINE if a.streams == 1 else 0
    if a.pipeline and a.streams != 1:
        raise SystemExit("--pipeline 1 overlaps the phases of ONE context; use it with --streams 1")
    if a.enc_cus < 0:
        a.enc_cus = DEFAULT_ENC_CUS
    if a.dec_cus < 0:
        a.dec_cus = wb.N_CUS - a.enc_cus if a.enc_cus else 0
    dev = local_rank
    model = wb.Model(f"synthetic:{a.preset}:{a.seed}", dev, prec)
    assert a.clips % a.streams == 0, "--clips must be a multiple of --streams"
    per_stream = a.clips // a.streams
    ces = {"auto": None, "on": True, "off": False}[a.cross_es]
    if a.pipeline:
        # encoder stream on the low mask bits, token loop on the bits above them (disjoint unless --dec-cus says otherwise)
        em = wb.cu_mask(0, a.enc_cus) if a.enc_cus else None
        dm = wb.cu_mask(wb.N_CUS - a.dec_cus, a.dec_cus) if a.dec_cus else None
        ctxs = [wb.Context(model, per_stream, enc_cu_mask=em, dec_cu_mask=dm, two_streams=True, cross_es=ces)]
    else:
        ctxs = [wb.Context(model, per_stream, cross_es=ces) for _ in range(a.streams)]
    ctx = ctxs[0]
    hip = Hip()

    # this rank's shard of the clip set: clip ids rank*clips .. (rank+1)*clips-1 (weak scaling: per-GPU work is fixed;
    # --total-clips: the job's clip set is fixed and dealt to the ranks).  Pipeline mode alternates between two such
    # batches (the second one: the same ids + 1,000,000) so that a step's prefetched encoder pass reads other clips
    # than the batch being decoded.
    from concurrent.futures import ThreadPoolExecutor
    n_bufs = 2 if a.pipeline else 1
    d_bufs = []
    for k in range(n_bufs):
        # clip synthesis threads: this rank's share of the host cores the job may use (8 ranks x 16 threads would oversubscribe a cgroup share)
        with ThreadPoolExecutor(max_workers=max(1, min(16, cpu_budget() // world))) as gen:   # numpy releases the GIL in the bulk of it
            pcm = np.stack(list(gen.map(lambda i: ms.synth_clip(1000000 * k + rank * a.clips + i), range(a.clips))))
        d_bufs.append(hip.upload(dev, pcm))
        del pcm
    d_pcm = d_bufs[0]
    if dims.vocab > 50400:
        prompt, eot = [50258, 50259, 50359, 50363], 50257  # reference src/main.rs:549-566
    else:
        prompt, eot = [3, 5, 7, 9], 2
    params = wb.DecodeParams(prompt, a.max_new_tokens, eot, suppress_tokens=[eot])

    def barrier():
        hip.sync()
        if dist is not None:
            dist.barrier()

    pool = ThreadPoolExecutor(max_workers=a.streams)

    step_no = [0]

    def run_step(prefetch: bool = True):
        """One pass over this GPU's clips: every stream transcribes its slice concurrently (ctypes
        releases the GIL; each context owns a HIP stream).  Pipeline mode: one context; the call also puts log-mel +
        encoder of the NEXT step's batch on the encoder stream, where they run beside this step's token loop — every
        step therefore executes one full encoder pass and one full decode pass, nothing is cached or skipped."""
        if a.pipeline:
            k = step_no[0]
            step_no[0] += 1
            cur, nxt = d_bufs[k % 2], d_bufs[(k + 1) % 2]
            return ctx.transcribe_batch_device(cur, a.clips, params, next_ptr=nxt if prefetch else None, next_n=a.clips)
        def one(i):
            return ctxs[i].transcribe_batch_device(d_pcm + i * per_stream * 480000 * 4, per_stream, params)
        if a.streams == 1:
            return one(0)
        res = list(pool.map(one, range(a.streams)))
        return [t for r in res for t in r]

    # untimed: warmup + one fully profiled pass to find the dominant kernel group
    toks = None
    for _ in range(max(1, a.warmup)):
        toks = run_step(prefetch=False)
    assert all(len(t) == len(prompt) + a.max_new_tokens for t in toks), "EOT suppressed: every clip decodes max_new tokens"
    # parity of the timed entry (wh_transcribe_batch_device*, a full device batch) with a one-clip call on the same
    # context: rows 0 and 3 must be identical — a clip decodes the same alone or in a batch
    row_check = None
    if rank == 0 and not a.no_row_check:
        step_no[0] = 0
        full = run_step(prefetch=False)                 # batch 0 again, unpipelined
        for r in (0, min(3, a.clips - 1)):
            alone = ctx.transcribe_batch_device(d_bufs[0] + r * 480000 * 4, 1, params)[0]
            assert alone.tolist() == full[r].tolist(), f"row {r} of the device batch differs from the one-clip call"
        if a.pipeline:                                  # and the pipelined call returns what the plain call returns
            step_no[0] = 0
            piped = run_step(prefetch=True)
            piped2 = run_step(prefetch=False)           # consumes the prefetched encoder states of batch 1
            step_no[0] = 1
            plain2 = run_step(prefetch=False)           # batch 1 from scratch
            assert [t.tolist() for t in piped] == [t.tolist() for t in full], "pipelined call differs from the plain call"
            assert [t.tolist() for t in piped2] == [t.tolist() for t in plain2], "prefetched encoder states differ from recomputed ones"
        row_check = "rows 0 and 3 of the device batch == one-clip calls on the same context" + (
            "; pipelined == plain call on both batches" if a.pipeline else "")
    step_no[0] = 0
    for cx in ctxs:
        cx.profile_enable(True)
    run_step(prefetch=False)
    breakdown = {k: {"ms": sum(cx.profile_get()[k]["ms"] for cx in ctxs) / a.streams,
                     "launches": sum(cx.profile_get()[k]["launches"] for cx in ctxs)} for k in wb.KG_NAMES}
    # timed region: only the dominant kernel (decoder cross-attention) is bracketed by HIP events
    for cx in ctxs:
        if a.graph_timed:
            cx.profile_enable(False)
        elif a.profile_all:
            cx.profile_enable(True)
        else:  # sampled live timing: every 16th token position runs eagerly with events around the kernel
            cx.profile_enable(["dec_cross_attn"], stride=16)
    live = {"ms": 0.0, "launches": 0}

    # timed region: EXACTLY K steps
    lat = []
    stage = {"preprocess_s": 0.0, "encode_s": 0.0, "decode_s": 0.0}
    step_no[0] = 0
    if a.pipeline:
        # steady state: the last untimed step prefetches the first timed step's encoder pass, and EVERY timed step
        # (the last one too) runs the next batch's encoder pass beside its token loop — K encoder passes and K decode
        # passes execute inside the timed region; the closing barrier waits for the last encoder pass as well
        step_no[0] = 1
        run_step(prefetch=True)      # decodes batch 1, prefetches batch 0
        step_no[0] = 0
    barrier()
    t0 = time.perf_counter()
    t_prev = t0
    for _ in range(a.steps):
        ts = time.perf_counter()
        toks = run_step()   # returns after the last D2H of the step
        # per-clip latency: every clip of a batch completes with its batch; in pipeline mode a batch's encoder pass
        # was enqueued at the start of the PREVIOUS step, so its clips have been in flight since then
        lat.append(time.perf_counter() - (t_prev if a.pipeline else ts))
        t_prev = ts
        for cx in ctxs:
            tm = cx.timings()
            for k in stage:
                stage[k] += tm[k] / a.streams
            pg = cx.profile_get()["dec_cross_attn"]
            live["ms"] += pg["ms"]
            live["launches"] += pg["launches"]
    barrier()
    elapsed = time.perf_counter() - t0
    cross_es_mode = ctxs[0].cross_mode == 1
    placement = ctxs[0].placement   # (before the contexts are closed) the start-up step that looks where the workspace lies: DESIGN.md section 5e
    if dist is not None:
        from whisper_rust_ort_amd import sharding
        dev_t = "cuda" if backend == "nccl" else "cpu"
        elapsed = sharding.max_over_ranks(dist, elapsed, dev_t)
        # the path's only exchange: fixed-stride result records gathered to every rank
        rec = sharding.pack_records(sharding.shard_clip_ids(rank, world, a.clips), toks, len(prompt) + a.max_new_tokens)
        allrec = sharding.gather_records(dist, rec, dev_t)
        n_results = len(sharding.unpack_records(allrec))
        assert n_results == world * a.clips
    else:
        n_results = len(toks)
    # (the untimed side measurements below — one GPU only: with more ranks the others would sit in the process group's teardown meanwhile —
    # run AFTER the timed region: run before it, the dominant kernel read 265 instead of 250 us per launch in the
    # timed steps)
    # BASELINE configs[1] (batch = 1 clip on one GPU): per-clip end-to-end latency, untimed extra
    b1_ms = []
    if rank == 0 and world == 1 and not a.no_batch1:
        ctx1 = wb.Context(model, 1)
        for i in range(6):
            t1 = time.perf_counter()
            ctx1.transcribe_batch_device(d_pcm + (i % a.clips) * 480000 * 4, 1, params)
            if i:
                b1_ms.append((time.perf_counter() - t1) * 1e3)
        ctx1.close()
    # BASELINE configs[2] at its own 8-GPU shard size (512 clips / 8 = 64 per GPU) as ONE 64-clip batch: untimed extra
    b64 = None
    if rank == 0 and world == 1 and not a.no_batch1 and a.clips >= 64 and a.clips != 64:
        ctx64 = wb.Context(model, 64)
        t64 = []
        for i in range(4):
            t1 = time.perf_counter()
            ctx64.transcribe_batch_device(d_pcm, 64, params)
            if i:
                t64.append((time.perf_counter() - t1) * 1e3)
        ctx64.close()
        b64 = {"ms_per_batch": float(np.median(t64)), "rtfx": 64 * 30e3 / float(np.median(t64)),
               "note": "BASELINE configs[2]'s 8-GPU shard (64 clips) as one batch on one GPU; p95 per clip = the batch time"}
    # the 256-clip device batch rounds 1 and 2 quoted as the headline, for round-to-round comparison: untimed extra
    b256 = None
    if rank == 0 and world == 1 and not a.no_batch1 and a.clips > 256:
        ctx256 = wb.Context(model, 256)
        t256 = []
        for i in range(4):
            t1 = time.perf_counter()
            ctx256.transcribe_batch_device(d_pcm, 256, params)
            if i:
                t256.append((time.perf_counter() - t1) * 1e3)
        ctx256.close()
        b256 = {"ms_per_batch": float(np.median(t256)), "rtfx": 256 * 30e3 / float(np.median(t256)),
                "note": "one 256-clip device batch per call (the per-step workload of the round-1/2 lines)"}
    # ... and the 1024-clip device batch of the round-2 / round-3 lines
    b1024 = None
    if rank == 0 and world == 1 and not a.no_batch1 and a.clips > 1024:
        ctx1024 = wb.Context(model, 1024)
        t1024 = []
        for i in range(3):
            t1 = time.perf_counter()
            ctx1024.transcribe_batch_device(d_pcm, 1024, params)
            if i:
                t1024.append((time.perf_counter() - t1) * 1e3)
        ctx1024.close()
        b1024 = {"ms_per_batch": float(np.median(t1024)), "rtfx": 1024 * 30e3 / float(np.median(t1024)),
                 "note": "one 1024-clip device batch per call (the per-step workload of the round-2 / round-3 lines)"}

    # ---- the same clips from page-locked HOST memory (the reference's end_to_end_s counts the load, src/main.rs:1190): wh_transcribe_batch_next
    # copies batch i + 1 to the device on a copy stream beside batch i's work, so a step pays what is left of its own copy
    host_res = None
    if rank == 0 and world == 1 and not a.no_batch1 and not a.pipeline and a.streams == 1:
        try:
            hbuf = hip.host_alloc((a.clips, 480000), np.float32)
            with ThreadPoolExecutor(max_workers=max(1, min(16, cpu_budget()))) as gen:
                list(gen.map(lambda i: hbuf.__setitem__(i, ms.synth_clip(rank * a.clips + i)), range(a.clips)))
            rows = [hbuf[i] for i in range(a.clips)]
            ctx.profile_enable(False)
            got = ctx.transcribe_batch_next(rows, params, rows)          # untimed: copies this batch now, prefetches the next
            assert [t.tolist() for t in got[:4]] == [t.tolist() for t in toks[:4]], "host-resident entry differs from the device-resident one"
            hl = []
            hip.sync()
            th0 = time.perf_counter()
            for _ in range(3):
                t1 = time.perf_counter()
                ctx.transcribe_batch_next(rows, params, rows)
                hl.append(time.perf_counter() - t1)
            hip.sync()
            hel = time.perf_counter() - th0
            tmh = ctx.timings()
            host_res = {"rtfx": 30.0 * a.clips * 3 / hel, "ms_per_step": hel / 3 * 1e3, "p95_ms_per_clip": float(np.percentile(np.asarray(hl) * 1e3, 95)),
                        "h2d_wait_ms_last_step": tmh["h2d_s"] * 1e3, "h2d_bytes_per_step": int(a.clips) * 480000 * 4,
                        "note": "the same clips in page-locked host memory through wh_transcribe_batch_next: every step copies its successor's PCM "
                                "(1.92 MB per clip) host-to-device on a copy stream beside its own log-mel / encoder / token loop; 3 steps"}
            hip.host_free(hbuf)
        except Exception as e:   # a side measurement must not cost the line
            host_res = {"error": f"{type(e).__name__}: {e}"}
    # ---- BASELINE configs[2]'s own reading of scaling: a FIXED job of 512 clips dealt to the ranks (strong scaling), beside the weak-scaling headline
    strong = None
    if scaling == "weak" and not a.no_batch1 and not a.pipeline and (512 % world) == 0 and 512 // world <= a.clips:
        cs = 512 // world
        ctx_s = wb.Context(model, cs)
        ctx_s.transcribe_batch_device(d_pcm, cs, params)
        barrier()
        ts0 = time.perf_counter()
        for _ in range(3):
            ctx_s.transcribe_batch_device(d_pcm, cs, params)
        barrier()
        tse = time.perf_counter() - ts0
        ctx_s.close()
        if dist is not None:
            from whisper_rust_ort_amd import sharding
            tse = sharding.max_over_ranks(dist, tse, "cuda" if backend == "nccl" else "cpu")
        strong = {"total_clips_per_step": 512, "clips_per_gpu": cs, "steps": 3, "ms_per_step": tse / 3 * 1e3, "value": 30.0 * 512 * 3 / tse,
                  "note": "BASELINE configs[2]: 512 clips per step sharded over the GPUs of the job (fixed total work); the headline `value` is the weak-scaling reading"}
    # ---- the precisions that meet north_star's tolerance (tokens identical to the f32 reference, logits within 1e-3): the headline dtype is
    # BASELINE's bf16, which does not (tests/test_hip_parity.py bounds it at 0.16); these are measured in the same run, after the main context
    # has given its memory back
    in_tol = None
    if rank == 0 and world == 1 and not a.no_batch1 and prec in (wb.WH_PREC_BF16, wb.WH_PREC_FP8) and a.preset == "base":
        try:
            for cx in ctxs:
                cx.close()
            ctxs = []
            mx = wb.Model(f"synthetic:{a.preset}:{a.seed}", dev, wb.WH_PREC_F16X3)
            cxx = wb.Context(mx, a.clips)
            tx = cxx.transcribe_batch_device(d_pcm, a.clips, params)
            hip.sync()
            t1 = time.perf_counter()
            for _ in range(2):
                tx = cxx.transcribe_batch_device(d_pcm, a.clips, params)
            hip.sync()
            ex = (time.perf_counter() - t1) / 2
            tmx = cxx.timings()
            cxx.close()
            m32 = wb.Model(f"synthetic:{a.preset}:{a.seed}", dev, wb.WH_PREC_F32)
            n32 = min(512, a.clips)
            c32 = wb.Context(m32, n32)
            t32 = c32.transcribe_batch_device(d_pcm, n32, params)
            hip.sync()
            t1 = time.perf_counter()
            t32 = c32.transcribe_batch_device(d_pcm, n32, params)
            hip.sync()
            e32 = time.perf_counter() - t1
            c32.close()
            same = sum(int(tx[i].tolist() == t32[i].tolist()) for i in range(n32))
            in_tol = {"dtype": "f16x3", "clips_per_step": a.clips, "steps": 2, "ms_per_step": ex * 1e3, "rtfx": 30.0 * a.clips / ex, "p95_ms_per_clip": ex * 1e3,
                      "stage_ms": {k: tmx[k] * 1e3 for k in ("preprocess_s", "encode_s", "decode_s")},
                      "tokens_identical_to_exact_f32": f"{same}/{n32} clips x {a.max_new_tokens} tokens",
                      "exact_f32": {"dtype": "f32", "clips_per_step": n32, "steps": 1, "ms_per_step": e32 * 1e3, "rtfx": 30.0 * n32 / e32},
                      "note": "WH_PREC_F16X3: every contraction on the fp16 matrix cores with both operands as two fp16 limbs (22 significant bits), three MFMAs "
                              "per product, f32 everywhere else; held to the golden vectors at 1e-3 with tokens exact by tests/test_hip_parity.py "
                              "(measured 5e-5) like the exact-f32 MFMA mode beside it"}
        except Exception as e:
            in_tol = {"error": f"{type(e).__name__}: {e}"}
    # ---- BASELINE configs[4]'s precision on the same clips, in the same run (bf16 headline only): e4m3 weights, MX activations, cross-attention on e4m3 encoder states
    fp8_side = None
    if rank == 0 and world == 1 and not a.no_batch1 and prec == wb.WH_PREC_BF16 and a.preset == "base":
        try:
            for cx in ctxs:
                cx.close()
            ctxs = []
            m8 = wb.Model(f"synthetic:{a.preset}:{a.seed}", dev, wb.WH_PREC_FP8)
            c8 = wb.Context(m8, a.clips)
            c8.transcribe_batch_device(d_pcm, a.clips, params)
            hip.sync()
            t1 = time.perf_counter()
            for _ in range(3):
                c8.transcribe_batch_device(d_pcm, a.clips, params)
            hip.sync()
            e8 = (time.perf_counter() - t1) / 3
            tm8 = c8.timings()
            fp8_side = {"dtype": "fp8", "clips_per_step": a.clips, "steps": 3, "ms_per_step": e8 * 1e3, "rtfx": 30.0 * a.clips / e8,
                        "cross_mode": c8.cross_mode, "stage_ms": {k: tm8[k] * 1e3 for k in ("preprocess_s", "encode_s", "decode_s")},
                        "note": "WH_PREC_FP8 (BASELINE configs[4]): e4m3 weights, MX activations on the fp8 matrix cores in the encoder, the token loop's cross-attention on "
                                "e4m3 encoder states (k_dec_cross_attn_es8); accuracy: profiles/r04_accuracy_64clips.json, tests/test_fp8_gpu.py"}
            c8.close()
        except Exception as e:
            fp8_side = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        cross_es = cross_es_mode
        # bytes per element of what the cross-attention streams: e4m3 (fp8 mode), fp16 + e4m3 remainder (split-fp16 mode on the encoder states, unless WH_ES3=0), else the compute dtype
        es3 = cross_es and prec == wb.WH_PREC_F16X3 and os.environ.get("WH_ES3", "1") != "0"
        work = algorithmic_work(dims, per_stream, len(prompt), a.max_new_tokens, 1 if prec == wb.WH_PREC_FP8 else 3 if es3 else esz, cross_es)
        audio_s = 30.0 * a.clips * a.steps * world
        ms_per_step = elapsed / a.steps * 1e3
        # Roofline of the dominant kernel (the decoder cross-attention: largest single-kernel share in every rocprofv3 --stats
        # summary under profiles/).  `achieved` = the algorithmic bytes of one launch — the encoder states of every clip of the launch
        # once (k_dec_cross_attn_es*), or K and V of one decoder layer (k_dec_cross_attn*) — ÷ the average launch duration
        # measured with HIP events on the launch stream over the TIMED region.
        tot_ms = sum(v["ms"] for v in breakdown.values())
        if a.graph_timed:
            live = {"ms": breakdown["dec_cross_attn"]["ms"] * a.streams, "launches": breakdown["dec_cross_attn"]["launches"]}
        avg_s = live["ms"] * 1e-3 / max(1, live["launches"])
        ach = work["cross_attn_bytes_per_launch"] / avg_s / 1e9
        traffic = None
        traffic_stamp = None
        # the dominant kernel's variants: e4m3 cache (fp8 mode), one workgroup per 256-column group (wide models), all heads per workgroup
        dom_kernel = ("k_dec_cross_attn_es3" if es3 else "k_dec_cross_attn_es2" if (cross_es and prec == wb.WH_PREC_F16X3) else "k_dec_cross_attn_es8" if (cross_es and prec == wb.WH_PREC_FP8) else
                      "k_dec_cross_attn_es" if cross_es else "k_dec_cross_attn8" if prec == wb.WH_PREC_FP8 else
                      "k_dec_cross_attn_cg" if (prec == wb.WH_PREC_BF16 and dims.d_model > 512 and dims.d_model % 256 == 0) else "k_dec_cross_attn")
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):   # HBM bytes per launch from rocprofv3 --pmc passes (see profiles/README.md)
            tj = json.load(open(tpath))
            key = f"{a.preset}_{a.precision}_b{per_stream}"
            traffic = tj.get(dom_kernel, {}).get(key)
            traffic_stamp = tj.get("_collected_at", {}).get(key)
        roofline = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": traffic, "traffic_source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profiles/pmc_traffic.json (profiles/collect.sh); "
                                                          "NOT measured in this run — counters of the build named in traffic_collected_at",
                    "traffic_collected_at": traffic_stamp,
                    "kernel": dom_kernel, "avg_launch_us": avg_s * 1e6,
                    "launches_timed": live["launches"], "alg_bytes_per_launch": work["cross_attn_bytes_per_launch"],
                    "streams": ("the encoder states [clips][1500][d] once per launch (cross-attention on the encoder states; the projected K + V "
                                "form of the same attention reads twice these bytes)" if cross_es else "K and V of one decoder layer for every clip of the launch"),
                    "share_of_kernel_time": breakdown["dec_cross_attn"]["ms"] / tot_ms}
        d_, F_, T_, Le_ = dims.d_model, dims.ffn, dims.n_audio_ctx, dims.enc_layers
        attn_flop = Le_ * 2 * 2 * T_ * T_ * d_ * a.clips
        gemm_flop = work["enc_flop_per_clip"] * a.clips - attn_flop
        # TFLOP/s dense bf16 / fp16; exact-f32 MFMA is 1/16 of it; the split-fp16 mode spends three fp16 MFMAs per product
        mfma_peak = {wb.WH_PREC_F32: 2500.0 / 16.0, wb.WH_PREC_F16X3: 2500.0 / 3.0}.get(prec, 2500.0)

        def sec(bound, work_units, ms, peak, unit, note):
            ach = work_units / (ms * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9) if ms > 0 else 0.0
            return {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "ms_per_step": ms, "note": note}
        secondary = {
            "enc_attn": sec("mfma", attn_flop, breakdown["enc_attn"]["ms"], mfma_peak, "TFLOP/s",
                            "k_enc_attn; VALU-bound below the MFMA roof (one v_exp_f32 per 256 MFMA flops at head_dim 64)"),
            "enc_gemm": sec("mfma", gemm_flop, breakdown["enc_gemm"]["ms"], mfma_peak, "TFLOP/s",
                            "k_gemm8 / k_gemm8_mx (Conv1d-as-GEMM, QK, V^T, out-proj, fc1, fc2); the group's time also holds the 13 encoder LayerNorm launches"),
            "mel": sec("hbm", work["mel_bytes_per_clip"] * a.clips, breakdown["mel"]["ms"], HBM_PEAK_GBS, "GB/s",
                       "k_mel_stft (mixed-radix FFT in LDS + f32-MFMA filterbank) + k_mel_tokens; latency-bound per 16-frame workgroup"),
        }
        # rocprof-reported MFMA utilisation of the same configuration (profiles/collect_mfma.sh: SQ_VALU_MFMA_BUSY_CYCLES over
        # GRBM_GUI_ACTIVE, i.e. matrix-pipe busy cycles over the SIMD cycles that actually elapsed — the shader clock under
        # these kernels is 2.0-2.2 GHz, not the 2.4 GHz behind the 2.5 PFLOP/s peak; a calibration loop of nothing but MFMAs
        # reads 0.91-0.98 at 1.9-2.0 GHz, profiles/r02_mfma_util_calibration.txt)
        upath = os.path.join(ROOT, "profiles", "mfma_util.json")
        if os.path.exists(upath):
            ujall = json.load(open(upath))
            ukey = f"{a.preset}_{a.precision}_b{per_stream}"
            uj = ujall.get(ukey, {})
            ustamp = ujall.get("_collected_at", {}).get(ukey)
            def grp(prefixes):
                ks = [k for k in uj if k.startswith(prefixes) and uj[k].get("shader_clock_GHz")]
                busy = sum(uj[k]["mfma_util"] * uj[k]["total_ms_under_pmc"] * uj[k]["shader_clock_GHz"] for k in ks)
                cyc = sum(uj[k]["total_ms_under_pmc"] * uj[k]["shader_clock_GHz"] for k in ks)
                return {"mfma_util": busy / cyc, "kernels": {k: uj[k]["mfma_util"] for k in ks},
                        "collected_at": ustamp, "note": "counter pass of the build named in collected_at, not of this run"} if cyc > 0 else None
            secondary["enc_attn"]["rocprof"] = grp(("k_enc_attn",))
            secondary["enc_gemm"]["rocprof"] = grp(("k_gemm8", "k_gemm<"))
        out = {
            "metric": f"rtfx: audio seconds transcribed per wall second (whisper-{a.preset}, 30 s clips, greedy {a.max_new_tokens} new tokens)",
            "value": audio_s / elapsed, "unit": "x real time", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"whisper-{a.preset} dims, hash-seeded weights, {a.clips} synthetic 30 s clips per GPU per step "
                                   f"(BASELINE configs[2]'s clip generator and decode settings, model/dtype of configs[1]; one device batch "
                                   f"per step), greedy, max_new_tokens={a.max_new_tokens}, EOT suppressed, PCM resident in HBM",
                       "clips_per_gpu": a.clips, "streams_per_gpu": a.streams, "parallelism": f"clip-sharded x{world}",
                       "total_clips_per_step": a.clips * world,
                       "pipeline": ({"enc_cus": a.enc_cus or "all", "dec_cus": a.dec_cus or "all",
                                     "note": "step i's token loop and step i+1's log-mel + encoder run side by side on disjoint compute units; "
                                             "every timed step executes one encoder pass and one decode pass"} if a.pipeline else None),
                       "cross_attention": ("on the encoder states (k_dec_cross_attn_es: S x d bf16 per clip, layer and token; the K projection as a query-side expansion kernel, the V "
                                           "projection as a grouped decode GEMM behind it)" if cross_es else "on the projected K / V cache (2 S x d per clip, layer and token)"),
                       "gather": backend, "results_gathered": n_results, "row_check": row_check},
            "rtf": elapsed / audio_s,   # reference definition: latency / duration (src/main.rs:1191)
            "clips_per_s": a.clips * a.steps * world / elapsed,
            "p95_ms_per_clip": float(np.percentile(np.asarray(lat) * 1e3, 95)),  # every clip of a batch completes with its batch
            "batch1": ({"p95_ms_per_clip": float(np.percentile(b1_ms, 95)), "median_ms_per_clip": float(np.median(b1_ms)),
                        "rtfx": 30e3 / float(np.median(b1_ms)), "note": "BASELINE configs[1]: one clip per call on one GPU"}
                       if b1_ms else None),
            "batch64": b64,
            "batch256": b256,
            "batch1024": b1024,
            "host_resident": host_res,
            "workspace_placement": placement,
            "in_tolerance": in_tol,
            "fp8": fp8_side,
            "scaling_weak": {"clips_per_gpu": a.clips, "total_clips_per_step": a.clips * world, "ms_per_step": ms_per_step, "value": audio_s / elapsed} if scaling == "weak" else None,
            "scaling_strong": strong if scaling == "weak" else {"total_clips_per_step": a.clips * world, "clips_per_gpu": a.clips, "ms_per_step": ms_per_step, "value": audio_s / elapsed},
            "stage_ms_per_step": {k: v / a.steps * 1e3 for k, v in stage.items()},
            "kernel_group_ms_per_step": {k: round(v["ms"], 3) for k, v in breakdown.items()},
            "kernel_group_launches": {k: v["launches"] for k, v in breakdown.items()},
            "kernel_time_frac_of_step": tot_ms / ms_per_step,
            "roofline": roofline,
            # the other kernel groups against their own roofs (event-timed in the untimed profiled pass; SURVEY §8d:
            # MFMA utilisation of the encoder attention / GEMMs, HBM rate of the log-mel path)
            "secondary_rooflines": secondary,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(dims, a.seed, prompt, eot, a.max_new_tokens)
            out["cpu_baseline"]["note"] = ("the reference's Rust/ONNX Runtime binary cannot be built or run here (no Rust, "
                                           "no ORT, no network); published: 14.03 s model time for 12 windows on 4 EPYC-9654 cores")
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
