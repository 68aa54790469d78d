#!/usr/bin/env python
"""bench.py — the reference's headline benchmark on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model bear] [--size 1024]

A "step" is one whole frame of the hot path: Context::render3D of benchmark/files/bear.frep
at 1024^3 with the reference's benchmark view (identity, T(3,2) = 0.3;
benchmark/render_3d_table.cpp:48-49), a blocking call like the reference's get_stats protocol
times (benchmark/stats.cpp:19-47).  The tape is resident in HBM before the timed region.
With N > 1 the top-level tile columns of the SAME frame are dealt to the N ranks and the
image is all-gathered over RCCL (strong scaling; mpr_amd/multigpu.py).

Rank 0 prints ONE JSON line.  Extra objects on that line:
  roofline      dominant kernel (eval_voxels_f): algorithmic bytes / HIP-event duration vs 8 TB/s; the
                kernel's VALU / scalar / LDS issue fractions (its real limiter) from the committed
                counter passes; the frame-level B_alg of SURVEY.md 8(d)
  cpu_baseline  the CPU oracle (a port, not the reference): all host cores at 512^3 and one core at
                256^3, warm-up + 3 frames each, beside the GPU's time at the same size; the oracle's
                frame must equal the GPU's
  full_frames   the same frame when every frame pushes the last tile stage's tapes (the first frame of a
                tape or view; the timed frames repeat one view and do not need them)
  also          the reference's other headline config (prospero render2D 1024^2), for which
                BASELINE.md holds the only published number (V100, 3.856 ms/frame)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
V100_PROSPERO_1024_MS = 3.85596     # BASELINE.md / reference README.md:111
# issue rates measured on this chip, wave-instructions per clock per CU at the nominal 2.4 GHz
# (profiles/r02a_issue_rates.txt, scripts/ubench/issue_rates2.hip): v_fma_f32 / v_add_f32 chains, s_add_u32, ds_read_b32
VALU_RATE, SALU_RATE, LDS_RATE = 1.75, 0.96, 0.48
ISSUE_CLOCK_HZ, ISSUE_CUS = 2.4e9, 256


def view3():
    T = np.eye(4, dtype=np.float32)
    T[3, 2] = 0.3
    return T


def stats(ms):
    ms = np.asarray(ms, dtype=np.float64)
    return float(ms.mean()), float(ms.std(ddof=1)) if ms.size > 1 else 0.0


def time_frames(fn, warmup, steps, barrier, sync):
    for _ in range(warmup):
        fn()
    barrier()
    sync()
    per = []
    t0 = time.perf_counter()
    for _ in range(steps):
        s = time.perf_counter()
        fn()
        per.append((time.perf_counter() - s) * 1e3)
    sync()
    barrier()
    total = time.perf_counter() - t0
    return total, per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # benchmark/stats.hpp:14: 100 timed
    ap.add_argument("--warmup", type=int, default=20)     #                         20 warm-up
    ap.add_argument("--model", default="bear")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the prospero 2-D side measurement")
    args = ap.parse_args()

    import torch
    import mpr_amd as m

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # MPR_BENCH_BACKEND=gloo + MPR_BENCH_SHARE_GPU=1: development check of the N > 1 code path on a
    # box with a single GPU (all ranks on device 0, gather through gloo); not a measurement
    share = os.environ.get("MPR_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("MPR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        torch.cuda.synchronize()

    m.build()
    tape = m.Tape(m.model(args.model))
    S = args.size
    T = view3()

    # ---- instrumented frame (counters on): algorithmic work of the dominant kernel ----
    cctx = m.Context(S, device=local_rank, flags=m.CTX_COUNTERS)
    cctx.render3D(tape, T)
    work = cctx.counters()
    cctx.close()

    ctx = m.Context(S, device=local_rank, flags=m.CTX_TIMING)
    kernel_ms = {}
    frames_timed = [0]

    if world > 1:
        from mpr_amd.multigpu import TileParallelRenderer

        def make_buffer(n):
            t = torch.empty(n, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            return t, t.data_ptr()

        # the collective is ordered after the context's stream (render + pack) and the unpack after
        # the collective, all on the device: no host synchronisation inside a frame
        ctx_stream = torch.cuda.ExternalStream(ctx.stream)

        def all_gather(out, inp):
            with torch.cuda.stream(ctx_stream):
                dist.all_gather_into_tensor(out, inp)

        tpr = TileParallelRenderer(ctx, m, rank, world, make_buffer, all_gather, dim=3)
        tpr.plan(tape, T)
        # this rank's share of the dominant kernel's algorithmic work: an instrumented frame of its columns
        pctx = m.Context(S, device=local_rank, flags=m.CTX_COUNTERS)
        pctx.render3D_part(tape, T, tpr.owner, rank)
        work = pctx.counters()
        pctx.close()

        def frame():
            tpr.render(tape, T)
    else:
        def frame():
            ctx.render3D(tape, T)

    def timed_frame():
        frame()
        for name, ms in ctx.timings():
            kernel_ms[name] = kernel_ms.get(name, 0.0) + ms
        frames_timed[0] += 1

    verified = None
    if world > 1 and os.environ.get("MPR_BENCH_VERIFY") == "1":
        # the gathered frame of every rank against a full single-GPU frame
        frame()
        got_h, got_n = ctx.image.copy(), ctx.normals.copy()
        ref = m.Context(S, device=local_rank)
        ref.render3D(tape, T)
        verified = bool((got_h == ref.image).all() and (got_n == ref.normals).all())
        ref.close()
        vt = torch.tensor([1 if verified else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(vt, op=dist.ReduceOp.MIN)
        verified = bool(vt.item())
    for _ in range(args.warmup):
        frame()
    barrier()
    sync()
    t0 = time.perf_counter()
    per = []
    for _ in range(args.steps):
        s = time.perf_counter()
        timed_frame()
        per.append((time.perf_counter() - s) * 1e3)
    sync()
    barrier()
    total_s = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([total_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total_s = float(tt.item())
    ms_per_step = total_s * 1e3 / args.steps
    mean_ms, std_ms = stats(per)

    # ---- roofline of the dominant kernel (eval_voxels_f) ----
    nframes = max(frames_timed[0], 1)
    avg = {k: v / nframes for k, v in kernel_ms.items()}
    vox_ms = avg.get("eval_voxels_f", 0.0)
    kname = ctx.float_kernel()
    # algorithmic bytes per launch: one 8-byte clause per wave-group visit (group = the 64
    # voxels of one smallest tile) + the 12-byte tile record of every smallest tile
    b_alg = 8 * work["clauses_fwd_voxels"] + 12 * work["voxel_tiles"]      # N > 1: rank 0's columns, rank 0's kernel time
    # ... and of the whole frame: B_alg = 8(F + R) + 8W + 12(T_in + T_out) + 4(I_w + I_r), SURVEY.md 8(d)
    t_in = sum(work["tiles_in"]) + work["voxel_tiles"]
    t_out = 64 * sum(work["tiles_active"][:2]) + work["tiles_active"][2]
    levels = sum((S // px) ** 2 for px in (64, 16, 4))
    i_w = 2 * S * S + levels            # heights + normals + the level images
    i_r = 2 * S * S                     # copy_filled + the normals pass
    b_frame = 8 * (work["clauses_fwd"] + work["clauses_bwd"]) + 8 * work["clauses_written"] + 12 * (t_in + t_out) + 4 * (i_w + i_r)
    roofline = None
    if vox_ms > 0 and b_alg:
        achieved = b_alg / (vox_ms * 1e-3) / 1e9
        traffic = traffic_source = issue = None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        if world == 1 and os.path.exists(pmc):         # the counter passes were collected on the full single-GPU frame
            try:
                with open(pmc) as f:
                    summ = json.load(f)
                rec = summ.get("eval_voxels_f", {})
                if rec.get("kernel") == kname:          # counters of another kernel say nothing about this one
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/pmc_summary.json: " + summ.get("_source", "?") + " (separate rocprofv3 --pmc passes of this command, not this run)"
                    sq = rec.get("sq")
                    if sq:
                        # fractions of the issue rates measured on this chip (profiles/r02a_issue_rates.txt): wave-instructions
                        # per second = rate per clock per CU x 2.4 GHz x CUs
                        dur = vox_ms * 1e-3
                        peak = lambda r: r * ISSUE_CLOCK_HZ * ISSUE_CUS
                        issue = {"valu_per_launch": sq.get("SQ_INSTS_VALU"), "salu_per_launch": sq.get("SQ_INSTS_SALU"),
                                 "lds_per_launch": sq.get("SQ_INSTS_LDS"),
                                 "valu_frac": round(sq.get("SQ_INSTS_VALU", 0) / dur / peak(VALU_RATE), 4),
                                 "salu_frac": round(sq.get("SQ_INSTS_SALU", 0) / dur / peak(SALU_RATE), 4),
                                 "lds_frac": round(sq.get("SQ_INSTS_LDS", 0) / dur / peak(LDS_RATE), 4),
                                 "rates_per_clk_per_cu": {"valu": VALU_RATE, "salu": SALU_RATE, "lds": LDS_RATE},
                                 "clock_ghz": ISSUE_CLOCK_HZ / 1e9, "cus": ISSUE_CUS,
                                 "busy_cu_clock_ghz": round(sq.get("SQ_BUSY_CU_CYCLES", 0) / ISSUE_CUS / dur / 1e9, 3) if sq.get("SQ_BUSY_CU_CYCLES") else None,
                                 "source": traffic_source}
            except Exception:
                traffic = traffic_source = issue = None
        frame_s = ms_per_step * 1e-3
        roofline = {"kernel": kname + (" (rank 0 of %d)" % world if world > 1 else ""), "bound": "hbm",
                    "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": traffic_source,
                    "algorithmic_bytes": int(b_alg), "kernel_ms": round(vox_ms, 4),
                    "limiter": "valu issue (see issue.valu_frac)" if issue else "valu issue",
                    "issue": issue,
                    "kernel_ms_all": {k: round(v, 4) for k, v in avg.items()},
                    "frame": {"algorithmic_bytes": int(b_frame), "achieved": round(b_frame / frame_s / 1e9, 2),
                              "frac": round(b_frame / frame_s / 1e9 / HBM_PEAK_GBS, 5),
                              "terms": {"F": int(work["clauses_fwd"]), "R": int(work["clauses_bwd"]), "W": int(work["clauses_written"]),
                                        "T_in": int(t_in), "T_out": int(t_out), "I_w": int(i_w), "I_r": int(i_r)}},
                    "lane_clauses_per_frame": int(work["lane_clauses"])}

    out = None
    if rank == 0:
        out = {
            "metric": "Mpixel/s, render3D bear 1024^3 (ms/frame in ms_per_step)",
            "value": round(S * S / (ms_per_step * 1e-3) / 1e6, 3),
            "unit": "Mpixel/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_frame_mean": round(mean_ms, 4),
            "ms_per_frame_std": round(std_ms, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "benchmark model %s.frep (copied from the reference's benchmark/files), identity view with T(3,2)=0.3" % args.model,
            "config": {"workload": "%s.frep render3D heightmap+normals at %d^3" % (args.model, S),
                       "image_size_px": S, "tape_clauses": tape.length - 2, "tape_slots": tape.num_slots,
                       "parallelism": "tile-columns x%d" % world,
                       "voxel_tiles": int(work["voxel_tiles"])},
            "roofline": roofline,
        }
        if verified is not None:
            out["verified_against_single_gpu"] = verified

    # ---- the same frame when every frame pushes the last tile stage's tapes (what the FIRST frame of a tape / view costs:
    #      repeated frames do not need those tapes, DESIGN.md 3 "Frames whose last tile stage pushes no tapes") ----
    if rank == 0 and world == 1:
        os.environ["MPR_LAST_STAGE_PUSH"] = "1"
        fctx = m.Context(S, device=local_rank)
        del os.environ["MPR_LAST_STAGE_PUSH"]
        _, fper = time_frames(lambda: fctx.render3D(tape, T), min(args.warmup, 5), min(args.steps, 30), lambda: None, sync)
        fm, fs = stats(fper)
        same = bool(np.array_equal(fctx.image, ctx.image) and np.array_equal(fctx.normals, ctx.normals))
        fctx.close()
        out["full_frames"] = {"ms_per_frame_mean": round(fm, 4), "ms_per_frame_std": round(fs, 4),
                              "value": round(S * S / (fm * 1e-3) / 1e6, 3), "unit": "Mpixel/s", "same_images": same,
                              "note": "every frame pushes the last tile stage's tapes (MPR_LAST_STAGE_PUSH=1): the cost of the first "
                                      "frame of a tape or view; the timed frames above repeat one view, as benchmark/stats.cpp does"}

    # ---- side measurement: prospero render2D 1024^2 (the published V100 number's config) ----
    if rank == 0 and world == 1 and not args.no_also:
        ptape = m.Tape(m.model("prospero"))
        pctx = m.Context(1024, device=local_rank)
        _, pper = time_frames(lambda: pctx.render2D(ptape), args.warmup, args.steps, lambda: None, sync)
        pm, ps = stats(pper)
        pctx.close()
        out["also"] = [{"workload": "prospero.frep render2D at 1024^2", "ms_per_frame_mean": round(pm, 4),
                        "ms_per_frame_std": round(ps, 4), "value": round(1024 * 1024 / (pm * 1e-3) / 1e6, 2),
                        "unit": "Mpixel/s", "vs_baseline": round(V100_PROSPERO_1024_MS / pm, 3),
                        "baseline": "3.85596 ms/frame on 1x V100 (reference README.md:111)"}]

    # ---- CPU baseline: the oracle (a port of the algorithm, NOT libfive's renderer), as the checker of two
    #      smaller frames of the same model and as the timed CPU leg: one core, and all cores ----
    if rank == 0 and world == 1 and not args.no_cpu:
        from oracle import orc
        orc.lib()
        cores = os.cpu_count() or 1
        legs = []
        for cs, threads_list, frames in ((min(S, 256), [1], 3), (min(S, 512), sorted({cores, min(cores, 32)}), 3)):
            gctx = m.Context(cs, device=local_rank)
            _, gper = time_frames(lambda: gctx.render3D(tape, T), 5, 20, lambda: None, sync)
            gpu_ms, _ = stats(gper)
            gimg, gnrm = gctx.image.copy(), gctx.normals.copy()
            gctx.close()
            # warm-up frame per thread count (also picks the better count when there are two), then `frames` timed ones
            best = None
            for th in threads_list:
                t1 = time.perf_counter()
                fr = orc.Frame(tape.data, 3, cs, m.colmajor(T, 4), threads=th, keep_pool=False)
                dt = time.perf_counter() - t1
                if best is None or dt < best[1]:
                    best = (th, dt)
            same = bool(np.array_equal(fr.filled[3], gimg) and np.array_equal(fr.normals, gnrm))
            th = best[0]
            per_cpu = []
            for _ in range(frames):
                t1 = time.perf_counter()
                orc.Frame(tape.data, 3, cs, m.colmajor(T, 4), threads=th, keep_pool=False)
                per_cpu.append((time.perf_counter() - t1) * 1e3)
            cm, csd = stats(per_cpu)
            legs.append({"value": round(cs * cs / (cm * 1e-3) / 1e6, 4), "unit": "Mpixel/s", "cores": th, "kind": "port",
                         "ms_per_frame": round(cm, 1), "ms_per_frame_std": round(csd, 1), "frames": frames, "warmup": len(threads_list),
                         "gpu_ms_per_frame_same_size": round(gpu_ms, 4), "gpu_over_cpu": round(cm / gpu_ms, 1),
                         "frame_matches_gpu": same,
                         "sample": "%s.frep render3D at %d^3 (1/%d of the voxels of the GPU workload), oracle/mpr_oracle.c%s" %
                                   (args.model, cs, (S // cs) ** 3, " with OpenMP over tile groups" if th > 1 else ", one thread")})
        out["cpu_baseline"] = dict(legs[-1])
        out["cpu_baseline"]["one_core"] = legs[0]
        if not all(l["frame_matches_gpu"] for l in legs):
            raise SystemExit("bench.py: the oracle's frame differs from the GPU's: " + json.dumps(out["cpu_baseline"]))
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
