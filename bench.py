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
  roofline      dominant kernel (eval_voxels_f).  The kernel is VALU-issue-bound, so the primary figure is
                bound = "valu": VALU wave-instructions per launch (SQ counters, committed passes of this command) over the
                HIP-event duration, against 2.0 wave-instr/clk/CU x 2.4 GHz x 256 CUs (MI355X_MICROARCH.md) and against
                the rate measured on this chip (profiles/r03a_issue_rates3.txt: it sustains ~2.1-2.2 GHz under VALU load),
                plus a rate-weighted fraction (half- and quarter-rate instructions at their own ceilings).  The HBM figure
                north_star asks for rides in roofline.hbm (algorithmic bytes / duration vs 8 TB/s) with `traffic`; the
                frame-level B_alg of SURVEY.md 8(d) in roofline.frame
  cpu_baseline  the CPU oracle (a port, not the reference) AT THE BENCH CONFIGURATION: all host cores, 1 warm-up + 3 frames,
                and one core on the columns a 16-way deal of the same frame gives rank 0; the oracle's frame must equal the GPU's
  full_frames   the same frame rendered the reference's way throughout (every tile stage evaluated and pushed; what a reader
                of tiles / tapes makes the context do), and first_frames: the first frame of a tape the context has not seen
  also          the reference's other headline config (prospero render2D 1024^2), for which
                BASELINE.md holds the only published number (V100, 3.856 ms/frame)

    python bench.py --all [--out profiles/r03_records.jsonl]     one JSON record per model x size
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
V100_PROSPERO_1024_MS = 3.85596     # BASELINE.md / reference README.md:111
# VALU issue ceilings, wave-instructions per second for the chip:
#  guide: a wave64 VALU op is two passes of a 32-lane SIMD, 4 SIMDs per CU -> 2.0 / clk / CU at the nominal 2.4 GHz (MI355X_MICROARCH.md)
#  measured (scripts/ubench/issue_rates3.hip -> profiles/r03a_issue_rates3.txt, 8 waves/SIMD, independent chains, s_setprio):
#  v_add / v_mul 4.18-4.27, v_fma 3.69, v_mov 4.51 wave-instr/ns/CU — the same 2.0 / clk at the 2.1-2.2 GHz the chip sustains
#  under VALU load (round 2's "1.75 / clk" was this rate divided by the nominal clock); half-rate classes ~half of it, v_exp /
#  v_rcp / v_sqrt 0.49 / clk at nominal = 1.18 / ns / CU
ISSUE_CLOCK_HZ, ISSUE_CUS = 2.4e9, 256
VALU_GUIDE_PER_CLK = 2.0
VALU_MEASURED_PER_NS, VALU_HALF_PER_NS, VALU_QUARTER_PER_NS = 4.2, 2.1, 1.18
SALU_PER_NS, LDS_PER_NS = 0.96 * 2.4, 0.48 * 2.4        # profiles/r02a_issue_rates.txt (per nominal clock)

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


def frame_b_alg(work, S, dim=3):
    """B_alg = 8(F + R) + 8W + 12(T_in + T_out) + 4(I_w + I_r); returns (bytes, terms)"""
    nst = 3 if dim == 3 else 2
    t_in = sum(work["tiles_in"][:nst]) + work["voxel_tiles"]
    t_out = 64 * sum(work["tiles_active"][:nst - 1]) + work["tiles_active"][nst - 1]
    levels = sum((S // px) ** 2 for px in ((64, 16, 4) if dim == 3 else (64, 8)))
    i_w = (2 if dim == 3 else 1) * S * S + levels        # heights (+ normals) + the level images
    i_r = (2 if dim == 3 else 1) * S * S                 # copy_filled (+ the normals pass)
    b = 8 * (work["clauses_fwd"] + work["clauses_bwd"]) + 8 * work["clauses_written"] + 12 * (t_in + t_out) + 4 * (i_w + i_r)
    return b, {"F": int(work["clauses_fwd"]), "R": int(work["clauses_bwd"]), "W": int(work["clauses_written"]),
               "T_in": int(t_in), "T_out": int(t_out), "I_w": int(i_w), "I_r": int(i_r)}


def roofline_of(work, S, vox_ms, kname, avg, ms_per_step, world, model, walked=None):
    if not (vox_ms > 0 and work["clauses_fwd_voxels"]):
        return None
    # algorithmic bytes per launch: one 8-byte clause per wave-group visit (group = the 64 voxels of one
    # smallest tile) + the 12-byte tile record of every smallest tile.  N > 1: rank 0's columns and kernel time
    b_alg = 8 * work["clauses_fwd_voxels"] + 12 * work["voxel_tiles"]
    units = None
    if walked and walked.get("walked", -1) >= 0:
        # round 6: counted over the tiles the TIMED frames' kernel actually walks, not over the tiles the reference's enclosures leave
        # ambiguous (the instrumented frame's `work`): the last tile stage's second verdict keeps most of those out of the float pass
        # (mpr_amd/csrc/interval_gen.hpp: tight code), and the pass by column stops a column at its first hidden tile.  Clauses per
        # walked tile: the instrumented frame's mean over its own tiles (the tapes are the same tapes)
        per_tile = work["clauses_fwd_voxels"] / max(work["voxel_tiles"], 1)
        b_alg = int(8 * per_tile * walked["walked"] + 12 * walked["listed"])
        units = {"tiles_walked": int(walked["walked"]), "tiles_listed": int(walked["listed"]), "tiles_the_reference_lists": int(work["voxel_tiles"]),
                 "clauses_per_walked_tile": round(per_tile, 1), "source": "a context made with MPR_DEBUG_WALKED=1 after the timed loop (same frames)"}
    b_frame, terms = frame_b_alg(work, S)
    dur = vox_ms * 1e-3
    hbm_achieved = b_alg / dur / 1e9
    traffic = traffic_source = sq = None
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
        except Exception:
            traffic = traffic_source = sq = None
    hbm = {"bound": "hbm", "achieved": round(hbm_achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(hbm_achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes": int(b_alg),
           "note": "the framing north_star asks for; the kernel reads one tape per 64 tiles and keeps slots in registers, so its "
                   "real traffic (`traffic`) is a fraction of the algorithmic bytes and HBM is not what bounds it"}
    frame_s = ms_per_step * 1e-3
    common = {"units": units, "traffic": traffic, "traffic_source": traffic_source, "kernel_ms": round(vox_ms, 4),
              "kernel_ms_all": {k: round(v, 4) for k, v in avg.items()},
              "frame": {"algorithmic_bytes": int(b_frame), "achieved": round(b_frame / frame_s / 1e9, 2),
                        "frac": round(b_frame / frame_s / 1e9 / HBM_PEAK_GBS, 5), "terms": terms},
              "lane_clauses_per_frame": int(work["lane_clauses"])}
    name = kname + (" (rank 0 of %d)" % world if world > 1 else "")
    if not sq or not sq.get("SQ_INSTS_VALU"):
        out = {"kernel": name}
        out.update({k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac", "algorithmic_bytes")})
        out["limiter"] = "valu issue (no committed SQ counters for this kernel: profiles/pmc_summary.json)"
        out["hbm_frac"] = hbm["frac"]
        out["hbm_achieved_gbs"] = hbm["achieved"]
        out.update(common)
        return out
    valu = float(sq["SQ_INSTS_VALU"])
    guide_peak = VALU_GUIDE_PER_CLK * ISSUE_CLOCK_HZ * ISSUE_CUS / 1e9            # G wave-instr / s
    measured_peak = VALU_MEASURED_PER_NS * ISSUE_CUS
    achieved = valu / dur / 1e9
    out = {"kernel": name, "bound": "valu", "achieved": round(achieved, 2), "peak": round(guide_peak, 1), "unit": "Gwave-instr/s",
           "frac": round(achieved / guide_peak, 4),
           "peak_source": "MI355X_MICROARCH.md: wave64 VALU op = 2 passes of a SIMD-32, 4 SIMDs/CU -> 2.0 wave-instr/clk/CU x 2.4 GHz x 256 CUs",
           "valu_per_launch": valu,
           "measured_ceiling": {"peak": round(measured_peak, 1), "frac": round(achieved / measured_peak, 4),
                                "source": "profiles/r03a_issue_rates3.txt: v_add/v_mul 4.2 wave-instr/ns/CU at 8 waves/SIMD = 2.0/clk at the "
                                          "2.1-2.2 GHz the chip sustains under VALU load"}}
    mixp = os.path.join(ROOT, "profiles", "valu_mix.json")
    if os.path.exists(mixp):
        try:
            with open(mixp) as f:
                mix = json.load(f)
            # (k_eval_voxels_gen_fp walks the same generated code as k_eval_voxels_gen: only the hand-out differs)
            m = mix.get(kname.split("<")[0].replace("_gen_fp", "_gen"), {}).get(model)
            if m:
                # the instruction stream's classes (scripts/valu_mix.py: generated-code templates + routines x the tape's opcode
                # histogram) scaled to the measured total; each class against its own measured ceiling
                t_issue = valu * (m["full"] / VALU_MEASURED_PER_NS + m["half"] / VALU_HALF_PER_NS + m["quarter"] / VALU_QUARTER_PER_NS) / ISSUE_CUS * 1e-9
                out["rate_weighted"] = {"frac": round(t_issue / dur, 4), "mix": m,
                                        "ceilings_per_ns_per_cu": {"full": VALU_MEASURED_PER_NS, "half": VALU_HALF_PER_NS, "quarter": VALU_QUARTER_PER_NS},
                                        "source": "profiles/valu_mix.json (scripts/valu_mix.py)"}
        except Exception:
            pass
    out["other_units"] = {"salu_frac": round(sq.get("SQ_INSTS_SALU", 0) / dur / (SALU_PER_NS * ISSUE_CUS * 1e9), 4),
                          "lds_frac": round(sq.get("SQ_INSTS_LDS", 0) / dur / (LDS_PER_NS * ISSUE_CUS * 1e9), 4),
                          "busy_cu_clock_ghz": round(sq.get("SQ_BUSY_CU_CYCLES", 0) / ISSUE_CUS / dur / 1e9, 3) if sq.get("SQ_BUSY_CU_CYCLES") else None}
    out["hbm"] = hbm
    out["hbm_frac"] = hbm["frac"]              # north_star's figure, where the driver's parser keeps it
    out["hbm_achieved_gbs"] = hbm["achieved"]
    out.update(common)
    return out


def cpu_baseline(m, tape, model, S, T, ctx, frames=3, dim=3, one_core_parts=16):
    """Times oracle/mpr_oracle.c on the GPU's own workload.  ctx holds the GPU's frame of (tape, T) at S."""
    from oracle import orc
    from mpr_amd.multigpu import column_weights
    orc.lib()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    mat = m.colmajor(T, dim + 1)
    gimg = ctx.image.copy()
    gnrm = ctx.normals.copy() if dim == 3 else None
    # warm-up frames, one per candidate thread count (the oracle's OpenMP loop stops scaling well before 256 hardware threads:
    # one atomic pool index, one heightmap); the better one is timed.  The last one is also the checker.
    best = None
    same = True
    for th in sorted({avail, min(avail, 32)}, reverse=True):
        t1 = time.perf_counter()
        fr = orc.Frame(tape.data, dim, S, mat, threads=th, keep_pool=False)
        dt = time.perf_counter() - t1
        same = same and bool(np.array_equal(fr.filled[3], gimg) and (dim == 2 or np.array_equal(fr.normals, gnrm)))
        del fr
        if best is None or dt < best[1]:
            best = (th, dt)
    cores, warm_s = best
    per_cpu = []
    for _ in range(frames if warm_s < 30 else 1):
        t1 = time.perf_counter()
        orc.Frame(tape.data, dim, S, mat, threads=cores, keep_pool=False, skip_normals=False)
        per_cpu.append((time.perf_counter() - t1) * 1e3)
    cm, csd = stats(per_cpu)
    rec = {"value": round(S * S / (cm * 1e-3) / 1e6, 4), "unit": "Mpixel/s", "cores": cores, "kind": "port",
           "ms_per_frame": round(cm, 1), "ms_per_frame_std": round(csd, 1), "frames": len(per_cpu), "warmup": 1,
           "host_threads_available": avail, "frame_matches_gpu": same,
           "sample": "%s.frep render%dD at %d^%d — the bench configuration itself, whole frame — oracle/mpr_oracle.c, OpenMP over tile groups, %d threads"
                     % (model, dim, S, dim, cores)}
    # one core: the columns rank 0 gets when the frame's 64x64 columns are dealt 16 ways by measured work (the multi-GPU deal)
    w = column_weights(ctx.stages[3].tiles, S, dim)
    owner = m.partition_columns((S // 64) ** 2, one_core_parts, w)
    share = float(w[owner == 0].sum() / max(w.sum(), 1.0))
    t1 = time.perf_counter()
    orc.Frame(tape.data, dim, S, mat, threads=1, owner=owner, rank=0, keep_pool=False)
    one_ms = (time.perf_counter() - t1) * 1e3
    rec["one_core"] = {"cores": 1, "kind": "port", "ms_sample": round(one_ms, 1), "share_of_frame": round(share, 4),
                       "ms_per_frame_extrapolated": round(one_ms / max(share, 1e-9), 1),
                       "value": round(S * S / (one_ms / max(share, 1e-9) * 1e-3) / 1e6, 4), "unit": "Mpixel/s", "frames": 1, "warmup": 0,
                       "sample": "%s.frep render%dD at %d^%d, the %d of %d top-level columns that a %d-way longest-processing-time deal gives "
                                 "rank 0 (%.1f %% of the frame's smallest tiles), one thread, one frame"
                                 % (model, dim, S, dim, int((owner == 0).sum()), owner.size, one_core_parts, 100 * share)}
    return rec


ALL_CONFIGS = [  # (model, dim, size, cpu legs?)   BASELINE.json configs first, then the reference's table sizes
    ("prospero", 2, 1024, True), ("involute_gear_2d", 2, 4096, True), ("bear", 3, 1024, True), ("architecture", 3, 2048, True),
    ("prospero", 2, 256, True), ("prospero", 2, 512, True), ("prospero", 2, 2048, False), ("prospero", 2, 4096, False),
    ("involute_gear_2d", 2, 1024, False), ("hello_world", 2, 1024, False),
    ("bear", 3, 256, True), ("bear", 3, 512, True), ("bear", 3, 2048, False),
    ("architecture", 3, 512, False), ("architecture", 3, 1024, False), ("involute_gear_3d", 3, 1024, False),
]


def run_all(args):
    """One JSON record per model x size with the fields of BASELINE.md 4 (benchmark/render_2d_table.cpp:50-62 prints
    `size mean stdev`; so does this, on stderr)."""
    import torch
    import mpr_amd as m
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X")
    m.build()
    records = []
    for model, dim, S, with_cpu in ALL_CONFIGS:
        tape = m.Tape(m.model(model))
        T = view3() if dim == 3 else np.eye(3, dtype=np.float32)
        render = (lambda c: c.render3D(tape, T)) if dim == 3 else (lambda c: c.render2D(tape, T))
        cctx = m.Context(S, flags=m.CTX_COUNTERS | m.CTX_SERIAL_STAGES)
        render(cctx)
        work = cctx.counters()
        cctx.close()
        # per-kernel times from a context that records HIP events (ten frames); the frames themselves from one that does not (the
        # events cost a small frame a tenth of its time): get_stats' 20 + 100 (benchmark/stats.cpp:19-47), fewer when a frame is slow
        tctx = m.Context(S, flags=m.CTX_TIMING)
        kernel_ms = {}
        for k in range(13):
            render(tctx)
            if k >= 3:
                for name, ms in tctx.timings():
                    kernel_ms[name] = kernel_ms.get(name, 0.0) + ms / 10
        tctx.close()
        ctx = m.Context(S)
        render(ctx)
        kname = ctx.float_kernel()
        t1 = time.perf_counter()
        render(ctx)
        one = time.perf_counter() - t1
        steps = 100 if one < 0.05 else 20
        per = []
        for _ in range(20 if one < 0.05 else 3):
            render(ctx)
        for _ in range(steps):
            t1 = time.perf_counter()
            render(ctx)
            per.append((time.perf_counter() - t1) * 1e3)
        mean, std = stats(per)
        # (a frame that takes many times the others must be seen, and explained: the slowest timed frame, where it
        # was, and what the context did since it was made — scripts/outlier_probe.py runs 2000 frames per configuration)
        fstats = (ctypes.c_int64 * 4)()
        m.lib().mpr_debug_frame_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        m.lib().mpr_debug_frame_stats(ctx._h, fstats)
        b_frame, terms = frame_b_alg(work, S, dim)
        rec = {"model": model, "dim": dim, "size": S, "gpus": 1, "ms_mean": round(mean, 4), "ms_std": round(std, 4),
               "ms_median": round(float(np.median(per)), 4), "ms_max": round(float(np.max(per)), 4), "slowest_timed_frame": int(np.argmax(per)),
               "pool_clauses": int(fstats[0]), "pool_growths": int(fstats[1]), "frames_restarted": int(fstats[2]), "skip0_vetoes": int(fstats[3]),
               "mpixel_per_s": round(S * S / (mean * 1e-3) / 1e6, 2), "B_alg_bytes": int(b_frame), "B_alg_terms": terms,
               "roofline_hbm": round(b_frame / (mean * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
               "roofline_lds": None, "roofline_valu": None,
               "roofline_note": "roofline_hbm = B_alg / frame time / 8 TB/s (SURVEY.md 8(d)); VALU / LDS fractions need the SQ counter passes, "
                                "collected for the bench configuration only (bench line: roofline)",
               "float_kernel": kname, "kernel_ms": {k: round(v, 4) for k, v in kernel_ms.items()},
               "voxel_tiles": int(work["voxel_tiles"]), "lane_clauses": int(work["lane_clauses"]),
               "cpu_ms_allcores": None, "cpu_cores": os.cpu_count() or 1, "cpu_ms_1core": None}
        if with_cpu and not args.no_cpu:
            cb = cpu_baseline(m, tape, model, S, T, ctx, frames=3, dim=dim)
            rec["cpu_ms_allcores"] = cb["ms_per_frame"]
            rec["cpu_ms_1core"] = cb["one_core"]["ms_per_frame_extrapolated"]
            rec["cpu_1core_sample"] = cb["one_core"]["sample"]
            rec["cpu_frame_matches_gpu"] = cb["frame_matches_gpu"]
        ctx.close()
        sys.stderr.write("%s %dD: %d %.4f %.4f\n" % (model, dim, S, mean, std))
        print(json.dumps(rec), flush=True)
        records.append(rec)
    if args.out:
        with open(args.out, "w") as f:
            for r in records:
                f.write(json.dumps(r) + "\n")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)     # benchmark/stats.hpp:14: 100 timed
    ap.add_argument("--warmup", type=int, default=20)     #                         20 warm-up
    ap.add_argument("--model", default="bear")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-also", action="store_true", help="skip the prospero 2-D side measurement")
    ap.add_argument("--all", action="store_true", help="one JSON record per model x size (BASELINE.md 4) instead of the bench line")
    ap.add_argument("--out", default=None, help="--all: also write the records to this file")
    args = ap.parse_args()
    if args.all:
        return run_all(args)

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
    cctx = m.Context(S, device=local_rank, flags=m.CTX_COUNTERS | m.CTX_SERIAL_STAGES)
    cctx.render3D(tape, T)
    work = cctx.counters()
    cctx.close()

    # HIP events around the dominant kernel only inside the timed loop (two per frame); the other kernels' times come from ten
    # frames of a context that times every launch, after the timed region
    ctx = m.Context(S, device=local_rank, flags=m.CTX_TIMING_FLOAT)
    kernel_ms = {}
    frames_timed = [0]
    timing_gather = [False]
    gather_events = []

    if world > 1:
        from mpr_amd.multigpu import TileParallelRenderer

        def make_buffer(n):
            t = torch.empty(n, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            return t, t.data_ptr()

        # the collective is ordered after the context's stream (render + pack) and the unpack after
        # the collective, all on the device: no host synchronisation inside a frame
        ctx_stream = torch.cuda.ExternalStream(ctx.stream)

        gather_events = []          # (start, end) around the collective on the context's stream, one pair per timed frame

        def all_gather(out, inp):
            with torch.cuda.stream(ctx_stream):
                if timing_gather[0]:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(ctx_stream)
                    dist.all_gather_into_tensor(out, inp)
                    e1.record(ctx_stream)
                    gather_events.append((e0, e1))
                else:
                    dist.all_gather_into_tensor(out, inp)

        tpr = TileParallelRenderer(ctx, m, rank, world, make_buffer, all_gather, dim=3)
        tpr.plan(tape, T, feedback=os.environ.get("MPR_BENCH_FEEDBACK") == "1")     # default: the stage-0 proxy deal, no frame in advance
        # this rank's share of the dominant kernel's algorithmic work: an instrumented frame of its columns
        pctx = m.Context(S, device=local_rank, flags=m.CTX_COUNTERS | m.CTX_SERIAL_STAGES)
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
    if world > 1:           # (always: one frame and a single-GPU frame; VERDICT r4 next-8 — one driver run tells the whole story)
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
    timing_gather[0] = True
    t0 = time.perf_counter()
    per = []
    for _ in range(args.steps):
        s = time.perf_counter()
        timed_frame()
        per.append((time.perf_counter() - s) * 1e3)
    sync()
    barrier()
    total_s = time.perf_counter() - t0
    timing_gather[0] = False
    gather_ms = [e0.elapsed_time(e1) for e0, e1 in gather_events]
    if dist is not None:
        tt = torch.tensor([total_s], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total_s = float(tt.item())
    ms_per_step = total_s * 1e3 / args.steps
    mean_ms, std_ms = stats(per)

    # ---- roofline of the dominant kernel (eval_voxels_f) ----
    nframes = max(frames_timed[0], 1)
    vox_ms = kernel_ms.get("eval_voxels_f", 0.0) / nframes
    kname = ctx.float_kernel()
    avg = {}
    actx = m.Context(S, device=local_rank, flags=m.CTX_TIMING)
    for k in range(13):
        if world > 1:
            actx.render3D_part(tape, T, tpr.owner, rank)
        else:
            actx.render3D(tape, T)
        if k >= 3:
            for name, ms in actx.timings():
                avg[name] = avg.get(name, 0.0) + ms / 10
    actx.close()
    avg["eval_voxels_f (timed loop)"] = vox_ms
    walked = None
    if world == 1:
        os.environ["MPR_DEBUG_WALKED"] = "1"
        wctx = m.Context(S, device=local_rank)
        del os.environ["MPR_DEBUG_WALKED"]
        for _ in range(4):
            wctx.render3D(tape, T)
        walked = {"walked": wctx.tiles_walked(), "listed": wctx.frame_tiles()[2]}
        wctx.close()
    roofline = roofline_of(work, S, vox_ms, kname, avg, ms_per_step, world, args.model, walked)

    out = None
    if rank == 0:
        out = {
            "metric": "Mpixel/s, render3D %s %d^3 (ms/frame in ms_per_step)" % (args.model, S),
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
                       "column_deal": ("previous frame's smallest tiles per column" if os.environ.get("MPR_BENCH_FEEDBACK") == "1"
                                       else "first tile stage's ambiguous tiles per column (no frame in advance)") if world > 1 else None,
                       "voxel_tiles": int(work["voxel_tiles"]),
                       "voxel_tiles_walked": None if not walked or walked["walked"] < 0 else int(walked["walked"])},
            "roofline": roofline,
        }
        if world > 1:
            gm, gs = stats(gather_ms) if gather_ms else (None, None)
            out["verified_against_single_gpu"] = verified
            out["collective"] = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(),
                                 "shared_one_gpu": share,      # (MPR_BENCH_SHARE_GPU=1: a check of the code path, not a measurement)
                                 "all_gather_ms_mean_rank0": None if gm is None else round(gm, 4), "all_gather_ms_std_rank0": None if gs is None else round(gs, 4),
                                 "bytes_per_rank": int(tpr.per_rank * 4), "note": "HIP events around all_gather_into_tensor on the context's stream, "
                                 "rank 0, the timed frames; the wait for the slowest rank's columns is inside"}
            out["kernel_ms_rank0"] = {k: round(v, 4) for k, v in avg.items()}

    # ---- the same frame the reference's way throughout (every tile stage evaluated from the 64^3 tiles down, every tape pushed:
    #      the state a reader of tiles / tapes asks for), and the FIRST frame of a tape the context has not seen ----
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
                              "note": "MPR_LAST_STAGE_PUSH=1: every frame leaves the reference's tiles and tapes behind (what a frame costs "
                                      "when they are read back); images identical to the timed frames"}
        fresh, made = [], []
        ftree = m.model(args.model)
        for _ in range(min(args.steps, 12)):                                            # new tapes: nothing the context has learned applies
            t1 = time.perf_counter()
            fresh.append(m.Tape(ftree))
            made.append((time.perf_counter() - t1) * 1e3)
        first = []
        for ft in fresh:
            sync()
            t1 = time.perf_counter()
            ctx.render3D(ft, T)
            first.append((time.perf_counter() - t1) * 1e3)
        f1, f1s = stats(first[2:] if len(first) > 4 else first)
        out["first_frames"] = {"ms_per_frame_mean": round(f1, 4), "ms_per_frame_std": round(f1s, 4), "frames": len(first),
                               "note": "render3D of a tape object the context sees for the first time (includes the tape's upload)",
                               "tape_construction_ms": round(sorted(made)[len(made) // 2], 3),
                               "tape_construction_note": "host time of mpr::Tape(tree), median: the clauses and the tape's walks as gfx950 code (not in ms_per_frame_mean)"}

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
        # the other GPU configurations of BASELINE.json, so that every one of them has a driver-timed number
        for model, dim, size in (("involute_gear_2d", 2, 4096), ("architecture", 3, 2048)):
            otape = m.Tape(m.model(model))
            octx = m.Context(size, device=local_rank)
            fn = (lambda: octx.render3D(otape, T)) if dim == 3 else (lambda: octx.render2D(otape))
            _, oper = time_frames(fn, min(args.warmup, 10), min(args.steps, 50), lambda: None, sync)
            om, osd = stats(oper)
            okern = octx.float_kernel()
            octx.close()
            out["also"].append({"workload": "%s.frep render%dD at %d^%d" % (model, dim, size, dim), "ms_per_frame_mean": round(om, 4),
                                "ms_per_frame_std": round(osd, 4), "value": round(size * size / (om * 1e-3) / 1e6, 2), "unit": "Mpixel/s",
                                "float_kernel": okern, "vs_baseline": None})
        # what a reader of tiles / tapes pays on top of an ordinary frame (mpr_read_tiles after render3D: the tile stages again,
        # the reference's way; heights and normals are left as they are)
        rd = []
        for _ in range(min(args.steps, 12)):
            ctx.render3D(tape, T)
            sync()
            t1 = time.perf_counter()
            n_tiles = ctx.stages[3].tile_array_size
            rd.append((time.perf_counter() - t1) * 1e3)
        rm, rs = stats(rd[2:] if len(rd) > 4 else rd)
        out["reader"] = {"ms_read_tiles_after_a_frame_mean": round(rm, 4), "ms_std": round(rs, 4), "tiles": int(n_tiles),
                         "frame_plus_read_over_full_frame": round((mean_ms + rm) / out["full_frames"]["ms_per_frame_mean"], 3),
                         "note": "stages[3].tile_array_size right after an ordinary frame: the context runs the frame's tile stages again the "
                                 "reference's way (every stage from the 64^3 tiles down, every tape pushed); no float pass, no normals pass"}

    # ---- CPU baseline: the oracle (a port of the algorithm, NOT libfive's renderer) at the bench configuration itself: all host
    #      cores on the whole frame (1 warm-up + 3 timed, the 750 ms rule of benchmark/render_3d_table.cpp:71 in spirit), one core on
    #      the columns a 16-way deal of the same frame gives rank 0.  The oracle's frame must equal the GPU's. ----
    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(m, tape, args.model, S, T, ctx, frames=3)
        if not out["cpu_baseline"]["frame_matches_gpu"]:
            raise SystemExit("bench.py: the oracle's frame differs from the GPU's: " + json.dumps(out["cpu_baseline"]))
    ctx.close()
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
