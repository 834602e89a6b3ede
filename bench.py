#!/usr/bin/env python3
"""Benchmark of the floe-interaction hot path (BASELINE.json: floe-timesteps/s and contact-pairs/s).

A "step" is one call of the accelerated part of floe_interactions_all (szStep) on the device-resident
floe field; every 10th step refreshes the ocean/wind forcing, as the reference driver does
(Subzero.m:153-166).  Workload at N = 1: BASELINE.json configs[3] geometry on one GPU -- 1e4 Voronoi floes,
concentration 1.0, periodic box, dt = 10 s (the configuration the metric's target is quoted on).

One JSON line on stdout (rank 0).  `value` is floe-timesteps/s with all inputs resident in HBM.
  roofline     : the pair-interaction kernel (k_narrow_team: a team of 16 or 32 lanes per candidate pair) priced by the
                 ALGORITHMIC bytes of SURVEY.md 8(d) (16 B per ring vertex of both rings + 2 x 56 B of body scalars +
                 56 B per contact row) summed over THE PAIRS THAT LAUNCH PROCESSES -- the hot, cold and wide lists the
                 classification leaves it (szStepStats.team_pairs / team_vertices; every contact row of a steady step
                 comes from one of them) -- over the HIP-event duration of that launch alone, against the 8 TB/s HBM3E
                 peak.  The kernel is bound by the vector issue rate of irregular int64/FP64 geometry, not by HBM: the
                 fraction is reported as computed ("limiter" says so).  `traffic` is the PMC figure of THIS round's
                 profile of the same command (profiles/r04_pmc_narrow.json, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
                 separate passes); null when that file is absent -- a bench run cannot collect PMC counters by itself.
  pair_stage   : the same figure for the whole pair stage -- ALL candidate pairs of the step (the quarter whose boxes
                 are apart is answered by the classification inside kf_pairs) over kf_pairs + k_narrow_team.
  cpu_baseline : the CPU oracle (C++ restatement of the MATLAB path around the reference's own Clipper)
                 on the host cores, a bounded sample of the same workload.  TEST INFRASTRUCTURE timed as a
                 baseline only; the product path never calls it.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def build_workload(n_floes, seed=4, Lx=1e5, Ly=1e5):
    from subzero_amd import synth
    S = synth.voronoi_field(n_floes, 1.0, Lx=Lx, Ly=Ly, seed=seed, vel_seed=2, vel_sigma=0.1)
    P = synth.StepParams(dt=10.0, Lx=Lx, Ly=Ly, periodic=True)
    if Lx > 2e5 or Ly > 2e5:      # the ocean / wind grid covers the box with the reference's margin (initialize_ocean.m: 4 x the box)
        P.Xo = np.arange(-Lx - 3e5, Lx + 3e5 + 1, 1e4); P.Yo = np.arange(-Ly - 3e5, Ly + 3e5 + 1, 1e4)
        shape = (len(P.Yo), len(P.Xo))
        P.Uocn = np.zeros(shape); P.Vocn = np.zeros(shape); P.Uwind = np.zeros(shape); P.Vwind = np.zeros(shape)
    P.Modulus = synth.reference_modulus(S)
    return S, P


def usable_cores():
    """Cores this process may really use: the cgroup CPU quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(S, P, budget_s=12.0):
    """Oracle steps on the host cores from the same (already colliding) state; bounded sample.
    Parallel structure of the reference: narrow phase and calc_trajectory are parfor loops (OpenMP here), the
    broad phase is its serial O(N^2) loop (floe_interactions_all.m:76-120)."""
    import oracle
    threads = usable_cores()
    A = S.copy()
    t0 = time.time()
    steps = 0
    it = 1
    while True:
        P.doInt = int(it % 10 == 0)
        oracle.step(A, P, threads=threads)
        steps += 1; it += 1
        el = time.time() - t0
        if el > budget_s or steps >= 200:
            break
    P.doInt = 0
    t1 = time.time()
    oracle.step(A, P, threads=1)          # also at one thread (one step, no forcing refresh)
    el1 = time.time() - t1
    return {"value": S.n * steps / el, "unit": "floe-timesteps/s", "cores": threads,
            "kind": "port", "clip": oracle.lib().clip_kind, "value_1_thread": S.n / el1,
            "sample": "%d oracle timesteps of the same %d-floe field (state after the GPU run), %d threads, %.1f s; "
                      "then 1 timestep on 1 thread, %.1f s" % (steps, S.n, threads, el, el1)}


def clip_kernel_line(ctx, with_cpu, n_pairs=20000, cpu_pairs=3000):
    """Kernel-level clips/s of the B1 surface (szClipBatch's kernel, inputs resident in HBM) on pairs of the
    reference's own floe outlines (test/test_conservation/FloeShapes.mat, decoded in tests/golden/): the workload of
    BASELINE.md's Clipper probe (real outlines paired and partially overlapped, intersection), next to the reference
    Clipper build on one host core over a bounded sample of the same pairs."""
    import clipgen
    ps = clipgen.shape_pairs(n_pairs, seed=2)
    ax, ay, aoff, bx, by, boff = ps.arrays()
    d = [ctx.to_device(v) for v in (ax, ay, aoff, bx, by, boff)]
    ms = []
    for it in range(4):
        st = ctx.clip_batch_device(len(ps), *d, len(ax), len(bx), 1)
        if it:
            ms.append(st.kernel_ms)
    for v in d:
        ctx.free(v)
    k_ms = sum(ms) / len(ms)
    out = {"workload": "%d pairs of FloeShapes.mat outlines, partially overlapped, intersection (polyclip 'int')" % n_pairs,
           "pairs": int(st.pairs), "mean_input_vertices_per_pair": st.input_vertices / max(st.pairs, 1),
           "kernel_ms": k_ms, "clips_per_s": st.pairs / (k_ms * 1e-3), "rings_out": int(st.rings_out),
           "redone_at_a_larger_workspace": int(st.retried), "failed": int(st.failed),
           # 16 B per input vertex + 16 B per output vertex
           "algorithmic_GBps": 16.0 * (st.input_vertices + st.vertices_out) / (k_ms * 1e-3) / 1e9}
    if with_cpu and clipgen.ref_lib() is not None:
        sub = clipgen.shape_pairs(cpu_pairs, seed=2)
        t0 = time.time(); clipgen.run_ref(sub, 1); el = time.time() - t0
        out["cpu_reference"] = {"clips_per_s": cpu_pairs / el, "cores": 1, "kind": "reference",
                                "sample": "the reference Clipper build (oracle/_ref) on the first %d of the same pairs, %.2f s" % (cpu_pairs, el)}
    return out


def post_step_line(sim, S, P, reps=3):
    """Call-level milliseconds (wall clock around the C-ABI call: kernels, transfers, host work inside the library; the
    Python wrapper's own time beside it) of the post-step entries of SURVEY.md 8(f) on the resident field, next to the
    cadence the reference driver calls them at."""
    from subzero_amd import capi, corners
    rng = np.random.default_rng(7)
    keep = corners.keep_mask(rng.uniform(size=S.n), S.OverlapArea, S.area, 0)
    ids = (np.nonzero(~keep)[0] + 1).astype(np.int32)                     # Subzero.m:341-348: ~70 % of the floes

    def timed(fn):
        """(ms of the C-ABI call itself, ms of the Python wrapper around it -- array allocation, result views --, result)"""
        fn()
        c = 0.0
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn(); c += sim.last_call_ms
        return (c / reps, (time.perf_counter() - t0) / reps * 1e3), out
    ms_c, (off, cnt, ang, da, n0, k_c) = timed(lambda: sim.corner_candidates(ids, 0, P.c2_boundary))
    grind, frac = corners.grind_flags(off, cnt, ang, da, rng.uniform(size=corners.draws_needed(cnt)))
    sel = np.nonzero(frac)[0]
    foff = np.zeros(len(sel) + 1, np.int32); fg = []
    for k, q in enumerate(sel):
        n = int(S.vert_off[ids[q]] - S.vert_off[ids[q] - 1])
        g = np.zeros(n, np.uint8); g[:cnt[q]] = grind[off[q]:off[q] + cnt[q]]; fg.append(g); foff[k + 1] = foff[k] + n
    fg = np.concatenate(fg) if fg else np.zeros(0, np.uint8)
    ms_k, (npieces, poff, pieces, rings, k_k) = timed(lambda: sim.corner_cuts(ids[sel], foff, fg))
    ms_e, (eul, k_e) = timed(lambda: sim.eulerian_data(40, 40, 0, P.c2_boundary, True))
    fids = ids[:max(1, len(ids) // 20)]
    ms_d, _ = timed(lambda: sim.fracture_deform(fids))
    return {"corner_candidates": {"call_ms": ms_c[0], "python_wrapper_ms": ms_c[1], "kernel_ms": k_c, "floes": int(len(ids)), "cadence": "every 10th step, ~70 % of the floes (Subzero.m:339-352)"},
            "corner_cuts": {"call_ms": ms_k[0], "python_wrapper_ms": ms_k[1], "kernel_ms": k_k, "floes": int(len(sel)), "cut_on_device": int((npieces > 0).sum()), "pieces": int(len(pieces)),
                            "cadence": "with corner_candidates, the floes with more than one corner to break (corners.m:94-98)"},
            "eulerian_data": {"call_ms": ms_e[0], "python_wrapper_ms": ms_e[1], "kernel_ms": k_e, "grid": "40 x 40", "cadence": "every 150th step, or every step with AVERAGE (Subzero.m:304-314)"},
            "fracture_deform": {"call_ms": ms_d[0], "python_wrapper_ms": ms_d[1], "floes": int(len(fids)), "cadence": "every 75th step, the floes picked by fracture.m:51-55 (Subzero.m:220)"}}


def other_fields_line(ctx):
    """ms/step of the same step on other fields (not the headline workload): floes between walls, the reference's evolved
    outlines, two BASELINE.json configs and a larger field; a few tens of steps each."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenes
    from subzero_amd import capi, synth
    out = {}

    def run(name, S, P, warm, steps):
        sim = capi.FloeInteractions(ctx, S)
        it = 0
        P.queue_ahead = 1                  # (as the headline loop: every step is followed by another with the same parameters)
        for _ in range(warm):
            P.doInt = int(it % 10 == 0); P.next_doInt = int((it + 1) % 10 == 0); sim.step(P); it += 1
        t = time.perf_counter(); general = 0
        for _ in range(steps):
            P.doInt = int(it % 10 == 0); P.next_doInt = int((it + 1) % 10 == 0); st = sim.step(P); it += 1; general += st.retried
        el = time.perf_counter() - t
        out[name] = {"floes": int(S.n), "candidate_pairs": int(st.n_pairs), "ms_per_step": el / steps * 1e3,
                     "floe_timesteps_per_s": S.n * steps / el, "pairs_through_the_general_sweep_per_step": general / steps}
        del sim

    S, P0 = build_workload(10000)
    P = synth.StepParams(dt=10.0, Lx=1e5, Ly=1e5, periodic=False); P.Modulus = P0.Modulus
    run("pack10000 between walls", S, P, 12, 20)
    S, P = scenes.concave(2500, periodic=True); run("2500 evolved outlines (FloeShapes.mat), periodic", S, P, 3, 10)
    S, P = synth.config("uniaxial200"); run("uniaxial200 (BASELINE.json configs[2])", S, P, 5, 20)
    S, P = synth.config("convex1000"); run("convex1000 (BASELINE.json configs[1])", S, P, 12, 30)
    S, P = build_workload(40000); run("pack40000 periodic", S, P, 12, 10)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--floes", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-clip-line", action="store_true")
    ap.add_argument("--no-other-fields", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Called without a launcher: start the N ranks ourselves -- fresh child processes (one per GPU, RCCL between
        # them), before this process has touched the GPU -- and pass their one JSON line through.  Never an n_gpus = 1
        # line for --gpus N.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))

    from subzero_amd import capi
    if world > 1 or os.environ.get("SUBZERO_FORCE_SHARDED") == "1":   # the env hook rehearses the N>1 code on one GPU
        import torch  # noqa: F401  (torch must load its HIP runtime before libsubzero_hip.so does; tests/conftest.py)
        from subzero_amd import multigpu
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        return multigpu.bench_main(args, rank, local_rank, world)

    S, P = build_workload(args.floes)
    ctx = capi.Context(local_rank)
    sim = capi.FloeInteractions(ctx, S)
    it = 0
    # szStepParams.flags: every step of this loop is followed by another with the same parameters, and says so
    # (SZ_STEP_QUEUE_AHEAD, the next doInt announced): the first kernels of the next step wait on the device while the host
    # returns here and calls again (include/subzero_hip.h)
    P.queue_ahead = 1
    for _ in range(args.warmup):
        P.doInt = int(it % 10 == 0); P.next_doInt = int((it + 1) % 10 == 0); sim.step(P); it += 1
    ctx.check(ctx.L.szSynchronize(ctx.h), "sync")
    narrow_ms = pair_bytes = team_bytes = stage_ms = 0.0
    pairs = rows = team_pairs = 0
    dev_ms = 0.0
    t0 = time.perf_counter()
    for q in range(args.steps):
        P.doInt = int(it % 10 == 0); P.next_doInt = int((it + 1) % 10 == 0)
        st = sim.step(P); it += 1
        # (the stage times a step returns are those of the step before -- include/subzero_hip.h szStepStats: the first call's
        # belong to the last warm-up step and are left out, the last timed step's are fetched after the loop)
        if q > 0:
            narrow_ms += st.narrow_ms; dev_ms += st.step_ms; stage_ms += st.pairs_ms + st.narrow_ms
        pairs += st.n_pairs; rows += st.n_rows
        # B_pair = 16 (n_i+1 + n_j+1) + 2*56 + 56 K, summed over the candidate pairs
        pair_bytes += 16.0 * st.pair_vertices + 112.0 * st.n_pairs + 56.0 * st.pair_rows
        # ... and over the pairs k_narrow_team itself processes (its three lists)
        team_bytes += 16.0 * st.team_vertices + 112.0 * st.team_pairs + 56.0 * st.pair_rows
        team_pairs += st.team_pairs
    ctx.check(ctx.L.szSynchronize(ctx.h), "sync")
    el = time.perf_counter() - t0
    st = sim.step_times()
    narrow_ms += st.narrow_ms; dev_ms += st.step_ms; stage_ms += st.pairs_ms + st.narrow_ms
    P.queue_ahead = 0; P.next_doInt = 0
    queued_ahead = sim.diagnostics()["steps_ahead"]

    off, inter = sim.interactions()
    collisions = capi.collision_count(inter)
    if np.isinf(inter[:, 0]).any():
        raise SystemExit("bench: boundary rows in a periodic workload")
    # calc_collisionNum.m:6 summed over the timed steps: without boundary rows every collision is two rows
    collisions_timed = rows / 2.0
    sim.download(capi.SYNC_ALL)

    k = args.steps
    achieved = (team_bytes / k) / (narrow_ms / k * 1e-3) / 1e9 if narrow_ms > 0 else 0.0
    stage = (pair_bytes / k) / (stage_ms / k * 1e-3) / 1e9 if stage_ms > 0 else 0.0
    traffic = traffic_src = None
    pmc = os.path.join(ROOT, "profiles", "r04_pmc_narrow.json")
    if os.path.exists(pmc):
        try:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            traffic_src = "profiles/r04_pmc_narrow.json"
        except Exception:
            traffic = None
    out = {
        "metric": "floe-timesteps/sec", "value": S.n * k / el, "unit": "floe-timesteps/s",
        "n_gpus": 1, "steps": k, "warmup": args.warmup, "ms_per_step": el / k * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64+int64", "data": "synthetic",
        "config": {"workload": "pack%d: %d Voronoi floes, concentration 1.0, periodic +-1e5 m box, dt=10 s, "
                               "collisions + ocean drag (forcing refreshed every 10th step)" % (S.n, S.n),
                   "floes": int(S.n), "floes_per_gpu": int(S.n), "parallelism": "1 GPU"},
        "timesteps_per_s": k / el,
        "contact_pairs_per_s": collisions_timed / el,
        "candidate_pairs_per_s": pairs / el,
        "contact_rows_last_step": int(len(inter)), "collisions_last_step": collisions,
        "device_ms_per_step": dev_ms / k,
        "steps_with_front_kernels_queued_ahead": int(queued_ahead),
        "roofline": {"bound": "hbm", "kernel": "k_narrow_team (pair interaction: clip + contact force of the pairs on its hot / cold / wide lists, "
                                               "i.e. every candidate pair whose boxes meet)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "limiter": "FP64 / int64 vector issue of irregular geometry (16 lanes per pair, ~5.4e3 vector instructions per wave of four pairs), not HBM",
                     "kernel_ms": narrow_ms / k, "algorithmic_bytes_per_launch": team_bytes / k,
                     "pairs_per_launch": team_pairs / k, "candidate_pairs_per_step": pairs / k},
        "pair_stage": {"kernels": "kf_pairs (pair records + classification: answers the pairs whose boxes are apart) + k_narrow_team",
                       "achieved": stage, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": stage / HBM_PEAK_GBS,
                       "kernel_ms": stage_ms / k, "algorithmic_bytes_per_step": pair_bytes / k, "pairs_per_step": pairs / k},
    }
    if not args.no_clip_line:
        out["post_step_calls"] = post_step_line(sim, S, P)
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(S, P)
    if not args.no_clip_line:
        out["clip_kernel"] = clip_kernel_line(ctx, not args.no_cpu_baseline)
    if not args.no_other_fields and not args.no_clip_line:
        del sim
        out["other_fields"] = other_fields_line(ctx)
    ctx.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
