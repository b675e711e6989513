#!/usr/bin/env python3
"""bench.py -- Schur nnz reduced / second on BASELINE config 3 (1M x 1M CSR, 20 nnz/row, p = 65521).

One "step" = one pass of the hot path over the synthetic matrix: the WHOLE Schur step of the round, as the reference's
spasm_schur does it (src/SpaSM.jl:761-762, the per-row solve :694-713 is inside): W = -(I + U_PP)^-1 U_PN rebuilt from the
round's pivot rows U level by level (every launch of every level, as a round makes them: nothing is skipped because an earlier
step found a list empty, no statistics are left out), the plan of every non-pivot row, the streaming scatter -- with the matrix,
U and all work buffers already resident in HBM.  Carried from step to step, and said so: U itself and the levels of its pivot
graph (the output of the pivot search and its bookkeeping, spasm_pivots_extract_structural: not part of the metric; `levels_ms`
reports the levels on their own), the sizes of the work buffers, and the knowledge that no row of THIS round needs the
multiplier-list fallbacks (four or five launches that would find their lists empty).  `echelonize_round0` is the same step timed
inside a real spasm_echelonize of the same matrix, where nothing is carried.
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (torch.distributed.run).
With N GPUs (one process per GPU) the non-pivot rows are dealt to the ranks (rank r: rows r, r + N, ...; BASELINE config 4,
strong scaling: the matrix is fixed); every rank elects the same pivots and holds the same U, rows never move.  For N > 1 every rank uploads only its row block and the round's pivot rows are exchanged once in
the setup (all-reduce(MIN) of the election keys + all-gather of the elected rows, RCCL over xGMI, timed on its own and
reported as `pivot_row_exchange_rank0`; SPASM_BENCH_EXCHANGE=0 replicates the matrix instead, explicitly); the timed
region has no data-path collective.  A failing exchange ends the run with a non-zero exit code.

Prints ONE JSON line on rank 0.  `roofline` describes the dominant kernel (the scatter launch of
the busiest hash-table class): algorithmic bytes per launch over its HIP-event duration on the
launch stream; `roofline.valu` holds the same launch against the ceiling it actually runs into (wave64 VALU issue: DESIGN.md section 8).
`cpu_baseline` times the CPU oracle (oracle/liboracle.so, OpenMP) on a bounded
sample of the same round, on rank 0 at N = 1 only.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# scatter launch of each hash-table class (spasm.jl_amd/csrc/engine.hip, kClasses): k_scatter<LOGT, ...>
SCATTER_KERNEL_PREFIX = ["k_scatter<8,", "k_scatter<9,", "k_scatter<10,", "k_scatter<11,", "k_scatter<12,", "k_scatter<13,",
                         "k_scatter<14,", None]
# streaming twin of class c (stats classes 8 .. 14): k_wstream<LOGT = 8 + c, ...>
STREAM_KERNEL_PREFIX = [f"k_wstream<{8 + c}," for c in range(7)]
TRAFFIC_FILES = ["r04_final_traffic.json", "r03_final_traffic.json", "r02_final_traffic.json", "r01_final_traffic.json"]  # newest first: PMC passes of tools/final_profile.sh
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=int(os.environ.get("SPASM_BENCH_N", 1_000_000)), help="rows = columns of the synthetic matrix")
    ap.add_argument("--row-nnz", type=int, default=20)
    ap.add_argument("--prime", type=int, default=65521)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: become one.  A child process, started before anything here touches the GPU (never an exec
        # from a process that has initialised HIP); rank 0 of the children prints the JSON line on the inherited stdout.
        import socket
        import subprocess

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # SPASM_BENCH_REHEARSE=1: every rank shares cuda:0 and the collectives run over gloo -- the only way to exercise the
    # N > 1 code path on a one-GPU box (the numbers of such a run mean nothing; the JSON says "rehearsal")
    rehearse = os.environ.get("SPASM_BENCH_REHEARSE", "0") == "1"
    if rehearse:
        local_rank = 0
    cdev = "cpu" if rehearse else "cuda"  # where the few scalars of the timing protocol are reduced
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import spasm_jl_amd as S

    lib = S._abi.lib()
    assert lib.spasm_amd_set_device(local_rank) == 0, S._abi.last_error()

    n = m = args.n
    seed = 0x5A5A0003  # configs 3 and 4 share the matrix
    t_gen = time.time()
    A = S.synth_csr(1, n, m, row_nnz=args.row_nnz, prime=args.prime, seed=seed)
    t_gen = time.time() - t_gen
    # rank r of G reduces rows r, r+G, r+2G, ...: contiguous blocks would be unbalanced (all rows have the same length, so the
    # election's tie-break puts the pivots in the first rows of the matrix)
    lo, hi, stride = rank, n, world
    my_rows = len(range(lo, hi, stride))

    t_setup = time.time()
    exchange = None
    engine = None
    if world > 1 and os.environ.get("SPASM_BENCH_EXCHANGE", "1") != "0":
        # every rank uploads only its row block; the round's pivot rows are elected with an all-reduce(MIN) and
        # exchanged with an all-gather (RCCL over xGMI), after which U is identical on all ranks
        from spasm_jl_amd import sharded

        # (an exchange that fails is fatal: a run that silently replicated the matrix instead would not measure config 4)
        engine = sharded.GpuShardEngine(A, lo, hi, stride=stride)
        xt = []
        for rep in range(3):  # the exchange alone, timed: all-reduce(MIN) of the keys + all-gather of the rows + import (U, Uinv, W build)
            if rep:
                engine.close()
                engine = sharded.GpuShardEngine(A, lo, hi, stride=stride)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t_x = time.time()
            npiv_x, exchange = sharded.exchange_pivot_rows(engine)
            torch.cuda.synchronize()
            xt.append(time.time() - t_x)
        exchange["seconds_incl_U_build"] = [round(x, 4) for x in xt]
        plan = engine.plan
    else:
        plan = lib.spasm_amd_schur_plan_create_strided(A.data, lo, hi, stride)
        if not plan:
            raise SystemExit("plan_create failed: " + S._abi.last_error())
    t_setup = time.time() - t_setup

    stream = torch.cuda.Stream()
    sptr = C.c_void_p(stream.cuda_stream)

    def step():
        if lib.spasm_amd_schur_plan_run(plan, sptr) != 0:
            raise SystemExit("plan_run failed: " + S._abi.last_error())

    def barrier():
        if world > 1:
            dist.barrier()

    lib.spasm_amd_schur_plan_class_timing(plan, 0)  # the timed steps carry no per-class event records
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one more step, outside the timed region, with an event pair around every scatter class (roofline of the dominant kernel)
    lib.spasm_amd_schur_plan_class_timing(plan, 1)
    step()
    torch.cuda.synchronize()
    st = S._abi.RoundStats()
    if lib.spasm_amd_schur_plan_stats(plan, C.byref(st)) != 0:
        raise SystemExit("plan_stats failed: " + S._abi.last_error())
    d = st.as_dict()
    counters = torch.tensor([d["nnz_reduced"], d["applications"], d["nnz_out"], d["read_bytes"], my_rows], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    nnz_reduced, applications, nnz_out, read_bytes, rows_total = [int(v) for v in counters.tolist()]

    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    value = nnz_reduced * args.steps / elapsed

    if rank == 0:
        # dominant kernel: the scatter launch of the busiest class (HIP events on the launch stream, last step)
        cls = max(range(16), key=lambda c: d["ms_class"][c])
        k_bytes = 8 * d["ent_class"][cls] + 16 * d["seg_class"][cls]
        k_ms = d["ms_class"][cls]
        k_name = (f"k_stream class {cls - 8}" if cls >= 8 else f"k_scatter class {cls}") + f" ({d['rows_class'][cls]} rows)"
        fused = d.get("rows_fused", 0) > 0 and d.get("ms_fused", 0.0) >= k_ms
        if fused:
            # the fused kernel (csrc/fused.hpp): 8 B per streamed entry + 16 B per row segment (the row itself and every run of W)
            cls = -1
            k_bytes = 8 * d["ent_fused"] + 16 * d["seg_fused"]
            k_ms = d["ms_fused"]
            k_name = f"k_schur_fused ({d['rows_fused']} rows)"
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        # HBM traffic of that kernel from the PMC counters (collected in separate rocprofv3 passes and calibrated for
        # this access shape, profiles/r02_final_traffic.json); only quoted for the workload it was measured on
        prefix = None
        if fused:
            prefix = "k_schur_fused<"
        elif args.prime < 65536:
            prefix = SCATTER_KERNEL_PREFIX[cls] if cls < 8 else (STREAM_KERNEL_PREFIX[cls - 8] if cls < 15 else None)
        traffic = None
        for tf in TRAFFIC_FILES:
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", tf)))["kernels"]
                if world == 1 and n == 1_000_000 and args.row_nnz == 20 and prefix:
                    hit = [v for k, v in prof.items() if k.startswith(prefix) and "true" in k]  # SMALL = true: p < 2^16
                    if len(hit) == 1:
                        traffic = hit[0]["fetch_bytes"] + hit[0]["write_bytes"]
                        break
            except (OSError, KeyError, ValueError):
                pass
        # the same kernel against the OTHER ceiling it runs into: a CU retires one wave64 VALU instruction per clock (four 16-lane
        # SIMDs, four clocks each).  SQ_INSTS_VALU per launch from the committed PMC pass (profiles/*_instruction_mix.txt, a separate
        # run as the guide prescribes), over the live duration of this run; only for the workload it was measured on.
        valu = None
        if world == 1 and n == 1_000_000 and args.row_nnz == 20 and args.prime == 65521 and prefix and not fused and k_ms > 0:
            try:
                lines = open(os.path.join(ROOT, "profiles", "r04_final_instruction_mix.txt")).read().splitlines()
                at = [i for i, ln in enumerate(lines) if ln.startswith("== " + prefix.rstrip(","))]
                if len(at) == 1:
                    for ln in lines[at[0] + 1: at[0] + 30]:
                        if ln.startswith("=="):
                            break
                        f = ln.split()
                        if len(f) == 2 and f[0] == "SQ_INSTS_VALU":
                            insts = float(f[1])
                            valu = {"wave_instructions_per_launch": int(insts), "peak_per_s": 256 * 2.4e9,
                                    "frac_of_valu_peak": round(insts / (256 * 2.4e9 * k_ms * 1e-3), 4), "source": "r04_final_instruction_mix.txt",
                                    "note": "one wave64 VALU instruction per CU and clock at 2.4 GHz; quarter-rate instructions count as one"}
            except (OSError, ValueError):
                pass
        roofline = {
            "bound": "hbm",
            "kernel": k_name,
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic,
            "valu": valu,
            "bytes_per_launch": k_bytes,
            "ms_per_launch": round(k_ms, 4),
            "round_algorithmic_read_GBs": round(read_bytes / (ms_per_step * 1e-3) / 1e9 / max(world, 1), 1) if ms_per_step > 0 else None,
            "round_frac_of_peak": round(read_bytes / (ms_per_step * 1e-3) / 1e9 / max(world, 1) / HBM_PEAK_GBS, 4) if ms_per_step > 0 else None,
            # the phases of one step (HIP events of the profiled extra step): W level by level, the plan of the rows, the scatter
            "round_ms": {"w_build": round(d["ms_wbuild"], 4), "plan": round(d["ms_solve"], 4), "scatter": round(d["ms_scatter"], 4)},
            "w": {"levels": d["w_levels"], "entries": d["w_entries"], "long_rows": d["w_long_rows"]},
            # once per round, outside the step: the levels of the pivot graph (part of the pivot bookkeeping) + sizing + first build
            # (wall times with their host synchronisations; the allocation of the build's buffers -- once per plan -- on its own)
            "levels_and_first_w_build_ms": round(d["ms_w"] - d.get("ms_w_sizing", 0.0), 4),
            "w_buffer_sizing_and_allocation_ms": round(d.get("ms_w_sizing", 0.0), 4),
            "per_class_ms": {"hash": [round(x, 4) for x in d["ms_class"][:8]], "stream": [round(x, 4) for x in d["ms_class"][8:15]]},
            "per_class_rows": {"hash": d["rows_class"][:8], "stream": d["rows_class"][8:15]},
            # streamed entries and (rows + chunk records) of the streaming classes: entries / (64 * chunks) = share of busy lanes
            "per_class_entries": {"stream": d["ent_class"][8:15]},
            "per_class_chunks": {"stream": [int(sg) - int(rw) for sg, rw in zip(d["seg_class"][8:15], d["rows_class"][8:15])]},
            "stream_fix": d["stream_fix"], "stream_redo": d["stream_redo"], "stream_fix_ms": round(d["ms_class"][15], 4),
            "fused": {"ms": round(d.get("ms_fused", 0.0), 4), "fix_ms": round(d.get("ms_fused_fix", 0.0), 4), "rows": d.get("rows_fused", 0),
                      "entries": d.get("ent_fused", 0), "segments": d.get("seg_fused", 0), "rows_left_to_general_path": d.get("rows_rejected", 0),
                      "s_entries_used": d.get("s_entries_used", 0)},
            "levels_ms": round(d.get("ms_levels", 0.0), 4),
        }
        # beside the algorithmic bytes, (i) the measured HBM bytes of a whole round and (ii) the compulsory floor
        # 8 (nnz(A) + nnz(U)) + 8 (n + r): every entry of A and of the pivot rows read once, one pointer pair per row
        try:
            Uc = lib.spasm_amd_schur_plan_fetch_U(plan, None, None)
            if Uc:
                nnz_u = int(lib.spasm_nnz(Uc))
                lib.spasm_csr_free(Uc)
                roofline["compulsory_floor_bytes"] = 8 * (int(d["nnz_in"]) + nnz_u) + 8 * (my_rows + int(d["npiv"]))
        except Exception:
            pass
        if traffic is not None:
            # the kernels every step launches once (streaming classes, plan, fix-ups, binning); the hash-table and combine kernels
            # only see the few rows handed back, and their per-dispatch means in the profile include the building of W
            step_kernels = ("k_wlevel", "k_wstream<", "k_wplan<", "k_stream_fix", "k_bin")
            try:
                fetch = sum(v.get("fetch_bytes_per_step") or v["fetch_bytes"] for k, v in prof.items() if k.startswith(step_kernels))
                write = sum(v.get("write_bytes_per_step") or v["write_bytes"] for k, v in prof.items() if k.startswith(step_kernels))
                roofline["round_traffic"] = {"fetch_bytes": fetch, "write_bytes": write, "source": tf,
                                             "kernels": "k_wlevel* (W build) + k_wplan + k_bin + k_wstream x7 + k_stream_fix (PMC, summed over the launches of one step)",
                                             "fetch_over_algorithmic_read": round(fetch / d["read_bytes"], 3) if d["read_bytes"] else None}
            except Exception:
                pass
        out = {
            "metric": "Schur nnz reduced/sec (GF(p) echelonize), 1Mx1M CSR",
            "value": value,
            "unit": "nnz/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "int32 (GF(p) balanced residues)",
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config 3: {n}x{m} CSR, {args.row_nnz} nnz/row, p={args.prime}, seed 0x5A5A0003, Schur round 0",
                "rows_per_gpu": (n + world - 1) // world,
                "npiv": d["npiv"],
                "nnz_reduced_per_step": nnz_reduced,
                "applications_per_step": applications,
                "nnz_out": nnz_out,
                "parallelism": (f"rows r, r+{world}, ... on rank r (x{world}), pivot rows exchanged by all-reduce(MIN)+all-gather" if world > 1 else "single GPU")
                + (" -- REHEARSAL: all ranks on one GPU over gloo, timings meaningless" if rehearse else ""),
            },
            "roofline": roofline,
            "setup_s": {"generate": round(t_gen, 2), "upload_elect_buildU": round(t_setup, 2)},
        }
        if world == 1:
            # the same Schur step inside a real spasm_echelonize (src/SpaSM.jl:863): one sparse round, leftmost pivots as in the plan,
            # no finish (enable_dense = enable_GPLU = 0: "Cannot finish", U = the round's pivot rows).  Device time of the W build, of
            # the plan of the rows (with the host synchronisation that reads the total of the bounds) and of the scatter, from the
            # events the round records; levels_ms = the levels of the pivot graph, density_probe_ms = the sample of 2048 rows that
            # sizes the pools (spasm_schur_estimate_density's place in the round)
            try:
                lib.spasm_amd_schur_plan_free(plan)
                plan = None
                S.echelonize(A, max_round=1, enable_dense=False, enable_GPLU=False, enable_greedy_pivot_search=False)
                r0 = S.last_rounds()[0]
                e0 = r0["ms_wbuild"] + r0["ms_solve"] + r0["ms_scatter"]
                out["echelonize_round0"] = {"ms": round(e0, 4), "w_build": round(r0["ms_wbuild"], 4), "plan": round(r0["ms_solve"], 4),
                                            "scatter": round(r0["ms_scatter"], 4), "levels_ms": round(r0["ms_levels"], 4),
                                            "w_levels_sizing_first_build_wall_ms": round(r0["ms_w"], 4), "w_buffer_sizing_and_allocation_ms": round(r0.get("ms_w_sizing", 0.0), 4), "pivots_and_probe_ms": round(r0["ms_pivots"], 4),
                                            "nnz_out": r0["nnz_out"], "over_bench_step": round(e0 / ms_per_step, 3) if ms_per_step > 0 else None}
            except Exception as exc:  # noqa: BLE001 - reported, not required
                out["echelonize_round0"] = {"error": repr(exc)}
        if exchange is not None:
            out["pivot_row_exchange_rank0"] = exchange
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if engine is not None:
        engine.close()
    elif plan is not None:
        lib.spasm_amd_schur_plan_free(plan)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(A, target_seconds):
    """The oracle (CPU restatement of libspasm's reach + scatter Schur round, OpenMP) on a bounded sample: the same matrix
    and the same pivots, but only the non-pivot rows of a leading row block are reduced.  Timed with all host threads
    (the reported value) and with one thread; a real libspasm named by $SPASM_LIB is timed beside it when it loads."""
    import oracle_ffi as O

    n = A.n
    probe = min(n, 20000)
    _, info = O.schur_round(A, row_lo=0, row_hi=probe)
    rate = info["nnz_reduced"] / max(info["sec_schur"], 1e-9)
    rows = int(min(n, max(probe, probe * 0.6 * target_seconds / max(info["sec_schur"], 1e-9))))
    if rows > probe:
        _, info = O.schur_round(A, row_lo=0, row_hi=rows)
        rate = info["nnz_reduced"] / max(info["sec_schur"], 1e-9)
    else:
        rows = probe
    out = {
        "value": rate,
        "unit": "nnz/s",
        "cores": info["threads"],
        "kind": "port",
        "sample": f"non-pivot rows among the first {rows} of {n} rows, same pivots/U as the full round "
                  f"({info['nnz_reduced']} nnz reduced in {info['sec_schur']:.2f} s; libspasm-algorithm CPU restatement, not libspasm)",
    }
    # one thread (SURVEY 8d asks for both): a sample sized for ~0.3 of the budget
    try:
        rows1 = int(min(n, max(2000, rows * 0.3 * target_seconds / max(info["sec_schur"], 1e-9) / max(info["threads"], 1) * 4)))
        O.set_threads(1)
        _, i1 = O.schur_round(A, row_lo=0, row_hi=rows1)
        out["one_thread"] = {"value": i1["nnz_reduced"] / max(i1["sec_schur"], 1e-9), "cores": 1,
                             "sample": f"first {rows1} rows, {i1['nnz_reduced']} nnz reduced in {i1['sec_schur']:.2f} s"}
    except Exception as exc:  # noqa: BLE001 - the baseline is reported, never required
        out["one_thread"] = {"error": repr(exc)}
    finally:
        try:
            O.set_threads(0)
        except Exception:  # noqa: BLE001
            pass
    # the real thing, if somebody put it on the box: spasm_echelonize of the same matrix, whole factorization
    lib_path = os.environ.get("SPASM_LIB")
    if lib_path:
        out["libspasm"] = real_libspasm_baseline(lib_path, A)
    return out


def real_libspasm_baseline(path, A):
    """Time a genuine libspasm (cbouilla/spasm) on the host: spasm_echelonize on the same CSR.  The struct layouts are the
    ones SpaSM.jl mirrors (include/spasm_amd.h); nothing here is needed by the product."""
    try:
        lib = C.CDLL(path)
        lib.spasm_echelonize.restype = C.c_void_p
        lib.spasm_echelonize.argtypes = [C.c_void_p, C.c_void_p]
        t0 = time.time()
        fact = lib.spasm_echelonize(A.data, None)
        dt = time.time() - t0
        return {"seconds_echelonize": round(dt, 3), "ok": bool(fact), "path": path}
    except Exception as exc:  # noqa: BLE001
        return {"error": repr(exc), "path": path}


if __name__ == "__main__":
    main()
