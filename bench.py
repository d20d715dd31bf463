#!/usr/bin/env python3
"""Headline benchmark of the hot path (BASELINE.json): batched greedy HNSW search on
1M x 768 f32 vectors, queries/sec at recall@10 >= 0.95, one process per GPU.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the search kernel over one batch of --nq synthetic queries per GPU,
inputs already resident in HBM.  Every rank holds the full store and graph (search
replicates, north_star) and searches its own query batch: weak scaling, no data-path
collective.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s; ~6.3 achievable)


class _DevArray:
    """expose a raw device pointer to torch (zero copy) via __cuda_array_interface__"""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--vectors", dest="n", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--queries", dest="nq", type=int, default=100_000, help="queries per step (one batch) per GPU")
    ap.add_argument("--dataset", default="clustered", choices=["clustered", "iid"])
    ap.add_argument("--ef", type=int, default=0, help="fix number_of_candidates (0 = sweep for recall@10>=0.95)")
    ap.add_argument("--probe-depth", type=int, default=0)
    ap.add_argument("--upper", type=int, default=0, help="upper_layer_candidate_count with --ef (0 = same as --ef)")
    ap.add_argument("--target-recall", type=float, default=0.95)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--skip-iid", dest="no_iid", action="store_true", help="skip the secondary iid-uniform measurement")
    ap.add_argument("--skip-pq", dest="no_pq", action="store_true", help="skip the BASELINE config-5 (PQ) measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import parallel_hnsw_amd as ph

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "gloo" + BENCH_SHARE_GPU=1: 1-GPU rehearsal
    if os.environ.get("BENCH_SHARE_GPU"):
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local)
    stream = torch.cuda.current_stream().cuda_stream

    def log(*a):
        if rank == 0:
            print("[bench]", *a, file=sys.stderr, flush=True)

    def make_store(kind, n, first):
        if kind == "clustered":
            return ph.VectorStore.clustered(n, args.dim, seed=42, first=first, n_clusters=1000, noise=1.0, device=local)
        return ph.VectorStore.synthetic(n, args.dim, seed=42, first=first, device=local)

    def tensor_of(store):
        return torch.as_tensor(_DevArray(store.rows_dev, (store.n, store.ld)), device=dev)

    gt_info = {}

    def ground_truth(store, q_store, k=10):
        """exact top-k by the library's brute-force kernel (f32 MFMA GEMM + top-k, G1)"""
        out_i = torch.empty((q_store.n, k), dtype=torch.int32, device=dev)
        out_d = torch.empty((q_store.n, k), dtype=torch.float32, device=dev)
        ms = store.bruteforce_topk_device(q_store.rows_dev, q_store.ld, q_store.n, k, out_i.data_ptr(),
                                          out_d.data_ptr())
        flops = 2.0 * q_store.n * store.n * store.ld
        gt_info.update({"kernel": "ph_gemm_nt_mfma_kernel (v_mfma_f32_32x32x2_f32)", "queries": q_store.n,
                        "gemm_ms": round(ms, 2), "tflops": round(flops / (ms * 1e-3) / 1e12, 1), "peak_tflops": 157.3,
                        "frac": round(flops / (ms * 1e-3) / 1e12 / 157.3, 3)})
        return out_i.to(torch.int64)

    def recall_at_10(ids_t, gt_t):
        hit = (ids_t[:, :10, None].to(torch.int64) == gt_t[:, None, :]).any(2).float().sum(1) / 10.0
        return float(hit.mean())

    class Runner:
        """owns the device output buffers of one query batch"""

        def __init__(self, index, q_store, ef_max=1024):
            self.ix = index
            self.q = q_store
            nq = q_store.n
            self.ids = torch.empty((nq, ef_max), dtype=torch.int32, device=dev)
            self.d = torch.empty((nq, ef_max), dtype=torch.float32, device=dev)
            self.len = torch.empty(nq, dtype=torch.int32, device=dev)
            self.stats = torch.empty((nq, 2), dtype=torch.int32, device=dev)
            self.status = torch.empty(nq, dtype=torch.int32, device=dev)

        def launch(self, sp, on_stream=None):
            self.ix.search_batch_device(self.q.n, sp, self.ids.data_ptr(), self.d.data_ptr(), self.len.data_ptr(),
                                        self.status.data_ptr(), queries=self.q.rows_dev, ldq=self.q.ld,
                                        out_stats=self.stats.data_ptr(), stream=on_stream or stream)

        def result_ids(self, ef):
            return self.ids.view(-1)[: self.q.n * ef].view(self.q.n, ef)

    def measure_dataset(kind, headline):
        t0 = time.time()
        store = make_store(kind, args.n, 0)
        torch.cuda.synchronize()
        log("%s store %d x %d generated in %.1f s" % (kind, args.n, args.dim, time.time() - t0))
        bp = ph.BuildParameters()
        build_mode = "single GPU"
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.time()
        index = None
        if world > 1:
            # index construction sharded over the ranks: node ranges per round, RCCL all-gather
            # of the per-node results (parallel_hnsw_amd/sharded.py)
            try:
                eng = ph.GpuEngine(store, bp, device=dev)
                comm = ph.TorchComm()
                index = ph.ShardedBuilder(eng, comm).generate(np.arange(args.n, dtype=np.uint64))
                build_mode = "sharded x%d, %.0f MB all-gathered per rank in %.0f ms" % (
                    world, comm.bytes_gathered / 1e6, comm.seconds * 1e3)
            except Exception as exc:  # keep the search measurement alive; say what happened
                log("sharded build failed (%r); every rank builds the full index instead" % (exc,))
                index = None
        if index is None:
            if world > 1:
                build_mode = "replicated (each rank built the full index)"
            index = ph.Hnsw.generate(store, np.arange(args.n, dtype=np.uint64), bp)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        build_s = time.time() - t0
        b_dist, b_hops = index.counters()  # every search of the build rounds (this rank's share when sharded)
        self_recall = index.stochastic_recall()  # the reference's own estimator (lib.rs:1463-1499): 10 % sample, self in the results
        build_bytes = b_dist * store.ld * 4 + b_hops * 48 * 4
        log("index built in %.1f s (%.0f vectors/s), layers %s" % (
            build_s, args.n / build_s, [index._layer(l).node_count() for l in range(index.layer_count())]))
        # calibration queries are the same on every rank => every rank picks the same parameters
        cal = make_store(kind, 8192, 2 ** 33)
        cal_gt = ground_truth(store, cal)
        cal_run = Runner(index, cal)
        if args.ef:
            grid = [(args.ef, args.upper or args.ef, args.probe_depth or 2)]
        else:
            # (number_of_candidates, upper_layer_candidate_count, probe_depth): the reference's three
            # SearchParameters (parameters.rs:3-14); its default keeps the upper count equal to the
            # bottom one, a narrower upper queue is the classic HNSW setting
            base = [(64, 2), (128, 2), (128, 4), (96, 8), (104, 8), (128, 5), (96, 16), (112, 8), (128, 6), (128, 8), (200, 4), (300, 2), (300, 4), (200, 8), (300, 8),
                    (128, 16), (300, 16), (512, 16), (512, 32), (1024, 64)]
            # (measured: a narrower upper count does not help -- the reference searches every layer with a
            # queue of number_of_candidates and only truncates its output, lib.rs:258-276)
            grid = [(ef, ef, pd) for ef, pd in base]
        sweep, chosen = [], None
        for ef, up, pd in grid:
            sp = ph.SearchParameters(ef, up, pd)
            cal_run.launch(sp)
            cal_run.launch(sp)
            torch.cuda.synchronize()
            ms = index.kernel_ms()
            rec = recall_at_10(cal_run.result_ids(ef), cal_gt)
            qps = cal.n / ms * 1e3
            sweep.append({"ef": ef, "upper": up, "probe_depth": pd, "recall_at_10": round(rec, 4), "qps_cal": round(qps)})
            log("sweep ef=%d upper=%d pd=%d recall@10=%.4f  %.0f q/s" % (ef, up, pd, rec, qps))
            # fastest setting that meets the target; settings within 3 % count as equal and the
            # earlier one is kept, so that run-to-run noise does not flip the choice
            # (0.003 of margin on the calibration set, so that the timed batch -- other queries -- meets it too)
            if rec >= args.target_recall + 0.003 and (chosen is None or qps > 1.03 * chosen[3]):
                chosen = (ef, up, pd, qps, rec)
        met = chosen is not None
        if not met:  # report honestly at the BASELINE configuration ef_search=128
            e = [s for s in sweep if s["ef"] == 128 and s["probe_depth"] == 2] or sweep[:1]
            chosen = (e[0]["ef"], e[0]["upper"], e[0]["probe_depth"], e[0]["qps_cal"], e[0]["recall_at_10"])
        ef, up, pd = chosen[0], chosen[1], chosen[2]
        if world > 1:
            # the sweep is timing based: all ranks adopt rank 0's choice (outside the timed region)
            t = torch.tensor([ef, up, pd, int(met)], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.broadcast(t, src=0)
            ef, up, pd, met = int(t[0]), int(t[1]), int(t[2]), bool(int(t[3]))
        sp = ph.SearchParameters(ef, up, pd)
        # this rank's own query batch (weak scaling: fixed work per GPU)
        qstore = make_store(kind, args.nq, 2 ** 32 + rank * args.nq)
        run = Runner(index, qstore, ef_max=ef)
        gt = ground_truth(store, qstore)
        # steps are issued on two streams alternately (two workspaces inside the library): the
        # tail of one batch overlaps the head of the next, like back-to-back batches in serving
        torch.cuda.synchronize()  # ground truth (default stream) done before side streams touch memory
        run_b = Runner(index, qstore, ef_max=ef)
        streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        for s_ in streams:
            s_.wait_stream(torch.cuda.current_stream())
        runs = [run, run_b]

        one_stream = bool(os.environ.get("BENCH_ONE_STREAM"))

        def step(i):
            if one_stream:
                run.launch(sp)
            else:
                runs[i & 1].launch(sp, streams[i & 1].cuda_stream)

        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        kms = []
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        # latency side of the same path (batch sizes 1, 64, 1 024, 10 000): one
        # isolated launch per measurement, the first nq queries of the timed batch
        batch_sweep = []
        if headline and rank == 0:
            for b in (1, 64, 1024, 10000):
                if b > args.nq:
                    break
                ms = []
                for _ in range(5):
                    index.search_batch_device(b, sp, run.ids.data_ptr(), run.d.data_ptr(), run.len.data_ptr(),
                                              run.status.data_ptr(), queries=qstore.rows_dev, ldq=qstore.ld,
                                              out_stats=run.stats.data_ptr(), stream=stream)
                    torch.cuda.synchronize()
                    ms.append(index.kernel_ms())
                batch_sweep.append({"queries": b, "kernel_ms": round(min(ms), 3), "queries_per_s": round(b / min(ms) * 1e3)})
            log("batch sweep: " + ", ".join("%d: %.2f ms" % (x["queries"], x["kernel_ms"]) for x in batch_sweep))
        # per-launch kernel time from HIP events on the launch stream (separate, untimed pass so
        # the event reads do not serialise the timed region)
        for _ in range(min(args.steps, 10)):
            run.launch(sp)
            torch.cuda.synchronize()
            kms.append(index.kernel_ms())
        if world > 1:
            t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        rec = recall_at_10(run.result_ids(ef), gt)
        st = run.stats.to(torch.int64)
        assert int(run.status.abs().sum()) == 0, "search reported per-query errors"
        w0 = index._layer(index.layer_count() - 1).neighborhood_size
        n_dist, n_hops = int(st[:, 0].sum()), int(st[:, 1].sum())
        row_bytes = store.ld * 4
        alg_bytes = n_dist * row_bytes + n_hops * w0 * 4 + args.nq * ef * 12
        k_ms = float(np.mean(kms))
        out = {
            "dataset": kind, "ef": ef, "upper": up, "probe_depth": pd, "recall_target_met": met, "recall_at_10": round(rec, 4),
            "elapsed": elapsed, "kernel_ms": k_ms, "alg_bytes": alg_bytes, "n_dist_per_query": n_dist / args.nq,
            "n_hops_per_query": n_hops / args.nq, "build_s": build_s, "build_mode": build_mode, "sweep": sweep,
            "batch_sweep": batch_sweep,
            "build_self_recall": round(self_recall, 5),
            "build_roofline": {"bound": "hbm", "distance_evals": b_dist, "hops": b_hops,
                               "algorithmic_bytes": build_bytes, "achieved": round(build_bytes / build_s / 1e9, 1),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(build_bytes / build_s / 1e9 / HBM_PEAK_GBS, 4),
                               "note": "searches of the build rounds only (K2: 92 % of the build's GPU time); whole "
                                       "build wall time incl. host control flow; per rank when sharded"},
            "dispatches": (1 + sum(1 for l in range(1, index.layer_count()) if index._layer(l).node_count() >= 32768))
            if args.nq >= 32768 else 1,
        }
        return out, store, index, qstore, run, sp, gt

    res, store, index, qstore, run, sp, gt = measure_dataset(args.dataset, True)
    value = world * args.nq * args.steps / res["elapsed"]
    achieved = res["alg_bytes"] / (res["kernel_ms"] * 1e-3) / 1e9

    traffic = None
    try:  # HBM bytes per launch from the committed rocprofv3 PMC pass of this very workload
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "search_kernel_summary.json"))):
            pm = json.load(open(f)).get("pmc", {})
            w = pm.get("workload", {})
            if (w.get("dataset"), w.get("n"), w.get("dim"), w.get("nq"), w.get("ef"), w.get("upper"),
                    w.get("probe_depth")) == (res["dataset"], args.n, args.dim, args.nq, res["ef"], res["upper"],
                                              res["probe_depth"]):
                traffic = pm.get("traffic_bytes_per_launch")
    except Exception:
        traffic = None

    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        # reference-algorithm CPU restatement (oracle/, kind "port") on this box's host cores,
        # same graph, same vectors, a bounded sample of the same query batch
        import oracle
        cores = os.cpu_count() or 1
        t0 = time.time()
        rows_h = store.read()
        oix = oracle.Index(rows_h, dim=store.dim, metric=oracle.METRIC_COSINE_HALF, sum_mode=oracle.SUM_SEQ)
        for l in range(index.layer_count()):
            L = index._layer(l)
            oix.push_layer(L.nodes, L.neighbors, L.neighborhood_size)
        qh = qstore.read()
        log("cpu baseline: copied store+graph to host in %.1f s, %d cores" % (time.time() - t0, cores))
        spt = (sp.number_of_candidates, sp.upper_layer_candidate_count, sp.probe_depth)
        t0 = time.time()
        oix.search(queries=qh[:2 * cores], sp=spt, threads=cores)
        per_q = (time.time() - t0) / (2 * cores)
        sample = int(max(2 * cores, min(args.nq, args.cpu_seconds / max(per_q, 1e-9))))
        t0 = time.time()
        ci, cd, cl = oix.search(queries=qh[:sample], sp=spt, threads=cores)
        dt = time.time() - t0
        crec = recall_at_10(torch.from_numpy(ci[:, :10].astype(np.int64)).to(dev), gt[:sample])
        gi = run.result_ids(sp.number_of_candidates)[:sample].cpu().numpy().astype(np.uint64)
        same = float((gi[:, :10] == ci[:, :10]).mean())
        cpu = {"value": round(sample / dt, 1), "unit": "queries/s", "cores": cores, "kind": "port",
               "sample": "%d of the %d timed queries, same graph and parameters, sequential-f32 reference arithmetic"
                         % (sample, args.nq),
               "recall_at_10": round(crec, 4), "top10_ids_equal_to_gpu": round(same, 5)}
        log("cpu baseline %.0f q/s on %d cores (%d queries in %.1f s)" % (sample / dt, cores, sample, dt))
        del oix
        # the second metric (index-build vectors/sec) on the host cores: the oracle's build of a
        # bounded prefix of the same vectors, reference defaults incl. promotion.  Build cost per
        # vector grows with n (more layers, longer searches), so this flatters the CPU.
        try:
            nb = min(args.n, 20_000)
            t0 = time.time()
            obp = oracle.default_build_params()
            ob = oracle.Index.generate(rows_h[:nb], np.arange(nb), obp, dim=store.dim, threads=cores)
            dtb = time.time() - t0
            cpu["build"] = {"value": round(nb / dtb, 1), "unit": "vectors/s", "cores": cores, "kind": "port",
                            "sample": "first %d of the %d vectors, reference default parameters" % (nb, args.n)}
            log("cpu baseline build: %d vectors in %.1f s (%.0f vectors/s)" % (nb, dtb, nb / dtb))
            del ob
        except Exception as exc:
            cpu["build"] = {"error": repr(exc)}
        del rows_h

    pq = None
    if rank == 0 and world == 1 and not args.no_pq and not args.ef:
        # BASELINE configs[4]: PQ m=96, 8-bit codes, per-query f32 ADC table, full-precision re-rank (pq.rs:346-364)
        try:
            t0 = time.time()
            qh = ph.QuantizedHnsw(256, store, ph.BuildParameters(promote=0), m=96 if args.dim % 96 == 0 else 4)
            torch.cuda.synchronize()
            pq_build = time.time() - t0
            log("pq: codebooks + codes + graph over codes in %.1f s" % pq_build)
            # the graph is built with the exact f32 table (symmetric distances); queries are then
            # scored through 8-bit table entries (phnsw_pq_set_table_mode 2: search-only, asymmetric)
            qh.store.set_table_mode("u8")
            ef_max = 1024
            pids = torch.empty((args.nq, ef_max), dtype=torch.int32, device=dev)
            pd_ = torch.empty((args.nq, ef_max), dtype=torch.float32, device=dev)
            pln = torch.empty(args.nq, dtype=torch.int32, device=dev)
            pst = torch.empty((args.nq, 2), dtype=torch.int32, device=dev)
            pstatus = torch.empty(args.nq, dtype=torch.int32, device=dev)
            best = None
            for ef, pdp in [(128, 8), (300, 8), (384, 16), (448, 12), (448, 16), (512, 12), (512, 16), (1024, 32)]:
                spq = ph.SearchParameters(ef, ef, pdp)
                for _ in range(2):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    qh.search_batch_device(args.nq, spq, qstore.rows_dev, qstore.ld, pids.data_ptr(), pd_.data_ptr(),
                                           pln.data_ptr(), pstatus.data_ptr(), pst.data_ptr(), stream=stream)
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - t0
                rec = recall_at_10(pids.view(-1)[: args.nq * ef].view(args.nq, ef), gt)
                log("pq sweep ef=%d pd=%d recall@10=%.4f %.0f q/s" % (ef, pdp, rec, args.nq / dt))
                cur = {"ef": ef, "probe_depth": pdp, "recall_at_10": round(rec, 4), "queries_per_s": round(args.nq / dt),
                       "distance_evals_per_query": float(pst[:, 0].float().mean()),
                       "hops_per_query": float(pst[:, 1].float().mean())}
                ok = rec >= args.target_recall
                if best is None or (ok and not best["met"]) or (ok and cur["queries_per_s"] > 1.03 * best["queries_per_s"]):
                    best = dict(cur, met=ok)
            m_ = qh.store.m
            bq = best["distance_evals_per_query"] * m_ + best["hops_per_query"] * 48 * 4 + best["ef"] * (store.ld * 4 + 12)
            pq = {"workload": "configs[4]: %dx%d PQ m=%d, 8-bit codes (%d B/vector), 8-bit per-query ADC table %d KiB per wave in "
                              "global memory (L2), search over codes + f32 re-rank; graph built with the f32 table, without "
                              "promotion" % (args.n, args.dim, m_, m_, m_ * 256 // 1024),
                  "build_s": round(pq_build, 1), "recall_target_met": best.pop("met"), **best,
                  "algorithmic_bytes_per_query": round(bq),
                  "roofline_gbs": round(best["queries_per_s"] * bq / 1e9, 1),
                  "note": "bound by the L1 miss rate of the table gathers (PMC TCP_TCC_READ_REQ: ~700 L2 requests per hop with f32 "
                          "entries, about half with 8-bit entries)"}
            del qh, pids, pd_
        except Exception as exc:
            pq = {"error": repr(exc)}
            log("pq measurement failed: %r" % (exc,))

    iid = None
    if rank == 0 and world == 1 and not args.no_iid and args.dataset != "iid" and not args.ef:
        # the reference's own data distribution (bigvec.rs:59-65) at the BASELINE setting ef=128
        del run, gt, qstore, index, store
        torch.cuda.empty_cache()
        saved = (args.ef, args.probe_depth)
        args.upper = 0
        args.ef, args.probe_depth = 128, 2
        r2, *_ = measure_dataset("iid", False)
        args.ef, args.probe_depth = saved
        iid = {"dataset": "iid-uniform (bigvec.rs:59-65)", "ef": 128, "probe_depth": 2,
               "queries_per_s": round(args.nq * args.steps / r2["elapsed"]),
               "recall_at_10": r2["recall_at_10"],
               "roofline_gbs": round(r2["alg_bytes"] / (r2["kernel_ms"] * 1e-3) / 1e9, 1),
               "build_vectors_per_s": round(args.n / r2["build_s"])}

    if rank == 0:
        line = {
            # BASELINE.json's metric, first clause (the second, index-build vectors/sec, is
            # build_vectors_per_sec below)
            "metric": "queries/sec at recall@10\u22650.95 on 1M\u00d7768 f32",
            "value": round(value, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(res["elapsed"] / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "configs[1]: %dx%d f32 cosine (1-dot)/2, batched greedy search, %d queries/step/GPU"
                            % (args.n, args.dim, args.nq),
                "dataset": "%s synthetic (1000 unit centres + uniform noise, normalised)" % res["dataset"]
                           if res["dataset"] == "clustered" else "iid uniform(-1,1) normalised (bigvec.rs:59-65)",
                "number_of_candidates": res["ef"], "upper_layer_candidate_count": res["upper"],
                "probe_depth": res["probe_depth"],
                "build": "reference defaults order=12 M=24 M0=48 ef_link=300 (parameters.rs:50-64), built on GPU",
                "parallelism": "replicated index, queries sharded x%d; steps issued on 2 streams" % world,
            },
            "recall_at_10": res["recall_at_10"],
            "recall_target_met": res["recall_target_met"],
            "build_vectors_per_sec": round(args.n / res["build_s"], 1),
            "build_mode": res["build_mode"],
            "queries_per_step_per_gpu": args.nq, "argv": " ".join(sys.argv[1:]),
            "distance_evals_per_query": round(res["n_dist_per_query"], 1),
            "hops_per_query": round(res["n_hops_per_query"], 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         # measured bytes on the L2's memory side (PMC pass: HBM + Infinity Cache) over the same launch time
                         "traffic_gbs": round(traffic / (res["kernel_ms"] * 1e-3) / 1e9, 1) if traffic else None,
                         "traffic_frac": round(traffic / (res["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                         if traffic else None,
                         "kernel": "ph_search_kernel", "kernel_ms": round(res["kernel_ms"], 4),
                         "algorithmic_bytes_per_launch": res["alg_bytes"],
                         # batches >= 32768 queries descend in several dispatches of the same kernel (small top
                         # layers; then each large layer in locality order): kernel_ms spans all (HIP events)
                         "dispatches_per_launch": res["dispatches"],
                         "note": "achieved = algorithmic bytes / time; it can exceed the HBM peak because "
                                 "neighbouring queries are scheduled together and share rows in L2 / the Infinity "
                                 "Cache (traffic = bytes measured on the L2's memory side: FETCH_SIZE x 2 + WRITE_SIZE)",
                         # the timed region pipelines launches on two streams; per-launch durations are
                         # measured on isolated launches (above); this is the steady-state rate
                         "achieved_pipelined": round(res["alg_bytes"] * args.steps / res["elapsed"] / 1e9, 1)},
            "ground_truth": dict(gt_info, note="exact top-10 by brute force on the f32 MFMA units; mfma bound"),
            "cpu_baseline": cpu,
            "secondary": iid,
            "pq": pq,
            "batch_sweep": res["batch_sweep"],
            "build_roofline": res["build_roofline"],
            "build_self_recall": res["build_self_recall"],
            "sweep": res["sweep"],
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
