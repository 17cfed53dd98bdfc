"""BASELINE.json configs C1 and C2 (and C3's Viterbi variant) on one MI355X, one JSON line each (bench.py embeds them as
`configs`):

  c1  benchmarks/ctc.cpp with batch = 1: T = 100, alphabet 28, U = 20 -- one utterance through the per-graph
      functions (reference names, bench_native/ctc_step.cpp: gtn_bench_single_utterance): latency per loss
  c2  forwardScore on 256 linear-chain emission graphs (T = 150, C = 32), one batched launch
  c3v viterbiScore / viterbiPath of intersect(ctc, emissions) at C3's shape, symbolic and built route

Each line carries `value` (losses/s resp. graphs/s), `ms`, a `roofline` for the dominant kernel family (bytes
from the engine's own accounting, hipEvent time) and a `cpu_baseline`: the UNMODIFIED reference
(oracle/_ref/libgtn_ref.so through tests/refbackend/gtn_ref.py) doing the same on the host cores.
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
HBM_PEAK_GBS = 8000.0


def roofline_of(gtn, steps):
    prof = {n: gtn.prof_get(n) for n in gtn.prof_names()}
    fams = {k: v for k, v in prof.items() if v["launches"] and v["total_ms"] > 0}
    if not fams:
        return None, {}
    dom = max(fams, key=lambda k: fams[k]["total_ms"])
    e = fams[dom]
    ms = e["total_ms"] / e["launches"]
    per = e["algorithmic_bytes"] / e["launches"]
    gbs = per / (ms * 1e-3) / 1e9 if per > 0 else None
    roof = {"bound": "hbm", "kernel_family": dom, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": (gbs / HBM_PEAK_GBS) if gbs else None, "traffic": None, "traffic_source": None,
            "ms_per_launch": ms, "algorithmic_bytes_per_launch": per,
            "note": "latency-bound at this size: a launch moves kilobytes"}
    return roof, {k: {"ms_per_step": v["total_ms"] / steps, "launches_per_step": v["launches"] / steps} for k, v in fams.items()}


def ref_api():
    sys.path.insert(0, os.path.join(ROOT, "tests", "refbackend"))
    import gtn_ref
    return gtn_ref


def c1():
    import torch
    import gtn_amd as gtn
    import graphgen as gg
    T, Cn, U = 100, 28, 20
    em, tg = gg.ctc_inputs(1234, 1, T, Cn, U)
    em_dev = torch.from_numpy(em).cuda()
    tg = np.ascontiguousarray(tg, np.int32)
    native = C.CDLL(os.path.join(ROOT, "bench_native", "libgtn_bench.so"))
    native.gtn_bench_single_utterance.restype = C.c_double
    native.gtn_bench_single_utterance.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    loss = C.c_float()
    iters = 300
    # (priming: 2 000 losses = 0.2 s -- pools, code objects, and the GPU's clocks, which take tens of milliseconds to
    #  leave idle: the 300 timed losses are 30 ms long)
    native.gtn_bench_single_utterance(em_dev.data_ptr(), tg.ctypes.data, T, Cn, U, 2000, C.byref(loss))
    gtn.prof_reset()
    gtn.prof_enable(True)
    ms = native.gtn_bench_single_utterance(em_dev.data_ptr(), tg.ctypes.data, T, Cn, U, iters, C.byref(loss))
    gtn.prof_enable(False)
    if ms < 0:
        native.gtn_bench_last_error.restype = C.c_char_p
        raise RuntimeError(native.gtn_bench_last_error().decode())
    roof, kernels = roofline_of(gtn, iters + 10)
    out = {"config": "C1: benchmarks/ctc.cpp CPU reference shape, batch=1, T=100, alphabet=28, target_len=20 "
                     "(ctcGraph, linearGraph + setWeights, intersect, 2 forwardScore, subtract, backward, item)",
           "metric": "CTC forward+backward losses/sec", "value": 1e3 / ms, "unit": "losses/s", "ms_per_loss": ms,
           "loss": float(loss.value), "host": "C++ per-graph functions, reference names (bench_native/ctc_step.cpp); no compose-mode "
                                              "hint: the engine's default keeps the product of a host-built target symbolic (band sweeps)",
           "roofline": roof, "kernels": kernels}
    # the unmodified reference, one thread (its benchmark's own case): same utterance
    try:
        ref = ref_api()
        g = gg.to_api(ref, gg.ctc_target_graph(tg[0].tolist()))
        t0 = time.perf_counter()
        n = 0
        while n < 50 or time.perf_counter() - t0 < 3.0:
            e = ref.linear_graph(T, Cn)
            e.set_weights(em[0].reshape(-1))
            l = ref.subtract(ref.forward_score(e), ref.forward_score(ref.intersect(g, e)))
            ref.backward(l)
            n += 1
        sec = (time.perf_counter() - t0) / n
        out["cpu_baseline"] = {"value": 1.0 / sec, "unit": "losses/s", "cores": 1, "kind": "reference", "ms_per_loss": sec * 1e3,
                               "loss": float(l.item()),
                               "sample": f"{n} repetitions of the same utterance through the unmodified reference (Python mirror "
                                         "over oracle/_ref), one host thread; the target graph is built once outside the loop"}
    except Exception as e:  # a baseline must not cost the line
        out["cpu_baseline"] = {"error": str(e)[:200]}
    print(json.dumps(out))


def c2():
    import torch
    import gtn_amd as gtn
    B, T, Cn = 256, 150, 32
    rng = np.random.default_rng(1234)
    em = (rng.random((B, T, Cn), dtype=np.float32) * 10 - 5).astype(np.float32)
    em_dev = torch.from_numpy(em).cuda()
    out_dev = torch.empty(B, dtype=torch.float32, device="cuda")

    def step():
        ems = gtn.linear_graph_n(B, T, Cn, em_dev)
        s = gtn.forward_score(ems)
        gtn.items_to_device(s, out_dev)
        return s

    for _ in range(10):
        step()
    gtn.synchronize()
    torch.cuda.synchronize()
    iters = 200
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    gtn.synchronize()
    torch.cuda.synchronize()
    ms_python = (time.perf_counter() - t0) / iters * 1e3
    # the same three calls from C++ (bench_native/ctc_step.cpp: gtn_bench_forward_score_linear): what `value` is
    native = C.CDLL(os.path.join(ROOT, "bench_native", "libgtn_bench.so"))
    native.gtn_bench_forward_score_linear.restype = C.c_double
    native.gtn_bench_forward_score_linear.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
    native.gtn_bench_forward_score_linear(em_dev.data_ptr(), B, T, Cn, out_dev.data_ptr(), 1500)  # priming (0.2 s), as in c1()
    out_dev.zero_()
    gtn.prof_reset()
    gtn.prof_enable(True)
    ms = native.gtn_bench_forward_score_linear(em_dev.data_ptr(), B, T, Cn, out_dev.data_ptr(), iters)
    gtn.prof_enable(False)
    if ms < 0:
        native.gtn_bench_last_error.restype = C.c_char_p
        raise RuntimeError(native.gtn_bench_last_error().decode())
    scores_vec = out_dev.cpu().numpy()
    # ... and through gtn::Batch (batch records: the 256 chains as ONE object, include/gtn/batch.h) -- the form the C3
    # headline uses; `value` is this one, the vector overloads (the reference binding's form) are reported beside it
    native.gtn_bench_forward_score_linear_batch.restype = C.c_double
    native.gtn_bench_forward_score_linear_batch.argtypes = native.gtn_bench_forward_score_linear.argtypes
    native.gtn_bench_forward_score_linear_batch(em_dev.data_ptr(), B, T, Cn, out_dev.data_ptr(), 5000)  # priming (0.2 s)
    out_dev.zero_()
    gtn.prof_reset()
    gtn.prof_enable(True)
    ms_batch = native.gtn_bench_forward_score_linear_batch(em_dev.data_ptr(), B, T, Cn, out_dev.data_ptr(), iters)
    gtn.prof_enable(False)
    if ms_batch < 0:
        native.gtn_bench_last_error.restype = C.c_char_p
        raise RuntimeError(native.gtn_bench_last_error().decode())
    roof, kernels = roofline_of(gtn, iters + 10)
    scores = out_dev.cpu().numpy()
    # fp64 log-sum-exp per row, summed: the exact answer
    want = np.logaddexp.reduce(em.astype(np.float64), axis=2).sum(axis=1)
    out = {"config": "C2: forwardScore on batch=256 linear-chain emission graphs (T=150, C=32), one batched launch "
                     "(the chains over one device tensor, forwardScore, scores left on the device)",
           "metric": "forwardScore graphs/sec", "value": B / (ms_batch * 1e-3), "unit": "graphs/s", "ms_per_batch": ms_batch,
           "max_rel_err_vs_fp64": float(np.max(np.abs(scores - want) / np.abs(want))),
           "host": "C++ (bench_native/ctc_step.cpp: gtn::Batch::linear + batched::forwardScore + itemsToDevice)",
           "value_reference_api": B / (ms * 1e-3), "ms_per_batch_reference_api": ms,
           "host_reference_api": "C++ (bench_native/ctc_step.cpp: linearGraphs + batched::forwardScore on std::vector<Graph> + "
                                 "gtnx_items_device_n: 256 graph handles in, 256 out)",
           "max_rel_err_vs_fp64_reference_api": float(np.max(np.abs(scores_vec - want) / np.abs(want))),
           "ms_per_batch_python_host": ms_python, "roofline": roof, "kernels": kernels}
    try:
        ref = ref_api()
        gs = []
        for b in range(B):
            e = ref.linear_graph(T, Cn)
            e.set_weights(em[b].reshape(-1))
            gs.append(e)
        t0 = time.perf_counter()
        n = 0
        while n < 5 or time.perf_counter() - t0 < 3.0:
            r = ref.forward_score(gs)  # the binding's vector overload: parallelMap over the host cores
            n += 1
        sec = (time.perf_counter() - t0) / n
        out["cpu_baseline"] = {"value": B / sec, "unit": "graphs/s", "cores": os.cpu_count(), "kind": "reference",
                               "ms_per_batch": sec * 1e3,
                               "sample": f"{n} repetitions of forward_score over the same 256 graphs through the unmodified "
                                         "reference's vector overload (parallelMap on all host cores); graphs built once"}
    except Exception as e:
        out["cpu_baseline"] = {"error": str(e)[:200]}
    print(json.dumps(out))


def c3v():
    """viterbiScore / viterbiPath of intersect(ctc_target, emissions) at BASELINE config C3's shape (B = 512, T = 1000,
    C = 256, U = 100; functions.cpp:324-330 -> shortest.cpp:190-272), symbolic route (band_viterbi_wave_kernel: the lattice
    is never built) and built route (compose_kernel, then the tropical sweep + pointer chase over the built lattices),
    labels of the first utterances compared EQUAL with the unmodified reference in this run."""
    import torch
    import gtn_amd as gtn
    import graphgen as gg
    B, T, Cn, U = 512, 1000, 256, 100
    if len(sys.argv) > 2:
        B = int(sys.argv[2])
    em, tg = gg.ctc_inputs(1234, B, T, Cn, U)
    em_dev = torch.from_numpy(em).cuda()
    ctcs = [gg.to_api(gtn, gg.ctc_target_graph(t.tolist())) for t in tg]
    for g in ctcs:
        g.arc_sort()
    ems = gtn.linear_graph_n(B, T, Cn, em_dev)
    N, A = 2 * U + 1, None

    def run(mode, what, iters):
        prev = gtn.compose_mode(mode)
        try:
            def step():
                comp = gtn.intersect(ctcs, ems)
                return gtn.viterbi_path(comp) if what == "path" else gtn.viterbi_score(comp)
            for _ in range(2):
                r = step()
            gtn.synchronize()
            gtn.prof_reset()
            gtn.prof_enable(True)
            t0 = time.perf_counter()
            for _ in range(iters):
                r = step()
            gtn.synchronize()
            ms = (time.perf_counter() - t0) / iters * 1e3
            gtn.prof_enable(False)
            prof = {n: gtn.prof_get(n) for n in gtn.prof_names()}
            fams = {k: {"ms_per_launch": v["total_ms"] / v["launches"], "launches_per_step": v["launches"] / iters,
                        "algorithmic_bytes_per_launch": v["algorithmic_bytes"] / v["launches"]}
                    for k, v in prof.items() if v["launches"] and v["total_ms"] > 0}
            return r, ms, fams
        finally:
            gtn.compose_mode(prev)

    out = {"config": "C3_viterbi: viterbiScore / viterbiPath of intersect(ctc_target, emissions), T=1000, C=256, U=100, "
                     f"batch={B}, 1 MI355X", "metric": "viterbiPath utterances/sec"}
    paths, ms_path, fam_path = run(2, "path", 10)
    scores, ms_score, fam_score = run(2, "score", 10)
    # roofline of the symbolic route's one kernel: 4TC in + T N / 2 back-pointers out and in + 20 T of path
    per = B * (4.0 * T * Cn + 0.5 * T * N + 20.0 * T)
    k = fam_path.get("band_viterbi_path")
    roof = None
    if k:
        gbs = per / (k["ms_per_launch"] * 1e-3) / 1e9
        # HBM bytes per launch from the committed counter passes of the same kernel on the same shape
        # (tools/ubench/viterbi_bench under rocprofv3 --pmc, tools/profile_round4.sh): 2 * FETCH_SIZE + WRITE_SIZE KiB
        import glob
        traffic, src = None, None
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_viterbi_pmc_hbm.json")))
        if files and (B, T, Cn, U) == (512, 1000, 256, 100):
            pm = json.load(open(files[-1]))
            e = next((v for kk, v in pm.items() if kk.startswith("band_viterbi_wave_kernel")), None)
            if e:
                traffic = (2 * e["FETCH_SIZE"]["mean_per_launch"] + e["WRITE_SIZE"]["mean_per_launch"]) * 1024
                src = "profiles/" + os.path.basename(files[-1])
        roof = {"bound": "hbm", "kernel": "band_viterbi_wave_kernel<4, false>", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gbs / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": src, "ms_per_launch": k["ms_per_launch"],
                "algorithmic_bytes_per_launch": per}
    out["symbolic_route"] = {"viterbi_path_ms_per_batch": ms_path, "viterbi_score_ms_per_batch": ms_score,
                             "kernels_path": fam_path, "kernels_score": fam_score, "roofline": roof,
                             "note": "ms per batch includes the host side of the Python mirror (512 path graphs are built on "
                                     "the host from one device->host copy); the roofline is the kernel's own launch time"}
    out["value"] = B / (ms_path * 1e-3)
    out["unit"] = "utterances/s"
    out["roofline"] = roof
    # the decode as the reference writes it: parallelMap over a per-utterance function returning
    # viterbiPath(intersect(ctc, emissions)), C++ host (bench_native/ctc_step.cpp: gtn_bench_viterbi_reference_loop)
    try:
        import ctypes as C
        native = C.CDLL(os.path.join(ROOT, "bench_native", "libgtn_bench.so"))
        native.gtn_bench_viterbi_reference_loop.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]
        native.gtn_bench_viterbi_reference_loop.restype = C.c_double
        tgc = np.ascontiguousarray(tg, dtype=np.int32)
        lab = np.full((B, T), -1, np.int32)
        # the loop's pool threads are made now and inherit this thread's mask: the cores of the GPU's NUMA node, as
        # bench.py places the training loop (the reference's own pool, made later for cpu_baseline, gets every core)
        sys.path.insert(0, ROOT)
        from bench import gpu_local_cpus
        cpus, node = gpu_local_cpus(0)
        before = os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
        try:
            # (60 repetitions: the loop's garbage -- 512 path graphs per batch -- is taken apart in bursts, a full list
            #  every half dozen batches; ten repetitions caught one burst or two and read 5-7 ms for a 3.9 ms mean)
            ms_loop = native.gtn_bench_viterbi_reference_loop(em_dev.data_ptr(), tgc.ctypes.data, B, T, Cn, U, 60, lab.ctypes.data)
        finally:
            os.sched_setaffinity(0, before)
        if ms_loop > 0:
            same = all([int(x) for x in lab[b] if x >= 0] == [int(x) for x in paths[b].labels_to_list(False)] for b in range(8))
            out["reference_api"] = {"viterbi_path_ms_per_batch": ms_loop, "utterances_per_s": B / (ms_loop * 1e-3),
                                    "host": "C++: parallelMap over viterbiPath(intersect(ctcGraph, linearGraph + setWeights)) per "
                                            "utterance, every path looked at (bench_native/ctc_step.cpp)",
                                    "labels_equal_python_route": bool(same),
                                    "placement": ("the %d logical CPUs of the GPU's NUMA node (%s)" % (len(cpus), node)) if cpus else "unbound"}
            out["value"] = B / (ms_loop * 1e-3)
            out["value_source"] = "reference_api (C++ host); the Python mirror's figure is symbolic_route.viterbi_path_ms_per_batch"
    except Exception as e:
        out["reference_api"] = {"error": str(e)[:200]}
    # the built route at the configuration's own batch when the device has room for it (about 40 MB per utterance with
    # the arrays viterbiPath asks for: 20 GB at B = 512), else on a slice of 64.  (Rounds 4-5 measured the slice: 64
    # workgroups on 256 CUs -- the chase is one latency chain per lattice, so a launch takes the same time for 64
    # lattices as for 512, and the per-launch roofline figure of the slice was an eighth of the batch's.)
    nb = min(B, 64)
    try:
        import torch
        if torch.cuda.mem_get_info()[0] > 60 * (1 << 30):
            nb = B
    except Exception:
        pass
    sub_c, sub_e = ctcs[:nb], ems[:nb]
    prev = gtn.compose_mode(0)
    try:
        def bstep():
            comp = gtn.intersect(sub_c, sub_e)
            return comp, gtn.viterbi_path(comp)
        bstep()
        gtn.synchronize()
        gtn.prof_reset()
        gtn.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(3):
            comp, bpaths = bstep()
        gtn.synchronize()
        ms_built = (time.perf_counter() - t0) / 3 * 1e3
        gtn.prof_enable(False)
        prof = {n: gtn.prof_get(n) for n in gtn.prof_names()}
        A = comp[0].num_arcs()
        Nn = comp[0].num_nodes()
        fb = {}
        for name, v in prof.items():
            if not v["launches"] or v["total_ms"] <= 0:
                continue
            msl = v["total_ms"] / v["launches"]
            e = {"ms_per_launch": msl, "launches_per_step": v["launches"] / 3}
            if name.startswith("viterbi") or name.startswith("forward_score") or "path" in name:
                bytes_ = nb * (8.0 * A + 8.0 * Nn + 4.0 * Nn)  # 8A + 8N of the sweep + a back-pointer per node
                e["roofline"] = {"bound": "hbm", "achieved": bytes_ / (msl * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": bytes_ / (msl * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_,
                                 "traffic": None}
            fb[name] = e
        out["built_route"] = {"batch": nb, "viterbi_path_ms_per_batch": ms_built, "composed_nodes": int(Nn), "composed_arcs": int(A),
                              "kernels": fb}
        same_built = all(bpaths[b].labels_to_list() == paths[b].labels_to_list() for b in range(min(nb, 8)))
        out["built_route"]["labels_equal_symbolic_route"] = bool(same_built)
    finally:
        gtn.compose_mode(prev)
    # in-run parity and the host baseline: the unmodified reference on the same utterances
    try:
        ref = ref_api()
        cores = os.cpu_count() or 1
        nref = int(min(B, max(4, cores)))  # one utterance per host thread: ALL the host's logical cores (count stated below)
        rc = [gg.to_api(ref, gg.ctc_target_graph(tg[b].tolist())) for b in range(nref)]
        for g in rc:
            g.arc_sort()
        re_ = []
        for b in range(nref):
            e = ref.linear_graph(T, Cn)
            e.set_weights(em[b].reshape(-1))
            re_.append(e)
        t0 = time.perf_counter()
        rp = ref.viterbi_path(ref.intersect(rc, re_))  # vector overloads: parallelMap on the host cores
        sec = time.perf_counter() - t0
        rs = ref.viterbi_score(ref.intersect(rc[:4], re_[:4]))
        ncmp = 4
        labels_equal = all(rp[b].labels_to_list() == paths[b].labels_to_list() for b in range(ncmp))
        got_s = np.array(gtn.items(scores[:ncmp]), np.float64)
        want_s = np.array(ref.items(rs), np.float64)
        out["parity_in_run"] = {"n": ncmp, "labels_equal": bool(labels_equal),
                                "score_max_rel": float(np.max(np.abs(got_s - want_s) / np.maximum(np.abs(want_s), 1e-30))),
                                "checker": "reference (oracle/_ref/libgtn_ref.so)", "ok": bool(labels_equal)}
        out["cpu_baseline"] = {"value": nref / sec, "unit": "utterances/s", "cores": min(cores, nref), "kind": "reference",
                               "sample": f"intersect + viterbiPath of {nref} utterances through the unmodified reference's vector "
                                         f"overloads (parallelMap), one pass, {sec:.1f}s wall; host has {cores} logical cores"}
    except Exception as e:  # a baseline must not cost the line
        out["cpu_baseline"] = {"error": str(e)[:200]}
    print(json.dumps(out))
    if not out.get("parity_in_run", {}).get("ok", True):
        sys.exit(3)


if __name__ == "__main__":
    {"c1": c1, "c2": c2, "c3v": c3v}[sys.argv[1]]()
