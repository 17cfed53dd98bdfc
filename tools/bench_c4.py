"""BASELINE config C4 at full size on one MI355X: ASG with a dense transitions graph,
B=512, T=1000, C=512 (262 M product arcs per utterance -- never built).
Times (a) Viterbi decode = viterbiPath(compose(emissions, transitions)) and (b) the
full-connect term forwardScore(compose(emissions, transitions)) forward + backward
(gradients of all emissions and of the shared transitions).  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def transitions(gtn, N, w):
    g = gtn.Graph()
    n = np.arange(N)
    g.add_nodes(np.array([1] + [0] * N, np.uint8), np.array([0] + [1] * N, np.uint8))
    src = np.concatenate([np.zeros(N, np.int32), np.tile(n + 1, N).astype(np.int32)])
    dst = np.concatenate([n + 1, np.repeat(n + 1, N)]).astype(np.int32)
    lab = np.concatenate([n, np.repeat(n, N)]).astype(np.int32)
    g.add_arcs(src, dst, lab, lab, w.astype(np.float32))
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=512)
    ap.add_argument("--T", type=int, default=1000)
    ap.add_argument("--C", type=int, default=512)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    import torch
    import gtn_amd as gtn
    B, T, C = args.B, args.T, args.C
    torch.manual_seed(0)
    em = (torch.rand(B, T, C, device="cuda") * 10 - 5).contiguous()
    tw = np.random.default_rng(0).random(C * C + C).astype(np.float32)
    out = {"config": "C4: ASG loss: compose with dense CxC transition WFST + viterbiPath, T=1000, C=512, batch=512, 1 MI355X",
           "workload": f"C4 ASG dense transitions B={B} T={T} C={C}", "product_arcs_per_utterance": C * C * (T - 1) + C}

    def sync():
        gtn.synchronize()
        torch.cuda.synchronize()

    # the transitions graph lives across steps (a trainer updates its weights, not its
    # structure), so it is built and uploaded once, outside the timed regions
    trans = transitions(gtn, C, tw)
    trans.arc_sort()  # by ilabel: compose(fal, transitions) then searches instead of scanning 513 arcs per node
    gtn.forward_score(gtn.compose(gtn.linear_graph_n(1, 2, C, em[:1, :2].contiguous()), [trans]))
    # (a) decode
    times = []
    for _ in range(args.steps):
        ems = gtn.linear_graph_n(B, T, C, em)
        sync()
        t0 = time.perf_counter()
        paths = gtn.viterbi_path(gtn.compose(ems, [trans]))
        sync()
        times.append(time.perf_counter() - t0)
    out["decode_ms_per_batch"] = min(times) * 1e3
    out["decode_utt_per_s"] = B / min(times)
    out["decode_path0_head"] = paths[0].labels_to_list()[:8]
    # (b) full-connect score forward + backward
    times = []
    for _ in range(args.steps):
        trans.zero_grad()
        ems = gtn.linear_graph_n(B, T, C, em)
        sync()
        t0 = time.perf_counter()
        fcc = gtn.forward_score(gtn.compose(ems, [trans]))
        sync()
        t1 = time.perf_counter()
        gtn.backward(fcc)
        sync()
        t2 = time.perf_counter()
        times.append((t1 - t0, t2 - t1))
    f, b = min(x[0] for x in times), min(x[1] for x in times)
    out["fcc_forward_ms"] = f * 1e3
    out["fcc_backward_ms"] = b * 1e3
    out["fcc_fwd_bwd_utt_per_s"] = B / (f + b)
    out["fcc_score0"] = float(gtn.items(fcc)[0])
    g = trans.grad().weights_to_numpy()
    out["transitions_grad_sum"] = float(g.sum())  # = B*T: one arc per step per utterance in expectation
    # (c) the whole ASG criterion of examples/asg.cpp:59-68 -- full-connect minus force-align
    # score, U=100 targets; the force-align lattices (compose(emissions, compose(fal, transitions)))
    # are small and go through the materialising compose, the full-connect term stays symbolic
    U = 100
    rng = np.random.default_rng(1)
    tgts = rng.integers(0, C, size=(B, U))
    times = []
    for _ in range(max(1, args.steps - 1)):
        trans.zero_grad()
        ems = gtn.linear_graph_n(B, T, C, em)
        fals = []
        for b in range(B):
            f = gtn.Graph()
            f.add_nodes(np.array([1] + [0] * U, np.uint8), np.array([0] * U + [1], np.uint8))
            lab = tgts[b].astype(np.int32)
            idx = np.arange(1, U + 1, dtype=np.int32)
            f.add_arcs(np.concatenate([idx - 1, idx]), np.concatenate([idx, idx]), np.concatenate([lab, lab]))
            fals.append(f)
        sync()
        t0 = time.perf_counter()
        fcc = gtn.forward_score(gtn.compose(ems, [trans]))
        fal = gtn.forward_score(gtn.compose(ems, gtn.compose(fals, [trans])))
        loss = gtn.subtract(fcc, fal)
        gtn.backward(loss)
        sync()
        times.append(time.perf_counter() - t0)
    out["asg_loss_fwd_bwd_ms"] = min(times) * 1e3
    out["asg_loss_utt_per_s"] = B / min(times)
    lv = gtn.items(loss)
    out["asg_loss_mean"] = float(lv.mean())
    out["asg_loss_min"] = float(lv.min())  # a loss is a -log probability ratio: never negative
    # (d) the same criterion as shipped: gtn_amd.torch_loss.asg_loss over libgtn_criteria.so (batch records:
    # the force-alignment acceptors composed with the transitions are built on the device)
    try:
        from gtn_amd.torch_loss import asg_loss
        emt = em.clone().requires_grad_(True)
        trt = torch.from_numpy(tw[C:].reshape(C, C).copy()).cuda().requires_grad_(True)
        stt = torch.from_numpy(tw[:C].copy()).cuda().requires_grad_(True)
        tl = [t.tolist() for t in tgts]
        times = []
        for _ in range(args.steps):
            for x in (emt, trt, stt):
                x.grad = None
            sync()
            t0 = time.perf_counter()
            l = asg_loss(emt, trt, tl, stt, reduction="none")
            l.sum().backward()
            sync()
            times.append(time.perf_counter() - t0)
        out["asg_criterion_fwd_bwd_ms"] = min(times) * 1e3
        out["asg_criterion_utt_per_s"] = B / min(times)
        out["asg_criterion_loss_mean"] = float(l.mean().item())
    except Exception as e:  # keep the line
        out["asg_criterion_error"] = str(e)[:200]
    # ---- roofline of the dominant kernel family: the dense-regime time steps on the matrix cores
    # (lazy.hip: lazy_mfma_step_kernel, v_mfma_f32_32x32x2_f32).  Algorithmic work of one pass: T products
    # [B x N] . [N x N] in float32 = 2 B N^2 T flops (N = C + 1 nodes of the transitions graph); hipEvent
    # time of the whole pass (T launches) from the engine's profiler; peak = 157.3 TFLOP/s float32 MFMA
    # (MI355X_MICROARCH.md)
    gtn.prof_reset()
    gtn.prof_enable(True)
    trans.zero_grad()
    ems = gtn.linear_graph_n(B, T, C, em)
    fcc = gtn.forward_score(gtn.compose(ems, [trans]))
    gtn.backward(fcc)
    sync()
    paths = gtn.viterbi_path(gtn.compose(ems, [trans]))
    sync()
    gtn.prof_enable(False)
    prof = {n: gtn.prof_get(n) for n in gtn.prof_names() if n.startswith("lazy") or n.startswith("maxplus")}
    out["kernels"] = prof
    N = C + 1
    flops = 2.0 * B * N * N * T
    if prof.get("lazy_forward_score", {}).get("total_ms"):
        ms = prof["lazy_forward_score"]["total_ms"]
        # Once a backward of such a product has been seen, the beta sweep runs BESIDE the alpha sweep (side stream,
        # DESIGN.md section 11.7): the span the profiler times then holds both passes' step launches, and the
        # backward family is left with the gradient contractions only (about half the forward span or less)
        bw = prof.get("lazy_forward_score_grad", {}).get("total_ms") or 0.0
        both = os.environ.get("GTNX_NO_EAGER_BETA") is None and 0.0 < bw < 0.7 * ms
        passes = 2 if both else 1
        tf = passes * flops / (ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma",
                           "kernel": "lazy_mfma_step_kernel<false> x T (+ prep)" + (" beside lazy_mfma_step_kernel<true> x T" if both else ""),
                           "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3, "traffic": None,
                           "ms_per_pass": ms, "flops_per_pass": flops, "passes_in_span": passes}
    # the tropical sweeps (maxplus.hip) are vector-ALU work: one packed add + one max3 per pair of product arcs,
    # i.e. one lane-instruction per arc at best; peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz
    if prof.get("maxplus_viterbi", {}).get("total_ms"):
        ms = prof["maxplus_viterbi"]["total_ms"]
        arcs = float(B) * T * (C * C) + float(B) * C
        out["decode_sweep"] = {"kernel": "maxplus_step_kernel x T (+ prep)", "ms_per_pass": ms,
                               "arc_relaxations_per_s": arcs / (ms * 1e-3), "valu_lane_instr_peak_per_s": 39.3e12,
                               "frac_of_valu_peak": arcs / (ms * 1e-3) / 39.3e12}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(C)
    # headline of the record: the shipped ASG criterion (full-connect minus force-align score, forward + backward) and
    # the Viterbi decode, utterances per second
    out["metric"] = "ASG criterion forward+backward utterances/sec (T=%d, C=%d); decode in decode_utt_per_s" % (T, C)
    out["value"] = out.get("asg_criterion_utt_per_s")
    out["unit"] = "utterances/s"
    # in-run parity: the engine in THIS process against the unmodified reference's answers at C4's alphabet
    # (tests/golden/asg_c512.npz, made by tests/golden/make_golden_c4.py over oracle/_ref: BASELINE's own size,
    # T = 1000, C = 512, two utterances of 262 M product arcs each) -- Viterbi labels EQUAL, forward and Viterbi
    # scores to 1e-4 relative, emission gradients within the reference's own float32 rounding (8 eps |score|; the
    # float64 triangulation at 1e-4 is tests/test_lazy_gpu.py::test_c4_alphabet_pinned_to_the_reference[1000]) --
    # the same kernels as the timed batch: matrix-core forward / backward, max-plus decode
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden_c4 as mk
        gold = np.load(os.path.join(ROOT, "tests", "golden", "asg_c512.npz"))
        Tg = 1000 if "T1000_forward" in gold.files and T >= 1000 else 17
        key = f"T{Tg}"
        Bg = gold[key + "_forward"].shape[0]
        emg, twg = mk.inputs(Tg, Bg, int(gold[key + "_seed"]))
        tr = mk.transitions(gtn, twg)
        prevm = gtn.compose_mode(1)
        try:
            eg = gtn.linear_graph_n(Bg, Tg, mk.C, torch.from_numpy(emg).cuda())
            cg = gtn.compose(eg, [tr])
            fg = gtn.forward_score(cg)
            fsg = np.asarray(gtn.items(fg), np.float64)
            vsg = np.asarray(gtn.items(gtn.viterbi_score(cg)), np.float64)
            pg = gtn.viterbi_path(cg)
            gtn.backward(fg)
            geg = np.stack([x.grad().weights_to_numpy().reshape(Tg, mk.C) for x in eg])
        finally:
            gtn.compose_mode(prevm)
        labels_equal = all(pg[b].labels_to_list() == gold[key + "_labels"][b].tolist() for b in range(Bg))
        rel = float(np.max(np.abs(fsg - gold[key + "_forward"]) / np.abs(gold[key + "_forward"])))
        relv = float(np.max(np.abs(vsg - gold[key + "_viterbi"]) / np.abs(gold[key + "_viterbi"])))
        gerr = float(np.max(np.abs(geg - gold[key + "_grad_emissions"])))
        gtol = float(max(1e-4, 8 * np.finfo(np.float32).eps * np.abs(gold[key + "_forward"]).max()))
        # third corner, utterance 0: the same recursion in float64 (tests/ctc_fp64.py: asg_fp64, ~4 s on the host) --
        # the engine's emission gradients within the north star's 1e-4 of it, and no further from it than the
        # float32 reference's own are
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from ctc_fp64 import asg_fp64
        z64, g64, _, v64, _ = asg_fp64(emg[0], twg)
        d_gpu = float(np.abs(geg[0] - g64).max())
        d_ref = float(np.abs(gold[key + "_grad_emissions"][0] - g64).max())
        rel64 = float(abs(fsg[0] - z64) / abs(z64))
        out["parity_in_run"] = {"n": int(Bg), "T": Tg, "C": int(mk.C), "labels_equal": bool(labels_equal), "forward_score_max_rel": rel,
                                "viterbi_score_max_rel": relv, "grad_emissions_max_abs_vs_reference": gerr,
                                "grad_tolerance_vs_reference": gtol,
                                "forward_score_rel_vs_fp64": rel64,
                                "grad_emissions_max_abs_vs_fp64": d_gpu, "reference_grad_emissions_max_abs_vs_fp64": d_ref,
                                "checker": "reference answers (tests/golden/asg_c512.npz, generated by oracle/_ref) + float64 "
                                           "restatement (tests/ctc_fp64.py: asg_fp64) of utterance 0",
                                "tolerance": "labels EQUAL; scores <= 1e-4 relative; emission gradients: max |engine - float64| <= 1e-4 AND "
                                             "<= max |reference - float64|, and |engine - reference| within the reference's own float32 "
                                             "rounding (8 eps |score|)",
                                "ok": bool(labels_equal and rel <= 1e-4 and relv <= 1e-4 and rel64 <= 1e-4 and gerr <= gtol
                                           and d_gpu <= 1e-4 and d_gpu <= d_ref + 1e-6)}
    except Exception as e:  # reported, and fails the run below
        out["parity_in_run"] = {"error": str(e)[:300], "ok": False}
    print(json.dumps(out))
    if not out.get("parity_in_run", {}).get("ok", False):
        sys.exit(3)


def cpu_baseline(C):
    """the unmodified reference (oracle/_ref) on this host's cores: forwardScore(compose(emissions, transitions)) +
    backward through the binding's VECTOR overloads (parallelMap: one utterance per host thread, as many utterances as
    the host has logical cores -- bounded by memory) at T = 20: the product has C^2 (T - 1) + C arcs, 5 M at C = 512
    (~0.25 GB per utterance in the reference's layout); T = 1000 is 262 M arcs = 11 GB and two minutes per utterance
    (tests/golden/make_golden_c4.py), so the full size is reported as the per-arc extrapolation"""
    path = os.path.join(ROOT, "oracle", "_ref", "libgtn_ref.so")
    if not os.path.exists(path):
        return None
    try:
        import subprocess
        cores = os.cpu_count() or 1
        try:
            import psutil
            room = int(psutil.virtual_memory().available / (0.6 * 2 ** 30))  # utterances that fit, with margin
        except Exception:
            room = 16
        n = int(max(1, min(cores, room, 512)))
        code = (
            "import os, sys, time, numpy as np\n"
            "sys.path.insert(0, %r)\n"
            "sys.path.insert(0, os.path.join(%r, 'tests', 'refbackend'))\n"
            "import gtn_ref as gtn\n"
            "C, T, n = %d, 20, %d\n"
            "rng = np.random.default_rng(0)\n"
            "g = gtn.Graph(); k = np.arange(C)\n"
            "g.add_nodes(np.array([1] + [0] * C, np.uint8), np.array([0] + [1] * C, np.uint8))\n"
            "src = np.concatenate([np.zeros(C, np.int32), np.tile(k + 1, C).astype(np.int32)])\n"
            "dst = np.concatenate([k + 1, np.repeat(k + 1, C)]).astype(np.int32)\n"
            "lab = np.concatenate([k, np.repeat(k, C)]).astype(np.int32)\n"
            "g.add_arcs(src, dst, lab, lab, rng.random(C * C + C).astype(np.float32))\n"
            "es = []\n"
            "for b in range(n):\n"
            "    e = gtn.linear_graph(T, C); e.set_weights((rng.random(T * C) * 10 - 5).astype(np.float32)); es.append(e)\n"
            "t0 = time.perf_counter()\n"
            "f = gtn.forward_score(gtn.compose(es, [g]))\n"
            "gtn.backward(f)\n"
            "dt = time.perf_counter() - t0\n"
            "print(dt)\n" % (ROOT, ROOT, C, n))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ))
        dt = float(r.stdout.strip().splitlines()[-1])
        arcs = C * C * 19 + C
        full = C * C * 999 + C
        return {"kind": "reference", "cores": int(min(cores, n)), "utterances": n, "seconds_T20": dt,
                "value": n / dt * arcs / full, "unit": "utterances/s (T=1000, extrapolated per product arc from T=20)",
                "utterances_per_s_T20": n / dt, "product_arcs_per_utterance_T20": arcs,
                "ns_per_product_arc_per_core": dt * min(cores, n) / (n * arcs) * 1e9,
                "sample": f"forwardScore(compose(emissions, transitions)) + backward of {n} utterances at T=20, C={C}, through the "
                          f"unmodified reference's vector overloads (parallelMap on {min(cores, n)} of the host's {cores} logical "
                          f"cores, one utterance each), {dt:.1f}s wall; T=1000 is 52.6x the arcs per utterance"}
    except Exception as e:
        return {"error": str(e)[:200]}


if __name__ == "__main__":
    main()
