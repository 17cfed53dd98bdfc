"""Fixtures that pin BASELINE config C4's alphabet (ASG, dense transitions, C = 512) to the UNMODIFIED reference:
tests/golden/asg_c512.npz.  Run where /root/reference has been compiled into oracle/_ref (build()):

    python tests/golden/make_golden_c4.py

For T in (17, 100) and B = 2 utterances each (5 M / 26 M product arcs per utterance) and for BASELINE's own
T = 1000 (B = 2: 262 M product arcs = ~11 GB and about a minute of the reference per utterance; run with
`--only T1000`, keys already in the .npz are kept as they are): forwardScore and viterbiScore of
compose(emissions, transitions), viterbiPath's labels, and after backward(forwardScore) the emission gradients and
the shared transitions' gradient (summed over the utterances, criterion_test.cpp:289-305).  Inputs are regenerated
from the seeds by the test (numpy Generator streams are stable); a checksum of them is stored.
What it mirrors: examples/asg.cpp:36-47 (transitions), :59-68 (full-connect score), test/criterion_test.cpp:308-345
(Viterbi labels)."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "refbackend"))

C = 512
CASES = [(17, 2, 101), (100, 2, 202), (1000, 2, 303)]  # (T, B, seed)


def inputs(T, B, seed):
    rng = np.random.default_rng(seed)
    em = rng.normal(0, 2, (B, T, C)).astype(np.float32)
    tw = rng.normal(0, 1, C * C + C).astype(np.float32)  # arc i: start -> label i; arc C + i*C + j: label j -> label i
    return em, tw


def transitions(api, tw):
    n = np.arange(C)
    g = api.Graph()
    g.add_nodes(np.array([1] + [0] * C, np.uint8), np.array([0] + [1] * C, np.uint8))
    g.add_arcs(np.concatenate([np.zeros(C, np.int32), np.tile(n + 1, C)]).astype(np.int32),
               np.concatenate([n + 1, np.repeat(n + 1, C)]).astype(np.int32),
               np.concatenate([n, np.repeat(n, C)]).astype(np.int32), None, tw)
    return g


def main():
    import gtn_ref as ref
    assert ref.backend() == "reference-cpu"
    path = os.path.join(HERE, "asg_c512.npz")
    only = [a for a in sys.argv[1:] if a.startswith("T")]
    out = dict(np.load(path)) if only and os.path.exists(path) else {}
    for T, B, seed in CASES:
        if only and f"T{T}" not in only:
            continue
        em, tw = inputs(T, B, seed)
        trans = transitions(ref, tw)
        fs, vs, labels, gem = [], [], [], []
        t0 = time.time()
        for b in range(B):
            e = ref.linear_graph(T, C)
            e.set_weights(em[b].reshape(-1))
            comp = ref.compose(e, trans)
            f = ref.forward_score(comp)
            fs.append(f.item())
            vs.append(ref.viterbi_score(comp).item())
            labels.append(np.asarray(ref.viterbi_path(comp).labels_to_list(), np.int32))
            ref.backward(f)
            gem.append(e.grad().weights_to_numpy().reshape(T, C).astype(np.float32))
            del comp, f
            print("  utterance", b, "%.1f s" % (time.time() - t0), flush=True)
        key = f"T{T}"
        out[key + "_seed"] = np.int64(seed)
        out[key + "_input_checksum"] = np.float64(em.astype(np.float64).sum() + 3.0 * tw.astype(np.float64).sum())
        out[key + "_forward"] = np.asarray(fs, np.float32)
        out[key + "_viterbi"] = np.asarray(vs, np.float32)
        out[key + "_labels"] = np.stack(labels)
        out[key + "_grad_emissions"] = np.stack(gem)
        out[key + "_grad_transitions"] = trans.grad().weights_to_numpy().astype(np.float32)
        print(key, "done in %.1f s" % (time.time() - t0), "forward", fs, "viterbi", vs, "labels[0][:8]", labels[0][:8])
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
