import json, sys
for line in sys.stdin:
    if line.startswith("{"):
        d = json.loads(line)
        r = d.get("roofline") or {}
        print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), "losses/s", round(d["ms_per_step"], 1), "ms/step",
              "fs_frac", round(r.get("frac", 0), 3), {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
