import json, sys, glob, os
d = sys.argv[1]
for f in sorted(glob.glob(os.path.join(d, "03_bench_*.json"))):
    try:
        j = json.load(open(f)); r = j["roofline"]; t = j["timing"]
        iss = r.get("issue") or {}
        print("%-22s %8.0f v/s %.4f ms (med %.4f) | dom %-9s %.1f us frac %.3f traffic %s MB | bound %s valu_busy %s | step bytes %s MB" % (
            os.path.basename(f)[9:-5], j["value"], j["ms_per_step"], t["ms_per_step_median"], r["kernel"], r["kernel_ms"] * 1e3, r["frac"],
            None if r["traffic"] is None else round(r["traffic"] / 1e6, 1), r["bound"], None if not iss else round(iss["valu_busy"], 2),
            None if r["step_measured_bytes"] is None else round(r["step_measured_bytes"] / 1e6)))
    except Exception as e:
        print(f, "ERR", e)
