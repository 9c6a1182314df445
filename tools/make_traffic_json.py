"""profiles/r02_traffic.json from the rocprofv3 --pmc passes (tools/gpu_pmc.sh): HBM bytes per
launch of every kernel class of marl_profile_begin, = 2 x FETCH_SIZE (gfx950 correction for wide
coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both reported in KB.
usage: python tools/make_traffic_json.py TAG"""
import csv
import json
import sys

tag = sys.argv[1]
CLASS_OF = [("gemm_nt_kernel<128, 128, 4, 1, true", 0), ("gemm_nt_kernel", 1), ("gemm_tn_kernel", 2),
            ("cnn_fwd", 3), ("panel_", 4), ("cnn_dgrad", 5), ("cnn_wgrad", 5)]


def load(c):
    return {r["kernel"]: (int(r["calls"]), float(r[c])) for r in
            csv.DictReader(open(f"gpurun_out/r02_pmc_{c}_{tag}.csv"))}


F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {}
for k, (calls, f) in F.items():
    cls = next((c for p, c in CLASS_OF if k.startswith(p)), None)
    if cls is None or k not in W:
        continue
    e = out.setdefault(str(cls), {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0, "kernels": []})
    e["launches"] += calls
    e["fetch_kb"] += calls * f
    e["write_kb"] += calls * W[k][1]
    e["kernels"].append(k)
for e in out.values():
    n = e.pop("launches")
    f, w = e.pop("fetch_kb") / n, e.pop("write_kb") / n
    e.update(fetch_size_kb_raw=round(f), write_size_kb=round(w), fetch_correction=2.0,
             traffic_bytes_per_launch=int((2 * f + w) * 1024))
out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (two separate passes) -- python3 "
                 f"bench.py --steps 2 --warmup 1 --no-cpu-baseline; per-launch averages per kernel class; summaries in "
                 f"profiles/r02_bench_c3_pmc_*.csv")
json.dump(out, open("profiles/r02_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
