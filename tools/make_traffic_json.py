"""profiles/<round>_traffic.json from the rocprofv3 --pmc passes (tools/round_profiles.sh): HBM bytes per
launch of every kernel class of marl_profile_begin, = 2 x FETCH_SIZE (gfx950 correction for wide
coalesced reads, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, both reported in KB.  The file is
stamped with the sha256 of csrc/*.hip|*.h at profiling time; bench.py only quotes it while the
sources it runs still hash to that value.
usage: python tools/make_traffic_json.py r06"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import csrc_sha256  # noqa: E402

rnd = sys.argv[1]
CLASS_OF = [("gemm_nt3_kernel<128, 128, 4, 1, 3, true", 0), ("gemm_nt3_kernel<256, 128, 8, 1, 4, true", 0),
            ("gemm_nt3_kernel", 1), ("gemm_tn3_kernel", 2), ("gemm_tn_batch_kernel", 2),
            ("gemm_nt_kernel<128, 128, 4, 1, true", 0), ("gemm_nt_split_kernel<128, true", 0),
            ("gemm_nt_kernel", 1), ("gemm_nt_split_kernel", 1), ("gemm_tn_kernel", 2),
            ("gemm_tn_split_kernel", 2), ("gemm_tn", 2),  # (any other row-contraction form: gemm_tn3_passes_kernel ...)
            ("gemm_nt", 1), ("cnn_fwd", 3), ("panel_", 4), ("cnn_dgrad", 5), ("cnn_wgrad", 5),
            ("cnn_bwd", 5)]


def load(c):
    return {r["kernel"]: (int(r["calls"]), float(r[c])) for r in
            csv.DictReader(open(f"profiles/{rnd}_bench_c3_pmc_{c.lower()}.csv"))}


F, W = load("FETCH_SIZE"), load("WRITE_SIZE")
out = {}
for k, (calls, f) in F.items():
    cls = next((c for p, c in CLASS_OF if k.startswith(p)), None)
    if cls is None and k.startswith("gemm"):  # (a GEMM kernel outside every class under-reports one)
        raise SystemExit(f"make_traffic_json: GEMM kernel {k[:80]} matches no class prefix")
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
tot_f = sum(c * f for c, f in F.values())
tot_w = sum(c * W[k][1] for k, (c, _) in F.items() if k in W)
out["whole_run_2xfetch_plus_write_gb"] = round((2 * tot_f + tot_w) * 1024 / 1e9, 3)
# iterations in the profiled command: 1 warm-up + 2 timed + one per kernel class of the sweep
lstm = next((c for k, (c, _) in F.items() if k.startswith(("gemm_nt3_kernel<128, 128, 4, 1, 3, true", "gemm_nt_split_kernel<128, true",
                                                            "gemm_nt_kernel<128, 128, 4, 1, true"))), 0)
out["whole_run_iterations"] = lstm // 16 if lstm else None  # (the LSTM kernel runs once per step, 16 steps)
if lstm:
    out["per_iteration_2xfetch_plus_write_gb"] = round(out["whole_run_2xfetch_plus_write_gb"] / (lstm // 16), 3)
out["src_sha256"] = csrc_sha256()
out["source"] = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace (two separate passes) -- python3 "
                 "bench.py --steps 2 --warmup 1 --no-cpu-baseline; per-launch averages per kernel class; summaries in "
                 f"profiles/{rnd}_bench_c3_pmc_*.csv")
json.dump(out, open(f"profiles/{rnd}_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
