#!/bin/bash
# usage (on the GPU box): R=r06 bash tools/round_profiles.sh
# Everything profiles/ holds for a round, from one box:
#   kernel trace (+ --stats-equivalent summary) of the default bench command  -> ${R}_bench_c3_kernel_stats.csv
#   separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES, waits, LDS, insts)
#   ${R}_traffic.json stamped with the sha256 of csrc/*.hip|*.h (tools/make_traffic_json.py)
#   un-profiled bench lines: c3 (with the CPU leg), rollout-only, c2 / c4 / c5 (+ their kernel stats)
R=${R:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
P=profiles
prof() {  # prof NAME bench-args... : kernel trace -> profiles/${R}_bench_NAME_kernel_stats.csv
  local name=$1; shift
  local D=/tmp/prof_$name; rm -rf $D
  rocprofv3 --kernel-trace -d $D -o r -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$name.log 2>&1
  local DB=$(find $D -name "*.db" | head -1)
  python3 tools/rocpd_stats.py $DB $P/${R}_bench_${name}_kernel_stats.csv > /dev/null 2>&1
  python3 tools/rocpd_stats.py $DB /dev/null --seq "" 420 > gpurun_out/${R}_seq_$name.txt 2>&1
}
pmc() {  # pmc NAME "COUNTERS" : one pass -> profiles/${R}_bench_c3_pmc_NAME.csv
  local name=$1 ctr=$2
  local D=/tmp/pmc_$name; rm -rf $D
  rocprofv3 --pmc $ctr --kernel-trace -d $D -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1
  local DB=$(find $D -name "*.db" | head -1)
  python3 tools/rocpd_pmc.py $DB $P/${R}_bench_c3_pmc_$name.csv > /dev/null 2>&1
}
prof c3 --steps 20 --warmup 5
pmc fetch_size FETCH_SIZE
pmc write_size WRITE_SIZE
pmc sq_valu_mfma_busy_cycles "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc sq_waits "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"
pmc sq_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
pmc sq_insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU"
python3 tools/make_traffic_json.py $R
python3 bench.py > gpurun_out/${R}_bench_c3.log 2>&1
grep '"metric"' gpurun_out/${R}_bench_c3.log > $P/${R}_bench_c3.json
python3 bench.py --rollout-only --no-cpu-baseline 2>/dev/null | grep '"metric"' > $P/${R}_bench_c3_rollout_only.json
for c in c2 c4 c5; do
  python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | grep '"metric"' > $P/${R}_bench_$c.json
  prof $c --config $c --steps 10 --warmup 3
done
prof c3_rollout_only --rollout-only --steps 10 --warmup 3
# the strong-scaling operating point of the C3 shapes (256 images over 8 GPUs = 32 per GPU)
python3 bench.py --batch 32 --no-cpu-baseline 2>/dev/null | grep '"metric"' > $P/${R}_bench_c3_b32.json
# the kernel-level lab records, regenerated from THIS tree (a stale one was cited as proof)
python3 tools/g3_lab.py > gpurun_out/${R}_g3_lab.log 2>&1 && cp gpurun_out/g3_lab.json $P/${R}_g3_lab.json
python3 - <<PY
import json
rows = json.load(open("$P/${R}_g3_lab.json"))
bad = [r for r in rows if r.get("max_err", 0) > 1e-3 or r.get("err_h", 0) > 1e-4 or r.get("err_c", 0) > 1e-4
       or r.get("pipelined_eq_safe") is False or r.get("h_image_ok") is False]
print("g3_lab rows:", len(rows), "bad:", bad)
assert not bad
PY
# parity records written by the GPU tests (tests/util.record) -> profiles/
for f in gpurun_out/${R}_achieved_errors.json gpurun_out/${R}_full_gradient.json gpurun_out/${R}_distinct_*.json gpurun_out/gemm_errors.json; do
  [ -f $f ] && cp $f $P/$(basename $f | sed "s/^gemm_errors/${R}_gemm_errors/")
done
cp -r $P gpurun_out/profiles_$R
for f in $P/${R}_bench_c3.json $P/${R}_bench_c3_rollout_only.json $P/${R}_bench_c2.json $P/${R}_bench_c4.json $P/${R}_bench_c5.json; do cut -c1-260 $f; done
head -12 $P/${R}_bench_c3_kernel_stats.csv
head -8 $P/${R}_bench_c3_pmc_sq_valu_mfma_busy_cycles.csv
