#!/bin/bash
# usage (on the GPU box): R=r06 bash tools/round_profiles.sh
# Everything profiles/ holds for a round, from one box:
#   kernel trace (+ --stats-equivalent summary) of the default bench command  -> ${R}_bench_c3_kernel_stats.csv
#   separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE, waits, LDS, insts)
#   ${R}_traffic.json stamped with the sha256 of csrc/*.hip|*.h (tools/make_traffic_json.py)
#   un-profiled bench lines: c3 (with the CPU leg), rollout-only, c2 / c4 / c5, c4 at 256 images (+ their kernel stats)
# A failed step ABORTS the script and every target is deleted before it is regenerated (a failed rocprofv3
# pass used to leave the previous round's record in place under the new prefix).
set -euo pipefail
R=${R:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out profiles
P=profiles
db_of() {  # the rocpd database of a profiler output directory, or abort
  local DB
  DB=$(find "$1" -name "*.db" | head -1)
  [ -n "$DB" ] || { echo "round_profiles: no rocpd database under $1 (see gpurun_out/*.log)"; exit 1; }
  echo "$DB"
}
bench_line() {  # bench_line OUT bench-args... : the JSON line of an un-profiled run, or abort
  local out=$1; shift
  rm -f "$out"
  python3 bench.py "$@" 2> gpurun_out/bench_err.log | grep '"metric"' > "$out" || { echo "round_profiles: no bench line for $out"; tail -5 gpurun_out/bench_err.log; exit 1; }
}
prof() {  # prof NAME bench-args... : kernel trace -> profiles/${R}_bench_NAME_kernel_stats.csv
  local name=$1; shift
  local D=/tmp/prof_$name; rm -rf $D; rm -f $P/${R}_bench_${name}_kernel_stats.csv
  rocprofv3 --kernel-trace -d $D -o r -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/prof_$name.log 2>&1
  local DB; DB=$(db_of $D)
  python3 tools/rocpd_stats.py $DB $P/${R}_bench_${name}_kernel_stats.csv > /dev/null
  python3 tools/rocpd_stats.py $DB /dev/null --seq "" 420 > gpurun_out/${R}_seq_$name.txt
}
pmc() {  # pmc NAME "COUNTERS" [bench-args...] : one pass -> profiles/${R}_bench_<cfg>_pmc_NAME.csv
  local name=$1 ctr=$2; shift 2
  local tag=c3; [ $# -gt 0 ] && tag=$1 && shift
  local D=/tmp/pmc_${tag}_$name; rm -rf $D; rm -f $P/${R}_bench_${tag}_pmc_$name.csv
  rocprofv3 --pmc $ctr --kernel-trace -d $D -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/pmc_${tag}_$name.log 2>&1
  local DB; DB=$(db_of $D)
  python3 tools/rocpd_pmc.py $DB $P/${R}_bench_${tag}_pmc_$name.csv > /dev/null
  if [ "$name" = sq_valu_mfma_busy_cycles ]; then python3 tools/rocpd_clock.py $DB $P/${R}_bench_${tag}_clock.csv 15 > /dev/null; fi
}
prof c3 --steps 20 --warmup 5
pmc fetch_size FETCH_SIZE
pmc write_size WRITE_SIZE
pmc sq_valu_mfma_busy_cycles "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
pmc sq_waits "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY"
pmc sq_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
pmc sq_insts "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM SQ_INSTS_SALU"
rm -f $P/${R}_traffic.json
python3 tools/make_traffic_json.py $R
bench_line $P/${R}_bench_c3.json
bench_line $P/${R}_bench_c3_rollout_only.json --rollout-only --no-cpu-baseline
for c in c2 c4 c5; do
  bench_line $P/${R}_bench_$c.json --config $c --no-cpu-baseline
  prof $c --config $c --steps 10 --warmup 3
done
prof c3_rollout_only --rollout-only --steps 10 --warmup 3
# the strong-scaling operating point of the C3 shapes (256 images over 8 GPUs = 32 per GPU)
bench_line $P/${R}_bench_c3_b32.json --batch 32 --no-cpu-baseline
# BASELINE configs[3]'s weak-scaling variant: AID at 256 images per GPU
bench_line $P/${R}_bench_c4_b256.json --config c4 --batch 256 --no-cpu-baseline --steps 30 --warmup 5
prof c4_b256 --config c4 --batch 256 --steps 6 --warmup 2
for c in c4 c5; do
  pmc fetch_size FETCH_SIZE $c --config $c
  pmc write_size WRITE_SIZE $c --config $c
  pmc sq_valu_mfma_busy_cycles "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" $c --config $c
done
# the kernel-level lab records, regenerated from THIS tree (a stale one was cited as proof)
rm -f $P/${R}_g3_lab.json gpurun_out/g3_lab.json
python3 tools/g3_lab.py > gpurun_out/${R}_g3_lab.log 2>&1
cp gpurun_out/g3_lab.json $P/${R}_g3_lab.json
python3 - <<PY
import json
rows = json.load(open("$P/${R}_g3_lab.json"))
bad = [r for r in rows if r.get("max_err", 0) > 1e-3 or r.get("err_h", 0) > 1e-4 or r.get("err_c", 0) > 1e-4
       or r.get("pipelined_eq_safe") is False or r.get("h_image_ok") is False]
print("g3_lab rows:", len(rows), "bad:", bad)
assert not bad
PY
# round 6's lab records (row contractions with asm LDS-DMA / the phase pipeline against round 5's library; the
# phase-pipelined NT / LSTM kernels; the pipeline with memory out of the way)
for t in tn_pipe_lab ntp_lab tn_pipe_probe; do
  rm -f gpurun_out/$t.json $P/${R}_$t.json
  python3 tools/$t.py > gpurun_out/${R}_$t.log 2>&1
  cp gpurun_out/$t.json $P/${R}_$t.json
done
python3 tools/lstm_p_probe.py > $P/${R}_lstm_p_probe.txt 2>&1
# parity records written by the GPU tests (tests/util.record) -> profiles/
for f in gpurun_out/${R}_achieved_errors.json gpurun_out/${R}_full_gradient.json gpurun_out/${R}_distinct_*.json gpurun_out/gemm_errors.json; do
  [ -f $f ] && cp $f $P/$(basename $f | sed "s/^gemm_errors/${R}_gemm_errors/")
done
rm -rf gpurun_out/profiles_$R; cp -r $P gpurun_out/profiles_$R
for f in $P/${R}_bench_c3.json $P/${R}_bench_c3_rollout_only.json $P/${R}_bench_c2.json $P/${R}_bench_c4.json $P/${R}_bench_c5.json $P/${R}_bench_c4_b256.json; do cut -c1-260 $f; done
head -12 $P/${R}_bench_c3_kernel_stats.csv
head -8 $P/${R}_bench_c3_clock.csv
