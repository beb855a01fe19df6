#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/collect_profiles.sh'): the measurements that
# profiles/rNN/ holds -- the default bench line, rocprofv3 kernel-trace stats of a 100k-bin
# bench run, and the HBM traffic counters of one full-size forward launch (separate --pmc
# passes, kernel trace only).  Output lands in gpurun_out/prof/; copy what should be judged
# into profiles/.
set -o pipefail
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/prof
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
echo "== bench (default: 1e6 bins x 64 penalties)"
python3 "$REPO/bench.py" > "$OUT/bench_1e6x64_full.json.log" 2>&1 || exit 1
grep metric "$OUT/bench_1e6x64_full.json.log" | cut -c1-400
echo "== rocprofv3 --kernel-trace --stats (100k bins)"
rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o bench100k -- \
  python3 "$REPO/bench.py" --bins 100000 --steps 3 --warmup 1 --no-cpu \
  > "$OUT/bench100k_under_rocprofv3.json.log" 2>&1 || exit 1
grep metric "$OUT/bench100k_under_rocprofv3.json.log" | cut -c1-300
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $c (one full-size launch)"
  rocprofv3 --pmc $c --kernel-trace -d "$OUT/pmc_$c" -o pmc -- \
    python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu > "$OUT/pmc_$c.log" 2>&1 || exit 1
done
# SQ counters of a 20k-bin run (what bounds the kernel: instruction mix, issue vs wait cycles)
echo "== rocprofv3 --pmc SQ_* (20k bins)"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS \
  SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d "$OUT/pmc_sq1" -o pmc -- \
  python3 "$REPO/bench.py" --bins 20000 --steps 1 --warmup 0 --no-cpu > "$OUT/pmc_sq1.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH \
  SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --kernel-trace \
  -d "$OUT/pmc_sq2" -o pmc -- \
  python3 "$REPO/bench.py" --bins 20000 --steps 1 --warmup 0 --no-cpu > "$OUT/pmc_sq2.log" 2>&1 || exit 1
# stamped diagnostic build: phase shares and wave-level Newton trips (input of tools/chain_floor.py)
echo "== phase profile (stamped build, 50k bins x 8 penalties)"
cd "$REPO" && python3 tools/phase_profile.py 50000 8 > "$OUT/phase_shares.log" 2>&1 || exit 1
tail -4 "$OUT/phase_shares.log" | cut -c1-200
find "$OUT" -name "*.db" | head -20
