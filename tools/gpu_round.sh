#!/bin/bash
# One GPU-box visit: tests, bench, rocprofv3 kernel stats and SQ counter passes.  Usage: tools/gpu_round.sh <tag> [stages...]
# stages: tests bench stats pmc traffic  (default: all).  Everything lands under gpurun_out/<tag>_*.
tag=$1; shift
stages=${@:-tests bench stats pmc}
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out
mkdir -p $O
for s in $stages; do
  case $s in
    tests) timeout 2000 python -m pytest tests -m gpu -q --durations=15 ${PYTEST_ARGS} > $O/${tag}_tests.txt 2>&1; echo "tests rc=$?"; tail -25 $O/${tag}_tests.txt ;;
    bench) timeout 900 python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.log; echo "bench rc=$?"; tail -3 $O/${tag}_bench.log; cat $O/${tag}_bench.json ;;
    stats) (cd /tmp && timeout 240 rocprofv3 --output-format csv --kernel-trace --stats -d $O/${tag}_stats -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-ddim --no-legs > $O/${tag}_stats.log 2>&1); echo "stats rc=$?" ;;
    pmc)
      (cd /tmp && timeout 240 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d $O/${tag}_pmcA -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim --no-legs > $O/${tag}_pmcA.log 2>&1); echo "pmcA rc=$?"
      (cd /tmp && timeout 240 rocprofv3 --output-format csv --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d $O/${tag}_pmcB -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim --no-legs > $O/${tag}_pmcB.log 2>&1); echo "pmcB rc=$?" ;;
    traffic)
      for c in FETCH_SIZE WRITE_SIZE; do
        (cd /tmp && timeout 240 rocprofv3 --output-format csv --pmc $c --kernel-trace -d $O/${tag}_pmc_$c -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-ddim --no-legs > $O/${tag}_pmc_$c.log 2>&1); echo "traffic $c rc=$?"
      done ;;
  esac
done
# keep only what fits the 64 MiB merge-back budget: counter CSVs are aggregated per kernel here
python $R/tools/summarize_pmc.py $O $tag || true
find $O -name "*.db" -size +20M -delete 2>/dev/null
du -sh $O | tail -1
